"""Histogram of a rocprofv3 PC-sampling CSV by kernel, source line and instruction (round 5: the per-phase budget).
Usage: pc_hist.py <pc_sampling csv> [<kernel_trace csv>]      -> text on stdout (gpurun_out/pcs_<tag>/hist.txt)"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
pcs, ktrace = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else None)
kname = {}
if ktrace:
    with open(ktrace, newline="") as f:
        for r in csv.DictReader(f):
            kname[r.get("Dispatch_Id")] = re.sub(r"\(.*", "", r.get("Kernel_Name", "?"))
by_kernel = collections.Counter()
by_line = collections.defaultdict(collections.Counter)     # kernel -> source line -> samples
by_inst = collections.defaultdict(collections.Counter)     # kernel -> (source line, instruction text) -> samples
by_op = collections.defaultdict(collections.Counter)       # kernel -> mnemonic -> samples
stall = collections.defaultdict(collections.Counter)       # kernel -> stall reason (stochastic) -> samples
issued = collections.defaultdict(collections.Counter)
n = 0
with open(pcs, newline="") as f:
    rd = csv.DictReader(f)
    cols = rd.fieldnames
    for r in rd:
        n += 1
        k = kname.get(r.get("Dispatch_Id"), "dispatch?")
        ins = r.get("Instruction", "")
        src = r.get("Instruction_Comment", "")
        src = re.sub(r"^.*/rbvfit_amd/csrc/", "", src)
        by_kernel[k] += 1
        by_line[k][src] += 1
        by_inst[k][(src, ins)] += 1
        by_op[k][ins.split(" ")[0] if ins else "?"] += 1
        if "Stall_Reason" in r:
            stall[k][r.get("Stall_Reason")] += 1
            issued[k][r.get("Wave_Issued_Instruction")] += 1
print("columns:", cols)
print("samples:", n)
for k, c in by_kernel.most_common(6):
    print(f"\n=== {k}: {c} samples ({100.0 * c / max(n, 1):.1f} %)")
    if stall[k]:
        print("  issued:", dict(issued[k].most_common(4)))
        print("  stall reasons:", dict(stall[k].most_common(12)))
    print("  -- by mnemonic")
    for op, v in by_op[k].most_common(40):
        print(f"  {v:8d} {100.0 * v / c:6.2f} %  {op}")
    print("  -- by source line")
    for src, v in by_line[k].most_common(160):
        print(f"  {v:8d} {100.0 * v / c:6.2f} %  {src}")
    print("  -- by instruction")
    for (src, ins), v in by_inst[k].most_common(120):
        print(f"  {v:8d} {100.0 * v / c:6.2f} %  {src:40s} {ins}")
