#!/bin/bash
# Round 5: the per-phase budget of walker_kernel / tile_kernel1 by ablation builds (VP_DIAG bits, csrc/voigt_kernels.h).
#   scripts/r5_ablate.sh build            (here, no GPU)  -> rbvfit_amd/lib/exp/abl_<name>.so for every variant
#   scripts/r5_ablate.sh run <tag> <bench args...>  (GPU box) -> gpurun_out/abl_<tag>/table.txt: per variant kernel time (bench.py's
#                                          HIP-event figure) and SQ_INSTS_VALU / SALU / SMEM / LDS / SQ_WAVES per launch (one --pmc pass)
# Variants: product (VP_DIAG=0) and one build per bit, plus the skeleton (all evaluation removed: what the loops and decisions cost).
VARIANTS="base:0 nofar:1 nonear:2 nocore:4 noexp:8 nolsf:16 nochi:32 entry:64 skel:47 nosmallexp:128 empty:512"
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-result -Wno-unused-value -Wno-invalid-offsetof -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=iterative-ilp"
if [ "$1" = "build" ]; then
  mkdir -p rbvfit_amd/lib/exp
  for v in $VARIANTS; do
    n=${v%%:*}; b=${v#*:}
    ( cd rbvfit_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DVP_DIAG=$b -o ../lib/exp/abl_$n.so capi.hip 2> ../lib/exp/abl_$n.log && echo built $n ) &
    while [ $(jobs -r | wc -l) -ge ${JOBS:-5} ]; do sleep 1; done
  done
  wait
  exit 0
fi
TAG=$2; shift 2
O=$PWD/gpurun_out/abl_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
: > $O/table.txt
for v in ${ONLY:-$VARIANTS}; do
  n=${v%%:*}
  lib=$PWD/rbvfit_amd/lib/exp/abl_$n.so
  [ -f $lib ] || continue
  RBVFIT_AMD_LIB=$lib python3 bench.py --no-cpu-baseline --no-extras --min-seconds 0.4 --steps 200 "$@" > $O/$n.json 2> $O/$n.err
  RBVFIT_AMD_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_$n -- python3 bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 5 --repeats 2 "$@" > /dev/null 2> $O/pmc_$n.err
  python3 - "$O" "$n" >> $O/table.txt <<'PY'
import sys, json, glob, csv, collections
O, n = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{O}/{n}.json").read().strip().splitlines()[-1]); r = d["roofline"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_{n}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
W = d["config"]["walkers_per_gpu"]
out = [f"{n:11s} us/step {1e3 * d['ms_per_step']:8.2f}  kernel us {1e3 * r['avg_kernel_ms']:8.2f}"]
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", [0]))):
    if "walker_kernel" in k or "tile_kernel" in k or "farfield" in k or "prep_lines" in k:
        m = {a: sum(v) / len(v) for a, v in c.items()}
        out.append(f"    {k[:44]:44s} launches {len(c.get('SQ_INSTS_VALU', []))//1:5d}  per eval: VALU {64 * m.get('SQ_INSTS_VALU', 0) / W / 64:9.0f} SALU {m.get('SQ_INSTS_SALU', 0) / W:9.0f} SMEM {m.get('SQ_INSTS_SMEM', 0) / W:8.0f} LDS {m.get('SQ_INSTS_LDS', 0) / W:8.0f}  waves {m.get('SQ_WAVES', 0):8.0f}")
print("\n".join(out))
PY
  rm -rf $O/pmc_$n
  tail -3 $O/table.txt
done
