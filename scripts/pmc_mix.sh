#!/bin/bash
# VALU instruction mix of the tile kernel (per launch): f64 arithmetic vs integer / conversions / the rest
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_mix; rm -rf $OUT; mkdir -p $OUT
W=${W:-512}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --walkers $W > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VSKIPPED --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --walkers $W > $OUT/b.json 2> $OUT/b.err
python3 - <<'PY'
import csv, glob, os, collections
out=os.getcwd()+"/gpurun_out/pmc_mix"
acc=collections.defaultdict(list)
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile_kernel<0, 0, false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
W=int(os.environ.get("W","512"))
for k,v in sorted(acc.items()): print(f"{k:28s} {sum(v)/len(v)/W:10.0f} per eval")
PY
