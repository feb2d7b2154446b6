#!/bin/bash
# PMC comparison of the tile kernel between the production build and an ablation build (GPU box)
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_cmp; rm -rf $OUT; mkdir -p $OUT
run() {  # tag, libpath
  RBVFIT_AMD_LIB=$2 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$1_a -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --walkers ${W:-2048} > $OUT/$1_a.json 2> $OUT/$1_a.err
  RBVFIT_AMD_LIB=$2 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/$1_b -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --walkers ${W:-2048} > $OUT/$1_b.json 2> $OUT/$1_b.err
}
run prod ""
run nocold $PWD/rbvfit_amd/lib/ablate/lib_ablate1.so
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get("OUT", os.getcwd()+"/gpurun_out/pmc_cmp")
for tag in ("prod","nocold"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile_kernel<0, 0>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: round(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
