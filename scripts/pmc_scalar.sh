#!/bin/bash
# Scalar data cache / instruction cache behaviour of the tile kernel (W walkers, default 512)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_scalar; rm -rf $OUT; mkdir -p $OUT
W=${W:-512}
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 --walkers $W > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 --walkers $W > $OUT/b.json 2> $OUT/b.err
python3 - <<PY
import csv, glob, os, collections
out=os.getcwd()+"/gpurun_out/pmc_scalar"
W=$W
acc=collections.defaultdict(list)
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile_kernel<0, 0, false>" in r["Kernel_Name"] and int(r["Grid_Size"]) % (W * 64) == 0 and int(r["Grid_Size"]) // (W * 64) in (10, 11, 12):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} {sum(v)/len(v):14.0f}   per launch ({len(v)} launches)")
PY
