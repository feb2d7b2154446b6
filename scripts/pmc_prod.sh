#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_prod; rm -rf $OUT; mkdir -p $OUT
W=${W:-512}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 --walkers $W > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 --walkers $W > $OUT/b.json 2> $OUT/b.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/c -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 --walkers $W > $OUT/c.json 2> $OUT/c.err
python3 - <<'PY'
import csv, glob, os, collections
out=os.getcwd()+"/gpurun_out/pmc_prod"
acc=collections.defaultdict(list); dur=[]
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile_kernel<0, 0, false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:28s} {sum(v)/len(v):14.0f}")
PY
