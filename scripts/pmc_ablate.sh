#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS / SMEM / VMEM per tile-kernel launch for the production and ablation builds
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_abl; rm -rf $OUT; mkdir -p $OUT
for tag in prod 1 2 3 5 6; do
  lib=""; [ "$tag" != "prod" ] && lib=$PWD/rbvfit_amd/lib/ablate/lib_ablate$tag.so
  RBVFIT_AMD_LIB=$lib rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --walkers ${W:-512} > $OUT/$tag.json 2> $OUT/$tag.err
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.getcwd()+"/gpurun_out/pmc_abl"
for tag in ("prod","1","2","3","5","6"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile_kernel<0, 0, false>" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d={k: round(sum(v)/len(v)) for k,v in sorted(acc.items())}
    print(tag, "VALU/eval=%.0f"%(d.get('SQ_INSTS_VALU',0)/512), d)
PY
