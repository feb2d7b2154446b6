#!/bin/bash
# VALU instruction mix (all / f64 fma,mul,add,trans / SALU / LDS / SMEM) per walker-eval for the production
# and the ablation builds (scripts/build_ablations.sh first).  W walkers (default 512).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_mix2; rm -rf $OUT; mkdir -p $OUT
W=${W:-512}
for tag in ${TAGS:-prod 1 2 3 5}; do
  lib=""; [ "$tag" != "prod" ] && lib=$PWD/rbvfit_amd/lib/ablate/lib_ablate$tag.so
  RBVFIT_AMD_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --walkers $W > $OUT/$tag.json 2> $OUT/$tag.err
done
python3 - <<PY
import csv, glob, os, collections
out=os.getcwd()+"/gpurun_out/pmc_mix2"
W=$W
for tag in ("prod","1","2","3","5"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile_kernel<0, 0, false>" in r["Kernel_Name"] and int(r["Grid_Size"]) % (W * 64) == 0 and int(r["Grid_Size"]) // (W * 64) in (10, 11, 12):   # only the W-walker two-pass launches
                acc[r["Counter_Name"].replace("SQ_INSTS_","")].append(float(r["Counter_Value"]))
    d={k: sum(v)/len(v)/W for k,v in acc.items()}
    f64=sum(d.get(k,0) for k in ("VALU_FMA_F64","VALU_MUL_F64","VALU_ADD_F64","VALU_TRANS_F64"))
    print(f"{tag:5s} VALU {d.get('VALU',0):7.0f}  f64 {f64:7.0f}  other VALU {d.get('VALU',0)-f64:7.0f}  SALU {d.get('SALU',0):7.0f}  LDS {d.get('LDS',0):6.0f}  SMEM {d.get('SMEM',0):6.0f}   per walker-eval")
PY
