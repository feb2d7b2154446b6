#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE, SQ sets) of
#   python3 bench.py --config <CFG> --no-cpu-baseline --no-extras --steps <S> --warmup 5 [extra bench args]
# Usage: scripts/profile_config.sh <tag> <CFG> <steps> [bench args...]   -> gpurun_out/prof_<tag>/{summary.txt,pmc.json,kernel_stats.csv,bench.json}
set -e
TAG=$1; CFG=$2; STEPS=$3; shift 3
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--config $CFG --no-cpu-baseline --no-extras --steps $STEPS --warmup 5 --repeats 3 $@"
echo "python3 bench.py $ARGS" > $OUT/command.txt
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || true
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/bench_pmc_sq2.json 2> $OUT/pmc_sq2.err || true
python3 scripts/summarize_prof.py $OUT $CFG > $OUT/summary.txt 2>&1 || true
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2
cat $OUT/summary.txt | head -60
