#!/bin/bash
# GPU box: the round's profile set for one config -- rocprofv3 kernel statistics, PMC passes (scripts/profile_config.sh) and the
# bench line of the driver's own command.  Usage: scripts/refresh_profiles.sh <CFG> <steps> [bench args...]
CFG=$1; STEPS=$2; shift 2
scripts/profile_config.sh $CFG $CFG $STEPS "$@" > /dev/null 2>&1
python3 bench.py --config $CFG --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof_$CFG/bench_driverform.json 2> gpurun_out/prof_$CFG/bench_driverform.err
head -8 gpurun_out/prof_$CFG/summary.txt
