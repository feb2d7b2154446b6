#!/bin/bash
# GPU box: which SQ level / wait counters the box offers, then three --pmc passes of the C1 line (average latency of scalar loads,
# vector loads and LDS operations = LEVEL / INSTS; what the waves wait for).  Usage: scripts/r5_levels.sh [walkers]
W=${1:-512}
O=gpurun_out/levels; mkdir -p $O
rocprofv3-avail list 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
wc -l $O/sq_counters.txt
grep -E "LEVEL|WAIT|IFETCH|CACHE|VMEM|SMEM" $O/sq_counters.txt | tr '\n' ' '; echo
for grp in "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  scripts/r5_pmc.sh lv "$grp" --walkers $W 2>&1 | grep "walker_kernel" | cut -c1-400 | tee -a $O/levels_$W.txt
done
