#!/bin/bash
O=gpurun_out/r4_step2; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/exp/lib_stamps.so python scripts/seam_stamps.py 1 256 512 > $O/seam_stamps.txt 2>&1; grep "^C1" $O/seam_stamps.txt
WS="" CFGS="C2 C3 C4" ROUNDS=1 scripts/r4_ab.sh r4_step2/cfg base=exp/lib_base.so t2=exp/lib_t2.so
