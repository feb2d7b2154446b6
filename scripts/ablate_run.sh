#!/bin/bash
echo "baseline"; bash scripts/wsweep.sh 128 512 8192
for a in 1 2 3 4; do echo "ablate $a (1=no cold path, 2=1-tap LSF, 3=no exp, 4=no line loop)"; RBVFIT_AMD_LIB=$PWD/rbvfit_amd/lib/ablate/lib_ablate$a.so bash scripts/wsweep.sh 128 512 8192; done
