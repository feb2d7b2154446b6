#!/bin/bash
# timing-only ablation builds of the tile kernel (outputs are WRONG by construction)
cd "$(dirname "$0")/../rbvfit_amd/csrc"
mkdir -p ../lib/ablate
for a in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DVP_ABLATE=$a -o ../lib/ablate/lib_ablate$a.so capi.hip &
done
wait
ls -la ../lib/ablate
