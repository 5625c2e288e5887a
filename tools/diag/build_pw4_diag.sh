#!/bin/bash
# builds tools/diag/pw4_diag_<abl> for each ablation mask given (default: 0 1 2 4 8 16 32 63)
cd "$(dirname "$0")/../.."
for abl in ${@:-0 1 2 4 8 16 32 63}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Wno-inline-asm -Wno-unused-value -DNDEBUG -DUV_PW4_DIAG -DUV_PW4_ABL=$abl \
      -I univid_amd/csrc -I tools/diag tools/diag/pw4_diag.hip -o tools/diag/pw4_diag_$abl || exit 1
done
ls -la tools/diag/
