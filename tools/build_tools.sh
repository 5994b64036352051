#!/bin/bash
# Builds the measurement executables under tools/bin/ (git-ignored; they travel with gpurun like the library): the field / mixed-addition ceilings of
# the library's own arithmetic and the FETCH_SIZE gather calibration. Cross-compiles for gfx950 without a GPU. Used by tools/collect_profiles.sh.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I snarkjs_amd/csrc -I include"
$H -o tools/bin/fieldbench29 tools/fieldbench29.hip &
$H -DZK_MAD_PLAIN -o tools/bin/fieldbench29_plain tools/fieldbench29.hip &
$H -DZK_MAD_PLAIN -DZK_MAD_ASM -o tools/bin/fieldbench29_asm tools/fieldbench29.hip &
$H -o tools/bin/maddbench29 tools/maddbench29.hip &
$H -o tools/bin/maddbench29_g2 tools/maddbench29_g2.hip &
$H -o tools/bin/gatherbench tools/gatherbench.hip &
wait
ls -la tools/bin
