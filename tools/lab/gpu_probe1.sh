#!/bin/bash
# first GPU probe: environment + instruction-rate calibration
mkdir -p gpurun_out
{
echo "== env"; which node && node --version; which rocprofv3; rocminfo | grep -E "Marketing|gfx" | head -4; nproc; free -g | head -2
echo "== ubench_int"; ./tools/ubench_int
for v in 1 2 3; do echo "== fieldbench v$v"; ./tools/fieldbench_v$v; done
} > gpurun_out/probe1.log 2>&1
tail -5 gpurun_out/probe1.log
