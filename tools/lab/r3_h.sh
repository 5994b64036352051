#!/bin/bash
# r03: diagnose the failing R'-form G2 reduction and the slow BLS12-381 B2 accumulation
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03h
echo "== bn g2=1"; ZKMI_R29_REDUCE_G2=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-napi-wall 2>&1 | tail -12 | cut -c1-600
echo "== bls g2=0 stats"; ZKMI_R29_REDUCE_G2=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h/st -o bls -- python bench.py --curve bls12381 --steps 4 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall 2>&1 | tail -3 | cut -c1-900
f=$(find gpurun_out/r03h/st -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
rm -f gpurun_out/r03h/st/*/*kernel_trace.csv gpurun_out/r03h/st/*kernel_trace.csv
