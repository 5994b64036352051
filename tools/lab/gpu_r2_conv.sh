#!/bin/bash
# round-2 GPU check of the ceremony-side conversions, the async addon calls and the extended replay
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "conversions or group_fft" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_node_boundary.py -q -x -m gpu 2>&1 | tail -15
timeout 300 python tools/gfft_probe.py 2>&1 | tail -12 | tee gpurun_out/gconv_probe.txt
