#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_node_boundary.py -m gpu -x -q -k "group_fft or addon_against or replay" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 600 python tools/gfft_probe.py > $O/gfft_probe.txt 2>&1; cat $O/gfft_probe.txt
