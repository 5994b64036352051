#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "ntt" 2>&1 | tail -3
for v in 0 1 0 1; do ZKMI_NTT29=$v python tools/lab/r4_ntt_probe.py 2>&1 | grep NTT29; done | tee $O/ntt.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall"
for v in 0 1 0 1; do ZKMI_NTT29=$v $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench NTT29=$v', d['value'], d['ms_per_step'], 'ntt_x6', round(d['stages_ms']['ntt_x6'],3), 'ntt_ms', d['submetrics']['ntt_ms'])"; done | tee -a $O/ntt.txt
for v in 0 1; do ZKMI_NTT29=$v timeout 400 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk NTT29=$v', d['value'], d['ms_per_step'])"; done | tee -a $O/ntt.txt
