#!/bin/bash
# r03 fourth GPU pass: NTT29 with 48-byte work records (A/B vs the 32-bit passes at 2^20 / 2^22 / 2^24), row/column sums with on-demand operand loads
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or golden or closed_form or resident or sharded" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for v in 1 0 1 0; do
  ZKMI_NTT29=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ntt29=$v', d['value'], d['ms_per_step'], d['stages_ms']['ntt_x6'], d['submetrics']['ntt_ms'], d['stages_ms']['reduce_g1'])"
done
for v in 1 0; do
  ZKMI_NTT29=$v timeout 300 python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk ntt29=$v', d['value'], d['ms_per_step'])"
done
for v in 1 0; do
ZKMI_NTT29=$v timeout 300 python bench.py --log-n 24 --steps 3 --warmup 1 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^24 ntt29=$v', d['value'], d['ms_per_step'], d['stages_ms']['ntt_x6'])"
done
timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls', d['value'], d['ms_per_step'], d['stages_ms'])"
