#!/bin/bash
# r03 fifth GPU pass: BLS12-381 G2 accumulation with a packed Jacobian accumulator (2 waves per SIMD), opt-in NTT29 parity test, BLS A/B of the G1 row/col sums
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bls or ntt29 or resident or golden" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for v in 1 0 1 0; do
  ZKMI_R29_REDUCE=$v timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls r29reduce=$v', d['value'], d['ms_per_step'], d['stages_ms'], d['accum_kernel_ms'])"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bn', d['value'], d['ms_per_step'], d['stages_ms'], d['box_calibration'])"
