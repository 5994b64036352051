#!/bin/bash
# G1 buckets kept in R'-form + row/column sums on 29-bit limbs (k_msm_rowcol_wave29): parity, then A/B against ZKMI_R29_REDUCE=0 on one box
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "not 24" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_plonk.py -q -x -m gpu -k "golden or verif or seeded" 2>&1 | tail -3
for v in 1 0 1 0; do
  for pipe in 1 2; do
  ZKMI_R29_REDUCE=$v timeout 600 python bench.py --steps 16 --warmup 3 --pipeline $pipe --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r29reduce=$v pipeline=$pipe', d['value'], d['ms_per_step'], {k: round(v,2) for k,v in d['stages_ms'].items()})"
  done
done
for v in 1 0; do
  ZKMI_R29_REDUCE=$v timeout 600 python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk r29reduce=$v', d['value'], d['ms_per_step'])"
done
