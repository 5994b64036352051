#!/bin/bash
# window size of the resident tables: c = 20 (13 digits, 2^19 buckets) vs 17 (15 digits, 2^16 buckets) vs 16 (16 digits, 2^15 buckets)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab10
for c in 20 17 16 20; do
  for pipe in 1 2; do
  ZKMI_PRECOMP=$c timeout 600 python bench.py --steps 12 --warmup 3 --pipeline $pipe --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c=$c pipeline=$pipe', d['value'], d['ms_per_step'], {k: round(v,2) for k,v in d['stages_ms'].items()})"
  done
done
