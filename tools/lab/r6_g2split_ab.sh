#!/bin/bash
# r06: BN254 G2 accumulation with one Fq2 component per lane (k_msm_accum29_g2s) against the LDS-parked layout (ZKMI_G2_SPLIT=0), one box: parity, then lines
O=$GRAFT_REPO_ROOT/gpurun_out/r6g2split; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or skew or groth16 or cache or variants or golden or edge" 2>&1 | tail -3) | tee $O/pytest.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], 'latency', d.get('latency_ms_single_proof'), 'B2 kernel_ms', (d.get('roofline') or {}).get('kernel_ms'))" | tee -a $O/ab.txt; }
for rep in 1 2 3; do
  for m in 1 0; do
    ZKMI_G2_SPLIT=$m python bench.py --steps 40 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "groth16 split=$m"
  done
done
for m in 1 0; do
  ZKMI_G2_SPLIT=$m python bench.py --steps 40 --warmup 3 --pipeline 1 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "groth16 serial split=$m"
  ZKMI_G2_SPLIT=$m python bench.py --coef-dist real --witness mixed --steps 40 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "circuit-shaped split=$m"
done
