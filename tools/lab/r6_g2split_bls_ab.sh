#!/bin/bash
# r06: BLS12-381 G2 accumulation with one Fq2 component per lane (XYZZ in registers, no spills) against the packed Jacobian in LDS (the default), one box
O=$GRAFT_REPO_ROOT/gpurun_out/r6g2bls; mkdir -p $O
(ZKMI_G2_SPLIT_BLS=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bls12381 and (msm or skew or groth16 or cache or golden or edge or synthetic or valid)" 2>&1 | tail -3) | tee $O/pytest_split.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], 'latency', d.get('latency_ms_single_proof'), 'B2 kernel_ms', (d.get('roofline') or {}).get('kernel_ms'), 'compact_code_mask', (d.get('box_calibration') or {}).get('compact_code_mask'))" | tee -a $O/ab.txt; }
for rep in 1 2; do
  for m in 1 0; do
    ZKMI_G2_SPLIT_BLS=$m python bench.py --curve bls12381 --steps 16 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "bls12381 split=$m"
  done
done
# what a slow-fetch box would run: the Compact instantiations (called products) for the LDS kernel against the split kernel, which has no Compact twin
ZKMI_COMPACT_CODE=14 ZKMI_G2_SPLIT_BLS=0 python bench.py --curve bls12381 --steps 12 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "bls12381 compact=14 split=0"
ZKMI_COMPACT_CODE=14 ZKMI_G2_SPLIT_BLS=1 python bench.py --curve bls12381 --steps 12 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "bls12381 compact=14 split=1"
