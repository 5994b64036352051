#!/bin/bash
# r04: 14-limb G2 accumulation with the accumulator in registers (one wave per SIMD) against the LDS-parked packed Jacobian form; peer copies on their own stream
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_peer.py -m gpu -x -q -k "golden_cases or closed_form_large or special_cases or groth16_golden_proof or peer or reset or key_curve" 2>&1 | tail -4
B="python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall"
for v in 1 0 1 0; do ZKMI_ACC29_G2_REG=$v $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bls G2_REG=$v', d['value'], d['ms_per_step'], {k.split('(')[-1][:-1]: round(v,3) for k,v in d['accum_kernel_ms'].items()}, round(d['stages_ms']['reduce_g1'],3))"; done | tee $O/bls_reg.txt
timeout 300 node tests/js/native_gpu.js 2>&1 | tail -12
