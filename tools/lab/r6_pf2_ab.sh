#!/bin/bash
# r06: G1 accumulation with TWO gathers in flight (msm29.cuh: ZK_PREFETCH2) — the product against a ZKMI_BUILD_VARIANT=pf2 ZKMI_EXTRA_FLAGS=-DZK_PREFETCH2=1 build, one box
O=$GRAFT_REPO_ROOT/gpurun_out/r6pf2; mkdir -p $O
V=$GRAFT_REPO_ROOT/snarkjs_amd/libzkmi_pf2.so
(ZKMI_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm_resident or msm_golden or skew or groth16_golden or synthetic_vs_oracle or special" 2>&1 | tail -3) | tee $O/pytest.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d.get('accum_kernel_ms') or {}
print('$1', d['value'], 'latency', d.get('latency_ms_single_proof'), 'dominant kernel_ms', (d.get('roofline') or {}).get('kernel_ms'), {k.split('(')[-1][:-1]: round(v, 3) for k, v in a.items()})" | tee -a $O/ab.txt; }
for rep in 1 2; do
  for lib in pf1 pf2; do
    if [ $lib = pf2 ]; then export ZKMI_LIB=$V; else unset ZKMI_LIB; fi
    python bench.py --steps 40 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "groth16 $lib"
    python bench.py --workload plonk --steps 32 --warmup 3 --no-cpu-baseline --no-napi-wall --no-other-configs 2>/dev/null | line "plonk   $lib"
  done
done
for lib in pf1 pf2; do
  if [ $lib = pf2 ]; then export ZKMI_LIB=$V; else unset ZKMI_LIB; fi
  python bench.py --curve bls12381 --steps 12 --warmup 2 --no-cpu-baseline --no-napi-wall --no-other-configs 2>/dev/null | line "bls     $lib"
  python bench.py --coef-dist real --witness mixed --steps 40 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "real    $lib"
done
unset ZKMI_LIB
