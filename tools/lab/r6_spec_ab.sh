#!/bin/bash
# r06: speculative gather in the accumulation loops (msm29.cuh: ZK_SPEC_GATHER) — product library against the variant built with -DZK_SPEC_GATHER=0
# (ZKMI_BUILD_VARIANT=nospec), one box, interleaved; parity first
O=$GRAFT_REPO_ROOT/gpurun_out/r6spec; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or skew or groth16 or cache or variants" 2>&1 | tail -3) | tee $O/pytest.log
(timeout 400 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q -k "golden or two_proofs or full_size" 2>&1 | tail -3) | tee -a $O/pytest.log
V=$GRAFT_REPO_ROOT/snarkjs_amd/libzkmi_nospec.so
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d.get('stage_ms') or {}
print('$1', d['value'], 'latency', d.get('latency_ms_single_proof'), 'kernel_ms', (d.get('roofline') or {}).get('kernel_ms'), {k: st[k] for k in list(st)[:12]} if isinstance(st, dict) else '')" | tee -a $O/ab.txt; }
for rep in 1 2; do
  for lib in spec nospec; do
    if [ $lib = nospec ]; then export ZKMI_LIB=$V; else unset ZKMI_LIB; fi
    python bench.py --steps 40 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | line "groth16 $lib"
    python bench.py --workload plonk --steps 32 --warmup 3 --no-cpu-baseline --no-napi-wall --no-other-configs 2>/dev/null | line "plonk   $lib"
    python bench.py --curve bls12381 --steps 12 --warmup 2 --no-cpu-baseline --no-napi-wall --no-other-configs 2>/dev/null | line "bls     $lib"
  done
done
unset ZKMI_LIB
