#!/bin/bash
# r05: whenever a gpurun call lands on a box whose code-fetch probe reads below 0.9, measure what the probe's border (0.85, zkmi_api.hip: compact_code) should
# select now that the inlined loops are 35-45 % smaller than when it was set (r03): BLS12-381 Groth16 under ZKMI_COMPACT_CODE = 31 (every kernel with called
# products: what a slow-fetch box gets), 14 (G1 accumulation inlined: its loop is 38 KB and fits the instruction cache), 0 (all inlined); PLONK under 31 / 15
# (bit 4: the quotient numerator by the r03 32-bit kernels or by the inlined 29-bit ones, 41-52 KB per part). On a healthy box: one line, nothing measured.
# usage: tools/lab/r5_slow_box_ab.sh out_file
out=${1:-gpurun_out/slow_box_ab.txt}
ratio=$(python - <<'PY'
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
from snarkjs_amd import zkmi
zkmi.init(0)
L = zkmi.lib()
a, b = ctypes.c_double(0), ctypes.c_double(0)
L.zkmi_calibrate_code_fetch(ctypes.byref(a), ctypes.byref(b))
print(round(b.value / a.value, 3))
PY
)
echo "code-fetch ratio of this box: $ratio" | tee -a $out
if python -c "import sys; sys.exit(0 if float('$ratio') < 0.9 else 1)"; then
  for m in 31 14 0 31 14; do
    ZKMI_COMPACT_CODE=$m python bench.py --curve bls12381 --steps 8 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']
print('bls12381 ZKMI_COMPACT_CODE=$m', d['repeats']['proofs_per_s'], 'B2', round(s['accum_B2'],2), 'A', round(s['accum_A'],2), 'C', round(s['accum_C'],2), 'reduce_g1', round(s['reduce_g1'],2))" | tee -a $out
  done
  for m in 31 15 31 15; do
    ZKMI_COMPACT_CODE=$m python bench.py --workload plonk --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk ZKMI_COMPACT_CODE=$m', d['value'], d['latency_ms_single_proof'])" | tee -a $out
  done
fi
