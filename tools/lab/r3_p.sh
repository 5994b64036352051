#!/bin/bash
# r03: the per-box decision on a slow-fetch box, final build (exits at once on a healthy box)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03p
ratio=$(python -c "
import ctypes as C
from snarkjs_amd import zkmi
zkmi.init(0); L = zkmi.lib(); a, b = C.c_double(0), C.c_double(0)
L.zkmi_calibrate_code_fetch(C.byref(a), C.byref(b)); print(round(b.value / a.value, 3))" 2>/dev/null | tail -1)
echo "code fetch big/small = $ratio"
if python -c "import sys; sys.exit(0 if float('$ratio') < 0.85 else 1)"; then
run() { python bench.py "$@" --no-cpu-baseline --no-napi-wall 2>gpurun_out/r03p/err.txt | tee gpurun_out/r03p/last.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stages_ms',{}); b=d['box_calibration']; print(d['value'], d['ms_per_step'], 'mask', b.get('compact_code_mask'), {k: round(v,2) for k,v in s.items() if k.startswith('accum') or k.startswith('reduce')})" || tail -3 gpurun_out/r03p/err.txt; }
echo -n "bn auto: "; run --steps 20 --warmup 3; cp gpurun_out/r03p/last.json gpurun_out/r03p/bench_slow_fetch_box_auto.json
echo -n "bn inlined: "; ZKMI_COMPACT_CODE=0 run --steps 20 --warmup 3
echo -n "bls auto: "; run --curve bls12381 --steps 8 --warmup 2; cp gpurun_out/r03p/last.json gpurun_out/r03p/bench_bls12381_slow_fetch_box_auto.json
echo -n "bls inlined: "; ZKMI_COMPACT_CODE=0 run --curve bls12381 --steps 8 --warmup 2
echo -n "plonk auto: "; run --workload plonk --log-n 20 --steps 12 --warmup 4; cp gpurun_out/r03p/last.json gpurun_out/r03p/bench_plonk_slow_fetch_box_auto.json
echo -n "bn auto again: "; run --steps 20 --warmup 3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03p/stp -o p -- python bench.py --workload plonk --log-n 20 --steps 4 --warmup 3 --pipeline 1 --no-cpu-baseline > /dev/null 2>&1
rm -f gpurun_out/r03p/stp/*kernel_trace.csv
python - gpurun_out/r03p/stp/p_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("  ", r["Name"].split("(")[0][-60:], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "valid_key_proof_verifies or synthetic_vs_oracle or resident_tables" 2>&1 | tail -2
fi
