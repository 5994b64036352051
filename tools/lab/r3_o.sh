#!/bin/bash
# r03: policy for boxes with slow instruction fetch beyond the instruction cache: exits at once on a healthy box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03o
ratio=$(python -c "
import ctypes as C
from snarkjs_amd import zkmi
zkmi.init(0); L = zkmi.lib(); a, b = C.c_double(0), C.c_double(0)
L.zkmi_calibrate_code_fetch(C.byref(a), C.byref(b)); print(round(b.value / a.value, 3))" 2>/dev/null | tail -1)
echo "code fetch big/small = $ratio"
if python -c "import sys; sys.exit(0 if float('$ratio') < 0.85 else 1)"; then
run() { python bench.py "$@" --no-cpu-baseline --no-napi-wall 2>gpurun_out/r03o/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; b=d['box_calibration']; print(d['value'], d['ms_per_step'], 'mask', b.get('compact_code_mask'), {k: round(v,2) for k,v in s.items() if k.startswith('accum') or k.startswith('reduce')})" || tail -3 gpurun_out/r03o/err.txt; }
ZKMI_COMPACT_CODE=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > gpurun_out/r03o/bench_slow_fetch_box_inlined.json 2>/dev/null
ZKMI_COMPACT_CODE=0 timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall > gpurun_out/r03o/bench_bls12381_slow_fetch_box_inlined.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > gpurun_out/r03o/bench_slow_fetch_box_auto.json 2>/dev/null
timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall > gpurun_out/r03o/bench_bls12381_slow_fetch_box_auto.json 2>/dev/null
for m in 0 4 8 12 2; do echo -n "bn compact=$m: "; ZKMI_COMPACT_CODE=$m run --steps 20 --warmup 3; done
for m in 0 4; do echo -n "bn generic-g2-rowcol compact=$m: "; ZKMI_R29_REDUCE_G2=0 ZKMI_COMPACT_CODE=$m run --steps 20 --warmup 3; done
for m in 0 1 2 3 7 15; do echo -n "bls compact=$m: "; ZKMI_COMPACT_CODE=$m run --curve bls12381 --steps 8 --warmup 2; done
for m in 3 7; do echo -n "bls generic-g2-rowcol compact=$m: "; ZKMI_R29_REDUCE_G2=0 ZKMI_COMPACT_CODE=$m run --curve bls12381 --steps 8 --warmup 2; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03o/st -o b -- python bench.py --steps 6 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall > /dev/null 2>&1
ZKMI_COMPACT_CODE=15 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03o/st15 -o b -- python bench.py --steps 6 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall > /dev/null 2>&1
ZKMI_COMPACT_CODE=15 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03o/stb15 -o b -- python bench.py --curve bls12381 --steps 4 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall > /dev/null 2>&1
rm -f gpurun_out/r03o/st*/*kernel_trace.csv
for d in st st15 stb15; do echo "== $d"; python - gpurun_out/r03o/$d/b_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:40]:
    n=r["Name"].split("(")[0][-70:]
    if "rowcol" in n or "accum" in n or "bitsums" in n: print("  ", n, r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
fi
