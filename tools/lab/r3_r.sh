#!/bin/bash
# r03: k_plonk_t with called products (mask bit 4) against the inlined kernel
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03r
ZKMI_COMPACT_CODE=16 timeout 900 python -m pytest tests/test_gpu_plonk.py -x -q -m gpu -k "golden_proof or compute_z_and_t or synthetic_plonk_key or two_proofs" 2>&1 | tail -2
for m in 0 16 0 16; do echo -n "plonk compact=$m: "; ZKMI_COMPACT_CODE=$m timeout 600 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('latency_ms_single_proof'), d['box_calibration'].get('code_fetch',{}).get('big_over_small'))"; done
for m in 0 16; do ZKMI_COMPACT_CODE=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03r/st$m -o p -- python bench.py --workload plonk --log-n 20 --steps 4 --warmup 3 --pipeline 1 --no-cpu-baseline > /dev/null 2>&1; rm -f gpurun_out/r03r/st$m/*kernel_trace.csv
python - gpurun_out/r03r/st$m/p_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_plonk_t" in r["Name"]: print("  ", r["Name"].split("(")[0][-46:], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
