#!/bin/bash
# r06: 2^24 after the staged level-1 scatter was limited to shapes with runs of >= 8 pairs and the chunk kernels got their full grid back where every partition is chunked
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6p24; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size_closed_form or skew or scalar_widths or msm_closed_form_large or sharded" 2>&1 | tail -3) | tee $O/pytest.log
timeout 900 python bench.py --log-n 24 --steps 3 --warmup 1 --no-cpu-baseline --no-napi-wall > $O/bench_bn128_2p24.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p24 -o p24 -- python bench.py --log-n 24 --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_p24_under_rocprof.json 2>/dev/null
find $O -name "*kernel_trace.csv" -delete
python - <<'PY'
import json,csv,glob
d=json.loads(open("gpurun_out/r6p24/bench_bn128_2p24.json").read().strip().splitlines()[-1]); print("2^24", d["value"], d["repeats"]["proofs_per_s"], d["latency_ms_single_proof"])
f=glob.glob("gpurun_out/r6p24/stats_p24/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "rsort" in r["Name"] or "msm_scan" in r["Name"] or "accum29" in r["Name"]: print(r["Name"].split("(")[0][:60].ljust(60), r["Calls"].rjust(4), "%10.1f us" % (float(r["AverageNs"])/1e3))
PY
