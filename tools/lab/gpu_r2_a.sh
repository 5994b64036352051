#!/bin/bash
# round-2 call A: GPU parity suite + A/B of the bucket-reduction rewrite (kernel stats under rocprofv3)
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err
ZKMI_ROWCOL_WAVE=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_old.json 2> $O/bench_old.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_new -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import json,csv
for t in ("new","old"):
    try:
        d=json.loads(open(f"gpurun_out/r2a/bench_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["stages_ms"])
    except Exception as e: print(t,"ERR",e)
rows=list(csv.DictReader(open("gpurun_out/r2a/stats_new/bench_kernel_stats.csv")))
for r in rows[:45]:
    if "precompute" in r["Name"] or "geometric" in r["Name"]: continue
    print(f"{r['Name'].split('(')[0][:80]:80s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
