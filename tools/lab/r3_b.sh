#!/bin/bash
# r03 second GPU pass: full GPU suite (Node host tests: typed pipeline / shard API, makeProver, shard processes; split sharded proof),
# BLS12-381 kernel trace (where does reduce_g1 go), FETCH_SIZE pass of the BN254 proof (16-byte list reads), default bench with the N-API leg
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bls -o bls -- python bench.py --curve bls12381 --steps 6 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_bls_prof.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall > /dev/null 2>&1
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json,glob,csv
d=json.loads(open("gpurun_out/r03b/bench.json").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d.get("wall_through_napi"), d.get("cpu_baseline",{}).get("value"))
for f in glob.glob("gpurun_out/r03b/stats_bls/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:22]: print(r["Name"].split("(")[0].replace("void ","")[:80], r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["TotalDurationNs"])/1e6,2))
PY
find $O -name "*.csv" | head; du -sh $O
