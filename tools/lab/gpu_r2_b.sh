#!/bin/bash
# round-2 call B: GPU parity suite (incl. 2^24 closed form, replay, cache, pipelining) + bench variants + gather calibration
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_dense_p2.json 2> $O/bench_dense_p2.err
timeout 300 python bench.py --steps 20 --warmup 3 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_dense_p1.json 2> $O/bench_dense_p1.err
timeout 300 python bench.py --steps 20 --warmup 3 --b-zero-every 3 --no-cpu-baseline --no-napi-wall > $O/bench_sparse_p2.json 2> $O/bench_sparse_p2.err
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gather -o g -- tools/bin/gatherbench > $O/gatherbench.txt 2>&1
python - <<'PY'
import json,csv,collections
for t in ("dense_p2","dense_p1","sparse_p2"):
    try:
        d=json.loads(open(f"gpurun_out/r2b/bench_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d.get("latency_ms_single_proof"), d["stages_ms"])
        if "cpu_baseline" in d: print(" cpu", {k:v for k,v in d["cpu_baseline"].items() if k!="reference_wasm"})
        if "wall_through_napi" in d: print(" napi", d["wall_through_napi"])
    except Exception as e: print(t,"ERR",e, open(f"gpurun_out/r2b/bench_{t}.err").read()[-1500:])
try:
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/r2b/pmc_gather/g_counter_collection.csv")):
        if r["Counter_Name"]=="FETCH_SIZE": acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, "FETCH_SIZE KB avg", sum(v)/len(v), "-> x1024 =", sum(v)/len(v)*1024/1e6, "MB")
    print(open("gpurun_out/r2b/gatherbench.txt").read()[-900:])
except Exception as e: print("pmc ERR", e)
PY
