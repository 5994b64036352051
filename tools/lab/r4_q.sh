#!/bin/bash
# closing collection after k_plonk_t29 became the default: PLONK lines + kernel trace into gpurun_out/r04, smoke, the whole GPU suite
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 600 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 > $O/bench_plonk_2p20.json 2>/dev/null
timeout 600 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --pipeline 1 --no-cpu-baseline > $O/bench_plonk_2p20_serial.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/stats_plonk -o plonk -- python bench.py --workload plonk --log-n 20 --steps 4 --warmup 3 --pipeline 1 --no-cpu-baseline > $O/bench_plonk_under_rocprof.json 2>/dev/null
for f in bench_plonk_2p20 bench_plonk_2p20_serial; do python - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms")
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 560 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
du -sh $O
