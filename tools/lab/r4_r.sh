#!/bin/bash
# PLONK lines with the rebuilt library (k_plonk_t29 the default) + the 32-bit kernels on the same box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 200 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 > $O/bench_plonk_2p20.json 2>/dev/null
timeout 100 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --pipeline 1 --no-cpu-baseline > $O/bench_plonk_2p20_serial.json 2>/dev/null
ZKMI_PLONK_T29=0 timeout 100 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_plonk_2p20_t32.json 2>/dev/null
timeout 100 rocprofv3 --kernel-trace --output-format csv -d $O/stats_plonk -o plonk -- python bench.py --workload plonk --log-n 20 --steps 4 --warmup 3 --pipeline 1 --no-cpu-baseline > $O/bench_plonk_under_rocprof.json 2>/dev/null
for f in bench_plonk_2p20 bench_plonk_2p20_serial bench_plonk_2p20_t32; do python - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms")
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
grep -c k_plonk_t29 $O/stats_plonk/plonk_kernel_trace.csv
