#!/bin/bash
# PLONK launch diet (r06): parity first, then the 2^20 line with two proofs in flight and serial, then a kernel trace of the serial run
mkdir -p gpurun_out/r6diet; O=gpurun_out/r6diet
timeout 1500 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q > $O/pytest_plonk.log 2>&1; echo "rc=$?" >> $O/pytest_plonk.log
tail -5 $O/pytest_plonk.log
timeout 600 python -m pytest tests/test_node_boundary.py -m gpu -x -q > $O/pytest_node.log 2>&1; echo "rc=$?" >> $O/pytest_node.log
tail -5 $O/pytest_node.log
for rep in 1 2; do
  timeout 300 python bench.py --workload plonk --steps 16 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs > $O/plonk_two_$rep.json 2>$O/plonk_two_$rep.err
  timeout 300 python bench.py --workload plonk --pipeline 1 --steps 12 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs > $O/plonk_serial_$rep.json 2>$O/plonk_serial_$rep.err
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob("gpurun_out/r6diet/plonk_*.json")):
    try:
        j=json.loads(open(p).read().strip().splitlines()[-1]); print(p.split('/')[-1], j["value"], j.get("latency_ms_single_proof"))
    except Exception as e: print(p, "ERR", str(e)[:80])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o plonk -- python $GRAFT_REPO_ROOT/bench.py --workload plonk --pipeline 1 --steps 6 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/$O/trace_bench.json 2>$GRAFT_REPO_ROOT/$O/trace.err
cd $GRAFT_REPO_ROOT; find $O/trace -name "*kernel_trace.csv" | head -2; ls -la $O/trace 2>/dev/null | head
