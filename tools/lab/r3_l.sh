#!/bin/bash
# r03: full GPU suite on the final build + clean PLONK A/B (serial against two in flight)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03l
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r03l/pytest.log
for pl in 1 2 1 2; do echo -n "plonk pipeline=$pl: "; timeout 600 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --pipeline $pl --no-cpu-baseline 2>gpurun_out/r03l/perr.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('latency_ms_single_proof'))" || tail -5 gpurun_out/r03l/perr.txt; done
