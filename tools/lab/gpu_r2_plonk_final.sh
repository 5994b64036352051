#!/bin/bash
# final-state PLONK kernel timeline (rocprofv3 --kernel-trace of the bench command) + its summary
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/plonk_trace -o plonk -- python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline > $O/plonk_profiled.json 2> $O/plonk_profiled.err
python tools/plonk_trace_summary.py $O/plonk_trace/plonk_kernel_trace.csv $O/plonk_profiled.json > $O/plonk_summary.md 2> $O/plonk_summary.err
head -20 $O/plonk_summary.md
