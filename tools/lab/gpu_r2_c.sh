#!/bin/bash
# round-2 call C: PLONK evidence (kernel trace of the bench + its own wall), counter list, NTT 2^22 fresh vs after PLONK
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translat|TCC_EA0_RDREQ|TCC_EA0_RD_|TCC_REQ|FETCH_SIZE|TCP_TCC_READ" | head -60 > $O/counters.txt
timeout 300 python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline > $O/plonk_plain.json 2> $O/plonk_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/plonk_trace -o plonk -- python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline > $O/plonk_profiled.json 2> $O/plonk_profiled.err
python tools/plonk_trace_summary.py $O/plonk_trace/plonk_kernel_trace.csv $O/plonk_profiled.json > $O/plonk_summary.md 2> $O/plonk_summary.err
cat $O/plonk_plain.json | cut -c1-400; head -40 $O/plonk_summary.md; cat $O/counters.txt | head -40
