#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03k; mkdir -p $O
for pl in 2 1; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr$pl -o p -- python bench.py --workload plonk --log-n 20 --steps 6 --warmup 4 --pipeline $pl --no-cpu-baseline > $O/bench_p$pl.json 2>$O/err$pl.txt
tail -1 $O/bench_p$pl.json | cut -c1-200
f=$(find $O/tr$pl -name "*kernel_trace.csv" | head -1); python tools/lab/plonk_overlap.py $f | tee $O/overlap_p$pl.txt | head -60
done
du -sh $O
