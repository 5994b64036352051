#!/bin/bash
# Collect the round's judged evidence on the GPU box (run through gpurun): bench JSON, rocprofv3 kernel stats of the SAME
# command, the two PMC passes (separate runs, --kernel-trace only), PLONK / BLS12-381 / 2^24 side benches.
# usage (repo root):  gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
# plain benches first: the PMC passes leave the device in a profiling clock state for the rest of the session (measured: the
# benches that followed them in one call were 10-20 % slower than the same commands run on their own)
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --workload plonk --log-n 20 --steps 8 --warmup 3 > $O/bench_plonk_2p20.json 2>/dev/null
timeout 600 python bench.py --curve bls12381 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_bls12381_2p20.json 2>/dev/null
timeout 300 python bench.py --workload fflonk --log-n 18 --steps 5 --warmup 1 > $O/bench_fflonk_2p18.json 2>/dev/null
timeout 900 python bench.py --log-n 24 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_bn128_2p24.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_plonk -o plonk -- python bench.py --workload plonk --log-n 20 --steps 3 --warmup 1 > /dev/null 2>&1
cat $O/bench.json
