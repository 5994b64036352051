#!/bin/bash
# r03: compact-code kernels (products called instead of inlined) against the inlined ones, per kernel class, on whatever box this lands on
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03n
run() { python bench.py "$@" --no-cpu-baseline --no-napi-wall 2>gpurun_out/r03n/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; b=d['box_calibration']; print(d['value'], d['ms_per_step'], 'mask', b.get('compact_code_mask'), 'fetch', b.get('code_fetch',{}).get('big_over_small'), {k: round(v,2) for k,v in s.items() if k.startswith('accum') or k.startswith('reduce')})" || tail -3 gpurun_out/r03n/err.txt; }
echo -n "bn auto: "; run --steps 20 --warmup 3
for m in 0 2 4 8 15 0 15; do echo -n "bn compact=$m: "; ZKMI_COMPACT_CODE=$m run --steps 20 --warmup 3; done
echo -n "bls auto: "; run --curve bls12381 --steps 8 --warmup 2
for m in 0 1 2 3 15 0 15; do echo -n "bls compact=$m: "; ZKMI_COMPACT_CODE=$m run --curve bls12381 --steps 8 --warmup 2; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "non_default_kernel_variants and (compact or inlined)" 2>&1 | tail -3
