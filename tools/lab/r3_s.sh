#!/bin/bash
# r03: the compact 14-limb kernels after the smaller call interface (their speed does not depend on the box: the code fits the cache)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03s
run() { python bench.py "$@" --no-cpu-baseline --no-napi-wall 2>gpurun_out/r03s/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; b=d['box_calibration']; print(d['value'], d['ms_per_step'], 'mask', b.get('compact_code_mask'), 'fetch', b.get('code_fetch',{}).get('big_over_small'), {k: round(v,2) for k,v in s.items() if k.startswith('accum') or k.startswith('reduce')})" || tail -3 gpurun_out/r03s/err.txt; }
for m in 31 0 31; do echo -n "bls compact=$m: "; ZKMI_COMPACT_CODE=$m run --curve bls12381 --steps 8 --warmup 2; done
ZKMI_COMPACT_CODE=31 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(valid_key_proof_verifies or synthetic_vs_oracle or resident_tables or full_size_closed_form) and bls" 2>&1 | tail -2
