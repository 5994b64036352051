#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4f; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall"
run() { env "$@" $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], [round(v,3) for v in d['accum_kernel_ms'].values()], round(d['stages_ms']['reduce_g1'],3), round(d['stages_ms']['ntt_x6'],3))"; }
for rep in 1 2; do
run ZKMI_X=0
run ZKMI_ACC29_BLOCK=64
run ZKMI_ACC29_BLOCK=128
run ZKMI_AUX_RC_SUMS=256
run ZKMI_AUX_RC_SUMS=1024
run ZKMI_AUX_PRIO=0
done | tee $O/ab.txt
