#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4d; mkdir -p $O
for sp in 1 2 4; do for c in 15 16 17; do ZKMI_MSM_SPLIT=$sp ZKMI_MSM_SPLIT_C=$c python tools/lab/r4_msm_probe.py 2>&1 | grep split; done; done > $O/msm_sweep.txt
cat $O/msm_sweep.txt
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall"
for v in 0 128 0 128; do ZKMI_ACC29_G2_BLOCK=$v $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g2block', '$v', d['value'], d['ms_per_step'], d['accum_kernel_ms'])"; done | tee $O/g2block.txt
timeout 400 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk', d['value'], d['ms_per_step'])" | tee $O/plonk.txt
