#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4e; mkdir -p $O
echo "== plain C multiply-adds"; tools/bin/fieldbench29 2>&1 | tee $O/fieldbench29.txt | tail -30
echo "== inline-asm multiply-adds (r03)"; tools/bin/fieldbench29_asm 2>&1 | tee $O/fieldbench29_asm.txt | tail -30
tools/bin/maddbench29 2>&1 | tee $O/maddbench29.txt | tail -12
for f in 1 0 1 0; do ZKMI_POOL_FENCE=$f timeout 400 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk fence', '$f', d['value'], d['ms_per_step'])"; done | tee $O/plonk_fence.txt
python tools/lab/r4_msm_probe.py 2>&1 | grep split
