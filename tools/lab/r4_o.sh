#!/bin/bash
# r04 closing collection after the 14-limb field moved to column statements: the lines and kernel statistics that changed, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall > $O/bench_bls12381_2p20.json 2>/dev/null
rm -rf $O/stats_bls; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bls -o bls -- python bench.py --curve bls12381 --steps 6 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_bls_under_rocprof.json 2>/dev/null
rm -f $O/stats_bls/*kernel_trace.csv
{ echo "== the shipped build: one asm statement per COLUMN of the product scanning (every field)"; tools/bin/fieldbench29;
  echo "== -DZK_MAD_PLAIN: multiply-adds in plain C everywhere"; tools/bin/fieldbench29_plain;
  echo "== -DZK_MAD_PLAIN -DZK_MAD_ASM: one inline-asm statement per multiply-add (the r03 build: one s_nop behind each)"; tools/bin/fieldbench29_asm; tools/bin/maddbench29; } > $O/fieldbench29.txt 2>&1
for f in bench bench_bls12381_2p20; do python - "$O/$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms", d["int_alu"]["valu_issue"]["frac"])
PY
done
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
