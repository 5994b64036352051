#!/bin/bash
# r04: multiply-adds as one asm statement per COLUMN (9-limb fields) against plain C (libzkmi_plain.so = -DZK_MAD_PLAIN), same box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or closed_form_large or special_cases or n1024 or ntt29" 2>&1 | tail -3
PL=$GRAFT_REPO_ROOT/snarkjs_amd/libzkmi_plain.so
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], [round(v,3) for v in d.get('accum_kernel_ms',{}).values()], d.get('stages_ms',{}).get('reduce_g1'), d.get('submetrics',{}).get('g1_msm_ms'))"; }
for rep in 1 2; do
$B 2>/dev/null | line "groth16 cols"
ZKMI_LIB=$PL $B 2>/dev/null | line "groth16 plain"
done | tee $O/ab.txt
for rep in 1 2; do
python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | line "plonk cols"
ZKMI_LIB=$PL python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | line "plonk plain"
done | tee -a $O/ab.txt
python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall 2>/dev/null | line "bls cols" | tee -a $O/ab.txt
ZKMI_LIB=$PL python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall 2>/dev/null | line "bls plain" | tee -a $O/ab.txt
tools/bin/fieldbench29 2>&1 | grep -A4 "bn254_f\|bls12381_fr" | grep -v "^--" | tee $O/fieldbench29_cols.txt
