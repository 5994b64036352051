#!/bin/bash
# r04: the 14-limb field in column statements too (libzkmi_colsall.so = -DZK_COLS_ALL) against the shipped build (14-limb plain C), BLS12-381 line, same box
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4n; mkdir -p $O
V=$GRAFT_REPO_ROOT/snarkjs_amd/libzkmi_colsall.so
B="python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], [round(v,3) for v in d.get('accum_kernel_ms',{}).values()], d.get('stages_ms',{}).get('reduce_g1'), d['box_calibration']['compact_code_mask'])"; }
for rep in 1 2; do
$B 2>/dev/null | line "bls shipped(plain14)"
ZKMI_LIB=$V $B 2>/dev/null | line "bls cols14"
done | tee $O/ab.txt
ZKMI_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or closed_form_large or special_cases" 2>&1 | tail -2
