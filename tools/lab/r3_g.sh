#!/bin/bash
# r03 seventh GPU pass: Fq2 row/column sums on unsaturated limbs (k_msm_rowcol_wave29_g2) against the generic kernel, both curves; the batched-affine
# arithmetic microbenchmark; parity of the non-default variants
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03g
run() { python bench.py "$@" --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print(d['value'], d['ms_per_step'], {k: round(v,2) for k,v in s.items()})"; }
for v in 1 0 1 0; do echo -n "bls g2_rowcol29=$v: "; ZKMI_R29_REDUCE_G2=$v run --curve bls12381 --steps 8 --warmup 2; done
for v in 0 1 0 1; do echo -n "bn g2_rowcol29=$v: "; ZKMI_R29_REDUCE_G2=$v run --steps 20 --warmup 3; done
timeout 120 tools/bin/maddbench29 2>&1 | tail -8
python tools/lab/check_affstep.py gpurun_out/affstep_lane0.txt
cp gpurun_out/affstep_lane0.txt gpurun_out/r03g/ 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "non_default_kernel_variants or full_size_closed_form or valid_key_proof_verifies or msm_resident_tables or synthetic_vs_oracle" 2>&1 | tail -5
