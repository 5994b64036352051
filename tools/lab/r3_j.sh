#!/bin/bash
# r03 eighth GPU pass: fixed Fq2 row/column sums (28/29-bit) A/B; PLONK with two proofs in flight
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03j
echo "== small G2 MSM patterns, R'-form reduction on"; ZKMI_R29_REDUCE_G2=1 timeout 300 python tools/lab/dbg_g2rc.py bn128 2>&1 | grep -c "True, True, True"; timeout 300 python tools/lab/dbg_g2rc.py bls12381 2>&1 | grep -c "True, True, True"
run() { python bench.py "$@" --no-cpu-baseline --no-napi-wall 2>gpurun_out/r03j/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print(d['value'], d['ms_per_step'], {k: round(v,2) for k,v in s.items() if k.startswith('accum') or k.startswith('reduce')})" || tail -3 gpurun_out/r03j/err.txt; }
for v in 1 0 1 0; do echo -n "bls g2_rowcol29=$v: "; ZKMI_R29_REDUCE_G2=$v run --curve bls12381 --steps 8 --warmup 2; done
for v in 0 1 0 1; do echo -n "bn g2_rowcol29=$v: "; ZKMI_R29_REDUCE_G2=$v run --steps 20 --warmup 3; done
echo "== plonk tests"; timeout 900 python -m pytest tests/test_gpu_plonk.py -x -q -m gpu -k "two_proofs_in_flight or golden_proof or synthetic_plonk_key" 2>&1 | tail -4
for pl in 2 1 2; do echo -n "plonk pipeline=$pl: "; timeout 600 python bench.py --workload plonk --log-n 20 --steps 8 --warmup 2 --pipeline $pl --no-cpu-baseline 2>gpurun_out/r03j/perr.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('latency_ms_single_proof'))" || tail -5 gpurun_out/r03j/perr.txt; done
echo "== parity subsets"; timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "non_default_kernel_variants or resident_tables_special_cases or valid_key_proof_verifies" 2>&1 | tail -4
