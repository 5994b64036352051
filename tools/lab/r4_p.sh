#!/bin/bash
# k_plonk_t on 29-bit limbs (ZKMI_PLONK_T29 = 1 inlined, 2 calls) against the 32-bit kernels (0): parity tests, then PLONK 2^20 A/B, then kernel stats
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4p; mkdir -p $O
ZKMI_PLONK_T29=1 timeout 420 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q -k "plonk and not fflonk" 2>&1 | tail -4 | tee $O/pytest_t29_1.txt
ZKMI_PLONK_T29=2 timeout 200 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q -k "golden_proof and not fflonk or synthetic_plonk" 2>&1 | tail -3 | tee $O/pytest_t29_2.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('stages_ms'))"; }
P="python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline"
for v in 0 1 2 0 1; do ZKMI_PLONK_T29=$v $P 2>/dev/null | line "plonk t29=$v"; done | tee $O/ab.txt
ZKMI_PLONK_T29=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t29 -- $P > $O/bench_prof.json 2>$O/prof.err
python - <<'PY' | tee gpurun_out/r4p/kernels.txt
import csv,glob
for f in glob.glob("gpurun_out/r4p/prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:40]:
        n=r["Name"]
        if "plonk_t" in n or "ntt" in n or "accum" in n: print(n[:90], r["Calls"], r["AverageNs"])
PY
