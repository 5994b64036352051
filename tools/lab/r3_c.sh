#!/bin/bash
# r03 third GPU pass: NTT on 9 x 29-bit limbs (ntt29.cuh) — parity suite, A/B against the 32-bit passes, PLONK profile, BLS PLONK fixture
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
for v in 1 0 1 0; do
  ZKMI_NTT29=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ntt29=$v', d['value'], d['ms_per_step'], d['stages_ms']['ntt_x6'], d['submetrics']['ntt_ms'])"
done
for v in 1 0; do
  ZKMI_NTT29=$v timeout 300 python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk ntt29=$v', d['value'], d['ms_per_step'])"
done
ZKMI_NTT29=1 timeout 300 python bench.py --log-n 24 --steps 3 --warmup 1 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^24 ntt29=1', d['value'], d['ms_per_step'], d['stages_ms']['ntt_x6'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_plonk -o plonk -- python bench.py --workload plonk --log-n 20 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_plonk_prof.json 2>/dev/null
python - <<'PY'
import glob,csv
for f in glob.glob("gpurun_out/r03c/stats_plonk/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:30]: print(r["Name"].split("(")[0].replace("void ","")[:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["TotalDurationNs"])/1e6,2))
PY
