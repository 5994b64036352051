#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab5; mkdir -p $O
run() { tag=$1; shift; env "$@" > $O/$tag.json 2> $O/$tag.err; }
run bls_cap X=1 timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall
run bls_nocap ZKMI_AUX_RC_SUMS=0 timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/plonk_trace -o plonk -- python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline > $O/plonk_profiled.json 2> $O/plonk_profiled.err
python tools/plonk_trace_summary.py $O/plonk_trace/plonk_kernel_trace.csv $O/plonk_profiled.json > $O/plonk_summary.md 2> $O/plonk_summary.err
python - <<'PY'
import json
for t in ("bls_cap","bls_nocap"):
    try:
        d=json.loads(open(f"gpurun_out/ab5/{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], {k:round(v,2) for k,v in d["stages_ms"].items() if k.startswith("accum") or k.startswith("reduce")})
    except Exception as e: print(t,"ERR",e)
PY
sed -n 3,16p $O/plonk_summary.md; sed -n 17,30p $O/plonk_summary.md
