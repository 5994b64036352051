#!/bin/bash
# r06: where the cycles of the split G2 accumulation go (rocprofv3 --pmc SQ_*, three passes of four counters, kernel-trace only; as tools/lab/r4_c.sh did for the LDS layout)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6sq; mkdir -p $O
for cv in bn128 bls12381; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${cv}_$tag -o p -- python bench.py --curve $cv --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall --no-other-configs > $O/bench_${cv}_$tag.json 2>/dev/null
done
done
python - <<'PY' | tee gpurun_out/r6sq/summary.txt
import csv,glob,collections,json
for cv in ("bn128","bls12381"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r6sq/pmc_{cv}_*/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_msm_accum29" in k:
                nm="G2 "+k.split("(")[0].split("::")[-1][:40] if "accum29_g2" in k else "G1"
                agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ms={}
    try:
        d=json.loads(open(f"gpurun_out/r6sq/bench_{cv}_SQ_WAVE_CYCLES.json").read().strip().splitlines()[-1]); ms=d.get("accum_kernel_ms",{})
    except Exception as e: pass
    print("==",cv,"accumulation ms under the counters:",{k.split('(')[-1][:-1]:round(v,3) for k,v in ms.items()})
    for nm,dd in agg.items():
        print(" ",nm,{k: round(sum(v)/len(v)) for k,v in sorted(dd.items())})
PY
find $O -name "*kernel_trace.csv" -delete; du -sh $O
