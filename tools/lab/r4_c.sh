#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|GRBM_[A-Z_]*\|TCP_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/counters.txt; wc -w $O/counters.txt
for sp in 1 2 4; do for c in 15 16 17; do ZKMI_MSM_SPLIT=$sp ZKMI_MSM_SPLIT_C=$c python tools/lab/r4_msm_probe.py 2>&1 | grep split; done; done
ZKMI_MSM_SPLIT=4 ZKMI_MSM_SPLIT_C=16 rocprofv3 --kernel-trace --output-format csv -d $O/tr4 -o t -- python tools/lab/r4_msm_probe.py > /dev/null 2>&1
ZKMI_MSM_SPLIT=1 ZKMI_MSM_SPLIT_C=16 rocprofv3 --kernel-trace --output-format csv -d $O/tr1 -o t -- python tools/lab/r4_msm_probe.py > /dev/null 2>&1
python - <<'PY'
import csv,glob
for tag in ("tr1","tr4"):
    f=glob.glob(f"gpurun_out/r4c/{tag}/**/*kernel_trace.csv",recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    # last MSM: find last k_msm_infmask launch
    idx=[i for i,r in enumerate(rows) if "k_msm_infmask" in r["Kernel_Name"]][-1]
    t0=int(rows[idx]["Start_Timestamp"])
    print("==",tag)
    for r in rows[idx:]:
        nm=r["Kernel_Name"].split("(")[0].replace("void zkmi::","").replace("zkmi::","")[:48]
        print(f"  {nm:50s} start {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  stream {r.get('Stream_Id','?')}")
PY
# where do the cycles of the G2 accumulation go: SQ counters (own pass, kernel-trace only)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r4c/pmc_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_msm_accum29" in k:
            nm="G2" if "accum29_g2" in k else "G1"
            agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
for nm,d in agg.items():
    print(nm, {k: round(sum(v)/len(v)) for k,v in sorted(d.items())})
PY
rm -rf $O/pmc_*/**/*kernel_trace.csv $O/tr*/ 2>/dev/null; du -sh $O
