#!/bin/bash
# FFLONK 2^18: kernel trace of serial proofs
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4s; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/stats -o fflonk -- python bench.py --workload fflonk --log-n 18 --steps 4 --warmup 2 --pipeline 1 --no-cpu-baseline > $O/bench.json 2>$O/err.txt
python - <<'PY'
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/r4s/stats/fflonk_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=sorted(d.items(), key=lambda kv:-sum(kv[1]))
T=sum(sum(v) for v in d.values())
for k,v in tot[:45]: print(f"{k:72s} n={len(v):5d} tot={sum(v)/1e3:8.2f}ms avg={sum(v)/len(v):8.1f}us")
print("total", T/1e3)
PY
tail -c 600 $O/bench.json
rm -f $O/stats/*kernel_trace.csv
