#!/bin/bash
# standalone kernel durations of one table MSM (sort kernels not overlapped with anything)
O=$GRAFT_REPO_ROOT/gpurun_out/r6sortprobe; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o probe -- python $GRAFT_REPO_ROOT/tools/lab/r6_sort_probe.py > $O/probe.txt 2>$O/probe.err
cd $GRAFT_REPO_ROOT
cat $O/probe.txt
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(r["Name"].split("(")[0][:70].ljust(70), r["Calls"].rjust(5), "%9.1f"%(float(r["AverageNs"])/1e3), "%9.1f"%(float(r["MinNs"])/1e3))
PY
find $O/t -name "*kernel_trace.csv" -delete
