"""Print the kernel timeline of one proof from a rocprofv3 --kernel-trace CSV (development aid).
usage: python tools/timeline.py <kernel_trace.csv> [proof_index]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_build_abc' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a = idx[k]
# the proof's aux-stream sorts start before k_build_abc: back up to the previous proof's last kernel
t_prev_end = max(int(r['End_Timestamp']) for r in rows[idx[k - 1]:a] if 'rsort' not in r['Kernel_Name'] and 'scan' not in r['Kernel_Name'] and 'fill' not in r['Kernel_Name'] and 'classif' not in r['Kernel_Name'] and 'class_scan' not in r['Kernel_Name'] and 'assign' not in r['Kernel_Name']) if k else 0
b = idx[k + 1] if k + 1 < len(idx) else len(rows)
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[idx[k - 1] if k else 0:b]:
    if int(r['Start_Timestamp']) < t_prev_end: continue
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('zkmi::', '')[:48]
    print(f"{s:9.1f} {e - s:8.1f} q{r['Queue_Id']} {nm}")
