"""Print the kernel timeline of the last proof in a rocprofv3 kernel trace CSV (development aid)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_build_abc' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    if 'k_gen_geom' in r['Kernel_Name']: break
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d < 20 and 'msm' not in r['Kernel_Name']: continue
    print(f"{s:10.1f} +{d:9.1f} us q={r['Queue_Id']} s={r['Stream_Id']} grid={r['Grid_Size_X']:>8} vgpr={r['VGPR_Count']:>3} scr={r['Scratch_Size']:>4} {r['Kernel_Name'][12:60]}")
