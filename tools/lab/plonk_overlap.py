"""r03: what the GPU does while two PLONK proofs are in flight (rocprofv3 --kernel-trace CSV): steady-state window = the last 60 % of the
kernel activity; union busy time, busy time per stream, the largest idle gaps (with the kernels around them), and the average duration of the
big kernels (to compare with the serial trace)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkmi::", "")) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
# steady state: skip set-up (key generation, table build): start at the first k_plonk_gather after 40 % of the trace
gathers = [e[0] for e in ev if e[3].startswith("k_plonk_gather")]
print("proofs in trace (k_plonk_gather launches):", len(gathers))
lo = gathers[len(gathers) // 2] if gathers else t0 + (t1 - t0) * 2 // 5
win = [e for e in ev if e[0] >= lo]
span = max(e[1] for e in win) - lo
def union(evs):
    tot, cur_s, cur_e = 0, None, None
    for s, e, *_ in sorted(evs):
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    return tot + (cur_e - cur_s if cur_e is not None else 0)
nproofs = sum(1 for g in gathers if g >= lo)
print(f"window: {span/1e6:.2f} ms, proofs started in it: {nproofs}, union busy {union(win)/1e6:.2f} ms ({union(win)/span:.1%}), sum of durations {sum(e[1]-e[0] for e in win)/1e6:.2f} ms")
bys = collections.defaultdict(list)
for e in win: bys[e[2]].append(e)
for s, evs in sorted(bys.items()): print(f"  stream {s}: launches {len(evs)}, busy {union(evs)/1e6:.2f} ms")
# idle gaps of the whole GPU
gaps, cur_e, last = [], None, None
for s, e, st, nm in sorted(win):
    if cur_e is not None and s > cur_e: gaps.append((s - cur_e, last, nm))
    if cur_e is None or e > cur_e: cur_e, last = e, nm
print("idle total %.2f ms; largest gaps (us, kernel before -> kernel after):" % (sum(g[0] for g in gaps) / 1e6))
for g in sorted(gaps, reverse=True)[:12]: print("   %7.1f  %s -> %s" % (g[0] / 1e3, g[1][:50], g[2][:50]))
agg = collections.defaultdict(list)
for s, e, st, nm in win: agg[nm].append(e - s)
print("kernels by total time in the window:")
for nm, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]: print("   %-60s n=%4d avg %8.1f us total %8.2f ms" % (nm[:60], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
