"""Per-proof summary of a rocprofv3 --kernel-trace CSV of `bench.py --workload plonk` (the kernel time of a proof
must fit inside its wall time). Proofs are delimited by k_plonk_gather launches; for every proof: span (first kernel start -> last
kernel end), busy time (union of kernel intervals), sum of kernel durations, launches, copy/fill launches; then the per-kernel table of
the median proof.  usage: python tools/plonk_trace_summary.py <kernel_trace.csv> [bench.json] > profiles/rNN_plonk_kernel_summary.md"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkmi::", "")
starts = [i for i, r in enumerate(rows) if "k_plonk_gather" in r["Kernel_Name"]]
proofs = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    seg = rows[a:b]
    if b == len(rows):                       # last proof: cut at the first gap > 20 ms (teardown / next phase)
        for k in range(1, len(seg)):
            if int(seg[k]["Start_Timestamp"]) - int(seg[k - 1]["End_Timestamp"]) > 20_000_000:
                seg = seg[:k]
                break
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    proofs.append({"span": (max(e for _, e in iv) - iv[0][0]) / 1e6, "busy": busy / 1e6, "sum": sum(e - s for s, e in iv) / 1e6, "n": len(seg),
                   "copies": sum(1 for r in seg if "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]), "seg": seg})
print("# PLONK 2^20: per-proof kernel timeline (rocprofv3 --kernel-trace of `python bench.py --workload plonk --log-n 20`)\n")
if len(sys.argv) > 2:
    try:
        d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
        print(f"bench line of the SAME (profiled) run: {d['value']} proofs/s, ms_per_step {d['ms_per_step']}\n")
    except Exception as e:
        print(f"(bench line unreadable: {e})\n")
print("| proof | span ms (first kernel start -> last kernel end) | busy ms (union) | sum of kernel durations ms | launches | copy/fill launches |\n|---|---|---|---|---|---|")
for i, p in enumerate(proofs):
    print(f"| {i} | {p['span']:.2f} | {p['busy']:.2f} | {p['sum']:.2f} | {p['n']} | {p['copies']} |")
med = sorted(proofs, key=lambda p: p["span"])[len(proofs) // 2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in med["seg"]:
    k = name(r)
    acc[k][0] += 1
    acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"\nMedian proof (span {med['span']:.2f} ms, busy {med['busy']:.2f} ms): kernels by total time\n\n| kernel | launches | total us | avg us |\n|---|---|---|---|")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"| `{k[:80]}` | {c} | {t:.1f} | {t / c:.1f} |")
