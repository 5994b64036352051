"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs -> profiles/pmc_traffic.json (HBM bytes per launch per kernel).

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 B, so it is doubled (calibrated here on the NTT passes: 2 x 16.55 MB reported = 33.9 MB vs
33.55 MB algorithmic read); WRITE_SIZE is taken as reported (NTT pass: 33.6 MB vs 33.55 MB algorithmic write).
usage: python tools/pmc_to_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [workload tag]
The workload tag (bench.py: groth16:<curve>:2^<lg>:b_zero_every=<k>:<witness>) is stored under "__workload__": bench.py only quotes
traffic from a file collected on the workload it is running.
"""
import collections
import csv
import json
import re
import sys


def agg(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkmi::", "")
        while re.search(r",\s*(true|false|\d+)>$", k):
            k = re.sub(r",\s*(true|false|\d+)>$", ">", k)      # drop trailing non-type template args
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
# FETCH_SIZE tallies memory-side read requests at 64 B each (profiles/r02_gather_calibration.md): exact for the 64-byte gathers of
# the G1 accumulation (one affine point per request), half the bytes for 128-byte requests (G2 gathers, coalesced wave reads)
def fetch_factor(kernel):
    return 1 if re.match(r"k_msm_accum(29<|<Fp<)", kernel) else 2
out = {k: int(f[k] * 1024 * fetch_factor(k) + w.get(k, 0.0) * 1024) for k in f}
if len(sys.argv) > 4:
    out["__workload__"] = sys.argv[4]
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps({k: round(v / 1e6, 1) for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, int)), key=lambda kv: -kv[1])[:8]}, indent=1))
