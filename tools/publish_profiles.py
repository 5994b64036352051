"""Copy the judged summaries of one gpurun_out/<tag>/ collection (tools/collect_profiles.sh) into profiles/ (tracked):
bench lines, rocprofv3 kernel-stats summaries (BN254 and BLS12-381, serial command), the PLONK per-proof timeline, HBM traffic of the
accumulation kernels from the two PMC passes WITH the per-access-class correction, the mul ceilings, the static instruction counts.
usage: python tools/publish_profiles.py r03"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src, dst = f"gpurun_out/{tag}", "profiles"
os.makedirs(dst, exist_ok=True)
for f in ("bench", "bench_serial", "bench_sparse_b", "bench_mixed_witness", "bench_real", "bench_plonk_2p20", "bench_plonk_2p20_serial", "bench_bls12381_2p20", "bench_bn128_2p24", "bench_fflonk_2p18", "bench_force_dist"):
    if os.path.exists(f"{src}/{f}.json"):
        shutil.copy(f"{src}/{f}.json", f"{dst}/{tag}_{f}.json")
line = lambda f: json.loads(open(f).read().strip().splitlines()[-1])
bench = line(f"{src}/bench.json")


def stats_md(csv_path, out, cmd):
    rows = list(csv.DictReader(open(csv_path)))
    shutil.copy(csv_path, out.replace("_summary.md", ".csv"))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd} ({tag}, MI355X)\n\n"
                f"Full CSV: {os.path.basename(out).replace('_summary.md', '.csv')}. Includes the one-off set-up kernels (k_gen_geometric_bases, k_msm_precompute, k_coef_*, k_scan_u32) and the\n"
                "sub-metric runs (plain-base G1 MSM, NTT) after the timed region.\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write(f"| `{r['Name'].split('(')[0].replace('void ', '')[:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |\n")


for sub, name, cmd in (("stats", "bench_kernel_stats", "python bench.py --steps 20 --warmup 3 --pipeline 1 --no-cpu-baseline --no-napi-wall"),
                       ("stats_bls", "bench_bls12381_kernel_stats", "python bench.py --curve bls12381 --steps 6 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall"),
                       ("stats_real", "bench_real_kernel_stats", "python bench.py --coef-dist real --witness mixed --steps 10 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall"),
                       ("stats_p24", "bench_p24_kernel_stats", "python bench.py --log-n 24 --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall")):
    c = glob.glob(f"{src}/{sub}/**/*kernel_stats.csv", recursive=True)
    if c:
        stats_md(c[0], f"{dst}/{tag}_{name}_summary.md", cmd)

# ---- PLONK per-proof timeline
pt = glob.glob(f"{src}/stats_plonk/**/*kernel_trace.csv", recursive=True)
if pt:
    md = subprocess.run([sys.executable, "tools/plonk_trace_summary.py", pt[0]], capture_output=True, text=True).stdout
    extra = ""
    if os.path.exists(f"{src}/bench_plonk_under_rocprof.json"):
        b = line(f"{src}/bench_plonk_under_rocprof.json")
        extra = f"\nBench line of the profiled run itself: {b['value']} proofs/s, {b['ms_per_step']} ms per proof.\n"
    open(f"{dst}/{tag}_plonk_kernel_summary.md", "w").write(md + extra)

# ---- HBM traffic per launch of the accumulation kernels, corrected per access class
# FETCH_SIZE tallies memory-side read requests at 64 B each (profiles/r02_gather_calibration.md): exact for 64-byte requests (a G1 table entry,
# the sector behind a 16-byte list read), HALF the bytes of a 128-byte request (a G2 table entry). So: G1 bytes = raw; G2 bytes = raw + 64 B x
# (number of 128-byte gathers) = raw + 64 B x mixed additions of the launch (counted on the device: bench line, accum_mixed_additions).
def agg(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zkmi::", "")].append(float(r["Counter_Value"]) * 1024)
    return {k: sum(v) / len(v) for k, v in acc.items()}


# ---- r06: the same for every other config of the driver line -> profiles/pmc_traffic.json["workloads"][<tag>] (bench.py looks its own workload up there)
# Correction per access class from THIS round's calibration (gatherbench under the same counter: <tag>_gather_calibration.md): a gather of w bytes
# is tallied at factor(w) of its bytes; G1 / G2 accumulation kernels gather one table entry per mixed addition (BN254 64 / 128 B, BLS12-381 96 / 192 B),
# so bytes = raw + (1 / factor - 1) x factor x w x additions = raw + (1 - factor) x w x additions; coalesced streaming kernels: x2 (the guide).
def gather_factors():
    out = {64: 1.0, 128: 0.5, 96: None, 192: None}
    g = glob.glob(f"{src}/pmc_gather/**/*counter_collection.csv", recursive=True)
    if not g:
        return out, None
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(g[0])):
        if r["Counter_Name"] == "FETCH_SIZE":
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024)
    lanes = 1 << 22
    rows = []
    for v, w in ((4, 64), (6, 96), (8, 128), (12, 192)):
        k = next((x for x in per if x.startswith(f"k_gather<{v}>")), None)
        if k:
            # the first launches of each kernel are over the 4 GiB table (the 832 MB variant of <4> comes later in every repetition: take the maximum)
            val = max(per[k])
            out[w] = val / (lanes * w)
            rows.append((k, lanes * w, val, out[w]))
    for kn, w in (("k_gather_pair192", 192), ("k_gather_pair", 128)):      # r06: an entry read by TWO lanes, half of it each (k_msm_accum29_g2s)
        kp = next((x for x in per if x.split("(")[0] == kn), None)
        if kp:
            val = max(per[kp])
            out[f"pair{w}"] = val / ((lanes // 2) * w)
            rows.append((kp, (lanes // 2) * w, val, out[f"pair{w}"]))
    ks = next((x for x in per if x.startswith("k_stream")), None)
    if ks:
        rows.append((ks, 1 << 30, max(per[ks]), max(per[ks]) / (1 << 30)))
    with open(f"{dst}/{tag}_gather_calibration.md", "w") as f:
        f.write(f"# FETCH_SIZE calibration in the MSM's gather widths ({tag}; tools/gatherbench under rocprofv3 --pmc FETCH_SIZE, 2^22 random gathers over a 4 GiB table)\n\n"
                "| kernel | bytes actually loaded | FETCH_SIZE x 1024 | tallied fraction |\n|---|---|---|---|\n")
        for k, b, v, fr in rows:
            f.write(f"| `{k}` | {b/1e6:.1f} MB | {v/1e6:.1f} MB | {fr:.3f} |\n")
        f.write("\n64 / 128 B = a BN254 G1 / G2 table entry, 96 / 192 B = a BLS12-381 entry (96 / 192-byte aligned only: half of them straddle a 128-byte line).\n"
                "`publish_profiles.py` scales the gather share of an accumulation kernel's FETCH_SIZE by 1 / fraction.\n")
    return out, rows


factors, _cal = gather_factors()

fc, wc = glob.glob(f"{src}/pmc_fetch/**/*counter_collection.csv", recursive=True), glob.glob(f"{src}/pmc_write/**/*counter_collection.csv", recursive=True)
if fc and wc:
    F, W = agg(fc[0], "FETCH_SIZE"), agg(wc[0], "WRITE_SIZE")
    adds = bench.get("accum_mixed_additions", {})
    g1_adds = next((v for k, v in adds.items() if "(A)" in k), 0)
    g2_adds = next((v for k, v in adds.items() if "(B2)" in k), 0)
    out = {"__workload__": f"groth16:{bench['config']['curve']}:2^{bench['config']['log_n']}:b_zero_every={0 if bench['config']['b_density'] == 1.0 else round(1 / (1 - bench['config']['b_density']))}:{bench['config']['witness']}"}
    rows = []
    for k in sorted(F):
        raw, wr = F[k], W.get(k, 0.0)
        is_g2 = k.startswith("k_msm_accum29_g2") or k.startswith("k_msm_accum<Fp2")
        # r06: the split layout (k_msm_accum29_g2s) reads a 128-byte entry as 2 x 32 bytes in each of two lanes: its own calibration factor
        g2_fr = (factors.get("pair128") or 0.5) if k.startswith("k_msm_accum29_g2s") else (factors.get(128) or 0.5)
        corr = raw + ((1.0 - g2_fr) * 128.0 * g2_adds if is_g2 else 0.0)
        if not (k.startswith("k_msm_accum") or k.startswith("k_ntt") or k.startswith("k_msm_rowcol") or k.startswith("k_build_abc")):
            corr = 2 * raw                                       # coalesced streaming kernels: 128-byte requests tallied at 64 B (MI355X_MICROARCH.md)
        key = k
        while key.endswith(">") and any(key.endswith(s) for s in (", true>", ", false>")):
            key = key[:key.rfind(",")] + ">"
        out[key] = int(corr + wr)
        rows.append((k, raw, wr, corr + wr))
    json.dump(out, open(f"{dst}/pmc_traffic.json", "w"), indent=1, sort_keys=True)
    shutil.copy(f"{dst}/pmc_traffic.json", f"{dst}/{tag}_pmc_traffic.json")
    with open(f"{dst}/{tag}_pmc_traffic.md", "w") as f:
        f.write(f"# HBM traffic per launch, accumulation kernels ({tag}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, `bench.py --steps 2 --warmup 1 --pipeline 1`)\n\n"
                "FETCH_SIZE tallies memory-side read requests at 64 B each (`r02_gather_calibration.md`): exact for the 64-byte gathers of a G1 table entry and for the sector\n"
                "behind a 16-byte list read, half the bytes of a 128-byte G2 gather by one lane. Correction per access class: G1 bytes = raw; G2 bytes = raw + (1 - tallied\n"
                f"fraction) x 128 B x gathers (= mixed additions counted on the device: {g2_adds}); r06: k_msm_accum29_g2s reads an entry as 2 x 32 bytes in each of two\n"
                f"lanes, tallied at {factors.get('pair128')} (k_gather_pair in the calibration).\n\n| kernel | FETCH raw MB | WRITE MB | corrected total MB |\n|---|---|---|---|\n")
        for k, raw, wr, tot in rows:
            if "accum" in k:
                f.write(f"| `{k}` | {raw/1e6:.0f} | {wr/1e6:.0f} | {tot/1e6:.0f} |\n")
        m1 = g1_adds * 64 + g1_adds / 4 * 64 + (1 << 19) * 128
        m2 = g2_adds * 128 + g2_adds / 4 * 64 + (1 << 19) * 256
        f.write(f"\nModel per 2^20 launch (c = 20: 13 table rows, 2^19 buckets): G1 = gathers {g1_adds * 64 / 1e6:.0f} + list sectors (one 64-byte sector per 4 entries) {g1_adds / 4 * 64 / 1e6:.0f} + bucket stores "
                f"{(1 << 19) * 128 / 1e6:.0f} = {m1 / 1e6:.0f} MB; G2 = {g2_adds * 128 / 1e6:.0f} + {g2_adds / 4 * 64 / 1e6:.0f} + {(1 << 19) * 256 / 1e6:.0f} = {m2 / 1e6:.0f} MB.\n"
                "r02 (4-byte list reads, one sector per ENTRY): G1 1 937 MB, G2 3 558 MB under r02's x2-everything correction = 2 870 MB under this one.\n"
                "The residual over the model (G1 ~15-20 %, G2 ~25 %) is not list traffic any more: page-table walks of 0.9 / 1.7 GB of random gathers are\n"
                "tallied by the same counter, and the G2 kernel's WRITE_SIZE includes its scratch stores.\n")

workloads = {}
try:
    workloads = json.load(open(f"{dst}/pmc_traffic.json")).get("workloads", {})
except Exception:
    workloads = {}
md_rows = []
for nm, entry_g1, entry_g2 in (("bls", 96, 192), ("plonk", 64, 128), ("real", 64, 128), ("p24", 64, 128)):
    fcs, wcs = glob.glob(f"{src}/pmc_{nm}_fetch/**/*counter_collection.csv", recursive=True), glob.glob(f"{src}/pmc_{nm}_write/**/*counter_collection.csv", recursive=True)
    bj = f"{src}/pmc_{nm}_bench.json"
    if not (fcs and wcs and os.path.exists(bj)):
        continue
    try:
        b = line(bj)
    except Exception:
        continue
    F, W = agg(fcs[0], "FETCH_SIZE"), agg(wcs[0], "WRITE_SIZE")
    cfg = b["config"]
    if b["metric"].startswith("groth16"):
        wl = f"groth16:{cfg['curve']}:2^{cfg['log_n']}:" + (f"b_zero_every={0 if cfg['b_density'] == 1.0 else round(1 / (1 - cfg['b_density']))}" if cfg.get("coef_dist", "flat") == "flat" else "coef_dist=real") + f":{cfg['witness']}"
        adds = b.get("accum_mixed_additions", {})
        g1_adds = max([v for k, v in adds.items() if "(B2)" not in k] or [0])
        g2_adds = next((v for k, v in adds.items() if "(B2)" in k), 0)
    else:
        wl = f"{b['metric'].split('_')[0]}:{cfg['curve']}:2^{cfg['log_n']}:additions={cfg.get('n_additions', 0)}"
        g1_adds, g2_adds = (b.get("int_alu") or {}).get("mixed_additions", 0), 0
    out = {}
    for k in sorted(F):
        raw, wr = F[k], W.get(k, 0.0)
        if k.startswith("k_msm_accum29_g2s"):
            fr = factors.get(f"pair{entry_g2}") or (0.5 if entry_g2 == 128 else 0.666)
            corr = raw + (1.0 - fr) * entry_g2 * g2_adds
        elif k.startswith("k_msm_accum29_g2") or k.startswith("k_msm_accum<Fp2"):
            fr = factors.get(entry_g2) or 0.5
            corr = raw + (1.0 - fr) * entry_g2 * g2_adds
        elif k.startswith("k_msm_accum"):
            fr = factors.get(entry_g1) or 1.0
            corr = raw + (1.0 - fr) * entry_g1 * g1_adds
        elif k.startswith("k_ntt") or k.startswith("k_msm_rowcol") or k.startswith("k_abc"):
            corr = raw
        else:
            corr = 2 * raw
        key = k
        while key.endswith(">") and any(key.endswith(x) for x in (", true>", ", false>")):
            key = key[:key.rfind(",")] + ">"
        out[key] = int(corr + wr)
        if "accum" in k:
            md_rows.append((wl, k, raw, wr, corr + wr))
    workloads[wl] = out
if workloads:
    try:
        top = json.load(open(f"{dst}/pmc_traffic.json"))
    except Exception:
        top = {}
    top["workloads"] = workloads
    json.dump(top, open(f"{dst}/pmc_traffic.json", "w"), indent=1, sort_keys=True)
    shutil.copy(f"{dst}/pmc_traffic.json", f"{dst}/{tag}_pmc_traffic.json")
    with open(f"{dst}/{tag}_pmc_traffic_other_configs.md", "w") as f:
        f.write(f"# HBM traffic per launch of the accumulation kernels, the other configs of the driver line ({tag}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,\n"
                "`bench.py <config> --steps 2 --warmup 1 --pipeline 1`). Corrected per access class with this round's gather calibration (" + f"{tag}_gather_calibration.md).\n\n"
                "| workload | kernel | FETCH raw MB | WRITE MB | corrected total MB |\n|---|---|---|---|---|\n")
        for wl, k, raw, wr, tot in md_rows:
            f.write(f"| {wl} | `{k}` | {raw/1e6:.0f} | {wr/1e6:.0f} | {tot/1e6:.0f} |\n")

if os.path.exists(f"{src}/fieldbench29.txt"):
    shutil.copy(f"{src}/fieldbench29.txt", f"{dst}/{tag}_fieldbench29.txt")
if os.path.exists(f"{src}/pytest_gpu.log"):
    shutil.copy(f"{src}/pytest_gpu.log", f"{dst}/{tag}_pytest_gpu.log")
with open(f"{dst}/{tag}_isa_counts.md", "w") as f:
    for obj in ("snarkjs_amd/build/msm_bn254.o", "snarkjs_amd/build/msm_bls12381.o"):
        f.write(subprocess.run([sys.executable, "tools/isa_counts.py", obj], capture_output=True, text=True).stdout + "\n")
    f.write("Reading (r04 build: the multiply-adds of one COLUMN of the product scanning are one asm statement — one `s_nop` per column where r03 carried one per\n"
            "multiply-add). Each kernel is: gather + unpack + the head of the addition (first big segment), the rare equal-points doubling (the segment skipped\n"
            "by a forward branch right after it), the body of the addition (the segment that ends in the backward jump), then three copies of the once-per-lane\n"
            "store. Main path of ONE mixed addition = head + body (+ the small gather / loop-control segments): BN254 G1 616 + 1 554 (+ ~60) = 2 230 VALU\n"
            "(1 467 MACs, 283 s_nop; r03: 2 238 VALU + 1 307 s_nop); BN254 G2, LDS-parked layout (`k_msm_accum29_g2`, ZKMI_G2_SPLIT=0) 1 354 + 4 434 (+ ~140) = 5 930 (4 374 MACs,\n"
            "767 s_nop; r03: 5 944 + 4 113 s_nop); r06 default `k_msm_accum29_g2s` (one Fq2 component per lane): 929 + 2 431 = 3 360 per LANE (2 196 MACs, 213 s_nop), two lanes\n"
            "per addition = 6 720 (+13 %: the DPP exchanges and the negations both lanes form);\n"
            "BLS12-381 G2, r06 default `k_msm_accum29_g2s` (XYZZ in registers, 8M + 2S): 1 860 + 5 571 = 7 431 per lane (5 503 MACs), 14 862 per addition, hot loop 61 KB; the\n"
            "packed-Jacobian LDS layout (`k_msm_accum29_g2`, ZKMI_G2_SPLIT=0):\n"
            "BLS12-381 G1 1 224 + 3 404 (+ ~370) = 5 002 (453 s_nop; r03: 5 012 + 3 262 s_nop); BLS12-381 G2 (packed Jacobian accumulator, 8M + 3S) 5 438 + 9 442 =\n"
            "14 880 VALU (1 387 s_nop; r03: 14 902 + 11 242 s_nop). These are the constants `bench.py` divides by (`VALU_PER_ADD`). Plain C multiply-adds\n"
            "(measured, not shipped: `-DZK_MAD_PLAIN`) carry no s_nop but 4-6 % more VALU (64-bit merge adds of the partial chains): 2 375 / 6 200 / 5 237 / 15 730.\n")
print(json.dumps({k: bench[k] for k in ("value", "ms_per_step", "roofline", "int_alu") if k in bench}, indent=1)[:2500])
