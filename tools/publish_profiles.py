"""Copy the judged summaries of one gpurun_out/<tag>/ collection (tools/collect_profiles.sh) into profiles/ (tracked)."""
import csv, json, os, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/{tag}", "profiles"
os.makedirs(dst, exist_ok=True)
for f, t in (("bench.json", f"{tag}_bench.json"), ("bench_serial.json", f"{tag}_bench_serial.json"), ("bench_sparse_b.json", f"{tag}_bench_sparse_b.json"), ("bench_plonk_2p20.json", f"{tag}_bench_plonk_2p20.json"), ("bench_bls12381_2p20.json", f"{tag}_bench_bls12381_2p20.json"),
             ("bench_bn128_2p24.json", f"{tag}_bench_bn128_2p24.json"), ("bench_fflonk_2p18.json", f"{tag}_bench_fflonk_2p18.json"), ("stats/bench_kernel_stats.csv", f"{tag}_bench_kernel_stats.csv"),
             ("stats_plonk/plonk_kernel_stats.csv", f"{tag}_plonk_kernel_stats.csv")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, t))
wl = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])["config"]
wl_tag = f"groth16:{wl['curve']}:2^{wl['log_n']}:b_zero_every={0 if wl['b_density'] == 1.0 else round(1 / (1 - wl['b_density']))}:{wl['witness']}"
subprocess.check_call([sys.executable, "tools/pmc_to_traffic.py", f"{src}/pmc_fetch/f_counter_collection.csv", f"{src}/pmc_write/w_counter_collection.csv", f"{dst}/pmc_traffic.json", wl_tag])
shutil.copy(f"{dst}/pmc_traffic.json", f"{dst}/{tag}_pmc_traffic.json")
if os.path.exists(f"{src}/fieldbench29.txt"):
    shutil.copy(f"{src}/fieldbench29.txt", f"{dst}/{tag}_fieldbench29.txt")
rows = list(csv.DictReader(open(f"{src}/stats/bench_kernel_stats.csv")))
with open(f"{dst}/{tag}_bench_kernel_stats_summary.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --pipeline 1 --no-cpu-baseline --no-napi-wall ({tag}, MI355X)\n\n"
            f"Full CSV: {tag}_bench_kernel_stats.csv. Includes the one-off set-up kernels (k_gen_geometric_bases, k_msm_precompute, k_coef_*, k_scan_u32) and the\n"
            "sub-metric runs (plain-base G1 MSM, NTT) after the timed region.\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:40]:
        f.write(f"| `{r['Name'].split('(')[0].replace('void ', '')[:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |\n")
print(open(f"{dst}/{tag}_bench.json").read()[:3000])
