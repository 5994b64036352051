#!/usr/bin/env python
"""bench.py — BASELINE.json metric on MI355X: Groth16 proofs/s (BN254, 2^20 constraints, synthetic zkey/wtns =
configs[1]) with G1-MSM Mscalar/s and NTT Melem/s as sub-metrics.

A "step" is one whole proof: buildABC -> 3 x (iNTT, coset NTT) -> joinABC -> 5 MSMs -> blinding/toAffine, with the
proving key AND the witness already resident in HBM when the timed region starts (zkmi_groth16_prove_dev).
N > 1: one process per GPU (torchrun), every rank proves its own stream of proofs over a replicated key — the proof
is the independent unit (SURVEY.md §8e "whole proofs"); no data-path collective; "scaling": "weak".

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU instructions of one mixed addition, main path of the shipped code objects (tools/isa_counts.py -> profiles/r04_isa_counts.md: the fetch /
# unpack segment + the addition segment of k_msm_accum29 / k_msm_accum29_g2 per curve). r04: one asm statement per column of the product scanning
# (the hand order of r03 — the same VALU counts — with 283 / 767 / 453 / 1 387 s_nop per addition instead of 1 307 / 4 113 / 3 262 / 11 242)
# r06: BN254 G2 accumulates with one Fq2 component per lane (k_msm_accum29_g2s): 3 360 VALU per LANE and addition (929 head + 2 431 body,
# profiles/r06_isa_counts.md), two lanes per addition = 6 720 per addition (the LDS-parked layout, ZKMI_G2_SPLIT=0: 5 930 in one lane)
# BLS12-381 G2 the same way (XYZZ in registers, 8M + 2S): 7 431 per lane (1 860 + 5 571) = 14 862 per addition (packed Jacobian in LDS, 8M + 3S: 14 880)
G2_SPLIT = os.environ.get("ZKMI_G2_SPLIT", "1") != "0"
G2_SPLIT_BLS = G2_SPLIT and os.environ.get("ZKMI_G2_SPLIT_BLS", "1") != "0"
VALU_PER_ADD = {"bn128": {"g1": 2230, "g2": 6720 if G2_SPLIT else 5930}, "bls12381": {"g1": 5002, "g2": 14862 if G2_SPLIT_BLS else 14880}}
VALU_ISSUE_PEAK_G = 614.4                      # 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave instruction
# measured Montgomery-multiply ceilings of the chip, Gmul/s at 8 waves per SIMD, in the limb form the accumulation kernels of the curve use
# (tools/fieldbench29 on the library's own mul29): BN254 Fq 9 x 29-bit limbs, BLS12-381 Fq 14 x 28-bit limbs; the saturated 32-bit forms they
# replaced measured 130 and 58.6 (tools/fieldbench, profiles/r01_fieldbench.txt). profiles/r04_fieldbench29.txt holds this round's run (column
# statements: BN254 171 at 8 waves per SIMD, 149 at 2; BLS12-381 81.7 / 76.8; the r03 per-instruction asm build beside it: 176 / 77.7 and 144 / 73.8).
FIELD_MUL_PEAK_G = {"bn128": 175.0, "bls12381": 78.3}
FIELD_MUL_PEAK_32 = {"bn128": 130.0, "bls12381": 58.6}


def _oracle():
    """The CPU oracle is test infrastructure: only this cpu_baseline leg of bench.py touches it (after the timed region)."""
    t = os.path.join(ROOT, "tests")
    if t not in sys.path:
        sys.path.insert(0, t)
    import oracle_lib as O
    return O


def drain_c_stdout_to_stderr():
    """RCCL prints its version banner through C stdio; with stdout redirected to a pipe or file that text sits in libc's buffer until exit
    and would land AFTER the JSON line. Flush it to stderr so that stdout carries the one JSON line and nothing else."""
    import ctypes
    sys.stdout.flush()
    keep = os.dup(1)
    try:
        os.dup2(2, 1)
        ctypes.CDLL(None).fflush(None)
    finally:
        os.dup2(keep, 1)
        os.close(keep)


def box_calibration(L):
    """Probes of the device this run landed on (include/zkmi.h: zkmi_calibrate_box, zkmi_calibrate_code_fetch), AFTER the timed region: the
    boxes differ, and a low line next to a `mul29_gmul_per_s` well under 150, a gather rate well under 6 TB/s or a code-fetch ratio well under
    one says so; rocm-smi's clocks / power state of the card go beside them."""
    import ctypes
    a, b = ctypes.c_double(0), ctypes.c_double(0)
    if L.zkmi_calibrate_box(ctypes.byref(a), ctypes.byref(b)) != 0:
        return None
    out = {"mul29_gmul_per_s": round(a.value, 1), "gather128_gb_per_s": round(b.value, 1), "healthy": {"mul29_gmul_per_s": 150.0, "gather128_gb_per_s": 6000.0},
           "note": "two dependent chains of Montgomery products on 29-bit limbs, 8 workgroups per CU; 16 dependent random 128-byte gathers per lane over a 2 GiB table; after the timed region; `healthy` = what this probe measures on a box that gives ~100 proofs/s"}
    c, d = ctypes.c_double(0), ctypes.c_double(0)
    if hasattr(L, "zkmi_calibrate_code_fetch") and L.zkmi_calibrate_code_fetch(ctypes.byref(c), ctypes.byref(d)) == 0 and c.value > 0:
        out["code_fetch"] = {"loop_17KB_gmul_per_s": round(c.value, 1), "loop_210KB_gmul_per_s": round(d.value, 1), "big_over_small": round(d.value / c.value, 3),
                             "note": "the same product chain as straight-line loops of ~17 KB and ~210 KB of code at 2 waves per SIMD: the cost of instruction fetch beyond the 64 KB instruction cache on this box"}
    if hasattr(L, "zkmi_compact_code"):
        out["compact_code_mask"] = int(L.zkmi_compact_code())     # which MSM kernels ran with called products on this box (include/zkmi.h)
    try:
        import subprocess
        smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        if smi.returncode == 0 and smi.stdout.strip():
            card = next(iter(json.loads(smi.stdout).values()))
            keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "performance level", "temperature (sensor junction)", "temperature (sensor edge)"))}
            out["rocm_smi"] = dict(list(keep.items())[:12])
    except Exception:
        pass
    return out


def relaunch_if_needed(args):
    """`python bench.py --gpus N` starts its own N ranks (one process per GPU over RCCL) when it was not already started by
    torch.distributed.run; a world size that does not match --gpus is refused rather than reported under a wrong n_gpus."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ or "LOCAL_RANK" in os.environ:
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line with the wrong n_gpus")
        return
    if args.gpus <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def reference_wasm_baseline():
    """The reference's own WASM + worker-thread path, measured in the BUILD container by tools/ref_wasm_baseline.py (the bundle cannot
    travel to the GPU box) and committed under profiles/: quoted next to the live C port, never as `value`."""
    runs, src, host = [], [], None
    for name in ("r02_ref_wasm_baseline.json", "r04_ref_wasm_baseline.json"):      # r02: 2^14 .. 2^18; r04: the bench size itself, 2^20
        f = os.path.join(ROOT, "profiles", name)
        if os.path.exists(f):
            d = json.load(open(f))
            host = host or d["host"]
            runs += [{k: r[k] for k in ("log_n", "threads", "ms_per_proof", "proofs_per_s")} | {"node": r.get("node")} for r in d["runs"]]
            src.append("profiles/" + name)
    if not runs:
        return None
    at20 = [r for r in runs if r["log_n"] == 20]
    best20 = max(at20, key=lambda r: r["proofs_per_s"]) if at20 else None
    return {"where": f"build container, {host['cpus']} cpus ({host['model']}), Node {runs[0]['node']}; NOT this box",
            "runs": [{k: r[k] for k in ("log_n", "threads", "ms_per_proof", "proofs_per_s")} for r in runs], "source": src,
            "measured_at_bench_size": None if best20 is None else {"log_n": 20, "threads": best20["threads"], "ms_per_proof": best20["ms_per_proof"], "proofs_per_s": best20["proofs_per_s"]},
            "note": "snarkjs groth16.prove of the reference bundle (WASM + worker threads) on the bench's own synthetic key recipe; the 2^20 figure is MEASURED (r04: one proof takes about a minute on the container's 8 threads), nothing here is extrapolated"}


def _ref_bundle():
    """the reference's bundle where a box has it: the staged copy oracle/_ref (make -C oracle _ref; git-ignored, shipped by gpurun), else /root/reference"""
    for p in (os.path.join(ROOT, "oracle", "_ref", "build", "snarkjs.min.js"), "/root/reference/build/snarkjs.min.js"):
        if os.path.exists(p):
            return p
    return None


def reference_wasm_same_box(proto, cases, curve="bn128", budget_s=200.0, warm=None):
    """The REFERENCE's own prover (snarkjs bundle: WASM + worker threads) on THIS box's host cores, after the timed region, rank 0 only
    (tools/ref_wasm_same_box.js; test / measurement infrastructure, never the product path). `cases` = [(log_n, zkey, wtns, draws_mont, device_proof_json)],
    smallest first; a case is run only while the remaining budget covers ~5x the previous one (a 2^20 Groth16 proof takes the reference 10 - 60 s
    depending on the host). The proof the reference emits for the same blinding draws must be the device's proof, byte for byte (`bit_identical`)."""
    import hashlib
    import shutil
    import subprocess
    import tempfile
    node, bundle = shutil.which("node"), _ref_bundle()
    if node is None or bundle is None:
        return {"skipped": "node or the reference bundle (oracle/_ref: `make -C oracle _ref` in the build container) is missing on this box"}
    threads = min(os.cpu_count() or 1, 64)                       # ffjavascript's own cap (threadman: concurrency > 64 -> 64)
    runs, t_start, last = [], time.perf_counter(), None
    need = max(len(c[1]) + len(c[2]) for c in cases) + (len(warm[0]) + len(warm[1]) if warm is not None else 0) + (16 << 20)
    base = None                                                   # key files in memory-backed storage where there is room, else the default temp dir
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * need:
            base = "/dev/shm"
    except OSError:
        pass
    with tempfile.TemporaryDirectory(dir=base) as td:
        warm_args = []
        if warm is not None:
            open(os.path.join(td, "w.zkey"), "wb").write(warm[0])
            open(os.path.join(td, "w.wtns"), "wb").write(warm[1])
            warm_args = [os.path.join(td, "w.zkey"), os.path.join(td, "w.wtns")]
        for lg, zkey, wtns, draws, dev_json in cases:
            left = budget_s - (time.perf_counter() - t_start)
            if last is not None and last * 5.5 > left:
                runs.append({"log_n": lg, "skipped": f"budget: the previous size took {last:.1f} s, {left:.0f} s left of {budget_s:.0f}"})
                continue
            zf, wf = os.path.join(td, "k.zkey"), os.path.join(td, "k.wtns")
            open(zf, "wb").write(zkey)
            open(wf, "wb").write(wtns)
            env = dict(os.environ, NTHREADS=str(threads), SNARKJS_REF_BUNDLE=bundle, CURVE=curve)
            env.pop("SINGLE", None)
            t0 = time.perf_counter()
            try:
                r = subprocess.run([node, "--harmony-optional-chaining", "--harmony-nullish", "--max-old-space-size=24000", os.path.join(ROOT, "tools", "ref_wasm_same_box.js"),
                                    proto, zf, wf, ",".join(bytes(d).hex() for d in draws)] + warm_args, capture_output=True, text=True, env=env, timeout=max(30.0, left))
            except subprocess.TimeoutExpired:
                runs.append({"log_n": lg, "error": f"timed out after {left:.0f} s"})
                break
            last = time.perf_counter() - t0
            if r.returncode != 0:
                runs.append({"log_n": lg, "error": (r.stderr or r.stdout)[-300:]})
                break
            d = json.loads(r.stdout.strip().splitlines()[-1])
            run = {"log_n": lg, "threads": d["threads"], "ms_per_proof": round(d["ms"], 1), "proofs_per_s": round(1e3 / d["ms"], 5), "warmup_proof_ms": round(d.get("warm_ms", 0.0), 1),
                   "process_wall_s": round(last, 1)}
            if dev_json is not None:
                run["bit_identical_to_device_proof"] = bool(hashlib.sha256(dev_json.encode()).hexdigest() == d["proof_json_sha256"])
            runs.append(run)
            host = {"cpus": d["cpus"], "model": d["cpu_model"], "node": d["node"]}
    done = [x for x in runs if "ms_per_proof" in x]
    if not done:
        return {"runs": runs}
    return {"where": f"THIS box: {host['cpus']} host cpus ({host['model']}), Node {host['node']}, {done[0]['threads']} worker threads (ffjavascript caps at 64)",
            "what": f"snarkjs {proto}.prove of the reference bundle (oracle/_ref) on the bench's own key, files in memory, one small untimed proof first (worker start, WASM tier-up), then ONE timed proof per size",
            "runs": runs, "at_bench_size": next((x for x in done if x["log_n"] == cases[-1][0]), None)}


def _plonk_same_box_into(base, same_box, proto, lg):
    """cpu_baseline of a PLONK / FFLONK line from the same-box reference run: the largest domain that was timed, scaled LINEARLY to the bench domain"""
    if not same_box:
        return
    base.setdefault("reference_wasm", {})
    base["reference_wasm"] = dict(base["reference_wasm"] or {}, same_box=same_box)
    done = [x for x in same_box.get("runs", []) if "ms_per_proof" in x]
    if not done:
        return
    big = max(done, key=lambda x: x["log_n"])
    scale = 2.0 ** (lg - big["log_n"])
    base.update({"value": round(1e3 / (big["ms_per_proof"] * scale), 6), "cores": big["threads"], "kind": "reference",
                 "bit_identical_to_device_proof": all(x.get("bit_identical_to_device_proof") for x in done),
                 "sample": f"ONE snarkjs {proto}.prove (the reference's bundle: WASM + {big['threads']} worker threads, Node) at the domain 2^{big['log_n']} on this box's host: {big['ms_per_proof'] / 1e3:.1f} s"
                           + (f", scaled linearly x{scale:g} to 2^{lg} (measured once at 2^18 / 2^20 by hand: profiles/r05_plonk_vs_reference.txt)" if scale != 1 else "")})


def reference_wasm_baseline_plonk(proto, lg):
    """The reference's own plonk.prove / fflonk.prove (WASM + worker threads), measured in the BUILD container and committed under profiles/.
    Sizes are PLONK DOMAINS (what --log-n means here). PLONK: r04 measured domains 2^11 and 2^16 (and 2^20 when the long run finished) on keys from
    the reference's setup over a known-tau ptau (tools/ref_wasm_baseline_plonk_big.js); r03's runs (real JS ceremony, 2^9 .. 2^13 as domains: that
    file labels them by r1cs constraints, one less) are listed beside them. A figure for a domain that was not measured is a LINEAR extrapolation
    from the largest measured one and says so."""
    runs = []
    f4 = os.path.join(ROOT, "profiles", "r04_ref_wasm_baseline_plonk.json")
    f3 = os.path.join(ROOT, "profiles", "r03_ref_wasm_baseline_plonk.json")
    host = None
    if os.path.exists(f4):
        d = json.load(open(f4))
        host = d["host"]
        runs += [{"log_domain": r["log_domain"], "threads": r["threads"], "ms_per_proof": r["ms_per_proof"], "node": r.get("node"), "source": "profiles/r04_ref_wasm_baseline_plonk.json"} for r in d["runs"] if r["proto"] == proto]
    if os.path.exists(f3):
        d = json.load(open(f3))
        host = host or d["host"]
        runs += [{"log_domain": r["log_n"] + 1, "threads": r["threads"], "ms_per_proof": r["ms_per_proof"], "node": r.get("node"), "source": "profiles/r03_ref_wasm_baseline_plonk.json (labelled log_n %d there)" % r["log_n"]}
                 for r in d["runs"] if r["proto"] == proto]
    if not runs:
        return None
    many = [r for r in runs if r["threads"] > 1] or runs
    exact = [r for r in many if r["log_domain"] == lg]
    out = {"where": f"build container, {host['cpus']} cpus ({host['model']}), Node {runs[0]['node']}; NOT this box",
           "measured": [{k: r[k] for k in ("log_domain", "threads", "ms_per_proof", "source")} for r in runs]}
    if exact:
        best = min(exact, key=lambda r: r["ms_per_proof"])
        out["measured_at_bench_size"] = {"log_domain": lg, "threads": best["threads"], "ms_per_proof": best["ms_per_proof"], "proofs_per_s": round(1e3 / best["ms_per_proof"], 6)}
    else:
        best = max(many, key=lambda r: r["log_domain"])
        scale = 2.0 ** (lg - best["log_domain"])
        out["extrapolated_proofs_per_s"] = round(1e3 / (best["ms_per_proof"] * scale), 6)
        out["extrapolation"] = f"linear x{scale:g} from the measured domain 2^{best['log_domain']} ({best['threads']} threads, {best['ms_per_proof']} ms) - not a measurement at 2^{lg}"
    return out


def pmc_traffic(workload_tag, kernel):
    """HBM bytes per launch of `kernel` from the separate rocprofv3 --pmc passes of THIS workload (tools/collect_profiles.sh -> tools/publish_profiles.py ->
    profiles/pmc_traffic.json, keyed by workload); a file collected on another workload (size / curve / B density / protocol) is not used: None."""
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tf):
        return None
    try:
        pj = json.load(open(tf))
        if pj.get("__workload__") == workload_tag and kernel in pj:
            return pj[kernel]
        return (pj.get("workloads", {}).get(workload_tag) or {}).get(kernel)
    except Exception:
        return None


def _synth_kw(args):
    return dict(witness=args.witness, b_zero_every=args.b_zero_every, coef_dist=args.coef_dist)


def cpu_baseline(args, zkey, wtns, log_n_full):
    """The CPU legs of a Groth16 line (rank 0, N = 1, after the timed region), by --cpu-baseline-mode:
      full    (the headline) the C restatement of the reference (oracle/zk_oracle.c, OpenMP over the reference's own task split) on a bounded sample —
              the bench's own key with >= 16 host threads, else a 2^18 key of the same recipe scaled linearly — AND the reference itself (WASM + worker
              threads, tools/ref_wasm_same_box.js) at 2^18 and at the bench size, its proof for the bench's (r, s) compared with the device's;
      ref     the reference itself at the bench size only (one proof, ~15 - 40 s) with the same comparison: the other BASELINE configs inside the
              driver run (other_configs) carry their OWN same-box reference parity this way; without node / the bundle it falls back to `port`;
      port    the C restatement alone, parity of the device proof against it on that sample;
      closed  (2^24: a reference proof would take minutes) the C restatement at 2^18 scaled linearly, and the device's proof AT THE BENCH SIZE checked
              against its closed form (every base of the synthetic key is a known multiple of the generator: tests/oracle_lib.py groth16_closed_form)."""
    O = _oracle()
    from snarkjs_amd import binfile
    from snarkjs_amd import groth16 as G
    from snarkjs_amd.workloads import synth_zkey
    mode = args.cpu_baseline_mode
    cid = 0 if args.curve == "bn128" else 1
    threads = O.threads()
    kw = _synth_kw(args)
    zkey_full, wtns_full = zkey, wtns
    rr, ss = 0x1234567, 0x7654321
    r_m, s_m = O.fr_e(cid, rr), O.fr_e(cid, ss)

    def device_proof(zk2, wt2):
        pk2 = G.ProvingKey(zk2)
        pts = pk2.prove_raw(binfile.read_wtns(wt2)["witness"], r_m, s_m)
        js = G.proof_to_json(G.raw_to_proof(pk2, *pts))
        pk2.release()
        return pts, js

    port, sample = None, None
    if mode in ("full", "port", "closed") or (mode == "ref" and (_ref_bundle() is None)):
        lg = args.cpu_log_n if args.cpu_log_n else (log_n_full if (threads >= 16 and mode != "closed" and log_n_full <= 20) else min(18, log_n_full))
        if lg != log_n_full:
            zkey, wtns = synth_zkey.make(args.curve, lg, seed=0xBA5E, **kw)
        zk, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)["witness"]
        t0 = time.perf_counter()
        ref = O.groth16_prove(cid, zk, w, r_m, s_m)
        dt = time.perf_counter() - t0
        scale = 1 << (log_n_full - lg)
        port = {"value": 1.0 / (dt * scale), "unit": "proofs/s", "cores": threads, "kind": "port",
                "sample": f"one full Groth16 proof at 2^{lg} constraints by oracle/zk_oracle.c ({threads} OpenMP threads, {dt:.1f} s wall)"
                          + (f", scaled linearly x{scale} to 2^{log_n_full} (optimistic: NTT is n log n)" if scale > 1 else ", no scaling")}
        sample = (zkey, wtns, ref, r_m, s_m)
    ref_wasm = (reference_wasm_baseline() or {}) if (cid == 0 and mode == "full") else {}
    if mode in ("full", "ref") and not args.no_ref_wasm:
        # the reference itself on this box's host cores, its proof for the same (r, s) compared with the device's
        cases = []
        for l2 in sorted({min(18, log_n_full), log_n_full} if mode == "full" else {log_n_full}):
            zk2, wt2 = (zkey_full, wtns_full) if l2 == log_n_full else synth_zkey.make(args.curve, l2, seed=0xBA5E, **kw)
            cases.append((l2, zk2, wt2, [r_m, s_m], device_proof(zk2, wt2)[1]))
        try:
            ref_wasm["same_box"] = reference_wasm_same_box("groth16", cases, curve=args.curve, warm=synth_zkey.make(args.curve, min(14, log_n_full), seed=0xBA5E, **kw), budget_s=args.ref_wasm_budget)
        except Exception as e:                               # noqa: BLE001 — a baseline leg must never cost the line
            ref_wasm["same_box"] = {"error": repr(e)[:300]}
    sb = (ref_wasm.get("same_box") or {}).get("at_bench_size")
    if sb:
        # the REFERENCE itself on this box's host cores is the baseline (north_star); the C port's figure stays beside it
        base = {"value": sb["proofs_per_s"], "unit": "proofs/s", "cores": sb["threads"], "kind": "reference",
                "sample": f"ONE snarkjs groth16.prove (the reference's bundle: WASM + {sb['threads']} worker threads, Node) of the bench's own 2^{log_n_full} key on this box's host, {sb['ms_per_proof'] / 1e3:.1f} s; "
                          f"its proof for the bench's (r, s) is {'bit-identical to' if sb.get('bit_identical_to_device_proof') else 'DIFFERENT from'} the device's",
                "bit_identical_to_device_proof": sb.get("bit_identical_to_device_proof"), "reference_wasm": ref_wasm}
        if port is not None:
            base["port"] = port
    else:
        if port is None:                                     # mode ref, the reference leg failed: the port after all
            args2 = argparse.Namespace(**dict(vars(args), cpu_baseline_mode="port"))
            base, sample = cpu_baseline(args2, zkey_full, wtns_full, log_n_full)
            base["reference_wasm"] = ref_wasm
            return base, sample
        base = dict(port, reference_wasm=ref_wasm)
    if mode == "closed":
        # the device's proof of the FULL-SIZE key against its closed form (no CPU MSM; buildABC / 6 NTTs / joinABC by the C restatement + O(n) sums)
        t0 = time.perf_counter()
        zkd, wv = binfile.read_groth16_zkey(zkey_full), binfile.read_wtns(wtns_full)["witness"]
        pts, _ = device_proof(zkey_full, wtns_full)
        a, b, cc = O.groth16_closed_form(cid, args.curve, zkd, wv, log_n_full, zkd["nPublic"], rr, ss, args.b_zero_every)
        want = (O.to_affine(cid, 1, O.generator_mul(cid, 1, a)), O.to_affine(cid, 2, O.generator_mul(cid, 2, b)), O.to_affine(cid, 1, O.generator_mul(cid, 1, cc)))
        base["closed_form_at_bench_size"] = bool(all(np.array_equal(x, y) for x, y in zip(pts, want)))
        base["closed_form_note"] = f"the device's proof of the bench's own 2^{log_n_full} key == the closed-form discrete logs of pi_a, pi_b, pi_c (tests/oracle_lib.py: groth16_closed_form), {time.perf_counter() - t0:.1f} s"
    return base, sample


T_PROCESS_START = time.perf_counter()


def other_configs(args):
    """The other BASELINE configs in the SAME driver run (rank 0, N = 1, after the headline line is complete): each is this script again as a child
    process with its own key — BLS12-381 Groth16 2^20 (configs[4]), PLONK 2^20 with addition gates (configs[3]), BN254 Groth16 2^24 on one GPU
    (configs[2] at N = 1), and the headline size on a circuit-shaped key (--coef-dist real) — >= 5 whole proofs each, the child's own line cut down to
    value / timing / roofline / int_alu / cpu_baseline. r06: every child carries its OWN same-box parity: the reference's WASM proof of that config's key
    (configs[4], real), the reference's plonk.prove at 2^14 / 2^16 (configs[3]), the closed form at full size (configs[2]; plus the same key opened and
    proved from Node). A wall-clock budget bounds the lot (--other-configs-budget): a config whose expected cost does not fit what is left is reported
    as skipped, never silently dropped."""
    import subprocess
    runs = [("configs[4]", ["--curve", "bls12381", "--steps", "8", "--warmup", "1", "--cpu-baseline-mode", "ref", "--no-napi-wall"], 42.0),
            ("configs[3]", ["--workload", "plonk", "--steps", "8", "--warmup", "1", "--no-napi-wall"], 46.0),
            ("configs[1] on a circuit-shaped key", ["--coef-dist", "real", "--witness", "mixed", "--steps", "10", "--warmup", "1", "--cpu-baseline-mode", "ref", "--no-napi-wall"], 22.0),
            ("configs[2] at N=1", ["--log-n", "24", "--steps", "5", "--warmup", "1", "--repeats", "1", "--cpu-baseline-mode", "closed", "--napi-wall-reps", "2"], 92.0)]
    t_start, res = time.perf_counter(), {}
    # two bounds: the budget of this leg, and the whole run's (--total-budget, from process start: the headline part takes 80 - 100 s of it) — whichever ends first
    deadline = min(t_start + args.other_configs_budget, T_PROCESS_START + args.total_budget)
    for tag, extra, expect_s in runs:
        left = deadline - time.perf_counter()
        if left < expect_s:
            res[tag] = {"skipped": f"budget: {left:.0f} s left (--other-configs-budget {args.other_configs_budget:.0f} / --total-budget {args.total_budget:.0f}), this config needs ~{expect_s:.0f} s (key synthesis + load + proofs + its reference leg)"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-other-configs"] + extra
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        if args.no_ref_wasm:
            cmd.append("--no-ref-wasm")
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left, env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
        except subprocess.TimeoutExpired:
            res[tag] = {"error": f"timed out after {left:.0f} s"}
            continue
        wall = time.perf_counter() - t0
        line = next((ln for ln in reversed(r.stdout.strip().splitlines()) if ln.startswith("{")), None)
        if r.returncode != 0 or line is None:
            res[tag] = {"error": (r.stderr or r.stdout)[-300:], "wall_s": round(wall, 1)}
            continue
        d = json.loads(line)
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "pipeline_depth", "proofs_in_flight", "latency_ms_single_proof", "stages_ms",
                "accum_kernel_ms", "roofline", "int_alu", "repeats", "cpu_baseline", "wall_through_napi", "coef_layout")
        res[tag] = {k: d[k] for k in keep if k in d}
        cb = res[tag].get("cpu_baseline") or {}
        # the parity verdict of this config, on top: same-box reference proof / closed form / restatement sample
        res[tag]["parity_in_run"] = {k: cb[k] for k in ("bit_identical_to_device_proof", "closed_form_at_bench_size", "parity_on_sample") if k in cb}
        if isinstance(cb.get("reference_wasm"), dict):       # keep the line readable: the build-container history stays in the headline's cpu_baseline only
            cb["reference_wasm"] = {k: v for k, v in cb["reference_wasm"].items() if k == "same_box"}
        res[tag]["wall_s"] = round(wall, 1)
        res[tag]["command"] = "python bench.py " + " ".join(cmd[2:])
    return res


def napi_wall(zkey, wtns, reps=5, draws=None, dropin=False, curve="bn128"):
    """Wall time through the N-API addon with host buffers (SURVEY.md 8d timing protocol): tools/napi_wall.js under Node. The key goes through a
    FILE (memory-backed where there is room) and Node opens it by offset, in pages (js/groth16_native.js: openZkey) — also a 2^24 key of 9.4 GB.
    draws = (r_mont, s_mont): the Node proof's hash for them comes back in `proof_sha256`. dropin: tools/dropin_wall.js too — unmodified snarkjs with
    the curve patched by register.js (JS buildABC1, every bulk call through the addon): what `snarkjs.groth16.prove` costs a drop-in user."""
    import shutil
    import subprocess
    import tempfile
    node = shutil.which("node")
    addon = os.path.join(ROOT, "snarkjs_amd", "napi", "zkmi_napi.node")
    if node is None or not os.path.exists(addon):
        return {"skipped": "node or the built addon is missing"}
    base = None
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2 * (len(zkey) + len(wtns)) + (64 << 20):
            base = "/dev/shm"
    except OSError:
        pass
    with tempfile.TemporaryDirectory(dir=base) as td:
        zf, wf = os.path.join(td, "k.zkey"), os.path.join(td, "k.wtns")
        with open(zf, "wb") as fh:
            fh.write(zkey)
        with open(wf, "wb") as fh:
            fh.write(wtns)
        dh = ",".join(bytes(d).hex() for d in draws) if draws is not None else ""
        r = subprocess.run([node, "--max-old-space-size=24000", os.path.join(ROOT, "tools", "napi_wall.js"), zf, wf, str(reps)] + ([dh] if dh else []), capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
        out = json.loads(r.stdout.strip().splitlines()[-1])
        if dropin and _ref_bundle() is not None and dh:
            try:
                r2 = subprocess.run([node, "--harmony-optional-chaining", "--harmony-nullish", "--max-old-space-size=24000", os.path.join(ROOT, "tools", "dropin_wall.js"), zf, wf, dh, "3"],
                                    capture_output=True, text=True, timeout=600, env=dict(os.environ, SNARKJS_REF_BUNDLE=_ref_bundle(), CURVE=curve))
                out["dropin"] = json.loads(r2.stdout.strip().splitlines()[-1]) if r2.returncode == 0 else {"error": (r2.stderr or r2.stdout)[-300:]}
            except Exception as e:                           # noqa: BLE001
                out["dropin"] = {"error": repr(e)[:300]}
        elif dropin:
            out["dropin"] = {"skipped": "the reference bundle (oracle/_ref) is not on this box"}
    return out


def bench_plonk(args, rank, world, dist, torch):
    """BASELINE configs[3]: BN254 PLONK prove at 2^log_n constraints on a synthetic VALID key (snarkjs_amd/workloads/synth_plonk.py); one proof
    stream per GPU. Key and witness are resident in HBM when the timed region starts (PlonkWitness; the figure with the 32 MB witness parsed and
    uploaded per proof, as the reference reads it from a file, is reported beside: latency_ms_with_witness_upload). The circuit has addition gates
    (--plonk-additions per multiplication gate): the internal signals are per-proof work and are computed on the device inside every timed proof."""
    from snarkjs_amd.workloads import synth_plonk
    from snarkjs_amd import fflonk, plonk, zkmi
    lg = args.log_n
    proto = args.workload
    if proto == "fflonk":               # not a BASELINE config; same kernels, MSMs over 8n / 16n coefficients
        zkey, wtns = synth_plonk.make_fflonk(lg, seed=3 + rank, additions=args.plonk_additions)
        key, plonk = fflonk.FflonkKey(zkey), fflonk
    else:
        zkey, wtns = synth_plonk.make("bn128", lg, seed=3 + rank, additions=args.plonk_additions)
        key = plonk.PlonkKey(zkey)
    wtns_host = wtns
    if hasattr(plonk, "PlonkWitness"):
        wtns = plonk.PlonkWitness(key, wtns)                            # inputs resident in HBM when the timed region starts (the upload-inclusive figure is reported beside)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        plonk.prove(key, wtns)
    two = args.pipeline == 2 and hasattr(plonk, "prove_many")          # throughput mode: two proofs in flight from this one host thread
    if two:
        plonk.prove_many(key, [wtns, wtns])                             # size the second slot's buffers outside the timed region
    per_proof = []
    import gc
    gc.collect()
    gc.disable()                                                        # a cyclic collection of the Python host (~45 ms) is not part of a proof
    barrier()
    t0 = time.perf_counter()
    if two:
        res = plonk.prove_many(key, [wtns] * args.steps)[-1]            # EXACTLY args.steps whole proofs
    else:
        for _ in range(args.steps):
            tp = time.perf_counter()
            res = plonk.prove(key, wtns)
            per_proof.append(time.perf_counter() - tp)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    lat = []
    for _ in range(6 if two else 0):                                    # single-proof latency beside the throughput figure
        tl = time.perf_counter()
        plonk.prove(key, wtns)
        lat.append(time.perf_counter() - tl)
    lat_up = []
    for _ in range(4 if wtns is not wtns_host else 0):                  # the same with the witness parsed and uploaded per proof (32 MB at 2^20 over PCIe)
        tl = time.perf_counter()
        plonk.prove(key, wtns_host)
        lat_up.append(time.perf_counter() - tl)
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n = 1 << lg
        # dominant kernel of a PLONK proof: the bucket accumulations of the commitments (G1 MSMs against the resident SRS table);
        # job slot 0 holds the last one (n + O(1) terms). Live HIP-event time around that launch on the library stream.
        acc_ms = zkmi.lib().zkmi_msm_accum_ms(0)
        alg = 96 * n                                            # 64-byte affine base + 32-byte scalar per term
        # mixed additions of that launch, counted on the device during one more proof (zkmi_msm_stats), and its time in that same proof
        int_alu = None
        try:
            Lz = zkmi.lib()
            zkmi.check(Lz.zkmi_msm_stats(1))
            plonk.prove(key, wtns)
            adds, acc2 = float(Lz.zkmi_msm_accum_additions(0)), float(Lz.zkmi_msm_accum_ms(0))
            zkmi.check(Lz.zkmi_msm_stats(0))
            if adds > 0 and acc_ms and acc_ms > 0:
                vpa = VALU_PER_ADD["bn128"]["g1"]
                int_alu = {"unit": "Gmul/s", "mixed_additions": int(adds), "field_muls": int(adds * 10), "achieved": round(adds * 10 / (acc_ms * 1e-3) / 1e9, 1), "peak": FIELD_MUL_PEAK_G["bn128"],
                           "frac": round(adds * 10 / (acc_ms * 1e-3) / 1e9 / FIELD_MUL_PEAK_G["bn128"], 4), "limb_form": "9 x 29-bit unsaturated limbs",
                           "valu_issue": {"valu_instr_per_addition": vpa, "achieved": round(adds * vpa / 64 / (acc_ms * 1e-3) / 1e9, 1), "peak": VALU_ISSUE_PEAK_G, "unit": "G wave-instr/s",
                                          "frac": round(adds * vpa / 64 / (acc_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_G, 4)},
                           "note": "the accumulation of the proof's last commitment: mixed additions counted on the device x 10 field multiplications / its launch time (HIP events, no counting kernel beside it)"}
        except Exception as e:                                  # noqa: BLE001 — a diagnostic, never the line
            int_alu = {"error": repr(e)[:200]}
        roof = None
        if acc_ms and acc_ms > 0:
            kname = "k_msm_accum29<Bn254Fq>" if os.environ.get("ZKMI_R29", "1") != "0" else "k_msm_accum<Fp<Bn254Fq>>"
            roof = {"bound": "hbm", "kernel": kname + " (last commitment, n terms)", "achieved": round(alg / (acc_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(alg / (acc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(f"{proto}:bn128:2^{lg}:additions={int(key.nAdditions)}", kname), "kernel_ms": round(acc_ms, 4), "algorithmic_bytes": alg,
                    "note": "integer-ALU-bound (256-bit Montgomery carry chains, no MFMA); traffic: average over the accumulation launches of a proof (9 commitments of n .. n + 6 terms)"}
        out = {
            "metric": f"{proto}_proofs_per_sec", "value": round(world * args.steps / elapsed, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"BN254 {proto.upper()} prove, 2^{lg} constraints, synthetic valid key" + (" (BASELINE configs[3])" if proto == "plonk" else "") + "; key " + ("and witness resident" if wtns is not wtns_host else "resident, witness uploaded per proof"),
                       "curve": "bn128", "log_n": lg, "n_constraints": int(key.nConstraints), "n_additions": int(key.nAdditions),
                       "additions": f"{args.plonk_additions} addition gate(s) per multiplication gate, internal signals in chains of that depth (plonk_setup.js reduceCoefs on an `x^2 + b` circuit gives 1); calculateAdditions (plonk_prove.js:174-204) runs on the device INSIDE every timed proof",
                       "parallelism": f"replica x{world}"},
            "roofline": roof, "int_alu": int_alu,
            "proofs_in_flight": 2 if two else 1, "host_gc": "Python's cyclic collector is off inside the timed region (timeit's convention)", "latency_ms_single_proof": round(min(lat) * 1e3, 3) if lat else None, "latency_ms_serial_proofs": [round(x * 1e3, 2) for x in lat], "timed_region_ms_per_proof": [round(x * 1e3, 2) for x in per_proof],
            "latency_ms_with_witness_upload": [round(x * 1e3, 2) for x in lat_up],
            "public_signal": res["publicSignals"][0][:24] + "..."}
        same_box = None
        if world == 1 and not args.no_cpu_baseline and not args.no_ref_wasm:
            # the reference's own prover on this box's host cores at bounded domains (a reference PLONK proof at 2^20 takes minutes: mostly single-threaded
            # JS loops), its proofs for the same draws compared with the device's; scaled linearly to the bench size and labelled as such
            mod = fflonk if proto == "fflonk" else plonk
            nd = 9 if proto == "fflonk" else 11
            fld = plonk._Field(0) if hasattr(plonk, "_Field") else fflonk._Field(0)
            draws = [bytes(fld.mont(4242 + 17 * i)) for i in range(nd)]
            cases = []
            for l2 in sorted({min(14, lg), min(16, lg)}):
                zk2, wt2 = (synth_plonk.make_fflonk(l2, seed=3, additions=args.plonk_additions) if proto == "fflonk" else synth_plonk.make("bn128", l2, seed=3, additions=args.plonk_additions))
                dev = json.dumps(mod.prove(zk2, wt2, blinding_mont=draws)["proof"], separators=(",", ":"))
                cases.append((l2, zk2, wt2, draws, dev))
            try:
                same_box = reference_wasm_same_box(proto, cases, budget_s=min(args.ref_wasm_budget, 120.0))
            except Exception as e:                           # noqa: BLE001
                same_box = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline and proto == "fflonk":
            out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "reference", "sample": "no live CPU leg for FFLONK: see reference_wasm",
                                   "reference_wasm": reference_wasm_baseline_plonk(proto, lg)}
            _plonk_same_box_into(out["cpu_baseline"], same_box, proto, lg)
        if world == 1 and not args.no_cpu_baseline and proto == "plonk":
            # CPU port: the Python restatement of src/plonk_prove.js (oracle/plonk_oracle.py, pure-Python field loops, 1 thread) on a
            # small valid key, scaled linearly — the reference's own PLONK prover spends most of its time in single-threaded JS loops too
            t = os.path.join(ROOT, "tests")
            if t not in sys.path:
                sys.path.insert(0, t)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import plonk_oracle
            slg = 8
            zk_s, wt_s = synth_plonk.make("bn128", slg, seed=3)
            f = plonk._Field(0)
            blind = [bytes(f.mont(900 + i)) for i in range(11)]
            t0 = time.perf_counter()
            ref_proof, _ = plonk_oracle.plonk_prove(zk_s, wt_s, blind)
            dt = time.perf_counter() - t0
            key_s = plonk.PlonkKey(zk_s)
            got = plonk.prove(key_s, wt_s, blinding_mont=blind)
            key_s.release()
            # value: null — a pure-Python proof scaled x4096 is not a baseline anyone can read; what the live leg still provides is the
            # parity check of the device proof against the restatement on that sample, and the reference's own WASM timings stand beside it
            out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 1, "kind": "port",
                                   "sample": f"one PLONK proof at 2^{slg} constraints by oracle/plonk_oracle.py (pure Python, 1 thread, {dt:.1f} s): used as the in-run parity check only, not scaled to the bench size",
                                   "parity_on_sample": bool(got["proof"] == ref_proof), "reference_wasm": reference_wasm_baseline_plonk(proto, lg)}
            _plonk_same_box_into(out["cpu_baseline"], same_box, proto, lg)
        out["box_calibration"] = box_calibration(zkmi.lib())
        drain_c_stdout_to_stderr()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        drain_c_stdout_to_stderr()


def multi_rank_extras(args, rank, world, dist, torch, barrier, cid, q8, lg, r_m, s_m, zkey, wtns, proof_pts):
    """What the ranks do TOGETHER (north star: "G1 MSM shards across the GPUs ... final RCCL reduce"; BASELINE configs[2]); reported next to the
    replica metric, never part of `value`:
      (a) one large G1 MSM sharded by base-index range over the ranks, bases resident as window tables on their rank (how a key shard holds
          them): every rank runs the device Pippenger on its slice, ONE all_gather of the partial points over RCCL, local fold;
      (b) ONE Groth16 proof stream over all ranks at --log-n and at --configs2-log-n (2^24 = BASELINE configs[2]) on key shards, with every
          rank's stage timeline (snarkjs_amd/distributed.py: groth16_prove_sharded)."""
    from snarkjs_amd import groth16, zkmi, binfile
    from snarkjs_amd import distributed as D
    from snarkjs_amd.workloads import synth, synth_zkey
    L = zkmi.lib()
    t_extras = time.perf_counter()
    res = {"world_size_rccl": dist.get_world_size(), "backend": dist.get_backend()}

    def max_over_ranks(sec):
        t = torch.tensor([sec], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- (a) sharded MSM over resident tables ----
    n_tot = 1 << (lg + 2)
    lo, hi = D.shard_range(n_tot, rank, world)
    k = hi - lo
    d_bs = zkmi.DeviceBuffer(max(k, 1) * 2 * q8)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(cid, 1, max(k, 1), 7 + rank, 11, d_bs.ptr))
    d_ss = zkmi.DeviceBuffer.from_host(synth.elems(0xD157 + rank, max(k, 1)))
    th = C.c_uint64(0)
    zkmi.check(L.zkmi_msm_table_build(cid, 1, d_bs.ptr, max(k, 1), C.byref(th)))
    d_bs.free()

    class _Cv:
        id = cid
        G1 = G2 = None

    def shard_msm(_b, _s):
        o = np.zeros(3 * q8, np.uint8)
        if k:
            zkmi.check(L.zkmi_msm_table_dev(th, d_ss.ptr, k, 32, zkmi.ptr(o)))
        return o
    D.msm_sharded(_Cv, 1, None, None, compute=shard_msm)
    barrier()
    ts = time.perf_counter()
    reps = 3
    for _ in range(reps):
        D.msm_sharded(_Cv, 1, None, None, compute=shard_msm)
    barrier()
    sec = max_over_ranks(time.perf_counter() - ts)
    res.update({"terms": n_tot, "ms": round(sec / reps * 1e3, 3), "mscalar_per_s": round(n_tot * reps / sec / 1e6, 2), "bases": "resident window tables (R'-form), one shard per rank",
                "exchange": "all_gather of %d x %d-byte partial points + host fold" % (world, 3 * q8)})
    zkmi.check(L.zkmi_msm_table_release(th))
    d_ss.free()

    # ---- (b) one proof over all ranks ----
    def one_proof_over_all_ranks(lgp, zkey0, wtns0, reference_pts):
        pks = groth16.ProvingKey(zkey0, shard=(rank, world))
        d_w0 = zkmi.DeviceBuffer.from_host(binfile.read_wtns(wtns0)["witness"])
        for _ in range(2):
            sh_proof = D.groth16_prove_sharded(pks, None, r_m, s_m, d_witness=d_w0.ptr)
        barrier()
        ts = time.perf_counter()
        reps = max(3, args.steps // 2) if lgp <= 20 else 3
        for _ in range(reps):
            sh_proof = D.groth16_prove_sharded(pks, None, r_m, s_m, d_witness=d_w0.ptr)
        barrier()
        sec = max_over_ranks(time.perf_counter() - ts)
        tl = {}
        D.groth16_prove_sharded(pks, None, r_m, s_m, d_witness=d_w0.ptr, timeline=tl)      # one more, with this rank's host-clock stage marks
        keys = ["chains_done", "w_enqueued", "slices_requested_done", "exchange_done", "sums_done", "gathered"]
        t = torch.tensor([tl.get(kk, 0.0) for kk in keys], device="cuda", dtype=torch.float64)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        stage = pks.stage_ms()
        out = {"ms_per_proof": round(sec / reps * 1e3, 3), "proofs_per_s": round(reps / sec, 3), "log_n": lgp, "scaling": "strong",
               "timeline_ms_per_rank": {str(j): dict(zip(keys, [round(float(x), 3) for x in allt[j].tolist()])) for j in range(world)},
               "rank0_device_stage_ms": {kk: round(v, 3) for kk, v in stage.items()},
               "exchange": "chain-parallel transforms (chain c on rank c %% world), point-to-point slices of the chain outputs over RCCL send/recv (%d bytes leave each chain owner), witness-side MSMs enqueued underneath, all_gather of %d x %d-byte MSM sums + host fold" % ((1 << lgp) * 32, world, 21 * q8)}
        if reference_pts is not None and rank == 0:
            out["equals_single_device_proof"] = bool(all(np.array_equal(a, b) for a, b in zip(sh_proof, reference_pts)))
        pks.release(); d_w0.free()
        return out

    if rank == 0:
        zkey0, wtns0 = zkey, wtns
    else:
        zkey0, wtns0 = synth_zkey.make(args.curve, lg, seed=0x5EED, **_synth_kw(args))
    res["groth16_one_proof_over_all_ranks"] = one_proof_over_all_ranks(lg, zkey0, wtns0, proof_pts)
    del zkey0, wtns0
    lg2 = args.configs2_log_n
    if lg2 and lg2 != lg:
        # BASELINE configs[2]: "BN254 Groth16 prove, 2^24 constraints, G1 MSM sharded across 8 x MI355X via RCCL/xGMI". Every rank synthesises the SAME
        # key (same seed) and keeps its shard. Skipped with a note when the host cannot hold `world` synthetic keys at once.
        # r05: ONE rank synthesises the key and the others map it from shared memory (r04: every rank synthesised its own copy — 8 x 9.4 GB of host
        # work at once on the first real 8-rank run); a wall-clock budget (--extras-budget) keeps this extra from running into the driver's timeout.
        import mmap
        import shutil
        key_bytes = ((5 * 2 + 4) * q8 + 110) << lg2
        spent = time.perf_counter() - t_extras
        expect = 25.0 + 8.0 * (1 << max(0, lg2 - 20))                     # key synthesis + load + 6 proofs: ~150 s at 2^24 on one rank
        go = torch.tensor([1.0 if spent + expect <= args.extras_budget else 0.0], device="cuda", dtype=torch.float64)
        dist.all_reduce(go, op=dist.ReduceOp.MIN)
        place = None
        if rank == 0:
            for dname in ("/dev/shm", "/tmp"):
                try:
                    if os.path.isdir(dname) and shutil.disk_usage(dname).free > key_bytes * 1.3 + (64 << 20 << max(0, lg2 - 20)):
                        place = dname
                        break
                except OSError:
                    pass
        ok = torch.tensor([1.0 if (rank != 0 or place is not None) else 0.0], device="cuda", dtype=torch.float64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(go.item()) < 1.0:
            res["groth16_configs2"] = {"skipped": "budget: %.0f s of the %.0f s for the multi-rank extras are spent, a 2^%d key needs ~%.0f s more (--extras-budget)" % (spent, args.extras_budget, lg2, expect)}
        elif float(ok.item()) < 1.0:
            res["groth16_configs2"] = {"skipped": "no shared place (/dev/shm, /tmp) with %d GB free for the synthetic 2^%d key" % ((key_bytes * 13 // 10) >> 30, lg2)}
        else:
            tk = time.perf_counter()
            names = [None, None]
            if rank == 0:
                zkey2, wtns2 = synth_zkey.make(args.curve, lg2, seed=0x5EED, **_synth_kw(args))
                base = os.path.join(place, "zkmi_bench_%d_%s_k%d" % (os.getpid(), os.environ.get("MASTER_PORT", "0"), lg2))
                names = [base + ".zkey", base + ".wtns"]
                for nm, blob in zip(names, (zkey2, wtns2)):
                    with open(nm, "wb") as fh:
                        fh.write(blob)
                del zkey2, wtns2
            dist.broadcast_object_list(names, src=0)
            maps = []
            for nm in names:
                fh = open(nm, "rb")
                maps.append(mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ))
                fh.close()
            try:
                r2 = one_proof_over_all_ranks(lg2, maps[0], maps[1], None)
                r2["config"] = "BASELINE configs[2] (2^%d constraints, one proof stream over %d rank(s), key sharded by base-index range)" % (lg2, world)
                r2["key_synthesis_and_load_s"] = round(time.perf_counter() - tk - r2["ms_per_proof"] * 6e-3, 1)
                r2["key_source"] = "synthesised by rank 0, mapped by every rank from %s" % place
                res["groth16_configs2"] = r2
            finally:
                dist.barrier()
                if rank == 0:
                    for nm in names:
                        try:
                            os.unlink(nm)
                        except OSError:
                            pass
    return res



def preflight(args, rank, local_rank, world, dist, torch):
    """First contact with a node (VERDICT r05 #5a): what every rank sees, before anything is proved. One JSON line from rank 0."""
    import ctypes
    info = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(local_rank), "devices_visible": torch.cuda.device_count(),
            "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES")}
    free, total = torch.cuda.mem_get_info(local_rank)
    info["hbm_free_gb"], info["hbm_total_gb"] = round(free / 2**30, 1), round(total / 2**30, 1)
    # peer-access row of this rank's device (hipDeviceCanAccessPeer): the Node shard driver pulls chain slices device to device over these links
    info["peer_access"] = [bool(torch.cuda.can_device_access_peer(local_rank, j)) if j != local_rank else True for j in range(torch.cuda.device_count())]
    from snarkjs_amd import zkmi
    info["zkmi"] = zkmi.lib().zkmi_version().decode()
    info["zkmi_device_count"] = int(zkmi.lib().zkmi_device_count())
    ok = True
    if dist is not None:
        t = torch.tensor([float(rank + 1)], device="cuda", dtype=torch.float64)
        dist.all_reduce(t)                                   # one collective over RCCL: the communicator works
        info["allreduce_ok"] = bool(abs(float(t.item()) - world * (world + 1) / 2) < 1e-9)
        ok = info["allreduce_ok"]
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
    else:
        gathered = [info]
    if rank == 0:
        out = {"preflight": True, "n_gpus": world, "world_size_rccl": dist.get_world_size() if dist is not None else 1, "backend": dist.get_backend() if dist is not None else None,
               "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "ranks": gathered, "ok": bool(ok and all(g is not None for g in gathered)),
               "distinct_devices": len({(g or {}).get("device") for g in gathered}) == world}
        drain_c_stdout_to_stderr()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        drain_c_stdout_to_stderr()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cpu-log-n", type=int, default=0, help="size of the CPU-baseline sample (0 = auto: the bench's own key with >= 16 host threads, else 2^18)")
    ap.add_argument("--b-zero-every", type=int, default=0, help="k: every k-th B1/B2 base is the point at infinity (B density 1 - 1/k); 0 = dense B sections (SURVEY.md 8d recipe)")
    ap.add_argument("--no-napi-wall", action="store_true")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2], help="proofs in flight per GPU (2: the tail of proof k overlaps the front of proof k+1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=3, help="how often the timed region is run (value = the first; min / median / max of all reported)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs[4] / [3] / [2] (child runs after the headline line; default run at N = 1 only)")
    ap.add_argument("--other-configs-budget", type=float, default=330.0, help="wall-clock seconds the other configs may take together")
    ap.add_argument("--total-budget", type=float, default=335.0, help="wall-clock seconds the whole default run may take (headline + other configs); a config that does not fit is reported as skipped")
    ap.add_argument("--no-ref-wasm", action="store_true", help="skip the reference's own WASM prover on this box's host cores (cpu_baseline.reference_wasm.same_box)")
    ap.add_argument("--ref-wasm-budget", type=float, default=200.0, help="seconds the same-box WASM leg may take; a size is skipped when ~5x the previous one does not fit")
    ap.add_argument("--witness", default="uniform", choices=["uniform", "mixed"])
    ap.add_argument("--curve", default="bn128", choices=["bn128", "bls12381"])
    ap.add_argument("--extras-budget", type=float, default=420.0, help="multi-rank runs only: wall-clock seconds the extras beside the replica line may take; the 2^24 proof is skipped (and says so) when it would not fit")
    ap.add_argument("--configs2-log-n", type=int, default=24, help="multi-rank runs only: size of the one-proof-over-all-ranks extra of BASELINE configs[2] (0 = skip)")
    ap.add_argument("--plonk-additions", type=int, default=1, help="PLONK / FFLONK workloads: addition gates (and internal signals) per multiplication gate of the synthetic circuit; 0 = none")
    ap.add_argument("--workload", default="groth16", choices=["groth16", "plonk", "fflonk"], help="plonk = BASELINE configs[3] (not the default metric)")
    ap.add_argument("--coef-dist", default="flat", choices=["flat", "real"], help="shape of the coefficient section of the synthetic Groth16 key: flat = one term per row (n_coef = 2.0 n); real = a compiled circuit's "
                    "(n_coef ~ 2.9 n, heavy-tailed rows up to 10^5 terms, B density 0.4; use with --witness mixed)")
    ap.add_argument("--cpu-baseline-mode", default="full", choices=["full", "ref", "port", "closed"], help="CPU legs of a Groth16 line (see cpu_baseline): full = C restatement + the reference's WASM prover at 2^18 and the bench size; "
                    "ref = the reference at the bench size only; port = the C restatement only; closed = restatement at 2^18 + the closed form of the full-size proof (2^24)")
    ap.add_argument("--napi-wall-reps", type=int, default=5)
    ap.add_argument("--preflight", action="store_true", help="every rank reports its device ordinal, free HBM, the peer-access row of its device and the RCCL world, rank 0 prints ONE JSON line, nothing is proved: "
                    "cheap first contact with a multi-GPU node before the real run")
    args = ap.parse_args()
    relaunch_if_needed(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:
        os.dup2(2, 1)                                       # only rank 0 owns stdout (the one JSON line)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ZKMI_FORCE_DIST"):      # ZKMI_FORCE_DIST: exercise the RCCL path with a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                      # ZKMI_FORCE_DIST without a launcher
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29517")):
                os.environ.setdefault(k, v)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from snarkjs_amd import groth16, zkmi, binfile
    from snarkjs_amd.workloads import synth, synth_zkey
    zkmi.init(local_rank)
    L = zkmi.lib()
    if args.preflight:
        return preflight(args, rank, local_rank, world, dist, torch)

    lg = args.log_n
    if args.workload in ("plonk", "fflonk"):
        return bench_plonk(args, rank, world, dist, torch)
    zkey, wtns = synth_zkey.make(args.curve, lg, seed=0x5EED + rank, **_synth_kw(args))
    cid = 0 if args.curve == "bn128" else 1
    q8 = 32 if cid == 0 else 48
    pk = groth16.ProvingKey(zkey)
    zk = pk.zk
    w = binfile.read_wtns(wtns)["witness"]
    d_w = zkmi.DeviceBuffer.from_host(w)
    R = synth_zkey.PRIMES[args.curve][2]
    mont = lambda v: np.frombuffer(((v << 256) % R).to_bytes(32, "little"), np.uint8).copy()
    r_m, s_m = mont(0x1234567), mont(0x7654321)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        return pk.prove_raw(None, r_m, s_m, d_witness=d_w.ptr)

    for _ in range(args.warmup):
        step()
    if args.pipeline == 2:                                   # size the second slot's buffers outside the timed region
        pk.submit(d_w.ptr, 1)
        pk.collect(1, r_m, s_m)
    stage_acc, accum_ms = {}, {k: [] for k in range(5)}

    def timed_region():
        """EXACTLY args.steps whole proofs between two barriers; returns (seconds, last proof)"""
        barrier()
        t0 = time.perf_counter()
        if args.pipeline == 1:
            for _ in range(args.steps):
                pts = step()
        else:
            # two in flight: proof i is enqueued before proof i-1 is collected; every proof is collected (folded, blinded, normalised) inside the
            # timed region
            for i in range(args.steps):
                pk.submit(d_w.ptr, i & 1)
                if i:
                    pts = pk.collect((i - 1) & 1, r_m, s_m)
            pts = pk.collect((args.steps - 1) & 1, r_m, s_m)
        barrier()
        return time.perf_counter() - t0, pts

    elapsed, proof_pts = timed_region()                     # THE timed region: `value` comes from this one
    # the same region again (args.repeats - 1 times): the spread of the line on this box, reported beside `value`, never instead of it
    region_s = [elapsed]
    for _ in range(max(0, args.repeats - 1)):
        region_s.append(timed_region()[0])
    if dist is not None and len(region_s) > 1:
        t = torch.tensor(region_s, device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(x) for x in t.tolist()]
        region_s[0] = elapsed                               # reduced below with the historical code path
    # stage / kernel times and single-proof latency: a few serial proofs after the timed region
    lat = []
    for _ in range(4):
        tl = time.perf_counter()
        serial_pts = step()
        lat.append(time.perf_counter() - tl)
        for k, v in pk.stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v / 4 * args.steps
        for k in range(5):
            accum_ms[k].append(L.zkmi_msm_accum_ms(k))
    assert all(np.array_equal(a, b) for a, b in zip(serial_pts, proof_pts)), "pipelined and serial proofs differ"
    # mixed additions per accumulation launch, counted ON THE DEVICE from the digit lists of one more serial proof (zkmi_msm_stats)
    zkmi.check(L.zkmi_msm_stats(1))
    step()
    additions = {k: L.zkmi_msm_accum_additions(k) for k in range(5)}
    zkmi.check(L.zkmi_msm_stats(0))
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stages = {k: v / args.steps for k, v in stage_acc.items()}

    # ---- one large G1 MSM sharded by base-index range over the ranks (north star: "G1 MSM shards across the GPUs ... final
    # RCCL reduce"): every rank runs the device Pippenger on its slice, one all_gather of the partial points, local fold.
    sharded = None
    extras_failed = False
    if dist is not None:
        # The replica metric above is complete at this point; the multi-rank extras below must never cost the line: any error in them is
        # reported inside the line and the process then leaves without further collectives.
        try:
            sharded = multi_rank_extras(args, rank, world, dist, torch, barrier, cid, q8, lg, r_m, s_m, zkey, wtns, proof_pts)
        except Exception as e:                               # noqa: BLE001 — report, do not die
            extras_failed = True
            sharded = {"error": repr(e)[:400]}

    out = None
    if rank == 0:
        # ---- sub-metrics (outside the timed region): G1 MSM and NTT at the same size, device-event time ----
        n = 1 << lg
        d_b = zkmi.DeviceBuffer(n * 2 * q8)
        zkmi.check(L.zkmi_gen_geometric_bases_dev(cid, 1, n, 7, 11, d_b.ptr))
        d_s = zkmi.DeviceBuffer.from_host(synth.elems(0x5EED, n))
        jac = np.zeros(3 * q8, np.uint8)
        ts, ta = [], []
        for _ in range(4):
            zkmi.check(L.zkmi_msm_dev(cid, 1, d_b.ptr, d_s.ptr, n, 32, zkmi.ptr(jac)))
            ts.append(L.zkmi_last_kernel_ms()); ta.append(L.zkmi_msm_accum_ms(0))
        msm_ms, msm_acc_ms = min(ts[1:]), min(ta[1:])
        # the same MSM with the bases resident as pre-computed window tables (how zkey sections / SRS points are used per proof)
        th = C.c_uint64(0)
        zkmi.check(L.zkmi_msm_table_build(cid, 1, d_b.ptr, n, C.byref(th)))
        tt = []
        for _ in range(4):
            zkmi.check(L.zkmi_msm_table_dev(th, d_s.ptr, n, 32, zkmi.ptr(jac)))
            tt.append(L.zkmi_last_kernel_ms())
        msm_tab_ms = min(tt[1:])
        zkmi.check(L.zkmi_msm_table_release(th))
        d_o = zkmi.DeviceBuffer(n * 32)
        tn = []
        for _ in range(5):
            zkmi.check(L.zkmi_ntt_dev(cid, d_s.ptr, d_o.ptr, lg, 0, None, None))
            tn.append(L.zkmi_last_kernel_ms())
        ntt_ms = min(tn[1:])
        # ---- roofline of the dominant kernel: bucket accumulation of the G2 MSM (B2) ----
        m = zk["nVars"]
        acc = {k: float(np.mean(v)) for k, v in accum_ms.items()}
        fq = "Bn254Fq" if cid == 0 else "Bls12381Fq"
        b1, b2 = 2 * q8 + 32, 4 * q8 + 32                     # affine base + 32-byte scalar per term
        r29 = os.environ.get("ZKMI_R29", "1") != "0" and not (cid == 1 and os.environ.get("ZKMI_R29_BLS", "1") == "0")     # accumulation kernels on unsaturated limbs (msm29.cuh)
        k1, k2 = (f"k_msm_accum29<{fq}>", f"k_msm_accum29_g2{'s' if (G2_SPLIT if cid == 0 else G2_SPLIT_BLS) else ''}<{fq}>") if r29 else (f"k_msm_accum<Fp<{fq}>>", f"k_msm_accum<Fp2<{fq}>>")
        names = {0: (f"{k1} (A)", b1), 1: (f"{k1} (B1)", b1), 2: (f"{k2} (B2)", b2), 3: (f"{k1} (C)", b1), 4: (f"{k1} (H)", b1)}
        dom = max(acc, key=lambda k: acc[k])
        units = {0: m, 1: m, 2: m, 3: m - zk["nPublic"] - 1, 4: zk["domainSize"]}[dom]
        alg_bytes = names[dom][1] * units                      # B/term (affine base + 32-B scalar) x terms
        achieved = alg_bytes / (acc[dom] * 1e-3) / 1e9
        # HBM bytes of that launch from separate rocprofv3 --pmc passes of THIS workload (tools/publish_profiles.py tags the file);
        # a file collected on another workload (size / curve / B density) is not used: traffic stays null
        wl_tag = f"groth16:{args.curve}:2^{lg}:" + (f"b_zero_every={args.b_zero_every}" if args.coef_dist == "flat" else "coef_dist=real") + f":{args.witness}"
        traffic = pmc_traffic(wl_tag, names[dom][0].split(" ")[0])
        roof = {"bound": "hbm", "kernel": names[dom][0], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "kernel_ms": round(acc[dom], 4),
                "algorithmic_bytes": alg_bytes,
                "note": "integer-ALU-bound (256-bit Montgomery carry chains, no MFMA): see int_alu"}
        # the resource that actually binds: field multiplications. Every non-zero signed digit of a scalar whose base is not at infinity is one
        # mixed addition (8M + 2S; x3 base-field products per Fq2 product in G2, Karatsuba's count); the additions of the launch are COUNTED ON
        # THE DEVICE from the digit lists of a serial proof (zkmi_msm_accum_additions), not modelled from the scalar distribution.
        adds = additions.get(dom, -1.0)
        peak_mul = FIELD_MUL_PEAK_G[args.curve] if r29 else FIELD_MUL_PEAK_32[args.curve]
        fmuls = int(adds * (28 if dom == 2 else 10))
        limb_form = ("9 x 29-bit" if cid == 0 else "14 x 28-bit") + " unsaturated limbs" if r29 else ("8" if cid == 0 else "12") + " x 32-bit saturated limbs"
        int_alu = {"unit": "Gmul/s", "mixed_additions": int(adds), "field_muls": fmuls, "achieved": round(fmuls / (acc[dom] * 1e-3) / 1e9, 1), "peak": peak_mul,
                   "frac": round(fmuls / (acc[dom] * 1e-3) / 1e9 / peak_mul, 4), "limb_form": limb_form,
                   "note": "mixed additions of this launch counted on the device (entries of the digit sort whose base is neither skipped nor at infinity) x 10 field multiplications in G1 / 28 in G2 (the Fq2 kernel's ~70 additions per mixed addition are not counted) / launch time; peak = measured Montgomery-multiply ceiling of the chip in the kernel's limb form at 8 waves per SIMD (tools/fieldbench29 / tools/fieldbench); the G2 kernel computes an Fq2 product as two double products with one reduction each (4 multiplications + 2 reductions instead of Karatsuba's 3 + 3): it is counted at Karatsuba's 3"}
        # VALU issue: static instruction count of the kernel's main path (tools/isa_counts.py on the shipped code object, profiles/r03_isa_counts.md)
        # x additions of the launch, against one wave instruction per SIMD every 4 cycles (1024 SIMDs x 2.4 GHz / 4)
        if r29:
            vpa = VALU_PER_ADD[args.curve]["g2" if dom == 2 else "g1"]
            wrate = adds * vpa / 64 / (acc[dom] * 1e-3) / 1e9
            int_alu["valu_issue"] = {"valu_instr_per_addition": vpa, "achieved": round(wrate, 1), "peak": VALU_ISSUE_PEAK_G, "unit": "G wave-instr/s", "frac": round(wrate / VALU_ISSUE_PEAK_G, 4),
                                     "note": "VALU instructions of the main path of one mixed addition (tools/isa_counts.py) x additions / 64 / launch time against one wave instruction per SIMD every 4 cycles; the G2 kernels (r06) hold one Fq2 component per lane in registers, two lanes per addition (their counts are per addition: BN254 2 x 3 360 at 3 waves per SIMD, BLS12-381 2 x 7 431 at 2 waves per SIMD), where a dependent v_mad_u64_u32 chain reaches 0.82 - 0.9 of that peak (tools/fieldbench29: wps=2)"}
        out = {
            "metric": "groth16_proofs_per_sec", "value": round(world * args.steps / elapsed, 4), "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "pipeline_depth": args.pipeline, "latency_ms_single_proof": round(float(np.median(lat)) * 1e3, 3),
            "repeats": (lambda v: {"what": f"the timed region ({args.steps} proofs between two barriers) run {len(v)} times back to back; `value` is the FIRST; proofs/s over all ranks",
                                   "proofs_per_s": [round(x, 3) for x in v], "min": round(min(v), 3), "median": round(float(np.median(v)), 3), "max": round(max(v), 3)})(
                [world * args.steps / x for x in ([elapsed] + region_s[1:])]),
            "config": {"workload": f"{'BN254' if cid == 0 else 'BLS12-381'} Groth16 prove, 2^{lg} constraints, synthetic zkey/wtns (BASELINE configs[{1 if (cid == 0 and lg == 20) else (2 if cid == 0 else 4)}]), "
                                   + (f"B density {1.0 if not args.b_zero_every else round(1 - 1 / args.b_zero_every, 3)} ({'every section dense, SURVEY 8d recipe' if not args.b_zero_every else f'every {args.b_zero_every}-th B1/B2 base at infinity'}), "
                                      if args.coef_dist == "flat" else "circuit-shaped coefficient section (heavy-tailed rows up to 10^5 terms, see coef_layout), B density 0.4, ")
                                   + f"{args.witness} witness; key + witness resident in HBM",
                       "curve": args.curve, "log_n": lg, "n_vars": m, "n_coef": int((zk['coeffs'].size - 4) // 44), "witness": args.witness, "coef_dist": args.coef_dist, "b_density": 0.4 if args.coef_dist == "real" else (1.0 if not args.b_zero_every else round(1 - 1 / args.b_zero_every, 4)),
                       "parallelism": f"replica x{world} (one proof stream per GPU, {args.pipeline} proof(s) in flight)"},
            "submetrics": {"g1_msm_mscalar_per_s": round(n / msm_ms / 1e3, 2), "g1_msm_ms": round(msm_ms, 4),
                           "g1_msm_resident_tables_mscalar_per_s": round(n / msm_tab_ms / 1e3, 2), "g1_msm_resident_tables_ms": round(msm_tab_ms, 4),
                           "g1_msm_note": "g1_msm_*: zkmi_msm_dev on caller-owned plain bases (R-form): the library's R'-form copy and its conversion pass are inside the time, sort / accumulate / reduce on one stream (r04 measured the MSM in 2 and 4 pieces with the sorts underneath the accumulations: slower, profiles/NOTES.md); resident_tables: the same MSM over pre-computed window tables (how a key holds its bases)",
                           "g1_msm_hbm_frac": round((2 * q8 + 32) * n / (msm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                           "ntt_melem_per_s": round(n / ntt_ms / 1e3, 2), "ntt_ms": round(ntt_ms, 4),
                           "ntt_hbm_frac": round(64 * n / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)},
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "g1_msm_sharded": sharded,
            "accum_kernel_ms": {names[k][0]: round(v, 4) for k, v in acc.items()},
            "accum_mixed_additions": {names[k][0]: int(v) for k, v in additions.items()},
            "roofline": roof,
            "int_alu": int_alu,
        }
        out["coef_layout"] = dict(pk.coef_layout(), note="the resident coefficient section: rows cut into segments of <= 32 terms, sorted by length into slices of 64 (csrc/groth16.hip); cut_rows = rows beyond 32 terms")
        if world == 1 and not args.no_napi_wall:
            # SURVEY.md 8d timing protocol: wall time THROUGH the N-API call with host buffers (H2D of inputs, D2H of results inside); the key is opened by
            # offset from a file, in pages (any size); at the headline config the drop-in (unmodified snarkjs + register.js) is timed beside the fused prover
            nw = napi_wall(zkey, wtns, reps=args.napi_wall_reps, draws=(r_m, s_m), dropin=(lg <= 20 and cid == 0 and args.coef_dist == "flat"), curve=args.curve)
            if isinstance(nw, dict) and "proof_sha256" in nw:
                import hashlib
                nw["proof_equals_python_mirror"] = bool(nw["proof_sha256"] == hashlib.sha256(b"".join(bytes(x) for x in proof_pts)).hexdigest())
                if isinstance(nw.get("dropin"), dict) and "proof_json_sha256" in nw["dropin"]:
                    pj = groth16.proof_to_json(groth16.raw_to_proof(pk, *proof_pts))
                    nw["dropin"]["proof_equals_fused_prover"] = bool(nw["dropin"]["proof_json_sha256"] == hashlib.sha256(pj.encode()).hexdigest())
            out["wall_through_napi"] = nw
        if world == 1 and not args.no_cpu_baseline:
            base, sample = cpu_baseline(args, zkey, wtns, lg)
            out["cpu_baseline"] = base
            if sample is not None:
                # the same sample through the device path must give the oracle's proof points (parity inside the bench run)
                zk_s, wt_s, ref, rs, ss = sample
                pk_s = groth16.ProvingKey(zk_s)
                got = pk_s.prove_raw(binfile.read_wtns(wt_s)["witness"], rs, ss)
                out["cpu_baseline"]["parity_on_sample"] = bool(all(np.array_equal(a, b) for a, b in zip(got, ref)))
                pk_s.release()
    if out is not None:
        out["box_calibration"] = box_calibration(L)
    if out is not None and world == 1 and not args.no_other_configs and args.curve == "bn128" and lg == 20 and args.coef_dist == "flat":
        # BASELINE configs[4], [3], [2] inside the same driver run, after everything above: this process's key and buffers are released first
        pk.release(); d_w.free()
        out["other_configs"] = other_configs(args)
    if dist is not None and not extras_failed:
        dist.barrier()
        dist.destroy_process_group()
    drain_c_stdout_to_stderr()                   # every rank: nothing but rank 0's JSON line may reach stdout
    if out is not None:
        print(json.dumps(out), flush=True)       # the ONE JSON line, after everything else this process may write
    if extras_failed:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)                              # the communicator may be unusable: no teardown collectives


if __name__ == "__main__":
    main()
