"""r06 A/B of buildABC on a circuit-shaped coefficient section.

  python tools/lab/r6_buildabc.py make <dir> [lg]      # writes <dir>/{flat,real}.zkey/.wtns (this tree's generator)
  python tools/lab/r6_buildabc.py run <dir>            # proves them with the package found first on sys.path (cwd): works in this tree and in a
                                                       # checkout of the r05 tree (build_r05src/), whose one-lane-per-constraint kernel is the "before"
Prints one JSON line per key: buildABC stage time of serial proofs (HIP events), proofs/s, sha256 of the proof points (must agree between trees).
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402


def main():
    mode, d = sys.argv[1], sys.argv[2]
    if mode == "make":
        from snarkjs_amd.workloads import synth_zkey
        lg = int(sys.argv[3]) if len(sys.argv) > 3 else 20
        os.makedirs(d, exist_ok=True)
        for dist in ("flat", "real"):
            zk, wt = synth_zkey.make("bn128", lg, seed=0x5EED, witness="mixed" if dist == "real" else "uniform", b_zero_every=0, coef_dist=dist)
            open(os.path.join(d, dist + ".zkey"), "wb").write(zk)
            open(os.path.join(d, dist + ".wtns"), "wb").write(wt)
        return
    from snarkjs_amd import groth16, zkmi, binfile
    zkmi.init(0)
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    mont = lambda v: np.frombuffer(((v << 256) % R).to_bytes(32, "little"), np.uint8).copy()
    r_m, s_m = mont(0x1234567), mont(0x7654321)
    for dist in ("flat", "real"):
        zkey = open(os.path.join(d, dist + ".zkey"), "rb").read()
        w = binfile.read_wtns(open(os.path.join(d, dist + ".wtns"), "rb").read())["witness"]
        t0 = time.perf_counter()
        pk = groth16.ProvingKey(zkey)
        load_s = time.perf_counter() - t0
        d_w = zkmi.DeviceBuffer.from_host(w)
        for _ in range(3):
            pts = pk.prove_raw(None, r_m, s_m, d_witness=d_w.ptr)
        st = []
        for _ in range(8):
            pk.prove_raw(None, r_m, s_m, d_witness=d_w.ptr)
            st.append(pk.stage_ms()["buildABC"])
        n = 20
        t0 = time.perf_counter()
        for i in range(n):
            pk.submit(d_w.ptr, i & 1)
            if i:
                pk.collect((i - 1) & 1, r_m, s_m)
        pk.collect((n - 1) & 1, r_m, s_m)
        dt = time.perf_counter() - t0
        out = {"key": dist, "n_coef": (binfile.read_groth16_zkey(zkey)["coeffs"].size - 4) // 44, "load_s": round(load_s, 2), "buildABC_ms_min": round(min(st), 4), "buildABC_ms_median": round(float(np.median(st)), 4),
               "proofs_per_s": round(n / dt, 2), "proof_sha256": hashlib.sha256(b"".join(bytes(x) for x in pts)).hexdigest()[:16]}
        if hasattr(pk, "coef_layout"):
            out["layout"] = pk.coef_layout()
            a, b, c = pk.build_abc(d_witness=d_w.ptr)
            out["build_abc_dev_ms"] = round(zkmi.lib().zkmi_last_kernel_ms(), 4)
        print(json.dumps(out), flush=True)
        pk.release(); d_w.free()


if __name__ == "__main__":
    main()
