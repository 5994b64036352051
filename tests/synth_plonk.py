"""Synthetic, structurally VALID PLONK proving key + witness of any size.

The PLONK prover throws on inconsistent inputs ("Copy constraints does not match", "Polynomial is not divisible",
polynomial.js:607-611), so unlike Groth16 the key must describe a satisfiable circuit.  Circuit: one public input x,
then the squaring chain w[i+1] = w[i]^2 (the Multiplier(n) shape of the reference's test/groth16/circuit.circom):

    row 0            : a = w[1]                ql = 1                    (public-input row, PI(X) = -L_1(X) w[1])
    row i (1..nc-1)  : a = b = w[i], c = w[i+1]   qm = 1, qo = -1
    rows nc..n-1     : a = b = c = 0 (signal 0)

SRS: [tau^i] G1, i < n + 6, for a KNOWN tau (never do this in production).  Layout follows src/plonk_setup.js /
src/zkey_utils.js:261-299: each Q / sigma / Lagrange section = n coefficients then 4n evaluations, Montgomery form.
The heavy lifting (NTTs, SRS points, commitments) uses the device library, the same way tests/synth_zkey.py uses the
device generator for Groth16 bases; the prover's own divisibility checks and the small reference-generated fixtures
(tests/golden/plonk_bn128_*) are what pins correctness.
"""
import struct

import numpy as np

from synth_zkey import PRIMES, _binfile


def make(name, lg, seed=7, tau=0x1F3D5B79):
    from snarkjs_amd import zkmi
    zkmi.init()
    L = zkmi.lib()
    cid = 0 if name == "bn128" else 1
    q8, q, r = PRIMES[name]
    n = 1 << lg
    nc = n - 4                                   # constraints (rows in use)
    R = pow(2, 256, r)
    mont = lambda v: (v % r * R % r).to_bytes(32, "little")
    one_m, mone_m, zero_m = mont(1), mont(r - 1), bytes(32)

    def root(i):
        out = np.zeros(32, np.uint8)
        zkmi.check(L.zkmi_fr_root(cid, i, zkmi.ptr(out)))
        return out

    # ---- witness: w[0] = 1 (ignored by the prover), w[1] = x, w[i+1] = w[i]^2
    w = [1, (seed * 0x9E3779B97F4A7C15 + 12345) % r]
    for _ in range(1, nc):
        w.append(w[-1] * w[-1] % r)
    n_vars = nc + 1
    assert len(w) == n_vars
    wt = _binfile(b"wtns", [(1, struct.pack("<I", 32) + r.to_bytes(32, "little") + struct.pack("<I", n_vars)),
                            (2, b"".join(v.to_bytes(32, "little") for v in w))])
    # ---- signal maps (sections 4-6)
    rows = np.arange(nc, dtype=np.uint32)
    map_a = rows.copy(); map_a[0] = 1
    map_b = rows.copy(); map_b[0] = 0
    map_c = rows + 1; map_c[0] = 0
    # ---- permutation: positions p = col*n + row, grouped by signal id, sigma = next position in the cycle
    sig = np.zeros((3, n), np.int64)
    sig[0, :nc], sig[1, :nc], sig[2, :nc] = map_a, map_b, map_c
    flat = sig.reshape(-1)
    order = np.argsort(flat, kind="stable")
    nxt = np.empty(3 * n, np.int64)
    srt = flat[order]
    start = np.r_[0, np.flatnonzero(srt[1:] != srt[:-1]) + 1]
    end = np.r_[start[1:], 3 * n]
    rolled = np.empty(3 * n, np.int64)
    rolled[:-1] = order[1:]
    rolled[end - 1] = order[start]                # last of each group -> first
    nxt[order] = rolled
    k1, k2 = 2, 3                                 # coset representatives used by snarkjs (plonk_setup.js)
    ones = np.frombuffer(one_m * n, np.uint8)
    ident = []
    d_in, d_out = zkmi.DeviceBuffer.from_host(ones), zkmi.DeviceBuffer(n * 32)
    for k in (1, k1, k2):                         # k * w^row for every row (Fr.batchApplyKey(1.., k, w))
        kb = np.frombuffer(mont(k), np.uint8)
        zkmi.check(L.zkmi_fr_batch_apply_key_dev(cid, d_in.ptr, d_out.ptr, n, zkmi.ptr(kb), zkmi.ptr(root(lg))))
        ident.append(d_out.to_host().reshape(n, 32).copy())
    ident = np.concatenate(ident)                 # (3n, 32)
    sigma_ev = ident[nxt].reshape(3, n, 32)

    d4 = zkmi.DeviceBuffer(4 * n * 32)

    def section(evals_bytes):
        """n evaluations (Montgomery) -> n coefficients + 4n evaluations, plus the coefficients on the device for the commitment"""
        zkmi.check(L.zkmi_memcpy_h2d(d_in.ptr, zkmi.ptr(np.ascontiguousarray(evals_bytes)), n * 32))
        zkmi.check(L.zkmi_ntt_dev(cid, d_in.ptr, d_out.ptr, lg, 1, None, None))
        zkmi.check(L.zkmi_memset_dev(d4.ptr, 0, 4 * n * 32))
        zkmi.check(L.zkmi_memcpy_d2d(d4.ptr, d_out.ptr, n * 32))
        zkmi.check(L.zkmi_ntt_dev(cid, d4.ptr, d4.ptr, lg + 2, 0, None, None))
        return d_out.to_host().tobytes() + d4.to_host().tobytes()

    # SRS: [tau^i] G1 (geometric table with f = 1, g = tau), and [tau] G2
    d_srs = zkmi.DeviceBuffer((n + 6) * 2 * q8)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(cid, 1, n + 6, 1, tau, d_srs.ptr))
    d_g2 = zkmi.DeviceBuffer(2 * 4 * q8)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(cid, 2, 2, 1, tau, d_g2.ptr))
    x2 = d_g2.to_host()[4 * q8:]
    d_sc = zkmi.DeviceBuffer(n * 32)

    def commit():                                 # [p(tau)]_1 of the coefficients currently in d_out
        zkmi.check(L.zkmi_fr_batch_dev(cid, 1, d_out.ptr, d_sc.ptr, n))
        jac, aff = np.zeros(3 * q8, np.uint8), np.zeros(2 * q8, np.uint8)
        zkmi.check(L.zkmi_msm_dev(cid, 1, d_srs.ptr, d_sc.ptr, n, 32, zkmi.ptr(jac)))
        zkmi.check(L.zkmi_to_affine(cid, 1, zkmi.ptr(jac), zkmi.ptr(aff)))
        return aff.tobytes()

    def const_rows(first_row, body):
        e = np.zeros((n, 32), np.uint8)
        e[0] = np.frombuffer(first_row, np.uint8)
        e[1:nc] = np.frombuffer(body, np.uint8)
        return e.reshape(-1)

    secs, commits = {}, {}
    for t, nm, ev in ((7, "Qm", const_rows(zero_m, one_m)), (8, "Ql", const_rows(one_m, zero_m)), (9, "Qr", const_rows(zero_m, zero_m)),
                      (10, "Qo", const_rows(zero_m, mone_m)), (11, "Qc", const_rows(zero_m, zero_m))):
        secs[t] = section(ev)
        commits[nm] = commit()
    sig_sec = b""
    for k, nm in enumerate(("S1", "S2", "S3")):
        sig_sec += section(sigma_ev[k].reshape(-1))
        commits[nm] = commit()
    secs[12] = sig_sec
    e0 = np.zeros((n, 32), np.uint8)
    e0[0] = np.frombuffer(one_m, np.uint8)
    secs[13] = section(e0.reshape(-1))            # Lagrange L_1 (nPublic = 1)
    hdr = (struct.pack("<I", q8) + q.to_bytes(q8, "little") + struct.pack("<I", 32) + r.to_bytes(32, "little")
           + struct.pack("<IIIII", n_vars, 1, n, 0, nc) + mont(k1) + mont(k2)
           + b"".join(commits[nm] for nm in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3")) + x2.tobytes())
    zkey = _binfile(b"zkey", [(1, struct.pack("<I", 2)), (2, hdr), (3, b""), (4, map_a.astype("<u4").tobytes()), (5, map_b.astype("<u4").tobytes()),
                              (6, map_c.astype("<u4").tobytes()), (7, secs[7]), (8, secs[8]), (9, secs[9]), (10, secs[10]), (11, secs[11]), (12, secs[12]),
                              (13, secs[13]), (14, d_srs.to_host().tobytes())])
    for b in (d_in, d_out, d4, d_srs, d_g2, d_sc):
        b.free()
    return zkey, wt
