"""Synthetic, structurally VALID PLONK proving key + witness of any size.

The PLONK prover throws on inconsistent inputs ("Copy constraints does not match", "Polynomial is not divisible",
polynomial.js:607-611), so unlike Groth16 the key must describe a satisfiable circuit.  Circuit: one public input x,
then the chain w[i+1] = w[i]^2 + sum_{j=1..d} alpha_j w[i-j], i = 1..m — the `x^2 + b` shape of the reference's test circuits, whose R1CS
constraint w[i] * w[i] = w[i+1] - sum_j alpha_j w[i-j] has a (d+1)-term linear combination on its C side. plonk.setup folds such a combination
pairwise (src/plonk_setup.js:180-210, reduceCoefs): every fold creates an internal signal, an ADDITION GATE row and an ADDITIONS record
(s_1 = c_1 x_1 + c_2 x_2, then s_j = c_{j+1} x_{j+1} + 1 s_{j-1}: a chain of depth d), and the prover recomputes the internal signals for
every proof (calculateAdditions, src/plonk_prove.js:174-204). `additions` = d (default 1: one addition gate per multiplication gate, what
plonk.setup emits for `x^2 + b`; 0: the plain squaring chain, no additions section).

    row 0                      : a = w[1]                       ql = 1            (public-input row, PI(X) = -L_1(X) w[1])
    constraint i, rows of j<=d : a = x, b = y, c = s_j          ql = -c_x, qr = -c_y, qo = 1       (addition gates)
    constraint i, last row     : a = b = w[i], c = s_d          qm = 1, qo = -1                    (c = w[i+1] when d = 0)
    rows nc..n-1               : a = b = c = 0 (signal 0)

SRS: [tau^i] G1, i < n + 6, for a KNOWN tau (never do this in production).  Layout follows src/plonk_setup.js /
src/zkey_utils.js:261-299: each Q / sigma / Lagrange section = n coefficients then 4n evaluations, Montgomery form.
The heavy lifting (NTTs, SRS points, commitments) uses the device library, the same way tests/synth_zkey.py uses the
device generator for Groth16 bases; the prover's own divisibility checks and the small reference-generated fixtures
(tests/golden/plonk_bn128_*) are what pins correctness.
"""
import struct

import numpy as np

from .synth_zkey import PRIMES, _binfile


def _pieces(name, lg, seed, tau, n_srs, free_rows=0, additions=1):
    """everything both PLONK-family key layouts share: witness, signal maps, selector / sigma / Lagrange sections, SRS, commit()"""
    from .. import zkmi
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

    # ---- witness: w[0] = 1 (ignored by the prover), w[1] = x, w[i+1] = w[i]^2 + sum_j alpha_j w[i-j]   (w[k] = 0 for k < 1: signal 0 reads as 0)
    d = int(additions)
    m = (nc - 1) // (d + 1)                       # multiplication gates (R1CS constraints); each brings d addition gates
    nc = 1 + m * (d + 1)                          # rows in use
    alpha = [(seed * 7919 + 104729 * j + 3) % r for j in range(1, d + 1)]
    w = [1, (seed * 0x9E3779B97F4A7C15 + 12345) % r]
    for i in range(1, m + 1):
        v = w[i] * w[i]
        for j in range(1, d + 1):
            if i - j >= 1:
                v += alpha[j - 1] * w[i - j]
        w.append(v % r)
    n_wit = m + 2                                 # signals of the .wtns file
    n_add = d * m
    n_vars = n_wit + n_add                        # the zkey header's nVars counts the internal signals too (plonk_setup.js: plonkNVars)
    assert len(w) == n_wit
    wt = _binfile(b"wtns", [(1, struct.pack("<I", 32) + r.to_bytes(32, "little") + struct.pack("<I", n_wit)),
                            (2, b"".join(v.to_bytes(32, "little") for v in w))])
    # ---- rows, signal maps (sections 4-6), selector evaluations and the additions section (section 3: 72-byte records u32 id1, u32 id2,
    #      factor1, factor2 in Montgomery form). Constraint i owns the internal signals n_wit + d (i-1) .. + d - 1 and the rows 1 + (i-1)(d+1) .. + d.
    map_a, map_b, map_c = (np.zeros(nc, np.uint32) for _ in range(3))
    sel = {k: np.zeros((n, 32), np.uint8) for k in ("qm", "ql", "qr", "qo", "qc")}
    bv = lambda b: np.frombuffer(b, np.uint8)
    map_a[0] = 1
    sel["ql"][0] = bv(one_m)
    ii = np.arange(1, m + 1, dtype=np.int64)
    base = 1 + (ii - 1) * (d + 1)
    rec = np.zeros((m, max(d, 1)), dtype=[("id1", "<u4"), ("id2", "<u4"), ("f1", "V32"), ("f2", "V32")])
    for j in range(1, d + 1):
        internal = n_wit + d * (ii - 1) + (j - 1)
        if j == 1:
            id1, id2, f1, f2 = ii + 1, np.maximum(ii - 1, 0), 1, r - alpha[0]
        else:
            id1, id2, f1, f2 = np.maximum(ii - j, 0), internal - 1, r - alpha[j - 1], 1
        rec["id1"][:, j - 1], rec["id2"][:, j - 1] = id1, id2
        rec["f1"][:, j - 1], rec["f2"][:, j - 1] = np.void(mont(f1)), np.void(mont(f2))
        rows_j = base + (j - 1)
        map_a[rows_j], map_b[rows_j], map_c[rows_j] = id1, id2, internal
        sel["ql"][rows_j], sel["qr"][rows_j], sel["qo"][rows_j] = bv(mont(r - f1)), bv(mont(r - f2)), bv(one_m)
    mul_rows = base + d
    map_a[mul_rows] = map_b[mul_rows] = ii
    map_c[mul_rows] = (n_wit + d * (ii - 1) + (d - 1)) if d else (ii + 1)
    sel["qm"][mul_rows], sel["qo"][mul_rows] = bv(one_m), bv(mone_m)
    add_sec = rec.tobytes() if d else b""
    assert len(add_sec) == 72 * n_add
    # ---- permutation: positions p = col*n + row, grouped by signal id, sigma = next position in the cycle
    sig = np.zeros((3, n), np.int64)
    sig[0, :nc], sig[1, :nc], sig[2, :nc] = map_a, map_b, map_c
    for j in range(free_rows):                    # FFLONK: the last two rows carry blinding values, sigma = identity there (fflonk_setup.js:357-361)
        sig[:, n - 1 - j] = -1 - 3 * j - np.arange(3)
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
    d_srs = zkmi.DeviceBuffer(n_srs * 2 * q8)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(cid, 1, n_srs, 1, tau, d_srs.ptr))
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

    secs, commits = {}, {}
    for t, nm, key in ((7, "Qm", "qm"), (8, "Ql", "ql"), (9, "Qr", "qr"), (10, "Qo", "qo"), (11, "Qc", "qc")):
        secs[t] = section(sel[key].reshape(-1))
        commits[nm] = commit()
    for k, nm in enumerate(("S1", "S2", "S3")):
        secs[nm] = section(sigma_ev[k].reshape(-1))
        commits[nm] = commit()
    e0 = np.zeros((n, 32), np.uint8)
    e0[0] = np.frombuffer(one_m, np.uint8)
    secs[13] = section(e0.reshape(-1))            # Lagrange L_1 (nPublic = 1)
    srs = d_srs.to_host().tobytes()

    def commit_coefs(coef_bytes):                 # [p(tau)]_1 for any coefficient vector no longer than the SRS
        k = len(coef_bytes) // 32
        dc, ds = zkmi.DeviceBuffer.from_host(np.frombuffer(coef_bytes, np.uint8)), zkmi.DeviceBuffer(k * 32)
        zkmi.check(L.zkmi_fr_batch_dev(cid, 1, dc.ptr, ds.ptr, k))
        jac, aff = np.zeros(3 * q8, np.uint8), np.zeros(2 * q8, np.uint8)
        zkmi.check(L.zkmi_msm_dev(cid, 1, d_srs.ptr, ds.ptr, k, 32, zkmi.ptr(jac)))
        zkmi.check(L.zkmi_to_affine(cid, 1, zkmi.ptr(jac), zkmi.ptr(aff)))
        dc.free(); ds.free()
        return aff.tobytes()
    out = dict(q8=q8, q=q, r=r, n=n, nc=nc, n_vars=n_vars, n_add=n_add, add_sec=add_sec, wt=wt, maps=(map_a, map_b, map_c), secs=secs, commits=commits, x2=x2.tobytes(), srs=srs,
               k1=k1, k2=k2, mont=mont, commit_coefs=commit_coefs, free=lambda: [b.free() for b in (d_in, d_out, d4, d_srs, d_g2, d_sc)])
    return out


def make(name, lg, seed=7, tau=0x1F3D5B79, additions=1):
    """PLONK zkey (protocol id 2, src/zkey_utils.js:261-299) + wtns; `additions` = addition gates per multiplication gate (module docstring)"""
    P = _pieces(name, lg, seed, tau, (1 << lg) + 6, additions=additions)
    q8, q, r, n, nc, secs, commits, mont = P["q8"], P["q"], P["r"], P["n"], P["nc"], P["secs"], P["commits"], P["mont"]
    map_a, map_b, map_c = P["maps"]
    hdr = (struct.pack("<I", q8) + q.to_bytes(q8, "little") + struct.pack("<I", 32) + r.to_bytes(32, "little")
           + struct.pack("<IIIII", P["n_vars"], 1, n, P["n_add"], nc) + mont(P["k1"]) + mont(P["k2"])
           + b"".join(commits[nm] for nm in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3")) + P["x2"])
    zkey = _binfile(b"zkey", [(1, struct.pack("<I", 2)), (2, hdr), (3, P["add_sec"]), (4, map_a.astype("<u4").tobytes()), (5, map_b.astype("<u4").tobytes()),
                              (6, map_c.astype("<u4").tobytes()), (7, secs[7]), (8, secs[8]), (9, secs[9]), (10, secs[10]), (11, secs[11]),
                              (12, secs["S1"] + secs["S2"] + secs["S3"]), (13, secs[13]), (14, P["srs"])])
    P["free"]()
    return zkey, P["wt"]


def make_fflonk(lg, seed=7, tau=0x1F3D5B79, additions=1):
    """FFLONK zkey (protocol id 10; src/fflonk_setup.js:213-500, src/zkey_utils.js:301-339) + wtns for the same circuit. BN254 only
    (the reference hard-codes w3 / wr for that field, fflonk_setup.js:525-547)."""
    n = 1 << lg
    P = _pieces("bn128", lg, seed, tau, 9 * n + 18, free_rows=2, additions=additions)
    q8, q, r, nc, secs, mont = P["q8"], P["q"], P["r"], P["nc"], P["secs"], P["mont"]
    map_a, map_b, map_c = P["maps"]
    coef = lambda key: np.frombuffer(secs[key][:n * 32], np.uint8).reshape(n, 32)
    # C0(X) = QL(X^8) + X QR(X^8) + X^2 QO(X^8) + X^3 QM(X^8) + X^4 QC(X^8) + X^5 S1(X^8) + X^6 S2(X^8) + X^7 S3(X^8)   (:441-458)
    c0 = np.stack([coef(k) for k in (8, 9, 10, 7, 11, "S1", "S2", "S3")], axis=1).reshape(-1).tobytes()
    w3 = pow(31624, 3648040478639879203707734290876212514758060733402672390616367364429301415936 // 3, r)   # computeW3 (:525-533)
    w4, w8 = (np.zeros(32, np.uint8) for _ in range(2))
    from .. import zkmi
    zkmi.check(zkmi.lib().zkmi_fr_root(0, 2, zkmi.ptr(w4)))
    zkmi.check(zkmi.lib().zkmi_fr_root(0, 3, zkmi.ptr(w8)))
    wr = pow(467799165886069610036046866799264026481344299079011762026774533774345988080, 2 ** (28 - lg), r)
    hdr = (struct.pack("<I", q8) + q.to_bytes(q8, "little") + struct.pack("<I", 32) + r.to_bytes(32, "little")
           + struct.pack("<IIIII", P["n_vars"], 1, n, P["n_add"], nc) + mont(P["k1"]) + mont(P["k2"]) + mont(w3) + w4.tobytes() + w8.tobytes() + mont(wr)
           + P["x2"] + P["commit_coefs"](c0))
    # FFLONK section ids (src/fflonk_constants.js): 7 QL, 8 QR, 9 QM, 10 QO, 11 QC, 12-14 sigma, 15 Lagrange, 16 PTau, 17 C0
    zkey = _binfile(b"zkey", [(1, struct.pack("<I", 10)), (2, hdr), (3, P["add_sec"]), (4, map_a.astype("<u4").tobytes()), (5, map_b.astype("<u4").tobytes()),
                              (6, map_c.astype("<u4").tobytes()), (7, secs[8]), (8, secs[9]), (9, secs[7]), (10, secs[10]), (11, secs[11]),
                              (12, secs["S1"]), (13, secs["S2"]), (14, secs["S3"]), (15, secs[13]), (16, P["srs"]), (17, c0)])
    P["free"]()
    return zkey, P["wt"]
