"""Synthetic Groth16 proving key + witness (the "fast" variant): structurally valid zkey / wtns
containers of any size whose proof does not verify (random bases) but is bit-comparable between implementations.

Layout follows src/zkey_utils.js:229-259 (header) / :20-45 (sections) and src/wtns_utils.js:62-72.
Bases come from the geometric table P_i = 7*11^i*G (device generator zkmi_gen_geometric_bases_dev, itself checked
against the oracle in test_gpu_parity.py::test_msm_closed_form_large) or, without a GPU, a caller-supplied generator.
"""
import struct

import numpy as np

from . import synth

PRIMES = {
    "bn128": (32, 21888242871839275222246405745257275088696311157297823662689037894645226208583,
              21888242871839275222246405745257275088548364400416034343698204186575808495617),
    "bls12381": (48, 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
                 52435875175126190479447740508185965837690552500527637822603658699938581184513),
}


def _section(typ, payload):
    return struct.pack("<IQ", typ, len(payload)) + payload


def _binfile(magic, sections):
    body = b"".join(_section(t, p) for t, p in sections)
    return magic + struct.pack("<II", 1, len(sections)) + body


def _tables(name, n1, n2, tables):
    """[G1 table of n1 points, G2 table of n2 points], P_i = 7*11^i*G affine Montgomery. tables: optional callable
    (curve_id, group, n) -> bytes for hosts without a GPU (tests/synth_zkey.py passes the CPU oracle's generator)."""
    cid = 0 if name == "bn128" else 1
    q8 = PRIMES[name][0]
    if tables is not None:
        return [tables(cid, 1, n1), tables(cid, 2, n2)]
    from .. import zkmi
    zkmi.init()
    out = []
    for group, n in ((1, n1), (2, n2)):
        d = zkmi.DeviceBuffer(n * 2 * group * q8)
        zkmi.check(zkmi.lib().zkmi_gen_geometric_bases_dev(cid, group, n, 7, 11, d.ptr))
        out.append(d.to_host())
        d.free()
    return out


def real_row_lengths(n, seed, mean, tail=True):
    """Row lengths of one constraint matrix shaped like a compiled circuit's: most rows hold 1 - 3 terms (P(len = k) ~ k^-2.3 up to 48, rescaled
    to the wanted mean), and a thin heavy tail on top (Num2Bits / packing / long linear combinations): ~n/2^13 rows of ~10^3 terms and — with
    `tail` — ~n/2^17 rows of ~10^4 and, from 2^17 constraints, one row of 10^5 (capped by the caller at the signals available)."""
    k = np.arange(1, 49, dtype=np.float64)
    pk = k ** -2.3
    pk /= pk.sum()
    u = (synth.words(seed, n).astype(np.float64) + 0.5) / 4294967296.0
    lens = 1 + np.searchsorted(np.cumsum(pk), u).astype(np.int64)
    have = lens.mean()
    if mean < have:                                      # thin out: a share of the rows drops to one term
        drop = (synth.words(seed ^ 0x7777, n) % 1000) < int(1000 * min(1.0, (have - mean) / max(have - 1.0, 1e-9)))
        lens[drop] = 1
    heavy = []
    n += 1                                                # the caller passes domain - 1 rows
    if n >= 1 << 12:
        heavy += [(int(x), 1000 + int(x) % 211) for x in synth.words(seed ^ 0x1111, max(1, n >> 13)) % n]
    if tail and n >= 1 << 15:
        heavy += [(int(x), 10000 + int(x) % 977) for x in synth.words(seed ^ 0x2222, max(1, n >> 17)) % n]
    if tail and n >= 1 << 17:
        heavy += [(int(synth.words(seed ^ 0x3333, 1)[0] % n), 100000)]
    for row, ln in heavy:
        lens[min(row, lens.size - 1)] = ln
    return lens


def real_coefs(n, m, n_public, seed):
    """-> (matrix, constraint, signal, in_b): the coefficient records of a circuit-shaped key over n constraints and m signals.
    A rows average ~1.5 terms and B rows ~1.0 before the heavy tail (together ~2.7 - 2.9 n with it, SURVEY a8's "2 - 3 n"); a row's signals are a contiguous run (as the terms of a
    linear combination over an array of signals are), distinct within the row; B rows only draw from the 40 % of the signals that occur in B
    (in_b: bool per signal). The public-input binding rows of src/zkey_new.js:290-300 are appended like in the flat recipe."""
    in_b = (np.arange(m) % 5 == 1) | (np.arange(m) % 5 == 3)
    in_b[0] = True
    b_sig = np.nonzero(in_b)[0].astype(np.uint64)
    out_m, out_c, out_s = [], [], []
    for mat, mean, pool in ((0, 1.5, None), (1, 1.0, b_sig)):
        avail = (m - 1) if pool is None else b_sig.size
        lens = np.minimum(real_row_lengths(n - 1, seed ^ (0xA0 + mat), mean, tail=mat == 0), avail)
        tot = int(lens.sum())
        rows = np.repeat(np.arange(n - 1, dtype=np.uint64), lens)
        starts = np.cumsum(lens) - lens
        kk = np.arange(tot, dtype=np.uint64) - np.repeat(starts, lens).astype(np.uint64)
        h = np.repeat(synth.words(seed ^ (0xB0 + mat), n - 1).astype(np.uint64), lens)
        idx = (h + kk) % np.uint64(avail)
        sig = (np.uint64(1) + idx) if pool is None else pool[idx]
        out_m.append(np.full(tot, mat, np.uint32)); out_c.append(rows); out_s.append(sig)
    pub = np.arange(n_public + 1, dtype=np.uint64)
    out_m.append(np.zeros(n_public + 1, np.uint32)); out_c.append((n - 1 - pub) % n); out_s.append(pub)
    return np.concatenate(out_m), np.concatenate(out_c), np.concatenate(out_s), in_b


def make(name, lg, seed=1, n_public=2, witness="mixed", tables=None, coef_per_row=1, b_zero_every=3, coef_dist="flat", n_vars=None):
    """-> (zkey_bytes, wtns_bytes). domain n = 2^lg, nVars m = n - 5 (min 4) unless n_vars is given (real circuits have anything from nVars << domainSize to
    nVars > domainSize: the witness-side and the H-side tables then differ in size and window width).
    b_zero_every = k: every k-th B1/B2 base (i % k == 1) is the point at infinity, as for signals absent from the B matrix of a real
    circuit (B density 1 - 1/k); 0 = dense B sections (SURVEY.md 8d recipe: every section filled from the geometric table).
    coef_dist = "flat": one coefficient per (matrix, constraint) row (n_coef = 2.0 n, the bottom of SURVEY a8's range);
    "real": the shape of a compiled circom circuit (real_coefs below): n_coef ~ 2.7 n, heavy-tailed rows up to 10^5 terms, B density 0.4
    (b_zero_every is ignored: the B1 / B2 bases of the signals absent from the B matrix are the point at infinity)."""
    q8, q, r = PRIMES[name]
    n = 1 << lg
    m = max(n - 5, n_public + 2) if n_vars is None else max(int(n_vars), n_public + 2)
    mc = m - n_public - 1
    T1, T2 = _tables(name, max(m + 1, n + 3, 8), max(m, 4), tables)
    g1, g2 = 2 * q8, 4 * q8
    T1 = T1.reshape(-1, g1)
    T2 = T2.reshape(-1, g2)
    A = T1[0:m].copy()
    B1 = T1[1:m + 1].copy()
    B2 = T2[0:m].copy()
    real = None
    if coef_dist == "real":
        real = real_coefs(n, m, n_public, seed)
        z = ~real[3]                                   # signals that never occur in the B matrix
        B1[z] = 0
        B2[z] = 0
    elif coef_dist != "flat":
        raise ValueError(f"coef_dist: {coef_dist!r} (flat | real)")
    elif b_zero_every:
        z = np.arange(m) % b_zero_every == 1          # real keys hold the point at infinity for signals absent from B
        B1[z] = 0
        B2[z] = 0
    Cb = T1[2:2 + mc].copy()
    H = T1[3:3 + n].copy()
    # coefficients (section 4): two records per constraint + one long row + the public-input binding rows
    cs = np.arange(n - 1, dtype=np.uint64)
    recs = []
    for k in range(coef_per_row):
        recs.append((np.zeros(n - 1, np.uint32), cs, 1 + (cs * (2 * k + 1) + k) % (m - 1)))
        recs.append((np.ones(n - 1, np.uint32), cs, 1 + (7 * cs + 3 + 5 * k) % (m - 1)))
    long_row = min(50, m)
    recs.append((np.zeros(long_row, np.uint32), np.zeros(long_row, np.uint64), np.arange(long_row, dtype=np.uint64) % m))
    pub = np.arange(n_public + 1, dtype=np.uint64)
    recs.append((np.zeros(n_public + 1, np.uint32), (n - 1 - pub) % n, pub))       # src/zkey_new.js:290-300 analogue
    if real is not None:
        recs = [real[:3]]
    mm = np.concatenate([x[0] for x in recs]).astype("<u4")
    cc = np.concatenate([x[1] for x in recs]).astype("<u4")
    ss = np.concatenate([x[2] for x in recs]).astype("<u4")
    ncoef = mm.size
    rec = np.zeros(ncoef, dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "u1", 32)])
    rec["m"], rec["c"], rec["s"] = mm, cc, ss
    rec["v"] = synth.elems(seed ^ 0x5151, ncoef).reshape(ncoef, 32)     # any residue < r is a valid (v*R^2) encoding
    coeffs = struct.pack("<I", ncoef) + rec.tobytes()
    le = lambda v, k: int(v).to_bytes(k, "little")
    hdr = (struct.pack("<I", q8) + le(q, q8) + struct.pack("<I", 32) + le(r, 32) + struct.pack("<III", m, n_public, n)
           + T1[5].tobytes() + T1[6].tobytes() + T2[1].tobytes() + T2[3].tobytes() + T1[7].tobytes() + T2[2].tobytes())
    ic = T1[8:8 + n_public + 1].tobytes() if T1.shape[0] >= 9 + n_public else bytes((n_public + 1) * g1)
    zkey = _binfile(b"zkey", [(1, struct.pack("<I", 1)), (2, hdr), (3, ic), (4, coeffs), (5, A.tobytes()), (6, B1.tobytes()),
                              (7, B2.tobytes()), (8, Cb.tobytes()), (9, H.tobytes())])
    w = (synth.witness_like(seed, m) if witness == "mixed" else synth.elems(seed, m)).copy()
    w[:32] = 0
    w[0] = 1
    wt = _binfile(b"wtns", [(1, struct.pack("<I", 32) + le(r, 32) + struct.pack("<I", m)), (2, w.tobytes())])
    return zkey, wt
