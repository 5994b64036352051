"""fflonk.prove on the MI355X — host-side mirror of the reference driver (src/fflonk_prove.js:51-1288, src/fflonk.js).

Same inputs (fflonk zkey + wtns containers), checks, error messages and output shape ({proof: {polynomials, evaluations,
protocol, curve}, publicSignals}) as the reference.  Every O(n) step runs in the HIP library on device-resident data:

    round 1  wires, 3 iNTT(n) + 3 NTT(4n), T0 over 4n points, divByZerofier(n, 1), C1 = interleave(A, B, C, T0), MSM(8n)
    round 2  computeZ, T1 over 2n points, T2 over 4n points, divisions, C2 = interleave(Z, T1, T2), MSM(16n)
    round 3  15 Horner evaluations at xi / xi*w
    round 4  R0/R1/R2 by Lagrange interpolation of 8/4/6 evaluations (host, O(1)), F = sum (C_i - R_i)/Z_{S_i}, MSM(16n)
    round 5  L(X), division by (X - y), MSM(16n); inv = Montgomery batched inverse (host, O(1))

The host keeps what is O(1) in the reference too: Keccak transcript, root-of-unity bookkeeping, the tiny R_i / ZT polynomials.
"""
import ctypes as C
import functools
import os
import struct

import numpy as np

from . import zkmi
from .groth16 import _curve_from_q
from .plonk import _Field, _Poly, _Transcript, evaluate_many, lincomb


class FflonkKey:
    """An FFLONK zkey resident on the device (selectors, permutations, Lagrange evaluations, C0, SRS table)."""

    def __init__(self, zkey_bytes):
        data = bytes(zkey_bytes)
        s = self.sections = {}
        off = 12
        for _ in range(struct.unpack_from("<I", data, 8)[0]):
            t, ln = struct.unpack_from("<IQ", data, off)
            s[t] = (off + 12, ln)
            off += 12 + ln
        if struct.unpack_from("<I", data, s[1][0])[0] != 10:
            raise ValueError("zkey file is not fflonk")                                   # fflonk_prove.js:71-73
        off = s[2][0]
        n8q = struct.unpack_from("<I", data, off)[0]
        q = int.from_bytes(data[off + 4:off + 4 + n8q], "little"); off += 4 + n8q
        n8r = struct.unpack_from("<I", data, off)[0]
        self.r = int.from_bytes(data[off + 4:off + 4 + n8r], "little"); off += 4 + n8r
        self.curve_id, self.curve_name = _curve_from_q(q)
        self.f = f = _Field(self.curve_id)
        self.nVars, self.nPublic, self.n, self.nAdditions, self.nConstraints = struct.unpack_from("<IIIII", data, off); off += 20
        self.power = self.n.bit_length() - 1
        for nm in ("k1", "k2", "w3", "w4", "w8", "wr"):                                   # src/zkey_utils.js:322-328
            setattr(self, nm, f.unmont(data[off:off + 32])); off += 32
        if self.curve_id != 0:
            # The reference's prover takes the curve and its roots from the zkey (src/fflonk_prove.js:51-110) — only its fflonk.setup hard-codes BN254
            # constants (computeW3: generator 31624 to BN254's (r - 1) / 3; getOmegaCubicRoot: a literal cube root of BN254's Fr.w[28],
            # src/fflonk_setup.js:533-556): the key THAT writes for BLS12-381 has w3^3 != 1 and the reference's own fflonk.prove throws "Polynomial is
            # not divisible" on a satisfied circuit (tests/golden/fflonk_bls12381_unsupported.json, oracle/gen_golden.js: fflonkBlsProbe). So the key is
            # held to what the protocol needs of it, not to its curve: a consistent key from a repaired or third-party setup takes the generic path
            # below, an inconsistent one fails here with the reference's own words, before the device is touched.
            r = self.r
            ok = (self.w3 != 1 and pow(self.w3, 3, r) == 1 and pow(self.w4, 2, r) == r - 1 and pow(self.w8, 4, r) == r - 1
                  and pow(self.wr, 3, r) == f.unmont(f.root(self.power)))
            if not ok:
                raise ValueError(f"Polynomial is not divisible: this {self.curve_name} FFLONK key is inconsistent (w3^3 != 1 or w4 / w8 / wr of the wrong order) — what the "
                                 "reference's fflonk.setup writes off bn128, where it hard-codes BN254 constants; the reference's fflonk.prove fails on such a key the same way")
        off += 4 * n8q                                                                    # X_2
        self.C0 = (f.unmont_q(data[off:off + n8q]), f.unmont_q(data[off + n8q:off + 2 * n8q]))
        zkmi.init()
        if s[3][1] < 72 * self.nAdditions:
            raise ValueError("zkey additions section is shorter than its header says")
        # section 3 (additions) too, as it lies in the file: calculateAdditions runs on the device, once per proof (zkmi_plonk_additions_dev)
        self.dev = {t: zkmi.DeviceBuffer.from_host(np.frombuffer(data, np.uint8, s[t][1], s[t][0])) for t in range(3, 18) if s[t][1]}
        self.n_ptau = s[16][1] // (2 * f.n8q)                                             # 9n + 18 points (fflonk_setup.js:430-438)
        self.ptau_table = C.c_uint64(0)
        zkmi.check(zkmi.lib().zkmi_msm_table_build(self.curve_id, 1, self.dev[16].ptr, self.n_ptau, C.byref(self.ptau_table)))

    def sec(self, t, elem_off=0):
        return self.dev[t].ptr + 32 * elem_off

    def release(self):
        for b in self.dev.values():
            b.free()
        self.dev = {}
        if self.ptau_table.value:
            zkmi.lib().zkmi_msm_table_release(self.ptau_table)
            self.ptau_table = C.c_uint64(0)


def _degree(p):
    d = C.c_size_t(0)
    zkmi.check(zkmi.lib().zkmi_poly_degree_dev(p.f.cid, p.ptr, p.n, C.byref(d)))
    return d.value


def _cpoly(f, polys, n):
    """CPolynomial.getPolynomial (cpolynomial.js:53-73) on the device"""
    degs = [0 if p is None else _degree(p) for p in polys]
    max_degree = max(0 if p is None else d * n + j for j, (p, d) in enumerate(zip(polys, degs)))
    length = 2 ** ((max_degree - 1).bit_length() - 1 + 1)
    out = _Poly(f, length, zero=False)
    ptrs = (C.c_void_p * n)(*[None if p is None else p.ptr for p in polys])
    lens = (C.c_size_t * n)(*[0 if p is None else min(d + 1, p.n) for p, d in zip(polys, degs)])
    zkmi.check(zkmi.lib().zkmi_cpoly_interleave_dev(f.cid, ptrs, lens, n, out.ptr, length))
    return out


def _commit(key, poly):
    """Polynomial.multiExponentiation over PTau (16n slots in the reference, zero beyond the 9n+18 SRS points: coefficients
    past the SRS multiply the point at infinity, so only the first n_ptau scalars matter)."""
    f, L = key.f, zkmi.lib()
    k = min(poly.n, key.n_ptau)
    sc = zkmi.DeviceBuffer(k * 32)
    zkmi.check(L.zkmi_fr_batch_dev(f.cid, zkmi.BATCH_FROM_MONTGOMERY, poly.ptr, sc.ptr, k))
    jac, aff = np.zeros(3 * f.n8q, np.uint8), np.zeros(2 * f.n8q, np.uint8)
    zkmi.check(L.zkmi_msm_table_dev(key.ptau_table, sc.ptr, k, 32, zkmi.ptr(jac)))
    sc.free()
    zkmi.check(L.zkmi_to_affine(f.cid, 1, zkmi.ptr(jac), zkmi.ptr(aff)))
    return (f.unmont_q(aff[:f.n8q]), f.unmont_q(aff[f.n8q:]))


def _commit_enqueue(key, poly):
    """first half of _commit: batchFromMontgomery + the MSM enqueued on the active pipeline slot (zkmi_msm_table_multi_enqueue_mont_dev, one MSM), nothing waits"""
    k = min(poly.n, key.n_ptau)
    ptrs, ks = (C.c_void_p * 1)(poly.ptr), (C.c_size_t * 1)(k)
    zkmi.check(zkmi.lib().zkmi_msm_table_multi_enqueue_mont_dev(key.ptau_table, ptrs, ks, 1))
    return key


def _commit_collect(key):
    f, L = key.f, zkmi.lib()
    jac, aff = np.zeros(3 * f.n8q, np.uint8), np.zeros(2 * f.n8q, np.uint8)
    zkmi.check(L.zkmi_msm_table_multi_collect(key.ptau_table, 1, zkmi.ptr(jac)))
    zkmi.check(L.zkmi_to_affine(f.cid, 1, zkmi.ptr(jac), zkmi.ptr(aff)))
    return (f.unmont_q(aff[:f.n8q]), f.unmont_q(aff[f.n8q:]))


def _evaluate_all(f, pairs):
    """[p(x)] for ((device pointer, length), x) pairs, eight per wait (zkmi_poly_evaluate_multi_dev); a generator: yields before every wait"""
    out = []
    for i in range(0, len(pairs), 8):
        chunk = pairs[i:i + 8]
        yield
        out += evaluate_many(f, [p for p, _ in chunk], [x for _, x in chunk])
    return out


def _div_zerofier(p, n, beta):
    zkmi.check(zkmi.lib().zkmi_poly_div_by_zerofier_dev(p.f.cid, p.ptr, p.n, n, zkmi.ptr(p.f.mont(beta))))


# ---- O(1) host algebra on tiny polynomials (lists of ints, lowest coefficient first) --------------------------------------------------
def _ev(c, x, r):
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % r
    return acc


def _mul_linear(p, x, r):             # p * (X - x)
    out = [0] * (len(p) + 1)
    for d, c in enumerate(p):
        out[d] = (out[d] - x * c) % r
        out[d + 1] = (out[d + 1] + c) % r
    return out


def _lagrange(xs, ys, r):             # Polynomial.lagrangePolynomialInterpolation (polynomial.js:896-930)
    out = [0] * len(xs)
    for i in range(len(xs)):
        num = [1]
        for j, x in enumerate(xs):
            if j != i:
                num = _mul_linear(num, x, r)
        k = ys[i] * pow(_ev(num, xs[i], r), -1, r) % r
        for d, c in enumerate(num):
            out[d] = (out[d] + c * k) % r
    return out


def _zerofier(xs, r):                 # Polynomial.zerofierPolynomial (:932-950)
    p = [1]
    for x in xs:
        p = _mul_linear(p, x, r)
    return p


def _small(f, coefs):
    p = _Poly(f, len(coefs), zero=False)
    b = np.concatenate([f.mont(c) for c in coefs])
    zkmi.check(zkmi.lib().zkmi_memcpy_h2d(p.ptr, zkmi.ptr(b), b.size))
    return p


def prove(zkey, witness_file, logger=None, options=None, blinding_mont=None):
    """fflonk.prove(zkeyFileName, witnessFileName). blinding_mont: the 9 Fr.random() draws (:321-324), Montgomery bytes."""
    steps = _prove_steps(zkey, witness_file, logger, options, blinding_mont)
    try:
        while True:
            next(steps)
    except StopIteration as done:
        return done.value


def prove_many(zkey, witness_files, blinding_monts=None, in_flight=2):
    """Throughput mode (r06), as plonk.prove_many: one proof per witness against one key, TWO in flight from this one host thread, each on its own pipeline slot. An FFLONK proof
    waits on the device some fifty times (eight degree checks, thirty-three evaluations, four commitments): alone it leaves the GPU idle through every one of those round
    trips; here the other proof's queued work runs meanwhile. Results in input order, equal to prove() for the same blinding values."""
    from .plonk import run_many
    key = zkey if isinstance(zkey, FflonkKey) else FflonkKey(zkey if isinstance(zkey, (bytes, bytearray)) else open(zkey, "rb").read())
    try:
        return run_many(lambda i: _prove_steps(key, witness_files[i], None, None, None if blinding_monts is None else blinding_monts[i]), len(witness_files), in_flight)
    finally:
        if not isinstance(zkey, FflonkKey):
            key.release()


def _prove_steps(zkey, witness_file, logger=None, options=None, blinding_mont=None):
    """fflonk.prove as a coroutine: `yield` stands right before every group of blocking calls (degree checks, evaluations, the collects of the four commitments); everything
    between two yields only enqueues work. The value of the generator is the proof."""
    def data(x):
        if isinstance(x, (bytes, bytearray, memoryview, np.ndarray)):
            return bytes(x)
        with open(x, "rb") as fh:
            return fh.read()

    key = zkey if isinstance(zkey, FflonkKey) else FflonkKey(data(zkey))
    f, L, r, n, power = key.f, zkmi.lib(), key.f.r, key.n, key.power
    wt = data(witness_file)
    ws, off = {}, 12
    for _ in range(struct.unpack_from("<I", wt, 8)[0]):
        t, ln = struct.unpack_from("<IQ", wt, off)
        ws[t] = (off + 12, ln)
        off += 12 + ln
    n8 = struct.unpack_from("<I", wt, ws[1][0])[0]
    if key.r != int.from_bytes(wt[ws[1][0] + 4:ws[1][0] + 4 + n8], "little"):
        raise ValueError("Curve of the witness does not match the curve of the proving key")
    n_witness = struct.unpack_from("<I", wt, ws[1][0] + 4 + n8)[0]
    nW = key.nVars - key.nAdditions
    if n_witness != nW:
        raise ValueError(f"Invalid witness length. Circuit: {key.nVars}, witness: {n_witness}, {key.nAdditions}")
    wit = np.frombuffer(wt, np.uint8, n_witness * 32, ws[2][0]).copy()
    public = [int.from_bytes(bytes(wit[32 * i:32 * i + 32]), "little") for i in range(1, key.nPublic + 1)]
    wit[:32] = 0
    if blinding_mont is None:
        bm = [None] + [bytes(f.mont(int.from_bytes(os.urandom(40), "little"))) for _ in range(9)]
    else:
        bm = [None] + [bytes(x) for x in blinding_mont]
    b = [0] + [f.unmont(x) for x in bm[1:]]                                               # logical values of b1..b9

    d_wit = zkmi.DeviceBuffer.from_host(wit)
    d_int = zkmi.DeviceBuffer(32 * max(key.nAdditions, 1))
    if key.nAdditions:                                                                    # calculateAdditions (:271-300) on the device, one launch
        zkmi.check(L.zkmi_plonk_additions_dev(f.cid, key.sec(3), key.nAdditions, d_wit.ptr, nW, d_int.ptr))
    mont = f.mont
    mp = lambda v: zkmi.ptr(mont(v))
    w_n, w_2n, w_4n = f.root(power), f.root(power + 1), f.root(power + 2)
    wv = f.unmont(w_n)
    pts, evs = {}, {}

    # ---- ROUND 1 (:318-556)
    A, B, Cw = _Poly(f, n, False), _Poly(f, n, False), _Poly(f, n, False)
    zkmi.check(L.zkmi_plonk_gather_wires_mont_dev(f.cid, d_wit.ptr, nW, d_int.ptr, key.nAdditions, key.sec(4), key.sec(5), key.sec(6), key.nConstraints, n, A.ptr, B.ptr, Cw.ptr))
    # the reference writes the blinding scalars (their Montgomery bytes) into the normal-form buffers BEFORE batchToMontgomery (:377-386): what ends up in the buffers is
    # toMontgomery of those bytes read as an integer — written here directly, behind the gather that already converted the rest
    for p, (k0, k1_) in ((A, (1, 2)), (B, (3, 4)), (Cw, (5, 6))):
        raw = np.concatenate([mont(int.from_bytes(bm[k0], "little")), mont(int.from_bytes(bm[k1_], "little"))])
        zkmi.check(L.zkmi_memcpy_h2d(p.at(n - 2), zkmi.ptr(raw), 64))
    pA, pB, pC = A.ntt(True), B.ntt(True), Cw.ntt(True)
    eA, eB, eC = pA.extended_evals(4), pB.extended_evals(4), pC.extended_evals(4)
    ev = zkmi.PlonkEvals(eA.ptr, eB.ptr, eC.ptr, None, key.sec(9, n), key.sec(7, n), key.sec(8, n), key.sec(10, n), key.sec(11, n), None, None, None, key.sec(15), A.ptr)
    T0 = _Poly(f, 4 * n, False)
    zkmi.check(L.zkmi_fflonk_t0_dev(f.cid, C.byref(ev), n, key.nPublic, T0.ptr))
    pT0 = T0.ntt(True, out=T0)
    _div_zerofier(pT0, n, 1)
    yield
    if _degree(pT0) >= 2 * n - 2:
        raise ValueError("T0 Polynomial is not well calculated")
    C1 = _cpoly(f, [pA, pB, pC, pT0], 4)
    if _degree(C1) >= 8 * n - 8:
        raise ValueError("C1 Polynomial is not well calculated")
    cm = _commit_enqueue(key, C1)
    yield
    pts["C1"] = _commit_collect(cm)

    # ---- ROUND 2 (:558-862)
    tr = _Transcript(f)
    tr.point(key.C0)
    for i in range(key.nPublic):
        tr.scalar(A.get(i))
    tr.point(pts["C1"])
    beta = tr.challenge()
    tr.reset(); tr.scalar(beta)
    gamma = tr.challenge()
    Zb = _Poly(f, n, False)
    # enqueue only: Z[0] == 1 ("Copy constraints does not match", :640-642) is read behind the next wait, as plonk.py does
    zkmi.check(L.zkmi_plonk_compute_z_enqueue(f.cid, A.ptr, B.ptr, Cw.ptr, key.sec(12, n), key.sec(13, n), key.sec(14, n), n, mp(beta), mp(gamma), mp(key.k1), mp(key.k2),
                                              zkmi.ptr(w_n), Zb.ptr))
    pZ, eZ = Zb.ifft_blinded([b[9], b[8], b[7]])
    b789 = np.concatenate([mont(b[7]), mont(b[8]), mont(b[9])])
    T1, T1z = _Poly(f, 2 * n, False), _Poly(f, 2 * n, False)
    zkmi.check(L.zkmi_fflonk_t1_dev(f.cid, eZ.ptr, key.sec(15), n, zkmi.ptr(b789), zkmi.ptr(w_2n), T1.ptr, T1z.ptr))
    pT1 = T1.ntt(True, out=T1)
    _div_zerofier(pT1, n, 1)
    pT1.axpy(T1z.ntt(True, out=T1z))
    yield
    if Zb.get(0) != 1:
        raise ValueError("Copy constraints does not match")
    if _degree(pT1) >= n + 2:
        raise ValueError("T1 Polynomial is not well calculated")
    ev2 = zkmi.PlonkEvals(eA.ptr, eB.ptr, eC.ptr, eZ.ptr, None, None, None, None, None, key.sec(12, n), key.sec(13, n), key.sec(14, n), None, None)
    T2, T2z = _Poly(f, 4 * n, False), _Poly(f, 4 * n, False)
    zkmi.check(L.zkmi_fflonk_t2_dev(f.cid, C.byref(ev2), n, zkmi.ptr(b789), mp(beta), mp(gamma), mp(key.k1), mp(key.k2), zkmi.ptr(w_n), zkmi.ptr(w_4n), T2.ptr, T2z.ptr))
    pT2 = T2.ntt(True, out=T2)
    _div_zerofier(pT2, n, 1)
    pT2.axpy(T2z.ntt(True, out=T2z))
    yield
    if _degree(pT2) >= 3 * n:
        raise ValueError("T2 Polynomial is not well calculated")
    C2 = _cpoly(f, [pZ, pT1, pT2], 3)
    if _degree(C2) >= 9 * n:
        raise ValueError("C2 Polynomial is not well calculated")
    cm = _commit_enqueue(key, C2)
    yield
    pts["C2"] = _commit_collect(cm)

    # ---- ROUND 3 (:864-963)
    tr = _Transcript(f)
    tr.scalar(gamma); tr.point(pts["C2"])
    xi_seed = tr.challenge()
    xs2 = xi_seed * xi_seed % r
    w8 = [pow(key.w8, i, r) for i in range(8)]
    w4 = [pow(key.w4, i, r) for i in range(4)]
    w3 = [1, key.w3, key.w3 * key.w3 % r]
    h0 = xs2 * xi_seed % r
    S0 = [h0 * x % r for x in w8]
    h1 = h0 * h0 % r
    S1 = [h1 * x % r for x in w4]
    h2 = h1 * xs2 % r
    S2 = [h2 * x % r for x in w3]
    h3 = h2 * key.wr % r
    S2p = [h3 * x % r for x in w3]
    xi = h2 * h2 % r * h2 % r
    xiw = xi * wv % r
    # fifteen evaluations, two waits; the selector and sigma polynomials are read where they lie in the key
    names = ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w")
    pairs = [((key.sec(t, 0), n), xi) for t in (7, 8, 9, 10, 11, 12, 13, 14)] + [((p.ptr, p.n), xi) for p in (pA, pB, pC, pZ)] + [((p.ptr, p.n), xiw) for p in (pZ, pT1, pT2)]
    vals = yield from _evaluate_all(f, pairs)
    evs.update(zip(names, vals))

    # ---- ROUND 4 (:965-1057)
    tr = _Transcript(f)
    tr.scalar(xi_seed)
    for k in ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w"):
        tr.scalar(evs[k])
    alpha = tr.challenge()
    C0p, C0n = key.sec(17, 0), 8 * n                                                     # C0 is read where it lies in the key
    # eighteen evaluations at the opening roots, three waits
    vals = yield from _evaluate_all(f, [((C0p, C0n), x) for x in S0] + [((C1.ptr, C1.n), x) for x in S1] + [((C2.ptr, C2.n), x) for x in S2 + S2p])
    R0, R1, R2 = _lagrange(S0, vals[:8], r), _lagrange(S1, vals[8:12], r), _lagrange(S2 + S2p, vals[12:], r)
    nF = max(C0n, C1.n, C2.n)
    neg = lambda v: -v % r
    # F = (C0 - R0) / ZT0 + alpha (C1 - R1) / ZT1 + alpha^2 (C2 - R2) / ZT2 (:1009-1042): each numerator one launch (zkmi_poly_lincomb_dev)
    F = lincomb(f, _Poly(f, nF, False), [(C0p, C0n, None), (_small(f, R0).ptr, len(R0), neg(1))])
    zkmi.check(L.zkmi_poly_div_by_zerofier_dev(f.cid, F.ptr, C0n, 8, mp(xi)))             # the division acts on C0's own length
    f2 = lincomb(f, _Poly(f, C1.n, False), [(C1.ptr, C1.n, alpha), (_small(f, R1).ptr, len(R1), neg(alpha))])
    _div_zerofier(f2, 4, xi)
    a2 = alpha * alpha % r
    f3 = lincomb(f, _Poly(f, C2.n, False), [(C2.ptr, C2.n, a2), (_small(f, R2).ptr, len(R2), neg(a2))])
    _div_zerofier(f3, 3, xi)
    _div_zerofier(f3, 3, xiw)
    lincomb(f, F, [(F.ptr, nF, None), (f2.ptr, f2.n, None), (f3.ptr, f3.n, None)])
    yield
    if _degree(F) >= 9 * n - 6:
        raise ValueError("F Polynomial is not well calculated")
    cm = _commit_enqueue(key, F)
    yield
    pts["W1"] = _commit_collect(cm)

    # ---- ROUND 5 (:1059-1180)
    tr = _Transcript(f)
    tr.scalar(alpha); tr.point(pts["W1"])
    y = tr.challenge()
    prod = lambda xs: functools.reduce(lambda a, x: a * ((y - x) % r) % r, xs, 1)
    mulL0, mulL1, mulL2 = prod(S0), prod(S1), prod(S2 + S2p)
    preL0, preL1, preL2 = mulL1 * mulL2 % r, alpha * mulL0 % r * mulL2 % r, alpha * alpha % r * mulL0 % r * mulL1 % r
    to_inv = {"denH1": mulL1, "denH2": mulL2}
    # L = preL0 (C0 - R0(y)) + preL1 (C1 - R1(y)) + preL2 (C2 - R2(y)) - ZT(y) F, times 1 / ZTS2(y) (:1061-1083): one launch. The scalar does not move the degree the
    # reference tests before it multiplies by it
    ZT = _zerofier(S0 + S1 + S2 + S2p, r)
    ZTS2 = _zerofier(S1 + S2 + S2p, r)
    inv2 = pow(_ev(ZTS2, y, r), -1, r)
    Lp = lincomb(f, _Poly(f, nF, False), [(C0p, C0n, preL0 * inv2 % r), (C1.ptr, C1.n, preL1 * inv2 % r), (C2.ptr, C2.n, preL2 * inv2 % r), (F.ptr, nF, neg(_ev(ZT, y, r) * inv2 % r))],
                 neg((preL0 * _ev(R0, y, r) + preL1 * _ev(R1, y, r) + preL2 * _ev(R2, y, r)) % r * inv2 % r))
    yield
    if _degree(Lp) >= 9 * n:
        raise ValueError("L Polynomial is not well calculated")
    try:
        _div_zerofier(Lp, 1, y)                     # L / (X - y): exact, so Euclidean division (:1085) = division by the zerofier
    except zkmi.ZkmiError as e:
        raise ValueError("Degree of L(X)/(ZTS2(y)(X-y)) remainder is not 0") from e
    if _degree(Lp) >= 9 * n - 1:
        raise ValueError("Degree of L(X)/(ZTS2(y)(X-y)) is not correct")
    cm = _commit_enqueue(key, Lp)
    yield
    pts["W2"] = _commit_collect(cm)

    # ---- getMontgomeryBatchedInverse (:1182-1287)
    to_inv["zh"] = (pow(xi, n, r) - 1) % r
    for name, roots in (("LiS0", S0), ("LiS1", S1)):
        ln = len(roots)
        den1 = ln * pow(roots[0], ln - 2, r) % r
        for i in range(ln):
            to_inv[f"{name}_{i + 1}"] = den1 * roots[((ln - 1) * i) % ln] % r * ((y - roots[i]) % r) % r
    den1 = 3 * S2[0] % r * ((xi - xiw) % r) % r
    for i in range(3):
        to_inv[f"LiS2_{i + 1}"] = den1 * (S2[2 * i % 3] * ((y - S2[i]) % r) % r) % r
    den1 = 3 * S2p[0] % r * ((xiw - xi) % r) % r
    for i in range(3):
        to_inv[f"LiS2_{i + 4}"] = den1 * (S2p[2 * i % 3] * ((y - S2p[i]) % r) % r) % r
    ww = 1
    for i in range(max(1, key.nPublic)):
        to_inv[f"Li_{i + 1}"] = n * ((xi - ww) % r) % r
        ww = ww * wv % r
    acc = 1
    for v in to_inv.values():
        acc = acc * v % r
    evs["inv"] = pow(acc, -1, r)

    proof = {"polynomials": {k: [str(pts[k][0]), str(pts[k][1]), "1"] for k in ("C1", "C2", "W1", "W2")},                # src/proof.js:61-83
             "evaluations": {k: str(evs[k]) for k in ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w", "inv")},
             "protocol": "fflonk", "curve": key.curve_name}
    if not isinstance(zkey, FflonkKey):
        key.release()
    return {"proof": proof, "publicSignals": [str(p) for p in public]}
