"""plonk.prove on the MI355X — host-side mirror of the reference driver (src/plonk_prove.js:47-888, src/plonk.js).

Same inputs (plonk zkey + wtns containers), same checks and error messages, same output ({proof, publicSignals} with
decimal strings in the reference's key order).  Every O(n) step runs in the HIP library and the data stays in device
memory between rounds; what remains on the host is what the reference also does with O(1) work: the Keccak-256
Fiat-Shamir transcript (src/Keccak256Transcript.js), challenge arithmetic, blinding of a few coefficients, proof
assembly — and calculateAdditions (:174-204), a data-dependent sequential chain.

    round 1  gather wires -> batchToMontgomery -> 3 iNTT(n) + 3 NTT(4n) -> 3 MSM           zkmi_plonk_gather_wires_dev, zkmi_ntt_dev, zkmi_msm_dev
    round 2  computeZ (factors, 2 product scans, batch inverse) -> iNTT, NTT(4n), MSM       zkmi_plonk_compute_z_dev
    round 3  computeT over 4n points (MulZ) -> 2 iNTT(4n), divZh, split -> 3 MSM            zkmi_plonk_compute_t_dev, zkmi_poly_div_zh_dev
    round 4  6 Horner evaluations                                                            zkmi_poly_evaluate_dev
    round 5  linearisation R (axpy chain), 2 x divByZerofier -> 2 MSM                        zkmi_poly_axpy_dev, zkmi_poly_div_by_zerofier_dev
"""
import ctypes as C
import os
import struct

import numpy as np

from . import zkmi
from .groth16 import _curve_from_q, _R

_Q = {0: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
      1: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab}

# ---- Keccak-256 (original padding 0x01, as @noble/hashes keccak_256 used by src/Keccak256Transcript.js) ----------------------
def keccak256(data):
    """host routine of the library (csrc/plonk.hip: zkmi_keccak256); needs no device"""
    data, out = bytes(data), C.create_string_buffer(32)
    zkmi.check(zkmi.lib().zkmi_keccak256(data, len(data), out))
    return out.raw


class _Field:
    """Fr / Fq conversions for one curve: Montgomery little-endian bytes <-> Python ints (normal form)."""

    def __init__(self, cid):
        self.cid, self.r, self.q = cid, _R[cid], _Q[cid]
        self.n8q = 32 if cid == 0 else 48
        self.Rr, self.Rq = pow(2, 256, self.r), pow(2, 8 * self.n8q, self.q)
        self.Rri, self.Rqi = pow(self.Rr, -1, self.r), pow(self.Rq, -1, self.q)

    def mont(self, v):        # int -> 32 Montgomery bytes (numpy)
        return np.frombuffer((v % self.r * self.Rr % self.r).to_bytes(32, "little"), np.uint8).copy()

    def unmont(self, b):      # 32 Montgomery bytes -> int
        return int.from_bytes(bytes(b), "little") * self.Rri % self.r

    def unmont_q(self, b):
        return int.from_bytes(bytes(b), "little") * self.Rqi % self.q

    def root(self, i):
        out = np.zeros(32, np.uint8)
        zkmi.check(zkmi.lib().zkmi_fr_root(self.cid, i, zkmi.ptr(out)))
        return out


class _Poly:
    """Coefficients (or evaluations) of `n` Montgomery Fr elements in device memory (the reference's Polynomial / Evaluations)."""

    def __init__(self, fld, n, zero=True):
        self.f, self.n = fld, n
        self.buf = zkmi.DeviceBuffer(max(n, 1) * 32)
        if zero:
            zkmi.check(zkmi.lib().zkmi_memset_dev(self.buf.ptr, 0, n * 32))

    @property
    def ptr(self):
        return self.buf.ptr

    def at(self, i):
        return self.buf.ptr + 32 * i

    def copy_from(self, src_ptr, count, dst_off=0):
        zkmi.check(zkmi.lib().zkmi_memcpy_d2d(self.at(dst_off), src_ptr, count * 32))
        return self

    def get(self, i):                     # getCoef -> int
        out = np.empty(32, np.uint8)
        zkmi.check(zkmi.lib().zkmi_memcpy_d2h(zkmi.ptr(out), self.at(i), 32))
        return self.f.unmont(out)

    def set(self, i, v):                  # setCoef
        b = self.f.mont(v)
        zkmi.check(zkmi.lib().zkmi_memcpy_h2d(self.at(i), zkmi.ptr(b), 32))

    def axpy(self, other, k=None, sub=False, count=None):
        """this.add(polynomial, blindingValue) / this.sub (polynomial.js:218-276); the caller sizes `self` to the max length"""
        cnt = other.n if count is None else count
        assert cnt <= self.n
        kb = None if k is None else zkmi.ptr(self.f.mont(k))
        zkmi.check(zkmi.lib().zkmi_poly_axpy_dev(self.f.cid, self.ptr, other.ptr, cnt, kb, int(sub)))

    def scale(self, k):
        zkmi.check(zkmi.lib().zkmi_poly_scale_dev(self.f.cid, self.ptr, self.n, zkmi.ptr(self.f.mont(k))))

    def add_scalar(self, v):
        zkmi.check(zkmi.lib().zkmi_poly_add_scalar_dev(self.f.cid, self.ptr, zkmi.ptr(self.f.mont(v))))

    def evaluate(self, x):
        out = np.empty(32, np.uint8)
        zkmi.check(zkmi.lib().zkmi_poly_evaluate_dev(self.f.cid, self.ptr, self.n, zkmi.ptr(self.f.mont(x)), zkmi.ptr(out)))
        return self.f.unmont(out)

    def tail_is_zero(self, start):
        z = C.c_int(1)
        zkmi.check(zkmi.lib().zkmi_poly_is_zero_dev(self.f.cid, self.at(start), self.n - start, C.byref(z)))
        return bool(z.value)

    def blinded(self, factors):
        """blindCoefficients (polynomial.js:68-93): length grows by len(factors)"""
        out = _Poly(self.f, self.n + len(factors))
        out.copy_from(self.ptr, self.n)
        fb = np.concatenate([self.f.mont(fct) for fct in factors])
        zkmi.check(zkmi.lib().zkmi_poly_blind_dev(self.f.cid, out.ptr, self.n, zkmi.ptr(fb), len(factors)))
        return out

    def ntt(self, inverse, out=None):
        out = out or _Poly(self.f, self.n, zero=False)
        zkmi.check(zkmi.lib().zkmi_ntt_dev(self.f.cid, self.ptr, out.ptr, self.n.bit_length() - 1, int(inverse), None, None))
        return out

    def extended_evals(self, ext):
        """Evaluations.fromPolynomial(p, 4): zero-pad to ext*n coefficients, then fft (evaluations.js:30-37)"""
        e = _Poly(self.f, self.n * ext, zero=False)                                       # the padding is read as zero by the first pass, never written (zkmi_ntt_padded_dev)
        zkmi.check(zkmi.lib().zkmi_ntt_padded_dev(self.f.cid, self.ptr, self.n, e.ptr, (self.n * ext).bit_length() - 1, 0))
        return e

    def ifft_blinded(self, factors):
        """The pattern of rounds 1 and 2 (plonk_prove.js:285-311, :441-455) in four launches less per polynomial: coefficients = ifft(self) into a buffer with room for the
        blinding tail, Evaluations.fromPolynomial(coefficients, 4) with the zero padding READ instead of written (zkmi_ntt_padded_dev), then blindCoefficients in place
        (zkmi_poly_blind_tail_dev) -> (blinded polynomial of n + len(factors) coefficients, 4n evaluations of the unblinded one)"""
        L, f, n, k = zkmi.lib(), self.f, self.n, len(factors)
        out = _Poly(f, n + k, zero=False)
        zkmi.check(L.zkmi_ntt_dev(f.cid, self.ptr, out.ptr, n.bit_length() - 1, 1, None, None))
        ev = _Poly(f, 4 * n, zero=False)
        zkmi.check(L.zkmi_ntt_padded_dev(f.cid, out.ptr, n, ev.ptr, (4 * n).bit_length() - 1, 0))
        fb = np.concatenate([f.mont(fct) for fct in factors])
        zkmi.check(L.zkmi_poly_blind_tail_dev(f.cid, out.ptr, n, zkmi.ptr(fb), k))
        return out, ev

    def free(self):
        self.buf.free()


def lincomb(f, out, terms, constant=None):
    """out[i] = sum_j k_j p_j[i] + (i == 0 ? constant : 0) in one launch (zkmi_poly_lincomb_dev). terms: (device pointer, length, k as an integer or None for 1)"""
    arr = (zkmi.PolyTerm * len(terms))()
    for t, (ptr, ln, k) in zip(arr, terms):
        t.d_p, t.len = ptr, ln
        if k is not None:
            t.has_k = 1
            C.memmove(t.k, zkmi.ptr(f.mont(k)), 32)
    cb = None if constant is None else zkmi.ptr(f.mont(constant))
    zkmi.check(zkmi.lib().zkmi_poly_lincomb_dev(f.cid, out.ptr, out.n, arr, len(terms), cb))
    return out


def evaluate_many(f, polys, xs):
    """[p(x)] for (device pointer, length) pairs and points, one wait (zkmi_poly_evaluate_multi_dev)"""
    cnt = len(polys)
    ptrs = (C.c_void_p * cnt)(*[p for p, _ in polys])
    lens = (C.c_size_t * cnt)(*[ln for _, ln in polys])
    xb = np.concatenate([f.mont(x) for x in xs])
    out = np.empty(32 * cnt, np.uint8)
    zkmi.check(zkmi.lib().zkmi_poly_evaluate_multi_dev(f.cid, ptrs, lens, zkmi.ptr(xb), cnt, zkmi.ptr(out)))
    return [f.unmont(out[32 * i:32 * i + 32]) for i in range(cnt)]


class PlonkKey:
    """A PLONK zkey resident on the device (selector / permutation sections, Lagrange evaluations, SRS points)."""

    def __init__(self, zkey_bytes):
        data = bytes(zkey_bytes)
        s = self.sections = {}
        nsec = struct.unpack_from("<I", data, 8)[0]
        off = 12
        for _ in range(nsec):
            t, ln = struct.unpack_from("<IQ", data, off)
            off += 12
            s[t] = (off, ln)
            off += ln
        if struct.unpack_from("<I", data, s[1][0])[0] != 2:
            raise ValueError("zkey file is not plonk")                                   # plonk_prove.js:60-62
        off = s[2][0]
        n8q = struct.unpack_from("<I", data, off)[0]
        q = int.from_bytes(data[off + 4:off + 4 + n8q], "little"); off += 4 + n8q
        n8r = struct.unpack_from("<I", data, off)[0]
        self.r = int.from_bytes(data[off + 4:off + 4 + n8r], "little"); off += 4 + n8r
        self.curve_id, self.curve_name = _curve_from_q(q)
        self.f = f = _Field(self.curve_id)
        self.nVars, self.nPublic, self.n, self.nAdditions, self.nConstraints = struct.unpack_from("<IIIII", data, off); off += 20
        self.power = self.n.bit_length() - 1
        self.k1, self.k2 = f.unmont(data[off:off + 32]), f.unmont(data[off + 32:off + 64]); off += 64
        self.commit = {}
        for nm in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):                       # src/zkey_utils.js:283-290
            self.commit[nm] = (f.unmont_q(data[off:off + n8q]), f.unmont_q(data[off + n8q:off + 2 * n8q])); off += 2 * n8q
        zkmi.init()
        if s[3][1] < 72 * self.nAdditions:
            raise ValueError("zkey additions section is shorter than its header says")
        # section 3 (additions) goes to the device as it lies in the file: calculateAdditions runs there, once per proof (zkmi_plonk_additions_dev)
        self.dev = {t: zkmi.DeviceBuffer.from_host(np.frombuffer(data, np.uint8, s[t][1], s[t][0])) for t in (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14) if s[t][1]}
        # the SRS is static: pre-computed window tables for the nine commitments of every proof (all use a prefix of PTau)
        self.n_ptau = s[14][1] // (2 * f.n8q)
        self.ptau_table = C.c_uint64(0)
        zkmi.check(zkmi.lib().zkmi_msm_table_build(self.curve_id, 1, self.dev[14].ptr, self.n_ptau, C.byref(self.ptau_table)))

    def sec(self, t, elem_off=0):
        return self.dev[t].ptr + 32 * elem_off

    def release(self):
        for b in self.dev.values():
            b.free()
        self.dev = {}
        if self.ptau_table.value:
            zkmi.lib().zkmi_msm_table_release(self.ptau_table)
            self.ptau_table = C.c_uint64(0)


class PlonkWitness:
    """A witness resident on the device for any number of proofs against `key` (the reference reads the .wtns file once per proof,
    src/plonk_prove.js:83-97): header checks, the public signals, the signals uploaded once. Read-only afterwards: proofs on both pipeline
    slots share it. calculateAdditions (:174-204) is NOT done here: it is per-proof work of the reference and runs on the device at the
    start of every prove() (zkmi_plonk_additions_dev), inside whatever the caller times."""

    def __init__(self, key, wt):
        f, r = key.f, key.f.r
        ws = {}
        off = 12
        for _ in range(struct.unpack_from("<I", wt, 8)[0]):
            t, ln = struct.unpack_from("<IQ", wt, off)
            ws[t] = (off + 12, ln)
            off += 12 + ln
        n8 = struct.unpack_from("<I", wt, ws[1][0])[0]
        wq = int.from_bytes(wt[ws[1][0] + 4:ws[1][0] + 4 + n8], "little")
        n_witness = struct.unpack_from("<I", wt, ws[1][0] + 4 + n8)[0]
        if key.r != wq:
            raise ValueError("Curve of the witness does not match the curve of the proving key")
        if n_witness != key.nVars - key.nAdditions:
            raise ValueError(f"Invalid witness length. Circuit: {key.nVars}, witness: {n_witness}, {key.nAdditions}")
        wit = np.frombuffer(wt, np.uint8, n_witness * 32, ws[2][0])                       # a view: the signal 0 slot is cleared on the device
        self.public = [int.from_bytes(bytes(wit[32 * i:32 * i + 32]), "little") for i in range(1, key.nPublic + 1)]
        self.nW = key.nVars - key.nAdditions
        self.d_wit = zkmi.DeviceBuffer.from_host(wit)
        zkmi.check(zkmi.lib().zkmi_memset_dev(self.d_wit.ptr, 0, 32))                     # :94-96
        zkmi.check(zkmi.lib().zkmi_synchronize())

    def release(self):
        self.d_wit.free()


class _Transcript:
    def __init__(self, f):
        self.f, self.parts = f, []

    def reset(self):
        self.parts = []

    def point(self, p):                                  # addPolCommitment -> G1.toRprUncompressed: x, y big-endian normal form
        self.parts.append(p[0].to_bytes(self.f.n8q, "big") + p[1].to_bytes(self.f.n8q, "big"))

    def scalar(self, v):                                 # addScalar -> Fr.toRprBE
        self.parts.append(v.to_bytes(32, "big"))

    def challenge(self):
        if not self.parts:
            raise ValueError("Keccak256Transcript: No data to generate a transcript")
        return int.from_bytes(keccak256(b"".join(self.parts)), "big") % self.f.r


def _commit_enqueue(key, *polys):
    """Polynomial.multiExponentiation (polynomial.js:970-977) for the commitments of one round, first half: batchFromMontgomery and the MSMs over PTau[0:len] (resident table;
    the bucket reductions of the round share one set of launches) are ENQUEUED on the active pipeline slot (zkmi_msm_table_multi_enqueue_dev); nothing waits. r06: with two
    proofs in flight the other proof's next segment — its own commitments included — is enqueued before this one is collected, so its accumulations run underneath this round's
    latency-bound reduction tail instead of behind a host that sits inside a blocking call."""
    L, cnt = zkmi.lib(), len(polys)
    ptrs = (C.c_void_p * cnt)(*[p.ptr for p in polys])
    ks = (C.c_size_t * cnt)(*[p.n for p in polys])
    # r06: the conversions of the round in one launch into the slot's scratch memory, inside the call (zkmi_msm_table_multi_enqueue_mont_dev)
    zkmi.check(L.zkmi_msm_table_multi_enqueue_mont_dev(key.ptau_table, ptrs, ks, cnt))
    return key, polys, cnt


def _commit_collect(state):
    """second half: wait for the round's MSMs, fold, toAffine -> [(x, y)] as integers"""
    key, _polys, cnt = state
    f, L = key.f, zkmi.lib()
    jac = np.zeros(cnt * 3 * f.n8q, np.uint8)
    zkmi.check(L.zkmi_msm_table_multi_collect(key.ptau_table, cnt, zkmi.ptr(jac)))
    out = []
    for i in range(cnt):
        aff = np.zeros(2 * f.n8q, np.uint8)
        zkmi.check(L.zkmi_to_affine(f.cid, 1, zkmi.ptr(jac[i * 3 * f.n8q:(i + 1) * 3 * f.n8q].copy()), zkmi.ptr(aff)))
        out.append((f.unmont_q(aff[:f.n8q]), f.unmont_q(aff[f.n8q:])))
    return out


def _commit(key, *polys):
    return _commit_collect(_commit_enqueue(key, *polys))


def prove(zkey, witness_file, logger=None, options=None, blinding_mont=None):
    """plonk.prove(zkeyFileName, witnessFileName). blinding_mont: the 11 Fr.random() draws (:224-227) as Montgomery bytes,
    for bit-exact reproduction; default = fresh randomness."""
    steps = _prove_steps(zkey, witness_file, logger, options, blinding_mont)
    try:
        while True:
            next(steps)
    except StopIteration as done:
        return done.value


def prove_many(zkey, witness_files, blinding_monts=None, in_flight=2):
    """Throughput mode: one proof per witness against the same key, TWO in flight from this one host thread. A PLONK proof is a chain of
    rounds separated by transcript hashes, so one proof alone leaves the GPU idle while the host hashes, folds window sums and launches, and
    runs its memory- and latency-bound phases (digit sorts, bucket reductions, small scans) with nothing beside them. Here every proof is a
    coroutine (_prove_steps) that yields right before each of its long blocking calls; the driver switches the library's pipeline slot
    (zkmi_pipeline_select: own stream, scratch, allocation pool) and lets the other proof enqueue up to ITS next blocking call first, so the
    GPU always holds queued work of the other proof while the host waits. Results come back in input order and are the proofs plonk.prove
    would return for the same blinding values."""
    key = zkey if isinstance(zkey, PlonkKey) else PlonkKey(zkey if isinstance(zkey, (bytes, bytearray)) else open(zkey, "rb").read())
    try:
        return run_many(lambda i: _prove_steps(key, witness_files[i], None, None, None if blinding_monts is None else blinding_monts[i]), len(witness_files), in_flight)
    finally:
        if not isinstance(zkey, PlonkKey):
            key.release()


def run_many(make_steps, n, in_flight=2):
    """The two-slot driver shared by plonk.prove_many and fflonk.prove_many: make_steps(i) -> the coroutine of proof i; at most two live, each on its own pipeline slot."""
    L = zkmi.lib()
    out = [None] * n
    live, free, nxt = [], list(range(max(1, min(2, in_flight)))), 0
    try:
        while nxt < n or live:
            while free and nxt < n:
                live.append((free.pop(0), nxt, make_steps(nxt)))
                nxt += 1
            for ent in list(live):
                slot, idx, steps = ent
                zkmi.check(L.zkmi_pipeline_select(slot))
                try:
                    next(steps)
                except StopIteration as done:
                    out[idx] = done.value
                    live.remove(ent)
                    free.append(slot)
    finally:
        for slot, _, steps in live:                    # an error in one proof: drop the other one too, leave no queued work behind
            try:
                L.zkmi_pipeline_select(slot)
                steps.close()
                L.zkmi_synchronize()
            except Exception:
                pass
        L.zkmi_pipeline_select(0)
    return out


def _prove_steps(zkey, witness_file, logger=None, options=None, blinding_mont=None):
    """plonk.prove as a coroutine: `yield` stands right before every long blocking call (the four commitment rounds, the divisibility check behind
    the T pipeline, the round-4 evaluations); the value of the generator is the proof. Everything between two yields only enqueues work."""
    def data(x):
        if isinstance(x, (bytes, bytearray)):
            return x                                                   # parsed in place: no copy of a 2^20-signal witness per proof
        if isinstance(x, (memoryview, np.ndarray)):
            return bytes(x)
        with open(x, "rb") as fh:
            return fh.read()

    key = zkey if isinstance(zkey, PlonkKey) else PlonkKey(data(zkey))
    f, L, r, n, power = key.f, zkmi.lib(), key.f.r, key.n, key.power
    wres = witness_file if isinstance(witness_file, PlonkWitness) else PlonkWitness(key, data(witness_file))
    public, d_wit, nW = wres.public, wres.d_wit, wres.nW
    if blinding_mont is None:
        b = [0] + [int.from_bytes(os.urandom(64), "little") % r for _ in range(11)]
    else:
        b = [0] + [f.unmont(x) for x in blinding_mont]

    tr = _Transcript(f)
    pts, evs = {}, {}
    w_n, w_4n, w_2 = f.root(power), f.root(power + 2), f.root(2)
    mont = f.mont

    # ---- ROUND 1 (:222-313)
    # calculateAdditions (:174-204): the internal signals, one launch on the device (this slot's own buffer: two proofs may be in flight)
    d_int = zkmi.DeviceBuffer(32 * max(key.nAdditions, 1))
    if key.nAdditions:
        zkmi.check(L.zkmi_plonk_additions_dev(f.cid, key.sec(3), key.nAdditions, d_wit.ptr, nW, d_int.ptr))
    A, B, Cw = _Poly(f, n, False), _Poly(f, n, False), _Poly(f, n, False)
    zkmi.check(L.zkmi_plonk_gather_wires_mont_dev(f.cid, d_wit.ptr, nW, d_int.ptr, key.nAdditions, key.sec(4), key.sec(5), key.sec(6), key.nConstraints, n,
                                                  A.ptr, B.ptr, Cw.ptr))                 # buffers + Fr.batchToMontgomery (:267-283) in one pass
    d_int.free()                                                                          # stream-ordered: the gather above is the last reader
    (pA, eA), (pB, eB), (pC, eC) = A.ifft_blinded([b[2], b[1]]), B.ifft_blinded([b[4], b[3]]), Cw.ifft_blinded([b[6], b[5]])
    cm = _commit_enqueue(key, pA, pB, pC)
    yield
    pts["A"], pts["B"], pts["C"] = _commit_collect(cm)

    # ---- ROUND 2 (:315-455)
    tr.reset()
    for nm in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        tr.point(key.commit[nm])
    for i in range(key.nPublic):
        tr.scalar(A.get(i))
    for nm in ("A", "B", "C"):
        tr.point(pts[nm])
    beta = tr.challenge()
    tr.reset(); tr.scalar(beta)
    gamma = tr.challenge()
    Zb = _Poly(f, n, False)
    zkmi.check(L.zkmi_plonk_compute_z_enqueue(f.cid, A.ptr, B.ptr, Cw.ptr, key.sec(12, n), key.sec(12, 6 * n), key.sec(12, 11 * n), n, zkmi.ptr(mont(beta)),
                                              zkmi.ptr(mont(gamma)), zkmi.ptr(mont(key.k1)), zkmi.ptr(mont(key.k2)), zkmi.ptr(w_n), Zb.ptr))
    pZ, eZ = Zb.ifft_blinded([b[9], b[8], b[7]])
    cm = _commit_enqueue(key, pZ)
    yield
    pts["Z"], = _commit_collect(cm)
    if Zb.get(0) != 1:                                                                   # computeZ's check (:437-439), read behind the commitment's
        raise ValueError("Copy constraints does not match")                              # own wait: no extra bubble between the transforms and the MSM

    # ---- ROUND 3 (:457-684)
    tr.reset(); tr.scalar(beta); tr.scalar(gamma); tr.point(pts["Z"])
    alpha = tr.challenge()
    ev = zkmi.PlonkEvals(eA.ptr, eB.ptr, eC.ptr, eZ.ptr, key.sec(7, n), key.sec(8, n), key.sec(9, n), key.sec(10, n), key.sec(11, n),
                         key.sec(12, n), key.sec(12, 6 * n), key.sec(12, 11 * n), key.sec(13), A.ptr)
    T, Tz = _Poly(f, 4 * n, False), _Poly(f, 4 * n, False)
    blind = np.concatenate([mont(b[i]) for i in range(1, 12)])
    zkmi.check(L.zkmi_plonk_compute_t_dev(f.cid, C.byref(ev), n, key.nPublic, zkmi.ptr(blind), zkmi.ptr(mont(beta)), zkmi.ptr(mont(gamma)), zkmi.ptr(mont(alpha)),
                                          zkmi.ptr(mont(key.k1)), zkmi.ptr(mont(key.k2)), zkmi.ptr(w_n), zkmi.ptr(w_4n), zkmi.ptr(w_2), T.ptr, Tz.ptr))
    pT = T.ntt(True, out=T)
    zkmi.check(L.zkmi_poly_div_zh_dev(f.cid, pT.ptr, 4 * n, n, 4))
    pTz = Tz.ntt(True, out=Tz)
    pT.axpy(pTz)
    yield
    if not pT.tail_is_zero(3 * n + 6):
        raise ValueError("T Polynomial is not well calculated")                          # :645-647
    T1, T2, T3 = _Poly(f, n + 1, False), _Poly(f, n + 1, False), _Poly(f, n + 6, False)
    zkmi.check(L.zkmi_plonk_split_t_dev(f.cid, pT.ptr, 4 * n, n, zkmi.ptr(mont(b[10])), zkmi.ptr(mont(b[11])), T1.ptr, T2.ptr, T3.ptr))      # :649-672 in one launch
    cm = _commit_enqueue(key, T1, T2, T3)
    yield
    pts["T1"], pts["T2"], pts["T3"] = _commit_collect(cm)

    # ---- ROUND 4 (:686-708)
    tr.reset(); tr.scalar(alpha)
    for nm in ("T1", "T2", "T3"):
        tr.point(pts[nm])
    xi = tr.challenge()
    xiw = xi * f.unmont(w_n) % r
    S1, S2, S3 = key.sec(12, 0), key.sec(12, 5 * n), key.sec(12, 10 * n)                  # the coefficient halves of the sigma section, read where they lie
    yield
    vals = evaluate_many(f, [(pA.ptr, pA.n), (pB.ptr, pB.n), (pC.ptr, pC.n), (S1, n), (S2, n), (pZ.ptr, pZ.n)], [xi, xi, xi, xi, xi, xiw])    # six evaluations, one wait
    for k, v_ in zip(("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"), vals):
        evs[k] = v_

    # ---- ROUND 5 (:710-888)
    tr.reset(); tr.scalar(xi)
    for k in ("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"):
        tr.scalar(evs[k])
    v = [0, tr.challenge()]
    for i in range(2, 6):
        v.append(v[i - 1] * v[1] % r)
    xin = pow(xi, n, r)
    zh = (xin - 1) % r
    wv = f.unmont(w_n)
    Lg, ww = [0], 1
    for i in range(1, max(1, key.nPublic) + 1):
        Lg.append(ww * zh % r * pow(n * (xi - ww) % r, -1, r) % r)
        ww = ww * wv % r
    eval_l1 = (xin - 1) * pow(n * (xi - 1) % r, -1, r) % r
    eval_pi = 0
    for i, pub in enumerate(public):
        eval_pi = (eval_pi - pub * Lg[i + 1]) % r
    ea, eb, ec, es1, es2, ezw = (evs[k] for k in ("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"))
    alpha2, betaxi = alpha * alpha % r, beta * xi % r
    e2 = (ea + betaxi + gamma) * (eb + betaxi * key.k1 + gamma) % r * (ec + betaxi * key.k2 + gamma) % r * alpha % r
    e3 = (ea + beta * es1 + gamma) * (eb + beta * es2 + gamma) % r * ezw % r * alpha % r
    e4 = eval_l1 * alpha2 % r
    # The linearisation polynomial R (:769-838) and the opening numerator Wxi = R + v1 (A - a) + ... (:840-866) are ONE linear combination of fifteen resident polynomials:
    # a single launch (zkmi_poly_lincomb_dev) instead of 18 add / sub, 2 mulScalar and 5 copies of selector polynomials; exact arithmetic, same coefficients.
    zh_n, xin2 = -zh % r, xin * xin % r
    r0 = (eval_pi - e3 * (ec + gamma) - e4) % r
    Wxi = lincomb(f, _Poly(f, n + 6, False), [
        (key.sec(7, 0), n, ea * eb % r), (key.sec(8, 0), n, ea), (key.sec(9, 0), n, eb), (key.sec(10, 0), n, ec), (key.sec(11, 0), n, None),
        (pZ.ptr, pZ.n, (e2 + e4) % r), (S3, n, -(e3 * beta) % r),
        (T1.ptr, T1.n, zh_n), (T2.ptr, T2.n, zh_n * xin % r), (T3.ptr, T3.n, zh_n * xin2 % r),
        (pA.ptr, pA.n, v[1]), (pB.ptr, pB.n, v[2]), (pC.ptr, pC.n, v[3]), (S1, n, v[4]), (S2, n, v[5])],
        (r0 - (v[1] * ea + v[2] * eb + v[3] * ec + v[4] * es1 + v[5] * es2)) % r)
    zkmi.check(L.zkmi_poly_div_by_zerofier_enqueue(f.cid, Wxi.ptr, Wxi.n, 1, zkmi.ptr(mont(xi))))
    Wxiw = lincomb(f, _Poly(f, pZ.n, False), [(pZ.ptr, pZ.n, None)], -ezw % r)
    zkmi.check(L.zkmi_poly_div_by_zerofier_enqueue(f.cid, Wxiw.ptr, Wxiw.n, 1, zkmi.ptr(mont(xiw))))
    cm = _commit_enqueue(key, Wxi, Wxiw)
    yield
    pts["Wxi"], pts["Wxiw"] = _commit_collect(cm)
    if not (Wxi.tail_is_zero(Wxi.n - 1) and Wxiw.tail_is_zero(Wxiw.n - 1)):             # divByZerofier's test (polynomial.js:665-669), read behind the commitments' wait
        raise ValueError("Polynomial is not divisible")

    proof = {}
    for nm in ("A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw"):                       # src/proof.js:61-83 (insertion order)
        proof[nm] = [str(pts[nm][0]), str(pts[nm][1]), "1"]
    for k in ("eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"):
        proof[k] = str(evs[k])
    proof["protocol"] = "plonk"
    proof["curve"] = key.curve_name
    if not isinstance(zkey, PlonkKey):
        key.release()
    return {"proof": proof, "publicSignals": [str(p) for p in public]}
