"""groth16.prove on the MI355X — host-side mirror of the reference driver (src/groth16_prove.js:28-144,
src/groth16.js:19-22): same inputs (zkey + wtns containers), same checks and error messages, same output
({proof, publicSignals} with decimal-string coordinates).  Everything between "sections read" and "proof points"
runs in the HIP library (zkmi_groth16_prove, snarkjs_amd/csrc/groth16.hip).
"""
import ctypes as C
import json
import os

import numpy as np

from . import binfile, zkmi

_BN128_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_BLS_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
_R = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
      1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}


def _curve_from_q(q):
    """src/curves.js:23-34 getCurveFromQ"""
    if q == _BN128_Q:
        return 0, "bn128"
    if q == _BLS_Q:
        return 1, "bls12381"
    raise ValueError(f"Curve not supported: {q}")


def _fr_random_mont(curve_id, rng=None):
    """Fr.random() stand-in: uniform element, returned in Montgomery form. `rng(nbytes) -> bytes`."""
    rng = rng or os.urandom
    r = _R[curve_id]
    v = int.from_bytes(rng(64), "little") % r
    return np.frombuffer(((v << 256) % r).to_bytes(32, "little"), np.uint8).copy()


def _from_mont_q(curve_id, b):
    q = _BN128_Q if curve_id == 0 else _BLS_Q
    n8 = 32 if curve_id == 0 else 48
    rinv = pow(1 << (8 * n8), -1, q)
    out = []
    for i in range(0, len(b), n8):
        out.append(int.from_bytes(bytes(b[i:i + n8]), "little") * rinv % q)
    return out


class ProvingKey:
    """A Groth16 zkey resident on the device (base tables + CSR coefficient table)."""
    _next = 1

    def __init__(self, zkey_bytes, shard=None, page_bytes=None, gaps=False):
        """shard = (rank, world): keep only this rank's index range of the five base sections on the device (multi-GPU proofs,
        BASELINE configs[2]); prove with snarkjs_amd.distributed.groth16_prove_sharded. Default: the whole key.
        page_bytes: hand the bulk sections over as pages of that many bytes (zkmi_groth16_load_paged: what the reference holds after
        readSection of a section beyond 2^30 bytes, src/groth16_prove.js:57-59); gaps (with shard and page_bytes): pages that lie wholly outside
        this shard's byte ranges are passed as gaps (NULL pointer), as a loader does that reads only its slice of the file."""
        self.zk = zk = binfile.read_groth16_zkey(zkey_bytes)
        self.curve_id, self.curve_name = _curve_from_q(zk["q"])
        self.key = ProvingKey._next
        ProvingKey._next += 1
        zkmi.init()
        self._keep = {k: np.ascontiguousarray(zk[k]) for k in
                      ("coeffs", "A", "B1", "B2", "C", "H", "vk_alpha_1", "vk_beta_1", "vk_beta_2", "vk_delta_1", "vk_delta_2")}
        p = lambda k: self._keep[k].ctypes.data
        q8 = zk["n8q"]
        for k, cnt, g in (("A", zk["nVars"], 2), ("B1", zk["nVars"], 2), ("B2", zk["nVars"], 4), ("C", zk["nVars"] - zk["nPublic"] - 1, 2), ("H", zk["domainSize"], 2)):
            if self._keep[k].size < cnt * g * q8:          # the library checks again (ZKMI_ERR_INVALID)
                raise ValueError(f"zkey section {k} is shorter than the header requires ({self._keep[k].size} < {cnt * g * q8} bytes)")
        self.shard = shard
        rng = None
        if shard is not None:
            from .distributed import shard_range
            rank, world = shard
            rng = shard_range(zk["nVars"], rank, world) + shard_range(zk["domainSize"], rank, world)
        L = zkmi.lib()
        if page_bytes:
            (v_lo, v_hi, h_lo, h_hi) = rng if rng is not None else (0, zk["nVars"], 0, zk["domainSize"])
            first_c = zk["nPublic"] + 1
            need = {"coeffs": (0, self._keep["coeffs"].size), "A": (v_lo * 2 * q8, v_hi * 2 * q8), "B1": (v_lo * 2 * q8, v_hi * 2 * q8), "B2": (v_lo * 4 * q8, v_hi * 4 * q8),
                    "C": (max(0, v_lo - first_c) * 2 * q8, max(0, v_hi - first_c) * 2 * q8), "H": (h_lo * 2 * q8, h_hi * 2 * q8)}
            self._pages = {}
            for k in ("coeffs", "A", "B1", "B2", "C", "H"):
                a = self._keep[k]
                pg = [a[o:o + page_bytes] for o in range(0, a.size, page_bytes)] or [a]
                ptrs = (C.c_void_p * len(pg))(*[x.ctypes.data for x in pg])
                if gaps and rng is not None:                 # pages wholly outside the byte range this shard reads: not provided
                    lo, hi = need[k]
                    for i in range(len(pg)):
                        if (i + 1) * page_bytes <= lo or i * page_bytes >= hi:
                            ptrs[i] = None
                lens = (C.c_size_t * len(pg))(*[x.size for x in pg])
                self._pages[k] = (pg, ptrs, lens, zkmi.Pages(C.cast(ptrs, C.POINTER(C.c_void_p)), C.cast(lens, C.POINTER(C.c_size_t)), len(pg)))
            P = lambda k: self._pages[k][3]
            self.desc = zkmi.Groth16ZkeyPaged(self.curve_id, zk["nVars"], zk["nPublic"], zk["domainSize"], P("coeffs"), P("A"), P("B1"), P("B2"), P("C"), P("H"),
                                              p("vk_alpha_1"), p("vk_beta_1"), p("vk_beta_2"), p("vk_delta_1"), p("vk_delta_2"))
            if rng is None:
                zkmi.check(L.zkmi_groth16_load_paged(C.byref(self.desc), self.key))
            else:
                zkmi.check(L.zkmi_groth16_load_shard_paged(C.byref(self.desc), self.key, *rng))
            return
        self.desc = zkmi.Groth16Zkey(self.curve_id, zk["nVars"], zk["nPublic"], zk["domainSize"], p("coeffs"), self._keep["coeffs"].size,
                                     p("A"), p("B1"), p("B2"), p("C"), p("H"),
                                     p("vk_alpha_1"), p("vk_beta_1"), p("vk_beta_2"), p("vk_delta_1"), p("vk_delta_2"),
                                     *(self._keep[k].size for k in ("A", "B1", "B2", "C", "H")))
        if rng is None:
            zkmi.check(L.zkmi_groth16_load(C.byref(self.desc), self.key))
        else:
            zkmi.check(L.zkmi_groth16_load_shard(C.byref(self.desc), self.key, *rng))

    def build_abc(self, witness=None, d_witness=None):
        """buildABC1 (src/groth16_prove.js:147-187) alone -> (A_T, B_T, C_T) as uint8 arrays of domainSize x 32 bytes (Montgomery)"""
        n = self.zk["domainSize"]
        buf = None
        if d_witness is None:
            buf = zkmi.DeviceBuffer.from_host(zkmi.u8(witness))
            d_witness = buf.ptr
        out = zkmi.DeviceBuffer(3 * n * 32)
        zkmi.check(zkmi.lib().zkmi_groth16_build_abc_dev(self.key, d_witness, out.ptr, out.ptr + n * 32, out.ptr + 2 * n * 32))
        h = out.to_host()
        out.free()
        if buf is not None:
            buf.free()
        return h[:n * 32], h[n * 32:2 * n * 32], h[2 * n * 32:]

    def coef_layout(self):
        """(records, segments, cut rows, partial-sum slots, padded terms) of the resident coefficient layout (include/zkmi_diag.h)"""
        out = (C.c_uint64 * 5)()
        zkmi.check(zkmi.lib().zkmi_groth16_coef_layout(self.key, out, 5))
        return dict(zip(("n_coef", "segments", "cut_rows", "partial_slots", "padded_terms"), [int(x) for x in out]))

    def sums_raw(self, witness, d_witness=None):
        """The five MSM sums of this key (shard) for `witness` (full witness): jA | jB1 | jB2 | jC | jH Jacobian bytes."""
        q = 32 if self.curve_id == 0 else 48
        sums = np.zeros(7 * 3 * q, np.uint8)
        buf = None
        if d_witness is None:
            buf = zkmi.DeviceBuffer.from_host(zkmi.u8(witness))
            d_witness = buf.ptr
        zkmi.check(zkmi.lib().zkmi_groth16_sums_dev(self.key, d_witness, zkmi.ptr(sums)))
        if buf is not None:
            buf.free()
        return sums

    def finish_raw(self, sums, r_mont, s_mont):
        """blinding + toAffine (src/groth16_prove.js:103-132) of complete MSM sums -> (pi_a, pi_b, pi_c) affine Montgomery bytes"""
        q = 32 if self.curve_id == 0 else 48
        pi_a, pi_b, pi_c = np.zeros(2 * q, np.uint8), np.zeros(4 * q, np.uint8), np.zeros(2 * q, np.uint8)
        r, s, sums = zkmi.u8(r_mont), zkmi.u8(s_mont), zkmi.u8(sums)
        zkmi.check(zkmi.lib().zkmi_groth16_finish(self.key, zkmi.ptr(sums), zkmi.ptr(r), zkmi.ptr(s), zkmi.ptr(pi_a), zkmi.ptr(pi_b), zkmi.ptr(pi_c)))
        return pi_a, pi_b, pi_c

    def prove_raw(self, witness, r_mont, s_mont, d_witness=None):
        """-> (pi_a, pi_b, pi_c) affine Montgomery bytes. d_witness: device pointer of an already uploaded witness."""
        q = 32 if self.curve_id == 0 else 48
        pi_a, pi_b, pi_c = np.zeros(2 * q, np.uint8), np.zeros(4 * q, np.uint8), np.zeros(2 * q, np.uint8)
        r, s = zkmi.u8(r_mont), zkmi.u8(s_mont)
        L = zkmi.lib()
        if d_witness is not None:
            zkmi.check(L.zkmi_groth16_prove_dev(self.key, d_witness, zkmi.ptr(r), zkmi.ptr(s), zkmi.ptr(pi_a), zkmi.ptr(pi_b), zkmi.ptr(pi_c)))
        else:
            w = zkmi.u8(witness)
            zkmi.check(L.zkmi_groth16_prove(None, self.key, zkmi.ptr(w), w.size, zkmi.ptr(r), zkmi.ptr(s), zkmi.ptr(pi_a), zkmi.ptr(pi_b), zkmi.ptr(pi_c)))
        return pi_a, pi_b, pi_c

    def submit(self, d_witness, slot=0):
        """Enqueue the device part of a proof into pipeline slot 0 | 1 (witness already in device memory); returns at once."""
        zkmi.check(zkmi.lib().zkmi_groth16_submit_dev(self.key, d_witness, slot))

    def collect(self, slot, r_mont, s_mont):
        """Wait for the proof in `slot`, fold and blind it -> (pi_a, pi_b, pi_c) affine Montgomery bytes."""
        q = 32 if self.curve_id == 0 else 48
        pi_a, pi_b, pi_c = np.zeros(2 * q, np.uint8), np.zeros(4 * q, np.uint8), np.zeros(2 * q, np.uint8)
        r, s = zkmi.u8(r_mont), zkmi.u8(s_mont)
        zkmi.check(zkmi.lib().zkmi_groth16_collect(self.key, slot, zkmi.ptr(r), zkmi.ptr(s), zkmi.ptr(pi_a), zkmi.ptr(pi_b), zkmi.ptr(pi_c)))
        return pi_a, pi_b, pi_c

    def stage_ms(self):
        out = (C.c_double * 11)()
        zkmi.check(zkmi.lib().zkmi_groth16_stage_ms(out, 11))
        names = ["buildABC", "ntt_x6", "joinABC", "sort_witness_B", "accum_B2", "accum_B1+sort_witness", "accum_A", "accum_C", "sort_H", "accum_H", "reduce_g1"]
        return dict(zip(names, list(out)))

    def release(self):
        if self.key:
            zkmi.lib().zkmi_groth16_release(self.key)
            self.key = 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def prove(zkey, witness_file, logger=None, options=None, r_mont=None, s_mont=None):
    """groth16.prove(zkeyFileName, witnessFileName) — files given as bytes or paths (or a ProvingKey to reuse).
    r_mont / s_mont: the two Fr.random() draws (:103-104) in Montgomery form, for bit-exact reproduction."""
    def data(x):
        if isinstance(x, (bytes, bytearray, memoryview, np.ndarray)):
            return bytes(x)
        with open(x, "rb") as f:
            return f.read()

    pk = zkey if isinstance(zkey, ProvingKey) else ProvingKey(data(zkey))
    zk = pk.zk
    wt = binfile.read_wtns(data(witness_file))
    if zk["r"] != wt["q"]:
        raise ValueError("Curve of the witness does not match the curve of the proving key")      # :41-43
    if wt["nWitness"] != zk["nVars"]:
        raise ValueError(f"Invalid witness length. Circuit: {zk['nVars']}, witness: {wt['nWitness']}")   # :45-47
    r = r_mont if r_mont is not None else _fr_random_mont(pk.curve_id)
    s = s_mont if s_mont is not None else _fr_random_mont(pk.curve_id)
    proof = raw_to_proof(pk, *pk.prove_raw(wt["witness"], r, s))
    w = wt["witness"]
    public = [str(int.from_bytes(bytes(w[i * 32:(i + 1) * 32]), "little")) for i in range(1, zk["nPublic"] + 1)]   # :123-128
    if not isinstance(zkey, ProvingKey):
        pk.release()
    return {"proof": proof, "publicSignals": public}


def raw_to_proof(pk, pi_a, pi_b, pi_c):
    """(pi_a, pi_b, pi_c) affine Montgomery bytes -> the reference's proof object (src/groth16_prove.js:130-141)"""
    n8q = pk.zk["n8q"]
    to_le = lambda vals: b"".join(int(v).to_bytes(n8q, "little") for v in vals)
    proof, _ = binfile.proof_json(pk.curve_name, n8q, to_le(_from_mont_q(pk.curve_id, pi_a)), to_le(_from_mont_q(pk.curve_id, pi_b)),
                                  to_le(_from_mont_q(pk.curve_id, pi_c)))
    return proof


def proof_to_json(proof):
    """JSON.stringify(proof) as snarkjs writes it (key order of src/groth16_prove.js:130-141)."""
    return json.dumps(proof, separators=(",", ":"))
