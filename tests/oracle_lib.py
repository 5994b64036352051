"""ctypes binding of oracle/libzkoracle.so — the CPU restatement used as the parity checker.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(_ORACLE_DIR, "libzkoracle.so")

BN128, BLS12381 = 0, 1
CURVE_ID = {"bn128": 0, "bls12381": 1}


def build():
    src = os.path.join(_ORACLE_DIR, "zk_oracle.c")
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "libzkoracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_fr_from_u64.argtypes = [C.c_int, C.c_uint64, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(x):
    if isinstance(x, (bytes, bytearray, memoryview)):
        x = np.frombuffer(bytes(x), dtype=np.uint8)
    return np.ascontiguousarray(x, dtype=np.uint8)


def n8q(curve):
    return lib().orc_n8q(curve)


def fr_w(curve, i):
    out = np.zeros(32, np.uint8)
    lib().orc_fr_w(curve, i, _p(out))
    return out


def fr_one(curve):
    out = np.zeros(32, np.uint8)
    lib().orc_fr_one(curve, _p(out))
    return out


def fr_e(curve, v):
    out = np.zeros(32, np.uint8)
    lib().orc_fr_from_u64(curve, v, _p(out))
    return out


def generator(curve, group):
    out = np.zeros(2 * group * n8q(curve), np.uint8)
    lib().orc_generator(curve, group, _p(out))
    return out


def ntt(curve, x, inverse=False):
    x = _u8(x)
    n = x.size // 32
    out = np.empty_like(x)
    rc = lib().orc_fr_ntt(curve, _p(x), _p(out), C.c_uint(max(n, 1).bit_length() - 1), int(inverse))
    assert rc == 0
    return out


def _batch(fn, curve, x):
    x = _u8(x)
    out = np.empty_like(x)
    assert fn(curve, _p(x), _p(out), C.c_size_t(x.size // 32)) == 0
    return out


def to_mont(curve, x):
    return _batch(lib().orc_fr_batch_to_mont, curve, x)


def from_mont(curve, x):
    return _batch(lib().orc_fr_batch_from_mont, curve, x)


def batch_inverse(curve, x):
    return _batch(lib().orc_fr_batch_inverse, curve, x)


def apply_key(curve, x, first, inc):
    x = _u8(x)
    out = np.empty_like(x)
    first, inc = _u8(first), _u8(inc)
    assert lib().orc_fr_batch_apply_key(curve, _p(x), _p(out), C.c_size_t(x.size // 32), _p(first), _p(inc)) == 0
    return out


def msm(curve, group, bases, scalars, n, scalar_bytes=32, naive=False):
    bases, scalars = _u8(bases), _u8(scalars)
    out = np.zeros(3 * group * n8q(curve), np.uint8)
    fn = lib().orc_msm_naive if naive else lib().orc_msm
    assert fn(curve, group, _p(bases), _p(scalars), C.c_size_t(n), scalar_bytes, _p(out)) == 0
    return out


def to_affine(curve, group, jac):
    jac = _u8(jac)
    out = np.zeros(2 * group * n8q(curve), np.uint8)
    lib().orc_to_affine(curve, group, _p(jac), _p(out))
    return out


def generator_mul(curve, group, k: int):
    kb = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    out = np.zeros(3 * group * n8q(curve), np.uint8)
    lib().orc_generator_mul(curve, group, _p(kb), 32, _p(out))
    return out


def geom_bases(curve, group, n):
    out = np.zeros(n * 2 * group * n8q(curve), np.uint8)
    lib().orc_geom_bases(curve, group, C.c_size_t(n), _p(out))
    return out


def geom_dot(curve, scalars, n, sb=32, f=7, g=11, skip_mod=0, skip_rem=0) -> int:
    """sum_i s_i * f * g^i mod r over entries with i % skip_mod != skip_rem (oracle/zk_oracle.c: orc_fr_geom_dot)."""
    scalars = _u8(scalars)
    out = np.zeros(32, np.uint8)
    assert lib().orc_fr_geom_dot(curve, _p(scalars), C.c_size_t(n), sb, C.c_uint64(f), C.c_uint64(g), skip_mod, skip_rem, _p(out)) == 0
    return int.from_bytes(out.tobytes(), "little")


def group_fft(curve, group, pts, inverse=False):
    """G.fft / G.ifft over n affine points (oracle/zk_oracle.c: orc_group_fft)"""
    pts = _u8(pts)
    n = pts.size // (2 * group * n8q(curve))
    out = np.empty_like(pts)
    assert lib().orc_group_fft(curve, group, _p(pts), _p(out), C.c_uint(max(n, 1).bit_length() - 1), int(inverse)) == 0
    return out


def group_apply_key(curve, group, pts, first, inc):
    pts, first, inc = _u8(pts), _u8(first), _u8(inc)
    out = np.empty_like(pts)
    assert lib().orc_group_apply_key(curve, group, _p(pts), _p(out), C.c_size_t(pts.size // (2 * group * n8q(curve))), _p(first), _p(inc)) == 0
    return out


CONVERT_KINDS = {"LEMtoU": 0, "UtoLEM": 1, "LEMtoC": 2, "CtoLEM": 3}


def group_convert(curve, group, kind, buf):
    """G.batchLEMtoU / batchUtoLEM / batchLEMtoC / batchCtoLEM (oracle/zk_oracle.c: orc_group_convert)"""
    buf = _u8(buf)
    k, full, half = CONVERT_KINDS[kind], 2 * group * n8q(curve), group * n8q(curve)
    n = buf.size // (half if k == 3 else full)
    out = np.zeros(n * (half if k == 2 else full), np.uint8)
    rc = lib().orc_group_convert(curve, group, k, _p(buf), C.c_size_t(n), _p(out))
    if rc:
        raise ValueError("compressed point is not on the curve")
    return out


def vec_op(curve, op, a, b):
    """element-wise Fr add / sub / mul (op = "add" | "sub" | "mul"), Montgomery in and out"""
    a, b = _u8(a), _u8(b)
    out = np.empty_like(a)
    assert lib().orc_fr_vec_op(curve, {"add": 0, "sub": 1, "mul": 2}[op], _p(a), _p(b), _p(out), C.c_size_t(a.size // 32)) == 0
    return out


def dot(curve, a_mont, w_plain) -> int:
    """sum a_i * w_i mod r (a Montgomery, w plain integers)"""
    a, w = _u8(a_mont), _u8(w_plain)
    out = np.zeros(32, np.uint8)
    assert lib().orc_fr_dot(curve, _p(a), _p(w), C.c_size_t(a.size // 32), _p(out)) == 0
    return int.from_bytes(out.tobytes(), "little")


def threads():
    return lib().orc_threads()


def set_threads(n):
    lib().orc_set_threads(int(n))


def fq_from_mont(curve, x):
    x = _u8(x)
    out = np.empty_like(x)
    lib().orc_fq_from_mont(curve, _p(x), _p(out), C.c_size_t(x.size // n8q(curve)))
    return out


def build_abc(curve, coeffs, witness, n_vars, domain):
    coeffs, witness = _u8(coeffs), _u8(witness)
    A, B, Cc = (np.zeros(domain * 32, np.uint8) for _ in range(3))
    rc = lib().orc_groth16_build_abc(curve, _p(coeffs), C.c_size_t(coeffs.size), _p(witness), C.c_size_t(n_vars),
                                     C.c_size_t(domain), _p(A), _p(B), _p(Cc))
    assert rc == 0
    return A, B, Cc


def join_abc(curve, A, B, Cc):
    A, B, Cc = _u8(A), _u8(B), _u8(Cc)
    out = np.empty_like(A)
    lib().orc_groth16_join_abc(curve, _p(A), _p(B), _p(Cc), C.c_size_t(A.size // 32), _p(out))
    return out


def groth16_prove(curve, zk, wtns_witness, r_mont, s_mont):
    """zk: dict from tests/binfile.py read_groth16_zkey(); returns (pi_a, pi_b, pi_c) affine Montgomery bytes."""
    q = n8q(curve)
    pi_a, pi_b, pi_c = np.zeros(2 * q, np.uint8), np.zeros(4 * q, np.uint8), np.zeros(2 * q, np.uint8)
    keep = [_u8(zk[k]) for k in ("coeffs", "A", "B1", "B2", "C", "H", "vk_alpha_1", "vk_beta_1", "vk_beta_2",
                                 "vk_delta_1", "vk_delta_2")]
    w, r, s = _u8(wtns_witness), _u8(r_mont), _u8(s_mont)
    rc = lib().orc_groth16_prove(curve, C.c_size_t(zk["nVars"]), C.c_size_t(zk["nPublic"]), C.c_size_t(zk["domainSize"]),
                                 _p(keep[0]), C.c_size_t(keep[0].size), _p(w), _p(keep[1]), _p(keep[2]), _p(keep[3]),
                                 _p(keep[4]), _p(keep[5]), _p(keep[6]), _p(keep[7]), _p(keep[8]), _p(keep[9]), _p(keep[10]),
                                 _p(r), _p(s), _p(pi_a), _p(pi_b), _p(pi_c))
    assert rc == 0
    return pi_a, pi_b, pi_c


def groth16_closed_form(c, name, zk_, w, lg, n_public, rr, ss, b_zero_every):
    """Closed-form discrete logs (a, b, cc) of pi_a, pi_b, pi_c for a tests/synth_zkey.py key: every base is a known multiple of
    the generator (T[i] = 7*11^i*G), so each MSM result is an O(n) field sum (oracle/zk_oracle.c: orc_fr_geom_dot) over the witness
    and over h, the odd-coset evaluations of A*B - C from the CPU restatement's buildABC / NTT chain / joinABC (:62-83). The five
    MSMs themselves are never run on the CPU."""
    r = {"bn128": 21888242871839275222246405745257275088548364400416034343698204186575808495617, "bls12381": 52435875175126190479447740508185965837690552500527637822603658699938581184513}[name]
    n, m = zk_["domainSize"], zk_["nVars"]
    A, B, Cc = build_abc(c, zk_["coeffs"], w, m, n)
    one, inc = fr_one(c), fr_w(c, lg + 1)
    A, B, Cc = (ntt(c, apply_key(c, ntt(c, x, inverse=True), one, inc)) for x in (A, B, Cc))
    h = join_abc(c, A, B, Cc)                                # joinABC already leaves normal form (:362)
    del A, B, Cc
    d = lambda i: 7 * pow(11, i, r) % r                        # discrete log of T[i]
    sa = geom_dot(c, w, m)                                   # A_i = T1[i]
    sb = geom_dot(c, w, m, skip_mod=b_zero_every, skip_rem=1) if b_zero_every else sa     # B2_i = T2[i] (or infinity)
    sc = geom_dot(c, w[(n_public + 1) * 32:], m - n_public - 1)                           # C_j = T1[2 + j]: shift applied below
    sh = geom_dot(c, h, n)                                   # H_i = T1[3 + i]
    a = (d(5) + sa + rr * d(7)) % r                            # alpha1 = T1[5], delta1 = T1[7]
    b = (d(1) + sb + ss * d(2)) % r                            # beta2 = T2[1], delta2 = T2[2]
    b1 = (d(6) + sb * 11 + ss * d(7)) % r                      # beta1 = T1[6], B1_i = T1[i + 1]
    cc = (sc * pow(11, 2, r) + sh * pow(11, 3, r) + ss * a + rr * b1 - rr * ss % r * d(7)) % r
    return a, b, cc
