"""Host-side mirror of the ffjavascript curve surface snarkjs calls on the prover hot path.

Same names, argument meaning and error behaviour as the reference objects returned by
`getCurveFromName` (/root/reference/src/curves.js:36-53 -> buildBn128/buildBls12381, bundle min.js:1@240638):

    curve = get_curve_from_name("bn128")
    jac   = curve.G1.multiExpAffine(bases, scalars)        # min.js:1@214996
    jac2  = curve.G2.multiExpAffine(bases2, scalars)
    X     = curve.Fr.fft(buf); x = curve.Fr.ifft(buf)      # min.js:1@215859
    y     = curve.Fr.batchApplyKey(buf, first, inc)        # min.js:1@211529
    curve.Fr.batchToMontgomery / batchFromMontgomery / batchInverse

Buffers are bytes-like (the reference's Uint8Array) or a list of bytes-like pages (the reference's BigBuffer,
min.js:1@183423); results come back as numpy uint8 arrays, or a list of them where the reference would return a
BigBuffer (see _alloc_like / _alloc_like_sliced).
All work happens in the HIP library behind include/zkmi.h; nothing here computes.
"""
import ctypes as C

import numpy as np

from . import zkmi

PAGE_SIZE = 1 << 30  # ffjavascript BigBuffer page size


def _is_paged(buf):
    return isinstance(buf, (list, tuple))


def _byte_length(buf):
    if _is_paged(buf):
        return sum(zkmi.u8(b).size for b in buf)
    return zkmi.u8(buf).size


def _alloc_like(buf, nbytes):
    """Same container type as the input (SURVEY.md §8b container rule)."""
    if _is_paged(buf):
        out, left = [], nbytes
        while left > 0:
            k = min(left, PAGE_SIZE)
            out.append(np.empty(k, np.uint8))
            left -= k
        return out
    return np.empty(nbytes, np.uint8)


def _alloc_like_sliced(buf, nbytes):
    """Container rule of Fr.fft / Fr.ifft / Fr.batchInverse: the reference first takes buff.slice(0, byteLength), which for a
    BigBuffer of at most one page is a flat Uint8Array (min.js:1@183423), and returns that type."""
    if _is_paged(buf) and nbytes > PAGE_SIZE:
        return _alloc_like(buf, nbytes)
    return np.empty(nbytes, np.uint8)


def _out_pages(out):
    import ctypes as C
    bufs = out if isinstance(out, list) else [out]
    n = len(bufs)
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    return C.cast(ptrs, C.POINTER(C.c_void_p)), C.cast(lens, C.POINTER(C.c_size_t)), n, (ptrs, lens)


class _Fr:
    n8 = 32

    def __init__(self, curve_id):
        self._c = curve_id

    def _ntt(self, buf, inverse):
        nbytes = _byte_length(buf)
        n = nbytes // 32
        if nbytes % 32 or n == 0 or (n & (n - 1)):
            raise ValueError("fft must be multiple of 2")   # reference message, min.js:1@215859
        pg = zkmi.pages_of(buf)
        out = _alloc_like_sliced(buf, nbytes)
        op, ol, no, _keep = _out_pages(out)
        zkmi.check(zkmi.lib().zkmi_ntt(self._c, pg.pages, op, ol, no, n.bit_length() - 1, int(inverse), None, None))
        return out

    def fft(self, buf, inType=None, outType=None, logger=None, name=None):
        return self._ntt(buf, False)

    def ifft(self, buf, inType=None, outType=None, logger=None, name=None):
        return self._ntt(buf, True)

    def batchApplyKey(self, buf, first, inc):
        nbytes = _byte_length(buf)
        pg = zkmi.pages_of(buf)
        out = _alloc_like(buf, nbytes)
        op, ol, no, _keep = _out_pages(out)
        f, g = zkmi.u8(first), zkmi.u8(inc)
        if f.size != 32 or g.size != 32:
            raise ValueError("first / inc must be 32-byte Montgomery elements")
        zkmi.check(zkmi.lib().zkmi_fr_batch_apply_key(self._c, pg.pages, op, ol, no, nbytes // 32, zkmi.ptr(f), zkmi.ptr(g)))
        return out

    def _batch(self, op_id, buf):
        nbytes = _byte_length(buf)
        if nbytes % 32:
            raise ValueError("Invalid buffer size")      # reference message (min.js:1@188677)
        pg = zkmi.pages_of(buf)
        out = (_alloc_like_sliced if op_id == zkmi.BATCH_INVERSE else _alloc_like)(buf, nbytes)
        op, ol, no, _keep = _out_pages(out)
        zkmi.check(zkmi.lib().zkmi_fr_batch(self._c, op_id, pg.pages, op, ol, no, nbytes // 32))
        return out

    def batchToMontgomery(self, buf):
        return self._batch(zkmi.BATCH_TO_MONTGOMERY, buf)

    def batchFromMontgomery(self, buf):
        return self._batch(zkmi.BATCH_FROM_MONTGOMERY, buf)

    def batchInverse(self, buf):
        return self._batch(zkmi.BATCH_INVERSE, buf)


class _Group:
    def __init__(self, curve_id, group, n8q):
        self._c, self._g = curve_id, group
        self.F_n8 = n8q * group           # bytes of one coordinate
        self.point_bytes = 2 * self.F_n8  # affine (x, y)

    def multiExpAffine(self, bases, scalars, logger=None, name=None, cache_key=0):
        """sum_i scalars[i] * bases[i]; bases affine Montgomery, scalars plain LE integers of
        byteLength(scalars)/n bytes; returns the Jacobian point (3 coordinates, Montgomery)."""
        nb = _byte_length(bases)
        n = nb // self.point_bytes
        ns = _byte_length(scalars)
        out = np.zeros(3 * self.F_n8, np.uint8)
        if n == 0:
            return out
        if ns % n:
            raise ValueError("Scalar size does not match")   # reference message, min.js:1@214651
        pb, ps = zkmi.pages_of(bases), zkmi.pages_of(scalars)
        zkmi.check(zkmi.lib().zkmi_msm(self._c, self._g, pb.pages, ps.pages, n, ns // n, cache_key, zkmi.ptr(out)))
        return out

    def _gfft(self, buff, inverse):
        nb = _byte_length(buff)
        n = nb // self.point_bytes
        if n == 0 or (n & (n - 1)) or n * self.point_bytes != nb:
            raise ValueError("fft must be multiple of 2")                  # reference message, min.js:1@215859
        pg = zkmi.pages_of(buff)
        out = np.empty(nb, np.uint8)
        op, ol = (C.c_void_p * 1)(out.ctypes.data), (C.c_size_t * 1)(nb)
        zkmi.check(zkmi.lib().zkmi_group_fft(self._c, self._g, pg.pages, op, ol, 1, n.bit_length() - 1, int(inverse)))
        return out

    def fft(self, buff, inType="affine", outType="affine", logger=None, name=None):
        """G.fft over affine points (ceremony side); only the affine -> affine form the reference's callers use"""
        if inType != "affine" or outType != "affine":
            raise ValueError("group fft: only affine in / affine out is implemented")
        return self._gfft(buff, False)

    def ifft(self, buff, inType="affine", outType="affine", logger=None, name=None):
        if inType != "affine" or outType != "affine":
            raise ValueError("group fft: only affine in / affine out is implemented")
        return self._gfft(buff, True)

    def lagrangeEvaluations(self, buff, inType="affine", outType="affine", logger=None, name=None):
        """G.lagrangeEvaluations for 2^k <= 2^Fr.s points = G.ifft (min.js: lagrangeEvaluations -> ifft)"""
        return self.ifft(buff, inType, outType)

    def batchApplyKey(self, buff, first, inc, inType="affine", outType="affine"):
        """out_i = (first * inc^i) * P_i; first / inc: Montgomery Fr elements (32 bytes)"""
        if inType != "affine" or outType != "affine":
            raise ValueError("group batchApplyKey: only affine in / affine out is implemented")
        nb = _byte_length(buff)
        n = nb // self.point_bytes
        pg = zkmi.pages_of(buff)
        out = np.empty(n * self.point_bytes, np.uint8)
        f, g = zkmi.u8(first), zkmi.u8(inc)
        op, ol = (C.c_void_p * 1)(out.ctypes.data), (C.c_size_t * 1)(out.size)
        zkmi.check(zkmi.lib().zkmi_group_batch_apply_key(self._c, self._g, pg.pages, op, ol, 1, n, zkmi.ptr(f), zkmi.ptr(g)))
        return out

    def _convert(self, kind, buff):
        """point-format conversions of the ceremony files (include/zkmi.h: zkmi_group_convert)"""
        nb = _byte_length(buff)
        in_sz = self.point_bytes // 2 if kind == zkmi.CONV_C_TO_LEM else self.point_bytes
        out_sz = self.point_bytes // 2 if kind == zkmi.CONV_LEM_TO_C else self.point_bytes
        if nb % in_sz:
            raise ValueError("Invalid buffer size")
        n = nb // in_sz
        pg = zkmi.pages_of(buff)
        out = np.empty(n * out_sz, np.uint8)
        op, ol = (C.c_void_p * 1)(out.ctypes.data), (C.c_size_t * 1)(out.size)
        zkmi.check(zkmi.lib().zkmi_group_convert(self._c, self._g, kind, pg.pages, op, ol, 1, n))
        return out

    def batchLEMtoU(self, buff):
        return self._convert(zkmi.CONV_LEM_TO_U, buff)

    def batchUtoLEM(self, buff):
        return self._convert(zkmi.CONV_U_TO_LEM, buff)

    def batchLEMtoC(self, buff):
        return self._convert(zkmi.CONV_LEM_TO_C, buff)

    def batchCtoLEM(self, buff):
        return self._convert(zkmi.CONV_C_TO_LEM, buff)

    def toAffine(self, jac):
        j = zkmi.u8(jac)
        out = np.zeros(2 * self.F_n8, np.uint8)
        zkmi.check(zkmi.lib().zkmi_to_affine(self._c, self._g, zkmi.ptr(j), zkmi.ptr(out)))
        return out


class Curve:
    def __init__(self, name):
        if name.lower() not in zkmi.CURVE_ID:
            raise ValueError(f"Curve not supported: {name}")   # src/curves.js:49
        self.name = "bn128" if zkmi.CURVE_ID[name.lower()] == 0 else "bls12381"
        self.id = zkmi.CURVE_ID[name.lower()]
        n8q = 32 if self.id == 0 else 48
        self.Fr = _Fr(self.id)
        self.G1 = _Group(self.id, 1, n8q)
        self.G2 = _Group(self.id, 2, n8q)

    def joinABC(self, a, b, c):
        """src/groth16_prove.js:320-374: fromMontgomery(a*b - c) element-wise."""
        nbytes = _byte_length(a)
        pa, pb, pc = zkmi.pages_of(a), zkmi.pages_of(b), zkmi.pages_of(c)
        out = _alloc_like(a, nbytes)
        op, ol, no, _keep = _out_pages(out)
        zkmi.check(zkmi.lib().zkmi_groth16_join_abc(self.id, pa.pages, pb.pages, pc.pages, op, ol, no, nbytes // 32))
        return out

    def terminate(self):
        pass


_curves = {}


def get_curve_from_name(name, device=None):
    """src/curves.js:36-53 getCurveFromName; the curve object is cached like globalThis.curve_bn128."""
    zkmi.init(device)
    key = "bn128" if zkmi.CURVE_ID.get(name.lower(), -1) == 0 else name.lower()
    if key not in _curves:
        _curves[key] = Curve(name)
    return _curves[key]


def get_curve_from_r(r, device=None):
    """src/curves.js:9-21 getCurveFromR."""
    if r == 21888242871839275222246405745257275088548364400416034343698204186575808495617:
        return get_curve_from_name("bn128", device)
    if r == 52435875175126190479447740508185965837690552500527637822603658699938581184513:
        return get_curve_from_name("bls12381", device)
    raise ValueError(f"Curve not supported: {r}")
