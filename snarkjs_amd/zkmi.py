"""ctypes binding of libzkmi.so (include/zkmi.h, include/zkmi_diag.h) — the C-ABI of the MI355X proving backend.

Thin plumbing only: every function maps 1:1 to a C entry point.  There is NO CPU fallback — if the HIP library is
missing or no device is visible, calls raise ZkmiError.  The library serves ONE caller per process (include/zkmi.h, conventions):
every entry point is taken under one process-wide lock here, as the N-API addon does with its mutex.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKMI_LIB", os.path.join(_HERE, "libzkmi.so"))

BN128, BLS12381 = 0, 1
CURVE_ID = {"bn128": 0, "bn254": 0, "bls12381": 1}
BATCH_TO_MONTGOMERY, BATCH_FROM_MONTGOMERY, BATCH_INVERSE = 0, 1, 2
ERR_NO_DEVICE = 1

# every symbol include/zkmi.h and include/zkmi_diag.h declare (tests check that the library exports all of them)
SYMBOLS = [
    "zkmi_groth16_load_paged", "zkmi_groth16_load_shard_paged", "zkmi_groth16_prove_paged", "zkmi_groth16_build_abc_dev", "zkmi_groth16_coef_layout", "zkmi_msm_dev_fallbacks",
    "zkmi_init", "zkmi_device_count", "zkmi_last_error", "zkmi_version", "zkmi_set_stream", "zkmi_synchronize",
    "zkmi_dev_alloc", "zkmi_dev_free", "zkmi_memcpy_h2d", "zkmi_memcpy_d2h", "zkmi_memcpy_d2d", "zkmi_memset_dev",
    "zkmi_msm", "zkmi_release_bases", "zkmi_msm_dev", "zkmi_msm_set_window_bits", "zkmi_msm_accum_ms", "zkmi_msm_stats", "zkmi_msm_accum_additions", "zkmi_msm_table_build", "zkmi_msm_table_dev", "zkmi_msm_table_multi_dev", "zkmi_msm_table_multi_enqueue_dev", "zkmi_msm_table_multi_collect", "zkmi_msm_table_release", "zkmi_msm_table_info",
    "zkmi_ipc_export", "zkmi_ipc_open", "zkmi_ipc_close", "zkmi_peer_copy", "zkmi_peer_copy_async", "zkmi_peer_fence", "zkmi_groth16_key_curve", "zkmi_groth16_reset",
    "zkmi_ntt", "zkmi_ntt_dev",
    "zkmi_fr_batch_apply_key", "zkmi_fr_batch_apply_key_dev", "zkmi_fr_batch", "zkmi_fr_batch_dev",
    "zkmi_groth16_join_abc", "zkmi_groth16_join_abc_dev",
    "zkmi_base_cache_stats", "zkmi_gen_bases_from_scalars_dev", "zkmi_group_fft", "zkmi_group_fft_dev", "zkmi_group_batch_apply_key", "zkmi_group_batch_apply_key_dev", "zkmi_group_convert", "zkmi_group_convert_dev", "zkmi_calibrate_box", "zkmi_calibrate_code_fetch", "zkmi_compact_code", "zkmi_groth16_load", "zkmi_groth16_prove", "zkmi_groth16_prove_dev", "zkmi_groth16_submit_dev", "zkmi_groth16_submit", "zkmi_groth16_sums_w_dev", "zkmi_groth16_collect", "zkmi_groth16_release", "zkmi_groth16_load_shard", "zkmi_groth16_sums_dev", "zkmi_groth16_chains_dev", "zkmi_groth16_sums_h_dev", "zkmi_groth16_finish", "zkmi_groth16_stage_ms",
    "zkmi_gen_geometric_bases_dev", "zkmi_host_register", "zkmi_host_unregister", "zkmi_to_affine", "zkmi_point_add", "zkmi_fr_root",
    "zkmi_plonk_gather_wires_dev", "zkmi_plonk_additions_dev", "zkmi_plonk_compute_z_dev", "zkmi_plonk_compute_z_enqueue", "zkmi_pipeline_select", "zkmi_pipeline_active", "zkmi_plonk_compute_t_dev", "zkmi_fflonk_t0_dev", "zkmi_fflonk_t1_dev",
    "zkmi_fflonk_t2_dev", "zkmi_poly_degree_dev", "zkmi_keccak256", "zkmi_poly_blind_dev", "zkmi_poly_add_scalar_dev", "zkmi_poly_axpy_dev", "zkmi_poly_scale_dev",
    "zkmi_msm_table_multi_enqueue_mont_dev", "zkmi_ntt_padded_dev", "zkmi_fr_batch_multi_dev", "zkmi_plonk_gather_wires_mont_dev", "zkmi_poly_blind_tail_dev", "zkmi_poly_lincomb_dev",
    "zkmi_poly_evaluate_multi_dev", "zkmi_poly_div_by_zerofier_enqueue", "zkmi_plonk_split_t_dev",
    "zkmi_poly_evaluate_dev", "zkmi_poly_is_zero_dev", "zkmi_poly_div_zh_dev", "zkmi_cpoly_interleave_dev", "zkmi_poly_div_by_zerofier_dev", "zkmi_last_kernel_ms",
]


CONV_LEM_TO_U, CONV_U_TO_LEM, CONV_LEM_TO_C, CONV_C_TO_LEM = 0, 1, 2, 3


class ZkmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"zkmi error {code}: {msg}")
        self.code = code


class Pages(C.Structure):
    _fields_ = [("ptr", C.POINTER(C.c_void_p)), ("len", C.POINTER(C.c_size_t)), ("n_pages", C.c_int)]


class PolyTerm(C.Structure):
    """zkmi_poly_term: one operand of zkmi_poly_lincomb_dev (56 bytes, no host pointers)"""
    _fields_ = [("d_p", C.c_void_p), ("len", C.c_uint64), ("k", C.c_uint8 * 32), ("has_k", C.c_uint32), ("reserved", C.c_uint32)]


class PlonkEvals(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("a", "b", "c", "z", "qm", "ql", "qr", "qo", "qc", "s1", "s2", "s3", "lagrange", "pub_a")]


class Groth16Zkey(C.Structure):
    _fields_ = [("curve", C.c_int), ("n_vars", C.c_uint32), ("n_public", C.c_uint32), ("domain_size", C.c_uint32),
                ("coeffs", C.c_void_p), ("coeffs_len", C.c_size_t),
                ("bases_a", C.c_void_p), ("bases_b1", C.c_void_p), ("bases_b2", C.c_void_p), ("bases_c", C.c_void_p),
                ("bases_h", C.c_void_p),
                ("vk_alpha_1", C.c_void_p), ("vk_beta_1", C.c_void_p), ("vk_beta_2", C.c_void_p),
                ("vk_delta_1", C.c_void_p), ("vk_delta_2", C.c_void_p),
                ("bases_a_len", C.c_size_t), ("bases_b1_len", C.c_size_t), ("bases_b2_len", C.c_size_t), ("bases_c_len", C.c_size_t),
                ("bases_h_len", C.c_size_t)]


class Groth16ZkeyPaged(C.Structure):
    _fields_ = [("curve", C.c_int), ("n_vars", C.c_uint32), ("n_public", C.c_uint32), ("domain_size", C.c_uint32)] + \
               [(k, Pages) for k in ("coeffs", "bases_a", "bases_b1", "bases_b2", "bases_c", "bases_h")] + \
               [(k, C.c_void_p) for k in ("vk_alpha_1", "vk_beta_1", "vk_beta_2", "vk_delta_1", "vk_delta_2")]


class _Locked:
    """The library behind one process-wide re-entrant lock: include/zkmi.h declares it single-caller per process (process-global pipeline
    slot, streams and scratch), and ctypes drops the GIL for the duration of a call. Attribute access hands out a locking wrapper of the
    ctypes function (hasattr() works as on the CDLL)."""

    def __init__(self, raw):
        object.__setattr__(self, "_raw", raw)
        object.__setattr__(self, "_lock", threading.RLock())

    def __getattr__(self, name):
        f = getattr(object.__getattribute__(self, "_raw"), name)
        lock = object.__getattribute__(self, "_lock")

        def call(*a):
            with lock:
                return f(*a)
        call.__name__ = name
        object.__setattr__(self, name, call)
        return call


_lib = None


def lib():
    """Load libzkmi.so (built in-tree by snarkjs_amd/build.py). Raises if it is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkmiError(-1, f"{LIB_PATH} not built (run python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(LIB_PATH)
    L.zkmi_last_error.restype = C.c_char_p
    L.zkmi_version.restype = C.c_char_p
    L.zkmi_last_kernel_ms.restype = C.c_double
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.c_void_p
    L.zkmi_init.argtypes = [C.c_int]
    L.zkmi_set_stream.argtypes = [vp]
    L.zkmi_dev_alloc.argtypes = [sz, C.POINTER(vp)]
    L.zkmi_dev_free.argtypes = [vp]
    L.zkmi_memcpy_h2d.argtypes = [vp, vp, sz]
    L.zkmi_memcpy_d2h.argtypes = [vp, vp, sz]
    L.zkmi_memcpy_d2d.argtypes = [vp, vp, sz]
    L.zkmi_memset_dev.argtypes = [vp, C.c_int, sz]
    L.zkmi_msm.argtypes = [C.c_int, C.c_int, Pages, Pages, sz, sz, C.c_uint64, u8p]
    L.zkmi_release_bases.argtypes = [C.c_uint64]
    L.zkmi_group_fft_dev.argtypes = [C.c_int, C.c_int, vp, vp, C.c_uint, C.c_int]
    L.zkmi_group_fft.argtypes = [C.c_int, C.c_int, Pages, C.POINTER(vp), C.POINTER(sz), C.c_int, C.c_uint, C.c_int]
    L.zkmi_group_batch_apply_key.argtypes = [C.c_int, C.c_int, Pages, C.POINTER(vp), C.POINTER(sz), C.c_int, sz, u8p, u8p]
    L.zkmi_group_batch_apply_key_dev.argtypes = [C.c_int, C.c_int, vp, vp, sz, u8p, u8p]
    L.zkmi_group_convert.argtypes = [C.c_int, C.c_int, C.c_int, Pages, C.POINTER(vp), C.POINTER(sz), C.c_int, sz]
    L.zkmi_group_convert_dev.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, sz]
    L.zkmi_calibrate_box.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.zkmi_gen_bases_from_scalars_dev.argtypes = [C.c_int, C.c_int, vp, sz, vp]
    L.zkmi_base_cache_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.zkmi_msm_dev.argtypes = [C.c_int, C.c_int, vp, vp, sz, sz, u8p]
    L.zkmi_msm_set_window_bits.argtypes = [C.c_int]
    L.zkmi_msm_table_build.argtypes = [C.c_int, C.c_int, vp, sz, C.POINTER(C.c_uint64)]
    L.zkmi_msm_table_dev.argtypes = [C.c_uint64, vp, sz, sz, u8p]
    L.zkmi_msm_table_multi_dev.argtypes = [C.c_uint64, C.POINTER(vp), C.POINTER(sz), C.c_int, sz, u8p]
    L.zkmi_msm_table_multi_enqueue_dev.argtypes = [C.c_uint64, C.POINTER(vp), C.POINTER(sz), C.c_int, sz]
    L.zkmi_msm_table_multi_collect.argtypes = [C.c_uint64, C.c_int, u8p]
    L.zkmi_msm_table_release.argtypes = [C.c_uint64]
    L.zkmi_msm_table_info.argtypes = [C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(sz)]
    L.zkmi_ipc_export.argtypes = [vp, u8p]
    L.zkmi_ipc_open.argtypes = [u8p, C.POINTER(vp), C.POINTER(sz)]
    L.zkmi_ipc_close.argtypes = [vp]
    L.zkmi_peer_copy.argtypes = [vp, vp, sz]
    L.zkmi_peer_copy_async.argtypes = [vp, vp, sz]
    L.zkmi_peer_fence.argtypes = []
    L.zkmi_groth16_key_curve.argtypes = [C.c_uint64]
    L.zkmi_groth16_reset.argtypes = [C.c_uint64]
    L.zkmi_msm_accum_ms.argtypes = [C.c_int]
    L.zkmi_msm_accum_ms.restype = C.c_double
    L.zkmi_msm_stats.argtypes = [C.c_int]
    L.zkmi_msm_accum_additions.argtypes = [C.c_int]
    L.zkmi_msm_accum_additions.restype = C.c_double
    L.zkmi_ntt.argtypes = [C.c_int, Pages, C.POINTER(vp), C.POINTER(sz), C.c_int, C.c_uint, C.c_int, u8p, u8p]
    L.zkmi_ntt_dev.argtypes = [C.c_int, vp, vp, C.c_uint, C.c_int, u8p, u8p]
    L.zkmi_fr_batch_apply_key.argtypes = [C.c_int, Pages, C.POINTER(vp), C.POINTER(sz), C.c_int, sz, u8p, u8p]
    L.zkmi_fr_batch_apply_key_dev.argtypes = [C.c_int, vp, vp, sz, u8p, u8p]
    L.zkmi_fr_batch.argtypes = [C.c_int, C.c_int, Pages, C.POINTER(vp), C.POINTER(sz), C.c_int, sz]
    L.zkmi_fr_batch_dev.argtypes = [C.c_int, C.c_int, vp, vp, sz]
    L.zkmi_groth16_join_abc.argtypes = [C.c_int, Pages, Pages, Pages, C.POINTER(vp), C.POINTER(sz), C.c_int, sz]
    L.zkmi_groth16_join_abc_dev.argtypes = [C.c_int, vp, vp, vp, vp, sz]
    L.zkmi_gen_geometric_bases_dev.argtypes = [C.c_int, C.c_int, sz, C.c_uint64, C.c_uint64, vp]
    L.zkmi_to_affine.argtypes = [C.c_int, C.c_int, u8p, u8p]
    L.zkmi_point_add.argtypes = [C.c_int, C.c_int, u8p, u8p, u8p]
    L.zkmi_fr_root.argtypes = [C.c_int, C.c_uint, u8p]
    u32 = C.c_uint32
    L.zkmi_plonk_gather_wires_dev.argtypes = [C.c_int, vp, u32, vp, u32, vp, vp, vp, u32, u32, vp, vp, vp]
    L.zkmi_plonk_additions_dev.argtypes = [C.c_int, vp, u32, vp, u32, vp]
    L.zkmi_plonk_compute_z_dev.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, u32, u8p, u8p, u8p, u8p, u8p, vp]
    L.zkmi_plonk_compute_z_enqueue.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, u32, u8p, u8p, u8p, u8p, u8p, vp]
    L.zkmi_pipeline_select.argtypes = [C.c_int]
    L.zkmi_pipeline_active.argtypes = []
    L.zkmi_plonk_compute_t_dev.argtypes = [C.c_int, C.POINTER(PlonkEvals), u32, u32, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p, u8p, vp, vp]
    L.zkmi_fflonk_t0_dev.argtypes = [C.c_int, C.POINTER(PlonkEvals), u32, u32, vp]
    L.zkmi_fflonk_t1_dev.argtypes = [C.c_int, vp, vp, u32, u8p, u8p, vp, vp]
    L.zkmi_fflonk_t2_dev.argtypes = [C.c_int, C.POINTER(PlonkEvals), u32, u8p, u8p, u8p, u8p, u8p, u8p, u8p, vp, vp]
    L.zkmi_poly_degree_dev.argtypes = [C.c_int, vp, sz, C.POINTER(sz)]
    L.zkmi_keccak256.argtypes = [C.c_char_p, sz, C.c_char_p]
    L.zkmi_poly_blind_dev.argtypes = [C.c_int, vp, sz, vp, C.c_int]
    L.zkmi_poly_add_scalar_dev.argtypes = [C.c_int, vp, vp]
    L.zkmi_poly_axpy_dev.argtypes = [C.c_int, vp, vp, sz, u8p, C.c_int]
    L.zkmi_poly_scale_dev.argtypes = [C.c_int, vp, sz, u8p]
    L.zkmi_poly_evaluate_dev.argtypes = [C.c_int, vp, sz, u8p, u8p]
    L.zkmi_poly_evaluate_multi_dev.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(sz), u8p, C.c_int, u8p]
    L.zkmi_poly_lincomb_dev.argtypes = [C.c_int, vp, sz, C.POINTER(PolyTerm), C.c_int, u8p]
    L.zkmi_poly_blind_tail_dev.argtypes = [C.c_int, vp, sz, vp, C.c_int]
    L.zkmi_poly_div_by_zerofier_enqueue.argtypes = [C.c_int, vp, sz, u32, u8p]
    L.zkmi_plonk_split_t_dev.argtypes = [C.c_int, vp, sz, u32, u8p, u8p, vp, vp, vp]
    L.zkmi_plonk_gather_wires_mont_dev.argtypes = [C.c_int, vp, u32, vp, u32, vp, vp, vp, u32, u32, vp, vp, vp]
    L.zkmi_msm_table_multi_enqueue_mont_dev.argtypes = [C.c_uint64, C.POINTER(vp), C.POINTER(sz), C.c_int]
    L.zkmi_ntt_padded_dev.argtypes = [C.c_int, vp, sz, vp, C.c_uint, C.c_int]
    L.zkmi_fr_batch_multi_dev.argtypes = [C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.c_int]
    L.zkmi_poly_is_zero_dev.argtypes = [C.c_int, vp, sz, C.POINTER(C.c_int)]
    L.zkmi_poly_div_zh_dev.argtypes = [C.c_int, vp, sz, u32, u32]
    L.zkmi_poly_div_by_zerofier_dev.argtypes = [C.c_int, vp, sz, u32, u8p]
    L.zkmi_cpoly_interleave_dev.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(sz), C.c_int, vp, sz]
    L.zkmi_groth16_load.argtypes = [C.POINTER(Groth16Zkey), C.c_uint64]
    L.zkmi_groth16_prove.argtypes = [C.POINTER(Groth16Zkey), C.c_uint64, u8p, C.c_size_t, u8p, u8p, u8p, u8p, u8p]
    L.zkmi_groth16_prove_dev.argtypes = [C.c_uint64, vp, u8p, u8p, u8p, u8p, u8p]
    L.zkmi_groth16_release.argtypes = [C.c_uint64]
    L.zkmi_groth16_submit_dev.argtypes = [C.c_uint64, vp, C.c_int]
    L.zkmi_groth16_submit.argtypes = [C.c_uint64, u8p, sz, C.c_int]
    L.zkmi_groth16_sums_w_dev.argtypes = [C.c_uint64, vp]
    L.zkmi_groth16_collect.argtypes = [C.c_uint64, C.c_int, u8p, u8p, u8p, u8p, u8p]
    L.zkmi_groth16_load_shard.argtypes = [C.POINTER(Groth16Zkey), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.zkmi_groth16_sums_dev.argtypes = [C.c_uint64, vp, u8p]
    L.zkmi_groth16_chains_dev.argtypes = [C.c_uint64, vp, C.c_uint, vp, vp, vp]
    L.zkmi_groth16_sums_h_dev.argtypes = [C.c_uint64, vp, vp, u8p]
    L.zkmi_groth16_finish.argtypes = [C.c_uint64, u8p, u8p, u8p, u8p, u8p, u8p]
    L.zkmi_groth16_stage_ms.argtypes = [C.POINTER(C.c_double), C.c_int]
    L.zkmi_groth16_load_paged.argtypes = [C.POINTER(Groth16ZkeyPaged), C.c_uint64]
    L.zkmi_groth16_load_shard_paged.argtypes = [C.POINTER(Groth16ZkeyPaged), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.zkmi_groth16_prove_paged.argtypes = [C.POINTER(Groth16ZkeyPaged), C.c_uint64, u8p, C.c_size_t, u8p, u8p, u8p, u8p, u8p]
    L.zkmi_groth16_build_abc_dev.argtypes = [C.c_uint64, vp, vp, vp, vp]
    L.zkmi_msm_dev_fallbacks.argtypes = []
    L.zkmi_msm_dev_fallbacks.restype = C.c_ulonglong
    L.zkmi_groth16_coef_layout.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.c_int]
    _lib = _Locked(L)
    return _lib


def check(rc):
    if rc != 0:
        raise ZkmiError(rc, lib().zkmi_last_error().decode(errors="replace"))


_device = None


def init(device=None):
    """Bind this process to one GPU (one process per GPU). device=None: keep the current binding, or bind to LOCAL_RANK
    (torchrun) / device 0 on first use."""
    global _device
    if device is None:
        if _device is not None:
            return
        device = int(os.environ.get("LOCAL_RANK", "0")) if lib().zkmi_device_count() > 1 else 0
    check(lib().zkmi_init(device))
    _device = device


def device_count():
    return lib().zkmi_device_count()


def u8(x):
    """View anything bytes-like as a contiguous uint8 numpy array (no copy when already one)."""
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x).view(np.uint8).reshape(-1)
    return np.frombuffer(x, dtype=np.uint8)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _PagesHolder:
    """Keeps the ctypes arrays of a zkmi_pages alive for the duration of a call."""

    def __init__(self, bufs):
        self.bufs = [u8(b) for b in bufs]
        n = len(self.bufs)
        self.ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in self.bufs])
        self.lens = (C.c_size_t * n)(*[b.size for b in self.bufs])
        self.pages = Pages(C.cast(self.ptrs, C.POINTER(C.c_void_p)), C.cast(self.lens, C.POINTER(C.c_size_t)), n)
        self.total = sum(b.size for b in self.bufs)


def pages_of(buf):
    """A logical buffer is one bytes-like object or a BigBuffer-like list of them (ffjavascript BigBuffer.buffers)."""
    if isinstance(buf, (list, tuple)):
        return _PagesHolder(list(buf))
    return _PagesHolder([buf])


class DeviceBuffer:
    """Device memory owned by the library (zkmi_dev_alloc)."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().zkmi_dev_alloc(nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes

    @classmethod
    def from_host(cls, data):
        a = u8(data)
        b = cls(a.size)
        check(lib().zkmi_memcpy_h2d(b.ptr, ptr(a), a.size))
        return b

    def to_host(self, nbytes=None, offset=0):
        n = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(n, np.uint8)
        check(lib().zkmi_memcpy_d2h(ptr(out), self.ptr + offset, n))
        return out

    def free(self):
        if self.ptr:
            lib().zkmi_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
