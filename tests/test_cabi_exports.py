"""The drop-in boundary is the C-ABI of include/zkmi.h (diagnostics: include/zkmi_diag.h): the built library must load (no GPU needed) and export EVERY function the
headers declare; the ctypes mirror must bind them all; and without a device every compute entry point must fail loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "zkmi.h")).read() + open(os.path.join(ROOT, "include", "zkmi_diag.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from snarkjs_amd import zkmi
    L = ctypes.CDLL(zkmi.LIB_PATH)
    names = declared()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    # the ctypes mirror knows every declared function (a symbol added to the header must be bound and checked by build())
    assert not [n for n in names if n not in zkmi.SYMBOLS], [n for n in names if n not in zkmi.SYMBOLS]
    assert not [n for n in zkmi.SYMBOLS if n not in names], [n for n in zkmi.SYMBOLS if n not in names]


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from snarkjs_amd import zkmi
    L = zkmi.lib()
    assert L.zkmi_device_count() == 0
    assert L.zkmi_init(0) != 0 and b"no HIP device" in L.zkmi_last_error()
    out = np.zeros(96, np.uint8)
    p = ctypes.c_void_p(0)
    assert L.zkmi_dev_alloc(64, ctypes.byref(p)) != 0
    assert L.zkmi_msm_dev(0, 1, None, None, 4, 32, zkmi.ptr(out)) != 0
    assert L.zkmi_ntt_dev(0, None, None, 4, 0, None, None) != 0
    assert L.zkmi_poly_scale_dev(0, None, 4, zkmi.ptr(out)) != 0
    with pytest.raises(zkmi.ZkmiError):
        zkmi.init(0)
    # the boundary header holds no diagnostics: statistics, probes, generators and knobs live in zkmi_diag.h
    boundary = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "zkmi.h")).read(), flags=re.S)
    for name in ("zkmi_msm_stats", "zkmi_calibrate_box", "zkmi_gen_geometric_bases_dev", "zkmi_msm_set_window_bits", "zkmi_last_kernel_ms"):
        assert name + "(" not in boundary, name
    # host-only helpers keep working: Fr.w[] and the transcript hash need no device
    assert L.zkmi_fr_root(0, 1, zkmi.ptr(out)) == 0
    h = ctypes.create_string_buffer(32)
    assert L.zkmi_keccak256(b"", 0, h) == 0 and h.raw.hex().startswith("c5d24601")
