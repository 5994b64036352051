"""Timing of the group-element FFT / batchApplyKey on the device, device-resident: python tools/gfft_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snarkjs_amd import zkmi
zkmi.init(0); L = zkmi.lib()
for cid, name in ((0, "bn128"),):
    q8 = 32
    for group, lgs in ((1, (12, 16, 20)), (2, (12, 16))):
        for lg in lgs:
            n = 1 << lg
            pb = 2 * group * q8
            d_b, d_o = zkmi.DeviceBuffer(n * pb), zkmi.DeviceBuffer(n * pb)
            zkmi.check(L.zkmi_gen_geometric_bases_dev(cid, group, n, 7, 11, d_b.ptr))
            ts = []
            for _ in range(2):
                zkmi.check(L.zkmi_group_fft_dev(cid, group, d_b.ptr, d_o.ptr, lg, 1)); zkmi.check(L.zkmi_synchronize()); ts.append(L.zkmi_last_kernel_ms())
            one = np.zeros(32, np.uint8); one[0] = 3
            t0 = time.perf_counter(); zkmi.check(L.zkmi_group_batch_apply_key_dev(cid, group, d_b.ptr, d_o.ptr, n, zkmi.ptr(one), zkmi.ptr(one))); zkmi.check(L.zkmi_synchronize())
            print(f"{name} G{group} 2^{lg}: ifft {min(ts):.2f} ms ({n * lg / 2 / min(ts) / 1e3:.2f} M butterflies/s), batchApplyKey {L.zkmi_last_kernel_ms():.2f} ms", flush=True)
            d_b.free(); d_o.free()
# point-format conversions (gconv.cuh): device-resident timings
for cid, name, q8 in ((0, "bn128", 32), (1, "bls12381", 48)):
    for group, lg in ((1, 20), (2, 18)):
        n, pb = 1 << lg, 2 * group * q8
        d_b, d_u, d_c, d_r = zkmi.DeviceBuffer(n * pb), zkmi.DeviceBuffer(n * pb), zkmi.DeviceBuffer(n * pb // 2), zkmi.DeviceBuffer(n * pb)
        zkmi.check(L.zkmi_gen_geometric_bases_dev(cid, group, n, 7, 11, d_b.ptr))
        res = {}
        for nm, kind, src, dst in (("LEMtoU", 0, d_b, d_u), ("UtoLEM", 1, d_u, d_r), ("LEMtoC", 2, d_b, d_c), ("CtoLEM", 3, d_c, d_r)):
            ts = []
            for _ in range(2):
                zkmi.check(L.zkmi_group_convert_dev(cid, group, kind, src.ptr, dst.ptr, n)); zkmi.check(L.zkmi_synchronize()); ts.append(L.zkmi_last_kernel_ms())
            res[nm] = min(ts)
        ok = np.array_equal(d_r.to_host(), d_b.to_host())
        print(f"{name} G{group} 2^{lg}: " + ", ".join(f"{k} {v:.3f} ms" for k, v in res.items()) + f"  (LEMtoU {2 * n * pb / res['LEMtoU'] / 1e6:.0f} GB/s moved; round trip {'ok' if ok else 'MISMATCH'})", flush=True)
        for d in (d_b, d_u, d_c, d_r): d.free()
