"""Quick device-timing probe (development aid): MSM and NTT at the BASELINE sizes, HIP-event timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import synth
from snarkjs_amd import zkmi

zkmi.init(0)
L = zkmi.lib()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
lgs = [int(a) for a in sys.argv[2:]] or [16, 20]

def msm(curve, group, lg, dist="uniform", cs=(0,)):
    n = 1 << lg
    q8 = 32 if curve == 0 else 48
    pb = 2 * group * q8
    d_b = zkmi.DeviceBuffer(n * pb)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(curve, group, n, 7, 11, d_b.ptr))
    sc = synth.elems(0x5EED, n) if dist == "uniform" else synth.witness_like(0x5EED, n)
    d_s = zkmi.DeviceBuffer.from_host(sc)
    out = np.zeros(3 * group * q8, np.uint8)
    for c in cs:
        L.zkmi_msm_set_window_bits(c)
        ts, ws = [], []
        for it in range(4):
            t0 = time.perf_counter()
            zkmi.check(L.zkmi_msm_dev(curve, group, d_b.ptr, d_s.ptr, n, 32, zkmi.ptr(out)))
            ws.append((time.perf_counter() - t0) * 1e3)
            ts.append(L.zkmi_last_kernel_ms())
        print(f"msm curve={curve} G{group} 2^{lg} {dist} c={c}: dev {min(ts[1:]):.3f} ms wall {min(ws[1:]):.3f} ms -> {n/min(ts[1:])/1e3:.1f} Mscalar/s", flush=True)
    L.zkmi_msm_set_window_bits(0)

def ntt(curve, lg):
    n = 1 << lg
    x = synth.elems(1, n)
    d_i, d_o = zkmi.DeviceBuffer.from_host(x), zkmi.DeviceBuffer(n * 32)
    ts = []
    for it in range(5):
        zkmi.check(L.zkmi_ntt_dev(curve, d_i.ptr, d_o.ptr, lg, 0, None, None))
        ts.append(L.zkmi_last_kernel_ms())
    t = min(ts[1:])
    print(f"ntt curve={curve} 2^{lg}: {t:.3f} ms -> {n/t/1e3:.1f} Melem/s, {64*n/t/1e6:.1f} GB/s algorithmic", flush=True)

if what in ("all", "ntt"):
    for lg in lgs: ntt(0, lg)
if what in ("all", "msm"):
    for lg in lgs:
        msm(0, 1, lg)
        msm(0, 1, lg, "witness")
if what == "msmc":
    for lg in lgs: msm(0, 1, lg, cs=(10, 11, 12, 13, 14, 15, 16))
if what in ("all", "g2"):
    for lg in lgs: msm(0, 2, lg)
