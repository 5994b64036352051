"""r06 probe: ONE table MSM at a time (zkmi_msm_table_dev, 2^20 BN254 G1, uniform 253-bit scalars) — the digit sort and the accumulation run one after
the other on one stream, so a rocprofv3 --kernel-trace of this script gives every sort kernel's OWN duration (inside a proof they run underneath a
full-chip accumulation and their per-launch averages are inflated).
usage: rocprofv3 --kernel-trace --stats --output-format csv -d out -o probe -- python tools/lab/r6_sort_probe.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snarkjs_amd import zkmi
from snarkjs_amd.workloads import synth
zkmi.init(0)
L = zkmi.lib()
n = 1 << int(os.environ.get("LOGN", "20"))
d_b = zkmi.DeviceBuffer(n * 64)
zkmi.check(L.zkmi_gen_geometric_bases_dev(0, 1, n, 7, 11, d_b.ptr))
tab = C.c_uint64(0)
zkmi.check(L.zkmi_msm_table_build(0, 1, d_b.ptr, n, C.byref(tab)))
d_s = zkmi.DeviceBuffer.from_host(synth.elems(0x5EED, n))
jac = np.zeros(96, np.uint8)
ts = []
for _ in range(int(os.environ.get("REPS", "8"))):
    zkmi.check(L.zkmi_msm_table_dev(tab, d_s.ptr, n, 32, zkmi.ptr(jac)))
    ts.append(L.zkmi_last_kernel_ms())
print("table msm ms", [round(t, 4) for t in ts], "jac", jac[:6].tolist(), flush=True)
