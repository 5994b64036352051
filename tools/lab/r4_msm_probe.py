"""r04 probe: the standalone G1 MSM (zkmi_msm_dev, 2^20 BN254) — pieces x window width; run under ZKMI_MSM_SPLIT / ZKMI_MSM_SPLIT_C"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snarkjs_amd import zkmi
from snarkjs_amd.workloads import synth
zkmi.init(0)
L = zkmi.lib()
n = 1 << 20
d_b = zkmi.DeviceBuffer(n * 64)
zkmi.check(L.zkmi_gen_geometric_bases_dev(0, 1, n, 7, 11, d_b.ptr))
d_s = zkmi.DeviceBuffer.from_host(synth.elems(0x5EED, n))
jac = np.zeros(96, np.uint8)
ts = []
for _ in range(5):
    zkmi.check(L.zkmi_msm_dev(0, 1, d_b.ptr, d_s.ptr, n, 32, zkmi.ptr(jac)))
    ts.append(L.zkmi_last_kernel_ms())
aff = np.zeros(64, np.uint8)
zkmi.check(L.zkmi_to_affine(0, 1, zkmi.ptr(jac), zkmi.ptr(aff)))
print("split", os.environ.get("ZKMI_MSM_SPLIT", "default"), "c", os.environ.get("ZKMI_MSM_SPLIT_C", "default"), "msm_ms", round(min(ts[1:]), 4), "affine", aff[:6].tolist(), flush=True)
