"""r04 probe: Fr NTT time (zkmi_ntt_dev, device events) at 2^16..2^24 and the batched in-proof chain; run under ZKMI_NTT29=0/1"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from snarkjs_amd import zkmi
from snarkjs_amd.workloads import synth
import hashlib
zkmi.init(0)
L = zkmi.lib()
out = []
for lg in (12, 16, 20, 22, 24):
    n = 1 << lg
    x = synth.elems(0x77 + lg, n)
    d_i = zkmi.DeviceBuffer.from_host(x); d_o = zkmi.DeviceBuffer(n * 32)
    ts = []
    for inv in (0, 1):
        t = []
        for _ in range(5):
            zkmi.check(L.zkmi_ntt_dev(0, d_i.ptr, d_o.ptr, lg, inv, None, None)); t.append(L.zkmi_last_kernel_ms())
        ts.append(min(t[1:]))
    h = hashlib.sha256(d_o.to_host().tobytes()).hexdigest()[:12]
    out.append(f"2^{lg}: fft {ts[0]:.4f} ms ifft {ts[1]:.4f} ms sha(ifft) {h}")
    d_i.free(); d_o.free()
print("NTT29=" + os.environ.get("ZKMI_NTT29", "0"), " | ".join(out), flush=True)
