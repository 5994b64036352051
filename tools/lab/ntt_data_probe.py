import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, synth
from snarkjs_amd import zkmi
zkmi.init(0); L=zkmi.lib()
lg=20; n=1<<lg
for name,x in (("random", synth.elems(1,n)), ("zeros", np.zeros(n*32,np.uint8)), ("delta", np.concatenate([synth.elems(2,1), np.zeros((n-1)*32,np.uint8)])), ("const", np.tile(synth.elems(3,1), n))):
    d_i, d_o = zkmi.DeviceBuffer.from_host(x), zkmi.DeviceBuffer(n*32)
    for inv in (0,1):
        ts=[]
        for it in range(4):
            zkmi.check(L.zkmi_ntt_dev(0, d_i.ptr, d_o.ptr, lg, inv, None, None)); ts.append(L.zkmi_last_kernel_ms())
        print(name, "inverse" if inv else "forward", [round(t,3) for t in ts])
