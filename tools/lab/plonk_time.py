import sys, time, os
mode = sys.argv[1]
if mode in ("torch", "cuda"):
    import torch
    if mode == "cuda":
        torch.cuda.set_device(0); torch.cuda.synchronize()
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import synth_plonk
from snarkjs_amd import plonk
zkey, wtns = synth_plonk.make("bn128", 20, seed=3)
key = plonk.PlonkKey(zkey)
for _ in range(3): plonk.prove(key, wtns)
ts=[]
for _ in range(6):
    t0=time.perf_counter(); plonk.prove(key, wtns); ts.append((time.perf_counter()-t0)*1e3)
print(mode, "prove wall ms min %.2f mean %.2f" % (min(ts), sum(ts)/len(ts)))
