import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import synth_plonk
from snarkjs_amd import plonk
zkey, wtns = synth_plonk.make("bn128", 20, seed=3)
key = plonk.PlonkKey(zkey)
plonk.prove(key, wtns)
ts=[]
for _ in range(4):
    t0=time.perf_counter(); plonk.prove(key, wtns); ts.append((time.perf_counter()-t0)*1e3)
print("prove wall ms", min(ts), ts)
