import sys, time, cProfile, pstats
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import synth_plonk
from snarkjs_amd import plonk
zkey, wtns = synth_plonk.make("bn128", 20, seed=3)
key = plonk.PlonkKey(zkey)
plonk.prove(key, wtns)
t0=time.perf_counter(); plonk.prove(key, wtns); print("prove wall ms", (time.perf_counter()-t0)*1e3)
pr=cProfile.Profile(); pr.enable(); plonk.prove(key, wtns); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
