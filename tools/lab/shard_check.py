import sys, time
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/oracle']
import numpy as np, oracle_lib as O, synth_zkey
from snarkjs_amd import groth16, binfile, distributed as D
lg, world = 22, 4
zkey, wtns = synth_zkey.make("bn128", lg, seed=3)
w = binfile.read_wtns(wtns)["witness"]
r_m, s_m = O.fr_e(0, 5), O.fr_e(0, 7)
full = groth16.ProvingKey(zkey); want=[bytes(x) for x in full.prove_raw(w, r_m, s_m)]; full.release()
parts=[]
for rank in range(world):
    pk = groth16.ProvingKey(zkey, shard=(rank, world))
    t0=time.perf_counter(); parts.append(pk.sums_raw(w)); t1=time.perf_counter(); pk.sums_raw(w); t2=time.perf_counter()
    print("rank", rank, "sums ms (incl. upload)", round((t2-t1)*1e3,2))
    if rank < world-1: pk.release()
got=[bytes(x) for x in pk.finish_raw(D.fold_groth16_sums(0, parts), r_m, s_m)]
print("EQUAL", got==want)
