"""r06: where does the HOST spend a PLONK proof? cProfile of plonk.prove_many (two proofs in flight) and plonk.prove at 2^20: if the enqueue-only library calls and the Python
arithmetic between them add up to a large share of the 24 ms, the second proof in flight cannot help ("two proofs in flight buy nothing")."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from snarkjs_amd import plonk, zkmi  # noqa: E402
from snarkjs_amd.workloads import synth_plonk  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
zkmi.init(0)
zkey, wtns = synth_plonk.make("bn128", lg, seed=3, additions=1)
key = plonk.PlonkKey(zkey)
w = plonk.PlonkWitness(key, wtns)
for _ in range(3):
    plonk.prove(key, w)
plonk.prove_many(key, [w, w])
for name, fn in (("serial x8", lambda: [plonk.prove(key, w) for _ in range(8)]), ("two in flight x8", lambda: plonk.prove_many(key, [w] * 8))):
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print(f"==== {name}: {dt / 8 * 1e3:.2f} ms per proof unprofiled"); print("\n".join(s.getvalue().splitlines()[:40]))
