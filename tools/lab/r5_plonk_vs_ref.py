"""r05 one-off (GPU box): the reference's own plonk.prove / fflonk.prove (bundle in oracle/_ref, WASM + worker threads) against the device-resident
provers on synthetic valid keys WITH addition gates at sizes the suite does not afford: same draws, proof JSON compared by hash; also the reference's
wall time on the box's host cores. usage: python tools/lab/r5_plonk_vs_ref.py plonk:16 plonk:18 fflonk:16   -> profiles/r05_plonk_vs_reference.txt"""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from snarkjs_amd import plonk, fflonk, zkmi
from snarkjs_amd.workloads import synth_plonk
zkmi.init(0)
f = plonk._Field(0)
sha = lambda o: hashlib.sha256(json.dumps(o, separators=(",", ":")).encode()).hexdigest()
for spec in sys.argv[1:]:
    proto, lg = spec.split(":"); lg = int(lg)
    zkey, wtns = (synth_plonk.make("bn128", lg, seed=21, additions=1) if proto == "plonk" else synth_plonk.make_fflonk(lg, seed=21, additions=1))
    mod, nd = (plonk, 11) if proto == "plonk" else (fflonk, 9)
    blind = [bytes(f.mont(77000 + 131 * i)) for i in range(nd)]
    t0 = time.perf_counter(); got = mod.prove(zkey, wtns, blinding_mont=blind); t_dev = time.perf_counter() - t0
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        zf, wf = os.path.join(td, "k.zkey"), os.path.join(td, "k.wtns")
        open(zf, "wb").write(zkey); open(wf, "wb").write(wtns)
        r = subprocess.run(["node", "--harmony-optional-chaining", "--harmony-nullish", "--max-old-space-size=32000", os.path.join(ROOT, "tools", "ref_wasm_same_box.js"), proto, zf, wf,
                            ",".join(b.hex() for b in blind)], capture_output=True, text=True, timeout=3000, env=dict(os.environ, NTHREADS="64"))
    if r.returncode:
        print(spec, "reference failed:", r.stderr[-300:], flush=True); continue
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"{proto} domain 2^{lg}: reference {d['ms'] / 1e3:.1f} s on {d['threads']} worker threads ({d['cpus']} cpus); device (cold call, key load inside) {t_dev:.2f} s; "
          f"proof JSON identical: {sha(got['proof']) == d['proof_json_sha256']}; public signals identical: {sha(got['publicSignals']) == d['public_signals_sha256']}", flush=True)
