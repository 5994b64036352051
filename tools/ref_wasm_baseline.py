#!/usr/bin/env python
"""The reference's own CPU path as the baseline (BASELINE.md §3 / SURVEY.md §8d "CPU baseline beside it"): snarkjs groth16.prove
from the reference bundle (WASM + worker threads, oracle/ref_shim.js) on synthetic keys of the bench recipe (tests/synth_zkey.py,
dense B sections), 1 thread and os.cpus().length threads. BUILD CONTAINER ONLY (needs /root/reference and Node); the bundle cannot
travel to the GPU box, so the result is committed under profiles/ and quoted by bench.py's cpu_baseline next to the live C port.

usage: python tools/ref_wasm_baseline.py [--out profiles/NAME.json] [log_n ...]     (default 14 16; r04: 20 -> profiles/r04_ref_wasm_baseline.json)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth_zkey  # noqa: E402  (tests/ shim: base tables from the CPU oracle, no GPU here)

JS = r"""
const fs=require('fs');
const snarkjs=require(process.argv[2]);
(async()=>{
  const zkey=new Uint8Array(fs.readFileSync(process.argv[3])), wtns=new Uint8Array(fs.readFileSync(process.argv[4]));
  const reps=parseInt(process.argv[5]);
  const curve=await snarkjs.curves.getCurveFromName('bn128');
  const t=[];
  let proof;
  for(let i=0;i<reps+1;i++){ const t0=process.hrtime.bigint(); proof=await snarkjs.groth16.prove(zkey,wtns); t.push(Number(process.hrtime.bigint()-t0)/1e6); }
  console.log(JSON.stringify({threads:snarkjs.nThreads, ms:t, node:process.version, pi_a0:proof.proof.pi_a[0]}));
  process.exit(0);
})().catch(e=>{console.error(e);process.exit(1)});
"""


def main():
    argv = sys.argv[1:]
    outf = os.path.join(ROOT, "profiles", "r02_ref_wasm_baseline.json")
    if "--out" in argv:
        i = argv.index("--out")
        outf = os.path.join(ROOT, argv[i + 1])
        del argv[i:i + 2]
    only_threads = None
    if "--threads" in argv:                      # measure one thread count only (and MERGE into an existing output file)
        i = argv.index("--threads")
        only_threads = int(argv[i + 1])
        del argv[i:i + 2]
    sizes = [int(x) for x in argv] or [14, 16]
    tmp = "/tmp/refbase"
    os.makedirs(tmp, exist_ok=True)
    js = os.path.join(tmp, "run.js")
    open(js, "w").write(JS)
    ncpu = os.cpu_count()
    out = {"what": "snarkjs groth16.prove (reference bundle build/snarkjs.min.js = snarkjs 0.7.6 + ffjavascript 0.3.1 WASM), synthetic key of tests/synth_zkey.py "
                   "(uniform 253-bit witness, dense B), files in memory, first call excluded (WASM tier-up / worker start)",
           "host": {"cpus": ncpu, "model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")},
           "runs": []}
    if only_threads and os.path.exists(outf):
        out["runs"] = json.load(open(outf)).get("runs", [])
    out["note"] = "every run on an otherwise idle build container (the first 8-thread figure of r04 was taken beside a compiler job and has been replaced)"
    for lg in sizes:
        t0 = time.time()
        zkey, wtns = synth_zkey.make("bn128", lg, seed=0x5EED, witness="uniform", use_device=False, b_zero_every=0)
        zf, wf = os.path.join(tmp, f"k{lg}.zkey"), os.path.join(tmp, f"k{lg}.wtns")
        open(zf, "wb").write(zkey)
        open(wf, "wb").write(wtns)
        print(f"2^{lg}: key built in {time.time() - t0:.1f} s", flush=True)
        for threads in ((only_threads,) if only_threads else (1, ncpu)):
            env = dict(os.environ, NTHREADS=str(threads))
            if threads == 1:
                env["SINGLE"] = "1"
            else:
                env.pop("SINGLE", None)
            reps = 2 if lg <= 14 else 1
            r = subprocess.run(["node", "--harmony-optional-chaining", "--harmony-nullish", "--max-old-space-size=12000", js,
                                os.path.join(ROOT, "oracle", "ref_shim.js"), zf, wf, str(reps)], capture_output=True, text=True, env=env, timeout=7200)
            if r.returncode != 0:
                print(r.stderr[-2000:])
                raise SystemExit(1)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            best = min(d["ms"][1:])
            out["runs"] = [x for x in out["runs"] if not (x["log_n"] == lg and x["threads"] == d["threads"])]
            out["runs"].append({"log_n": lg, "threads": d["threads"], "ms_per_proof": round(best, 1), "proofs_per_s": round(1e3 / best, 5), "all_ms": [round(x, 1) for x in d["ms"]],
                                "node": d["node"]})
            print(out["runs"][-1], flush=True)
        json.dump(out, open(outf, "w"), indent=1)      # after every size: a long run that is cut short keeps what it measured


if __name__ == "__main__":
    main()
