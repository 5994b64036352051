#!/bin/bash
# r06 one-off: BASELINE configs[2] FROM NODE over key shards — a 2^24-constraint key (7.9 GB file) proved by js/groth16_shards.js with 2 and 4 worker processes (all on
# the box's one GPU: the protocol, the by-offset slice reads with gaps and the peer copies are the multi-GPU ones, the placement is not), against the fused single-process
# prover's proof of the same key for the same (r, s). usage: gpurun -- 'bash tools/lab/r6_shards_2p24.sh [lg]'
LG=${1:-24}
D=/dev/shm/zkmi_k$LG; mkdir -p $D gpurun_out/r6_shards
python - <<PY
import sys, time
sys.path.insert(0, '.')
from snarkjs_amd.workloads import synth_zkey
t0 = time.time()
zk, wt = synth_zkey.make("bn128", $LG, seed=0x5EED, witness="uniform", b_zero_every=0)
open("$D/k.zkey", "wb").write(zk); open("$D/k.wtns", "wb").write(wt)
print("key synthesised and written in %.1f s, %d bytes" % (time.time() - t0, len(zk)), flush=True)
PY
cat > $D/run.js <<'JS'
const path = require("path"), crypto = require("crypto");
const root = process.argv[2], dir = process.argv[3];
const { makeProver } = require(path.join(root, "snarkjs_amd/js/groth16_native.js"));
const { ShardedProver, pointToObject } = require(path.join(root, "snarkjs_amd/js/groth16_shards.js"));
const now = () => Number(process.hrtime.bigint()) / 1e6;
const r = new Uint8Array(32), s = new Uint8Array(32); r[0] = 3; s[0] = 5;
const sha = (p) => crypto.createHash("sha256").update(JSON.stringify(p)).digest("hex");
(async () => {
    const out = {};
    const draws = [];
    const curve = { name: "bn128", Fr: { random: () => draws.shift() }, G1: { toObject: (b) => pointToObject(0, 1, b) }, G2: { toObject: (b) => pointToObject(0, 2, b) } };
    const prover = makeProver({ curves: { getCurveFromName: async () => curve } });
    draws.push(r, s);
    let t0 = now();
    const one = await prover.prove(path.join(dir, "k.zkey"), path.join(dir, "k.wtns"));
    out.fused_cold_ms = +(now() - t0).toFixed(1);
    draws.push(r, s); t0 = now(); await prover.prove(path.join(dir, "k.zkey"), path.join(dir, "k.wtns")); out.fused_warm_ms = +(now() - t0).toFixed(1);
    await prover.release();
    out.fused_proof = sha(one.proof);
    for (const world of [2, 4]) {
        t0 = now();
        const sp = new ShardedProver({ world, zkeyPath: path.join(dir, "k.zkey"), devices: Array(world).fill(0) });
        await sp.ready();
        const load = now() - t0;
        const t = [];
        let res;
        for (let i = 0; i < 3; i++) { t0 = now(); res = await sp.prove(path.join(dir, "k.wtns"), { r, s }); t.push(+(now() - t0).toFixed(1)); }
        out["shards_" + world] = { load_ms: +load.toFixed(0), prove_ms: t, exchange: res.exchange, devices: res.devices, equals_fused_proof: sha(res.proof) === out.fused_proof, timeline_ms: res.timeline_ms };
        await sp.close();
    }
    console.log(JSON.stringify(out));
    process.exit(0);
})().catch((e) => { console.log("ERROR", e && e.stack || e); process.exit(1); });
JS
node --max-old-space-size=24000 $D/run.js "$PWD" $D 2>&1 | tee gpurun_out/r6_shards/shards_2p$LG.txt
rm -rf $D
