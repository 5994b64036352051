// tools/ref_wasm_same_box.js — the REFERENCE's own prover (snarkjs bundle = WASM + worker threads, staged in oracle/_ref by `make -C oracle _ref`)
// timed on THIS box's host cores, on the bench's own key: bench.py's cpu_baseline.reference_wasm.same_box (north_star: "alongside the reference
// WASM+worker-thread path timed on the same box's host cores (core count stated)"). Test / measurement infrastructure: never in the product path.
//
//   node --harmony-optional-chaining --harmony-nullish --max-old-space-size=16000 tools/ref_wasm_same_box.js <proto> <zkey> <wtns> <drawsHex,..> [<warm zkey> <warm wtns>]
//
// proto = groth16 | plonk | fflonk. The optional warm pair (a small key of the same recipe) is proved first and not timed: it starts the worker
// threads and lets V8 tier the WASM up, which the reference's own users pay once per process, not per proof. `draws` = the blinding elements
// (Montgomery bytes, hex) handed out by curve.Fr.random in order, so that the proof can be compared with the device's proof for the same
// draws (bench.py checks the hash: bit-identical proofs at the bench size, on the same box).
// NTHREADS = worker threads (default: os.cpus().length; ffjavascript itself caps at 64).
"use strict";
const fs = require("fs"), path = require("path"), os = require("os"), crypto = require("crypto");
const snarkjs = require(path.join(__dirname, "..", "oracle", "ref_shim.js"));
const [proto, zf, wf, drawsHex, wzf, wwf] = process.argv.slice(2);
const rd = (p) => new Uint8Array(fs.readFileSync(p));
(async () => {
    const out = { proto, threads: Math.min(snarkjs.nThreads, 64), cpus: os.cpus().length, cpu_model: os.cpus()[0].model, node: process.version };
    let t0;
    if (wzf) {
        t0 = process.hrtime.bigint();
        await snarkjs[proto].prove(rd(wzf), rd(wwf));
        out.warm_ms = Number(process.hrtime.bigint() - t0) / 1e6;
    }
    const zkey = rd(zf), wtns = rd(wf);
    const curve = await snarkjs.curves.getCurveFromName(process.env.CURVE || "bn128");
    const draws = drawsHex ? drawsHex.split(",").map((h) => new Uint8Array(Buffer.from(h, "hex"))) : [];
    const real = curve.Fr.random;
    let k = 0;
    if (draws.length) curve.Fr.random = function () { return k < draws.length ? draws[k++].slice() : real.call(curve.Fr); };
    t0 = process.hrtime.bigint();
    const res = await snarkjs[proto].prove(zkey, wtns);
    out.ms = Number(process.hrtime.bigint() - t0) / 1e6;
    curve.Fr.random = real;
    out.draws_used = k;
    out.proof_json_sha256 = crypto.createHash("sha256").update(JSON.stringify(res.proof)).digest("hex");
    out.public_signals_sha256 = crypto.createHash("sha256").update(JSON.stringify(res.publicSignals)).digest("hex");
    if (process.env.VERIFY) {
        const vk = await snarkjs.zKey.exportVerificationKey(zkey);
        t0 = process.hrtime.bigint();
        out.verified = await snarkjs[proto].verify(vk, res.publicSignals, res.proof);
        out.verify_ms = Number(process.hrtime.bigint() - t0) / 1e6;
    }
    console.log(JSON.stringify(out));
    process.exit(0);
})().catch((e) => { console.error(e && e.stack || e); process.exit(1); });
