// tools/dropin_wall.js — what a snarkjs user gets from the DROP-IN: unmodified snarkjs (the reference's bundle, oracle/_ref) with
// registerAll-style patching of the curve (snarkjs_amd/js/register.js + the real addon), `snarkjs.groth16.prove(zkey, wtns)` as the reference's
// own driver runs it (src/groth16_prove.js:28-144): sections read by its binfile reader, buildABC1 as its single-threaded JS loop (:147-187),
// every bulk call (3 x ifft / batchApplyKey / fft, joinABC's batchFromMontgomery, 5 x multiExpAffine) on the MI355X with host buffers in and out.
// bench.py reports it in wall_through_napi.dropin beside the fused prover's figures (VERDICT r05 #9). Measurement infrastructure.
//
//   node --harmony-optional-chaining --harmony-nullish --max-old-space-size=24000 tools/dropin_wall.js <zkey> <wtns> <rHex,sHex> [reps]
// Prints ONE JSON line: cold_ms (first call: bases cross PCIe, window tables are built on second sight), warm_ms (median of `reps`), the time of the
// JS buildABC1 loop inside a warm proof, sha256 of the proof JSON (bench.py compares it with the fused prover's proof for the same draws).
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
process.env.NTHREADS = process.env.NTHREADS || "8";             // the WASM workers are idle once the curve is patched
const snarkjs = require(path.join(__dirname, "..", "oracle", "ref_shim.js"));
const { register, unregister, installFused, uninstallFused } = require(path.join(__dirname, "..", "snarkjs_amd", "js", "register.js"));
const [zf, wf, drawsHex, repsArg] = process.argv.slice(2);
const reps = parseInt(repsArg || "3");
const now = () => Number(process.hrtime.bigint()) / 1e6;
const med = (a) => { const s = a.slice().sort((x, y) => x - y); return s[s.length >> 1]; };
(async () => {
    const zkey = new Uint8Array(fs.readFileSync(zf)), wtns = new Uint8Array(fs.readFileSync(wf));
    const curve = await snarkjs.curves.getCurveFromName(process.env.CURVE || "bn128");
    register(curve, { immutableBases: !!process.env.ZKMI_DROPIN_IMMUTABLE });
    const draws = drawsHex.split(",").map((h) => new Uint8Array(Buffer.from(h, "hex")));
    const real = curve.Fr.random;
    const seeded = () => { let k = 0; curve.Fr.random = () => (k < draws.length ? draws[k++].slice() : real.call(curve.Fr)); };
    const out = { what: "snarkjs.groth16.prove of the unmodified bundle with the curve patched by register.js (bulk calls on the device, JS buildABC1, host buffers)", reps, node: process.version };
    // time spent in the patched bulk calls (awaited wall time inside them) of one warm proof: the rest is the reference's own JavaScript
    const inside = { ms: 0 };
    for (const [obj, names] of [[curve.G1, ["multiExpAffine"]], [curve.G2, ["multiExpAffine"]], [curve.Fr, ["fft", "ifft", "batchApplyKey", "batchFromMontgomery", "batchToMontgomery"]]]) for (const nm of names) {
        const f = obj[nm];
        obj[nm] = async function () { const t0 = now(); try { return await f.apply(this, arguments); } finally { inside.ms += now() - t0; } };
    }
    seeded();
    let t0 = now();
    let res = await snarkjs.groth16.prove(zkey, wtns);
    out.cold_ms = +(now() - t0).toFixed(1);
    seeded(); await snarkjs.groth16.prove(zkey, wtns);           // second sight of the base sections: window tables built
    const t = [], ins = [];
    for (let i = 0; i < reps; i++) { seeded(); inside.ms = 0; t0 = now(); res = await snarkjs.groth16.prove(zkey, wtns); t.push(now() - t0); ins.push(inside.ms); }
    out.warm_ms = +med(t).toFixed(1);
    out.warm_ms_all = t.map((x) => +x.toFixed(1));
    out.inside_device_calls_ms = +med(ins).toFixed(1);
    out.reference_js_ms = +(med(t) - med(ins)).toFixed(1);       // file parsing, buildABC1, the slices and copies between the calls
    out.proof_json_sha256 = crypto.createHash("sha256").update(JSON.stringify(res.proof)).digest("hex");
    // the same call with the FUSED prover installed behind snarkjs.groth16.prove (registerAll(snarkjs, { fused: true })): files given by PATH as a CLI-style caller would
    installFused(snarkjs, {});
    seeded();
    t0 = now();
    let fres = await snarkjs.groth16.prove(zf, wf);
    out.fused_cold_ms = +(now() - t0).toFixed(1);
    const tf = [];
    for (let i = 0; i < reps + 2; i++) { seeded(); t0 = now(); fres = await snarkjs.groth16.prove(zf, wf); tf.push(now() - t0); }
    out.fused_warm_ms = +med(tf).toFixed(1);
    out.fused_warm_ms_all = tf.map((x) => +x.toFixed(1));
    out.fused_proof_equals_dropin_proof = crypto.createHash("sha256").update(JSON.stringify(fres.proof)).digest("hex") === out.proof_json_sha256;
    out.fused_what = "registerAll(snarkjs, { fused: true }): snarkjs.groth16.prove(zkeyPath, wtnsPath) served by js/groth16_native.js (key resident after the first call; the .wtns is read from its path every call)";
    await uninstallFused(snarkjs);
    curve.Fr.random = real;
    unregister(curve);
    console.log(JSON.stringify(out));
    process.exit(0);
})().catch((e) => { console.error(e && e.stack || e); process.exit(1); });
