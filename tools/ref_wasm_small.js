// tools/ref_wasm_small.js — what a SMALL bulk call costs on the reference's own path (WASM + worker threads, oracle/ref_shim.js): G1.multiExpAffine
// and Fr.fft of 4 / 64 / 1024 elements, median of 12 calls. BUILD CONTAINER ONLY (needs /root/reference). Beside it bench.py reports the same calls
// through the N-API addon on the device (wall_through_napi.msm_small_ms / ntt_small_ms): the question of VERDICT r03 weak #2 — does the drop-in
// boundary need a small-n threshold (SURVEY.md 8b)?   NTHREADS=8 node --harmony-optional-chaining --harmony-nullish tools/ref_wasm_small.js
"use strict";
const path = require("path");
const snarkjs = require(path.join(__dirname, "..", "oracle", "ref_shim.js"));
(async () => {
    const curve = await snarkjs.curves.getCurveFromName("bn128");
    const G = curve.G1, Fr = curve.Fr, now = () => Number(process.hrtime.bigint()) / 1e6;
    const out = { what: "reference bundle (snarkjs 0.7.6 + ffjavascript 0.3.1 WASM), median of 12 calls, ms", threads: snarkjs.nThreads, node: process.version, msm: {}, fft: {} };
    for (const k of [4, 64, 1024]) {
        const one = new Uint8Array(k * 64), g = G.toAffine(G.g);
        for (let i = 0; i < k; i++) one.set(g, i * 64);
        const bases = await G.batchApplyKey(one, Fr.e(7), Fr.e(11));                       // P_i = 7 * 11^i * G (SURVEY.md 8d recipe)
        const sc = new Uint8Array(k * 32);
        for (let i = 0; i < sc.length; i++) sc[i] = (i * 131 + 7) & 0xff;
        for (let i = 0; i < k; i++) sc[i * 32 + 31] &= 0x1f;
        const t = [];
        for (let i = 0; i < 12; i++) { const t0 = now(); await G.multiExpAffine(bases, sc); t.push(now() - t0); }
        t.sort((a, b) => a - b); out.msm[k] = +t[6].toFixed(3);
        const u = [];
        for (let i = 0; i < 12; i++) { const t0 = now(); await Fr.fft(sc); u.push(now() - t0); }
        u.sort((a, b) => a - b); out.fft[k] = +u[6].toFixed(3);
    }
    console.log(JSON.stringify(out));
    process.exit(0);
})().catch((e) => { console.error(e); process.exit(1); });
