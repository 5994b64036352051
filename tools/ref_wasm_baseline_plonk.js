// tools/ref_wasm_baseline_plonk.js — the REFERENCE's own CPU path for PLONK and FFLONK (BASELINE.md §3; src/plonk_prove.js:47-172,
// src/fflonk_prove.js:51-267): plonk.prove / fflonk.prove of the bundle (WASM + worker threads, oracle/ref_shim.js) on Multiplier(n) circuits
// (SURVEY.md 8d recipe: the chain x_i = x_{i-1}^2 + b written directly in the r1cs format), keys made by the reference's own setup over a
// seeded ptau. BUILD CONTAINER ONLY (needs /root/reference); the result is committed under profiles/ and quoted by bench.py's cpu_baseline
// for the PLONK / FFLONK lines, labelled with the size it was measured at.
//   NTHREADS=8 node --harmony-optional-chaining --harmony-nullish tools/ref_wasm_baseline_plonk.js <ptau power> <log_n ...>   (one JSON line per run)
'use strict';
const path = require('path');
const snarkjs = require(path.join(__dirname, '..', 'oracle', 'ref_shim.js'));
const le = (v, k) => { const o = Buffer.alloc(k); let x = BigInt(v); for (let i = 0; i < k; i++) { o[i] = Number(x & 255n); x >>= 8n; } return o; };
function multiplierR1cs(r, n) {                               // wires [1, c = x_{n-1}, a, b, x_0 .. x_{n-2}]; constraint i: x_{i-1} * x_{i-1} = x_i - b
    const u32 = (v) => le(v, 4), u64 = (v) => le(v, 8), nWires = n + 3, wireOf = (i) => (i == n - 1) ? 1 : 4 + i;
    const lc = (terms) => Buffer.concat([u32(terms.length)].concat(terms.map(([w, v]) => Buffer.concat([u32(w), le(v, 32)]))));
    const cons = [];
    for (let i = 0; i < n; i++) { const prev = i == 0 ? 2 : wireOf(i - 1); cons.push(lc([[prev, 1n]]), lc([[prev, 1n]]), lc([[3, r - 1n], [wireOf(i), 1n]])); }
    const hdr = Buffer.concat([u32(32), le(r, 32), u32(nWires), u32(1), u32(1), u32(1), u64(nWires), u32(n)]);
    const sec = (t, b) => Buffer.concat([u32(t), u64(b.length), b]);
    return new Uint8Array(Buffer.concat([Buffer.from('r1cs'), u32(1), u32(3), sec(1, hdr), sec(2, Buffer.concat(cons)), sec(3, Buffer.concat(Array.from({ length: nWires }, (_, i) => u64(i))))]));
}
function multiplierWtns(r, n, a, b) {
    const xs = [(a * a + b) % r];
    for (let i = 1; i < n; i++) xs.push((xs[i - 1] * xs[i - 1] + b) % r);
    const sig = [1n, xs[n - 1], a, b].concat(xs.slice(0, n - 1));
    const hdrS = Buffer.concat([le(32, 4), le(r, 32), le(sig.length, 4)]), dataS = Buffer.concat(sig.map((v) => le(v, 32)));
    return new Uint8Array(Buffer.concat([Buffer.from('wtns'), le(2, 4), le(2, 4), le(1, 4), le(hdrS.length, 8), hdrS, le(2, 4), le(dataS.length, 8), dataS]));
}
const now = () => Number(process.hrtime.bigint()) / 1e6;
(async () => {
    const power = parseInt(process.argv[2]), sizes = process.argv.slice(3).map((x) => parseInt(x));
    const curve = await snarkjs.curves.getCurveFromName('bn128'), r = curve.Fr.p, mem = () => ({ type: 'mem' });
    let t0 = now();
    const p0 = mem(), p1 = mem(), pf = mem();
    await snarkjs.powersOfTau.newAccumulator(curve, power, p0);
    await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
    await snarkjs.powersOfTau.preparePhase2(p1, pf);
    console.error(`ptau 2^${power}: ${((now() - t0) / 1e3).toFixed(1)} s`);
    for (const lg of sizes) {
        // PLONK: one gate per constraint, domain = next power of two >= constraints + public rows; FFLONK needs a few spare rows too
        const n = (1 << lg) - 8, r1cs = multiplierR1cs(r, n), wtns = multiplierWtns(r, n, 11n, 2n);
        for (const proto of ['plonk', 'fflonk']) {
            if (proto === 'fflonk' && lg + 4 > power) { console.error(`fflonk 2^${lg}: needs a ptau of power ${lg + 4}, skipped`); continue; }   // 9n + 18 SRS points
            const z = mem();
            t0 = now();
            await snarkjs[proto].setup(r1cs, pf, z);
            const setup_s = (now() - t0) / 1e3;
            const t = [];
            let res;
            for (let i = 0; i < 2; i++) { t0 = now(); res = await snarkjs[proto].prove(z.data, wtns); t.push(now() - t0); }
            const vk = await snarkjs.zKey.exportVerificationKey(z.data);
            const ok = await snarkjs[proto].verify(vk, res.publicSignals, res.proof);
            console.log(JSON.stringify({ proto, log_n: lg, constraints: n, threads: snarkjs.nThreads, ms_per_proof: +Math.min(...t).toFixed(1), all_ms: t.map((x) => +x.toFixed(1)), setup_s: +setup_s.toFixed(1), verified: ok, node: process.version }));
        }
    }
    process.exit(0);
})().catch((e) => { console.error(e); process.exit(1); });
