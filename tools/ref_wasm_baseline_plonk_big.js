// tools/ref_wasm_baseline_plonk_big.js — the REFERENCE's own plonk.prove (WASM + worker threads, oracle/ref_shim.js) at sizes where a real ceremony
// in JavaScript is out of reach (powersOfTau.preparePhase2 alone is hours at 2^17): the ptau is written directly from a KNOWN toy tau — never do
// this in production — with exactly the sections plonk.setup reads (src/plonk_setup.js:41-42, :84-92, :127, :478): the header, tauG1 (domain + 6
// powers), tauG2 (two points) and, in section 12, the Lagrange basis of the circuit's power at its offset 2^power - 1 (L_i(tau) G computed as scalars
// in BigInt arithmetic, then G1.timesFr). Everything after that is the reference: plonk.setup on the Multiplier(n) r1cs (SURVEY.md 8d recipe),
// plonk.prove timed twice, plonk.verify as the judge (the pairing check holds because the SRS is a consistent one).
// BUILD CONTAINER ONLY. One JSON line per size:   NTHREADS=8 node --harmony-optional-chaining --harmony-nullish --max-old-space-size=12000 tools/ref_wasm_baseline_plonk_big.js 16
'use strict';
const path = require('path');
const snarkjs = require(path.join(__dirname, '..', 'oracle', 'ref_shim.js'));
const le = (v, k) => { const o = Buffer.alloc(k); let x = BigInt(v); for (let i = 0; i < k; i++) { o[i] = Number(x & 255n); x >>= 8n; } return o; };
const u32 = (v) => le(v, 4), u64 = (v) => le(v, 8);
function multiplierR1cs(r, n) {                               // wires [1, c = x_{n-1}, a, b, x_0 .. x_{n-2}]; constraint i: x_{i-1} * x_{i-1} = x_i - b
    const nWires = n + 3, wireOf = (i) => (i == n - 1) ? 1 : 4 + i;
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
const modpow = (b, e, m) => { let r = 1n; b %= m; while (e > 0n) { if (e & 1n) r = r * b % m; b = b * b % m; e >>= 1n; } return r; };
const now = () => Number(process.hrtime.bigint()) / 1e6;

// the sections of a .ptau that plonk.setup reads, for tau known: section 1 header, 2 tauG1, 3 tauG2, 12 Lagrange bases (only the block of `power`)
async function knownTauPtau(curve, power, tau) {
    const G1 = curve.G1, G2 = curve.G2, Fr = curve.Fr, r = Fr.p, n = 1 << power, sG1 = G1.F.n8 * 2, sG2 = G2.F.n8 * 2;
    const g1 = G1.toAffine(G1.g), rep = new Uint8Array((n + 6) * sG1);
    for (let i = 0; i < n + 6; i++) rep.set(g1, i * sG1);
    const tauG1 = await G1.batchApplyKey(rep, Fr.e(1), Fr.e(tau));                                   // [tau^i] G1, affine Montgomery = the ptau's own format
    const tauG2 = new Uint8Array(2 * sG2);
    tauG2.set(G2.toAffine(G2.g), 0); tauG2.set(G2.toAffine(G2.timesFr(G2.g, Fr.e(tau))), sG2);
    // L_i(tau) = w^i (tau^n - 1) / (n (tau - w^i)); inverses by Montgomery's trick
    const w = BigInt(Fr.toString(Fr.w[power])), zh = (modpow(tau, BigInt(n), r) - 1n + r) % r, ninv = modpow(BigInt(n), r - 2n, r);
    const wi = new Array(n), den = new Array(n), pre = new Array(n);
    let acc = 1n, cur = 1n;
    for (let i = 0; i < n; i++) { wi[i] = cur; den[i] = (tau - cur + r) % r; pre[i] = acc; acc = acc * den[i] % r; cur = cur * w % r; }
    let inv = modpow(acc, r - 2n, r);
    const L = new Uint8Array(((n - 1) + n) * sG1);                                                   // blocks of powers 0 .. power-1 stay zero: never read
    for (let i = n - 1; i >= 0; i--) {
        const dinv = inv * pre[i] % r; inv = inv * den[i] % r;
        const s = wi[i] * zh % r * ninv % r * dinv % r;
        L.set(G1.toAffine(G1.timesFr(G1.g, Fr.e(s))), ((n - 1) + i) * sG1);
    }
    const n8 = G1.F.n8, hdr = Buffer.concat([u32(n8), le(G1.F.p, n8), u32(power), u32(power)]);
    const sec = (t, b) => Buffer.concat([u32(t), u64(b.length), Buffer.from(b.buffer, b.byteOffset, b.byteLength)]);
    return new Uint8Array(Buffer.concat([Buffer.from('ptau'), u32(1), u32(4), sec(1, hdr), sec(2, tauG1), sec(3, tauG2), sec(12, L)]));
}

(async () => {
    const sizes = process.argv.slice(2).map((x) => parseInt(x));
    const curve = await snarkjs.curves.getCurveFromName('bn128'), r = curve.Fr.p;
    for (const lg of sizes) {                                 // lg = log2 of the PLONK DOMAIN (what bench.py --workload plonk --log-n means): every r1cs
        let t0 = now();                                       // constraint of the Multiplier chain becomes two PLONK gates, so the r1cs has 2^(lg-1) - 8 of them
        const ptau = await knownTauPtau(curve, lg, 0x1F3D5B79n);
        const ptau_s = (now() - t0) / 1e3;
        const n = (1 << (lg - 1)) - 8, r1cs = multiplierR1cs(r, n), wtns = multiplierWtns(r, n, 11n, 2n);
        const z = { type: 'mem' };
        t0 = now();
        const rc = await snarkjs.plonk.setup(r1cs, ptau, z, { debug() {}, info() {}, warn: console.error, error: console.error });
        if (rc === -1 || !z.data) throw new Error("plonk.setup refused the key");
        const setup_s = (now() - t0) / 1e3;
        const t = [];
        let res;
        for (let i = 0; i < 2; i++) { t0 = now(); res = await snarkjs.plonk.prove(z.data, wtns); t.push(now() - t0); }
        const vk = await snarkjs.zKey.exportVerificationKey(z.data);
        const ok = await snarkjs.plonk.verify(vk, res.publicSignals, res.proof);
        console.log(JSON.stringify({ proto: 'plonk', log_domain: lg, r1cs_constraints: n, plonk_domain: 1 << lg, threads: snarkjs.nThreads, ms_per_proof: +Math.min(...t).toFixed(1), all_ms: t.map((x) => +x.toFixed(1)), setup_s: +setup_s.toFixed(1),
                                     ptau_from_known_tau_s: +ptau_s.toFixed(1), verified: ok, node: process.version }));
    }
    process.exit(0);
})().catch((e) => { console.error(e); process.exit(1); });
