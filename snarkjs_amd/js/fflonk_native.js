// snarkjs_amd/js/fflonk_native.js — fflonk.prove on the MI355X from Node.js (opt-in fused driver, SURVEY.md §8 f2).
//
// JavaScript twin of snarkjs_amd/fflonk.py: same inputs, checks, error messages and proof object as the reference driver
// (src/fflonk_prove.js:51-1288, src/polynomial/cpolynomial.js:53-83); every O(n) step is a libzkmi call on device-resident
// polynomials through the addon's generic C-ABI binding.  Host side: the Keccak transcript, the roots bookkeeping, Lagrange
// interpolation of the tiny R0/R1/R2 polynomials, the batched inverse — all O(1), as in the reference.
//
//   const { prove, proveAsync } = require("snarkjs_amd/js/fflonk_native.js");
//   const { proof, publicSignals } = prove(zkeyBytes, wtnsBytes);
//   const res = await proveAsync(zkeyBytes, wtnsBytes, null, { device: 2 });    // fflonkProve is async in the reference (src/fflonk_prove.js:51): the four
//                                                                               // commitment waits run on a libuv pool thread
"use strict";
const crypto = require("crypto");
const { _internals: I } = require("./plonk_native.js");
const { addon, call, bindDevice, Field, Poly, Transcript, readSections, devAlloc, devFree, devFrom, mod, modinv, modpow, toLE, fromLE, Q_BN, Q_BLS } = I;

class FflonkKey {                                           // src/zkey_utils.js:301-339, sections of src/fflonk_constants.js
    constructor(zkey, options) {
        const data = zkey instanceof Uint8Array ? zkey : new Uint8Array(zkey);
        const { dv, s } = readSections(data);
        if (dv.getUint32(s[1][0], true) !== 10) throw new Error("zkey file is not fflonk");                 // fflonk_prove.js:71-73
        let off = s[2][0];
        const n8q = dv.getUint32(off, true), q = fromLE(data.subarray(off + 4, off + 4 + n8q)); off += 4 + n8q;
        const n8r = dv.getUint32(off, true); this.r = fromLE(data.subarray(off + 4, off + 4 + n8r)); off += 4 + n8r;
        if (q === Q_BN) { this.curveId = 0; this.curveName = "bn128"; } else if (q === Q_BLS) { this.curveId = 1; this.curveName = "bls12381"; } else throw new Error(`Curve not supported: ${q}`);
        const f = this.f = new Field(this.curveId);
        this.nVars = dv.getUint32(off, true); this.nPublic = dv.getUint32(off + 4, true); this.n = dv.getUint32(off + 8, true);
        this.nAdditions = dv.getUint32(off + 12, true); this.nConstraints = dv.getUint32(off + 16, true); off += 20;
        this.power = Math.log2(this.n);
        for (const nm of ["k1", "k2", "w3", "w4", "w8", "wr"]) { this[nm] = f.unmont(data.subarray(off, off + 32)); off += 32; }
        if (this.curveId !== 0) {
            // The reference's prover takes the curve and its roots from the zkey; only its fflonk.setup hard-codes BN254 constants (src/fflonk_setup.js:
            // 533-556) — the key THAT writes for BLS12-381 has w3^3 != 1 and the reference's own fflonk.prove fails on it with "Polynomial is not divisible"
            // (tests/golden/fflonk_bls12381_unsupported.json). The key is held to what the protocol needs, not to its curve.
            const r = this.r, ok = this.w3 !== 1n && modpow(this.w3, 3n, r) === 1n && modpow(this.w4, 2n, r) === r - 1n && modpow(this.w8, 4n, r) === r - 1n &&
                                   modpow(this.wr, 3n, r) === f.unmont(f.root(this.power));
            if (!ok) throw new Error(`Polynomial is not divisible: this ${this.curveName} FFLONK key is inconsistent (w3^3 != 1 or w4 / w8 / wr of the wrong order), as the reference's fflonk.setup writes it off bn128`);
        }
        off += 4 * n8q;                                                                                      // X_2
        this.C0 = [f.unmontQ(data.subarray(off, off + n8q)), f.unmontQ(data.subarray(off + n8q, off + 2 * n8q))];
        this.device = bindDevice(options && options.device);       // one device per process, chosen by the first key (plonk_native.js: bindDevice)
        if (s[3][1] < 72 * this.nAdditions) throw new Error("zkey additions section is shorter than its header says");
        this.dev = {};                                      // section 3 (additions) too, as it lies in the file: calculateAdditions runs on the device
        for (let t = 3; t <= 17; t++) if (s[t] && s[t][1]) this.dev[t] = devFrom(data.subarray(s[t][0], s[t][0] + s[t][1]));
        this.nPtau = s[16][1] / (2 * n8q);                                                                   // 9n + 18 points
        const h = new Uint8Array(8);
        call("zkmi_msm_table_build", this.curveId, 1, this.dev[16], this.nPtau, h);
        this.ptauTable = Number(new DataView(h.buffer).getBigUint64(0, true));
    }
    sec(t, elemOff = 0) { return this.dev[t] + 32 * elemOff; }
    release() { for (const t of Object.keys(this.dev)) devFree(this.dev[t]); this.dev = {}; if (this.ptauTable) { call("zkmi_msm_table_release", this.ptauTable); this.ptauTable = 0; } }
}

function degree(p) { const d = new BigUint64Array(1); call("zkmi_poly_degree_dev", p.f.cid, p.ptr, p.n, d); return Number(d[0]); }
function ceilLog2Len(maxDegree) { let bits = 0, v = BigInt(maxDegree - 1); while (v > 0n) { bits++; v >>= 1n; } return 2 ** bits; }   // 2^(log2(maxDegree-1)+1)

// CPolynomial.getPolynomial (cpolynomial.js:53-73) on the device
function cpoly(f, polys, n, track) {
    const degs = polys.map((p) => p ? degree(p) : 0);
    let maxDegree = 0;
    polys.forEach((p, j) => { if (p) maxDegree = Math.max(maxDegree, degs[j] * n + j); });
    const length = ceilLog2Len(maxDegree), out = track(new Poly(f, length, false));
    const ptrs = new BigUint64Array(n), lens = new BigUint64Array(n);
    polys.forEach((p, j) => { ptrs[j] = p ? BigInt(p.ptr) : 0n; lens[j] = p ? BigInt(Math.min(degs[j] + 1, p.n)) : 0n; });
    call("zkmi_cpoly_interleave_dev", f.cid, ptrs, lens, n, out.ptr, length);
    return out;
}

// Polynomial.multiExponentiation over PTau: coefficients past the 9n+18 SRS points multiply the point at infinity
function commitEnqueue(key, poly) { addon.msmTableMultiEnqueueMontDev(key.ptauTable, [poly.ptr], [Math.min(poly.n, key.nPtau)]); }   // batchFromMontgomery + the MSM, enqueued
function commitCollect(key) {
    const f = key.f, aff = new Uint8Array(2 * f.n8q);
    call("zkmi_to_affine", f.cid, 1, addon.msmTableMultiCollect(key.ptauTable, 1), aff);
    return [f.unmontQ(aff.subarray(0, f.n8q)), f.unmontQ(aff.subarray(f.n8q))];
}
function commit(key, poly) { commitEnqueue(key, poly); return commitCollect(key); }
async function commitAsync(key, poly) { commitEnqueue(key, poly); await addon.synchronizeAsync(0); call("zkmi_pipeline_select", 0); return commitCollect(key); }   // the wait on a libuv pool thread
const divZerofier = (p, n, beta) => call("zkmi_poly_div_by_zerofier_dev", p.f.cid, p.ptr, p.n, n, p.f.mont(beta));

// ---- O(1) host algebra on tiny polynomials (arrays of BigInt, lowest coefficient first) ---------------------------------
const evalSmall = (c, x, r) => c.reduceRight((acc, v) => (acc * x + v) % r, 0n);
function mulLinear(p, x, r) { const out = new Array(p.length + 1).fill(0n); p.forEach((c, d) => { out[d] = mod(out[d] - x * c, r); out[d + 1] = (out[d + 1] + c) % r; }); return out; }
function lagrange(xs, ys, r) {                               // Polynomial.lagrangePolynomialInterpolation (polynomial.js:896-930)
    const out = new Array(xs.length).fill(0n);
    for (let i = 0; i < xs.length; i++) {
        let num = [1n];
        xs.forEach((x, j) => { if (j !== i) num = mulLinear(num, x, r); });
        const k = ys[i] * modinv(evalSmall(num, xs[i], r), r) % r;
        num.forEach((c, d) => { out[d] = (out[d] + c * k) % r; });
    }
    return out;
}
const zerofier = (xs, r) => xs.reduce((p, x) => mulLinear(p, x, r), [1n]);    // Polynomial.zerofierPolynomial (:932-950)
function small(f, coefs, track) { const b = new Uint8Array(32 * coefs.length); coefs.forEach((c, i) => b.set(f.mont(c), 32 * i)); const p = track(new Poly(f, coefs.length, false)); call("zkmi_memcpy_h2d", p.ptr, b, b.length); return p; }

// The proof is a generator (proveSteps): `pts.X = yield poly` asks the driver for the commitment of `poly` — the call the host waits in.
function prove(zkey, wtns, blindingMont = null, options = null) {
    const key = zkey instanceof FflonkKey ? zkey : new FflonkKey(zkey, options);
    const polys = [];
    const track = (p) => { polys.push(p); return p; };
    try {
        const steps = proveSteps(key, wtns instanceof Uint8Array ? wtns : new Uint8Array(wtns), blindingMont, track);
        for (let s = steps.next(); ; s = steps.next(commit(key, s.value))) if (s.done) return s.value;
    } finally {
        polys.forEach((p) => p.free());
        if (!(zkey instanceof FflonkKey)) key.release();
    }
}
// async like the reference's fflonkProve (src/fflonk_prove.js:51): the commitment waits on a libuv pool thread; serialised per process
let asyncQueue = Promise.resolve();
function proveAsync(zkey, wtns, blindingMont = null, options = null) {
    const run = async () => {
        const key = zkey instanceof FflonkKey ? zkey : new FflonkKey(zkey, options);
        const polys = [];
        const track = (p) => { polys.push(p); return p; };
        try {
            const steps = proveSteps(key, wtns instanceof Uint8Array ? wtns : new Uint8Array(wtns), blindingMont, track);
            let s = steps.next();
            while (!s.done) { const pt = await commitAsync(key, s.value); s = steps.next(pt); }
            return s.value;
        } finally {
            polys.forEach((p) => p.free());
            if (!(zkey instanceof FflonkKey)) key.release();
        }
    };
    const p = asyncQueue.then(run, run);
    asyncQueue = p.catch(() => {});
    return p;
}

function* proveSteps(key, wt, blindingMont, track) {
    const f = key.f, r = f.r, n = key.n, power = key.power;
    const P = (len, zero = true) => track(new Poly(f, len, zero));
    const { dv, s: ws } = readSections(wt);
    const n8 = dv.getUint32(ws[1][0], true), nWitness = dv.getUint32(ws[1][0] + 4 + n8, true), nW = key.nVars - key.nAdditions;
    if (key.r !== fromLE(wt.subarray(ws[1][0] + 4, ws[1][0] + 4 + n8))) throw new Error("Curve of the witness does not match the curve of the proving key");
    if (nWitness !== nW) throw new Error(`Invalid witness length. Circuit: ${key.nVars}, witness: ${nWitness}, ${key.nAdditions}`);
    if (ws[2][0] + nWitness * 32 > wt.length) throw new Error("Invalid witness length: the wtns data section is shorter than its header says");
    const wit = wt.slice(ws[2][0], ws[2][0] + nWitness * 32);
    const pub = [];
    for (let i = 1; i <= key.nPublic; i++) pub.push(fromLE(wit.subarray(32 * i, 32 * i + 32)));
    wit.fill(0, 0, 32);
    const bm = [null];                                       // the 9 Fr.random() draws (:321-324) as Montgomery bytes
    for (let i = 0; i < 9; i++) bm.push(blindingMont ? Uint8Array.from(blindingMont[i]) : f.mont(fromLE(crypto.randomBytes(40))));
    const b = [0n].concat(bm.slice(1).map((x) => f.unmont(x)));
    // calculateAdditions (:271-300): the internal signals, ONE launch on the device (zkmi_plonk_additions_dev)
    let dWit = 0, dInt = 0;
    try {
        dWit = devFrom(wit); dInt = I.devAlloc(32 * Math.max(key.nAdditions, 1));
        if (key.nAdditions) call("zkmi_plonk_additions_dev", f.cid, key.sec(3), key.nAdditions, dWit, nW, dInt);
        const mont = (v) => f.mont(v), wN = f.root(power), w2N = f.root(power + 1), w4N = f.root(power + 2), wv = f.unmont(wN);
        const pts = {}, evs = {}, big = (a) => new BigUint64Array(a.map((x) => BigInt(x || 0)));

        // ---- ROUND 1 (:318-556)
        const A = P(n, false), B = P(n, false), Cw = P(n, false);
        call("zkmi_plonk_gather_wires_mont_dev", f.cid, dWit, nW, dInt, key.nAdditions, key.sec(4), key.sec(5), key.sec(6), key.nConstraints, n, A.ptr, B.ptr, Cw.ptr);
        // the reference writes the blinding scalars (their Montgomery bytes) into the normal-form buffers BEFORE batchToMontgomery (:377-386): what ends up in the buffers is
        // toMontgomery of those bytes read as an integer — written here directly, behind the gather that already converted the rest
        for (const [p, k0, k1] of [[A, 1, 2], [B, 3, 4], [Cw, 5, 6]]) {
            const raw = new Uint8Array(64); raw.set(mont(fromLE(bm[k0])), 0); raw.set(mont(fromLE(bm[k1])), 32);
            call("zkmi_memcpy_h2d", p.at(n - 2), raw, 64);
        }
        const pA = track(A.ntt(true)), pB = track(B.ntt(true)), pC = track(Cw.ntt(true));
        const eA = track(pA.extendedEvals(4)), eB = track(pB.extendedEvals(4)), eC = track(pC.extendedEvals(4));
        const T0 = P(4 * n, false);
        call("zkmi_fflonk_t0_dev", f.cid, big([eA.ptr, eB.ptr, eC.ptr, 0, key.sec(9, n), key.sec(7, n), key.sec(8, n), key.sec(10, n), key.sec(11, n), 0, 0, 0, key.sec(15), A.ptr]), n, key.nPublic, T0.ptr);
        const pT0 = T0.ntt(true, T0);
        divZerofier(pT0, n, 1n);
        if (degree(pT0) >= 2 * n - 2) throw new Error("T0 Polynomial is not well calculated");
        const C1 = cpoly(f, [pA, pB, pC, pT0], 4, track);
        if (degree(C1) >= 8 * n - 8) throw new Error("C1 Polynomial is not well calculated");
        pts.C1 = yield C1;

        // ---- ROUND 2 (:558-862)
        let tr = new Transcript(f);
        tr.point(key.C0);
        for (let i = 0; i < key.nPublic; i++) tr.scalar(A.get(i));
        tr.point(pts.C1);
        const beta = tr.challenge();
        tr.reset(); tr.scalar(beta);
        const gamma = tr.challenge();
        const Zb = P(n, false);
        call("zkmi_plonk_compute_z_dev", f.cid, A.ptr, B.ptr, Cw.ptr, key.sec(12, n), key.sec(13, n), key.sec(14, n), n, mont(beta), mont(gamma), mont(key.k1), mont(key.k2), wN, Zb.ptr);
        const [pZ, eZ] = Zb.ifftBlinded([b[9], b[8], b[7]]).map(track);
        const b789 = new Uint8Array(96); b789.set(mont(b[7]), 0); b789.set(mont(b[8]), 32); b789.set(mont(b[9]), 64);
        const T1 = P(2 * n, false), T1z = P(2 * n, false);
        call("zkmi_fflonk_t1_dev", f.cid, eZ.ptr, key.sec(15), n, b789, w2N, T1.ptr, T1z.ptr);
        const pT1 = T1.ntt(true, T1);
        divZerofier(pT1, n, 1n);
        pT1.axpy(T1z.ntt(true, T1z));
        if (degree(pT1) >= n + 2) throw new Error("T1 Polynomial is not well calculated");
        const T2 = P(4 * n, false), T2z = P(4 * n, false);
        call("zkmi_fflonk_t2_dev", f.cid, big([eA.ptr, eB.ptr, eC.ptr, eZ.ptr, 0, 0, 0, 0, 0, key.sec(12, n), key.sec(13, n), key.sec(14, n), 0, 0]), n, b789, mont(beta), mont(gamma), mont(key.k1), mont(key.k2), wN, w4N, T2.ptr, T2z.ptr);
        const pT2 = T2.ntt(true, T2);
        divZerofier(pT2, n, 1n);
        pT2.axpy(T2z.ntt(true, T2z));
        if (degree(pT2) >= 3 * n) throw new Error("T2 Polynomial is not well calculated");
        const C2 = cpoly(f, [pZ, pT1, pT2], 3, track);
        if (degree(C2) >= 9 * n) throw new Error("C2 Polynomial is not well calculated");
        pts.C2 = yield C2;

        // ---- ROUND 3 (:864-963)
        tr = new Transcript(f);
        tr.scalar(gamma); tr.point(pts.C2);
        const xiSeed = tr.challenge(), xs2 = xiSeed * xiSeed % r;
        const pw = (w, k) => Array.from({ length: k }, (_, i) => modpow(w, BigInt(i), r));
        const h0 = xs2 * xiSeed % r, S0 = pw(key.w8, 8).map((x) => h0 * x % r);
        const h1 = h0 * h0 % r, S1 = pw(key.w4, 4).map((x) => h1 * x % r);
        const h2 = h1 * xs2 % r, S2 = pw(key.w3, 3).map((x) => h2 * x % r);
        const h3 = h2 * key.wr % r, S2p = pw(key.w3, 3).map((x) => h3 * x % r);
        const xi = h2 * h2 % r * h2 % r, xiw = xi * wv % r;
        // fifteen evaluations, two waits (zkmi_poly_evaluate_multi_dev); the selector and sigma polynomials are read where they lie in the key
        const EV = ["ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w"];
        const evalAll = (pairs) => { let out = []; for (let i = 0; i < pairs.length; i += 8) { const c = pairs.slice(i, i + 8); out = out.concat(I.evaluateMany(f, c.map(([p]) => p), c.map(([, x]) => x))); } return out; };
        evalAll([7, 8, 9, 10, 11, 12, 13, 14].map((t) => [[key.sec(t, 0), n], xi]).concat([pA, pB, pC, pZ].map((p) => [[p.ptr, p.n], xi]), [pZ, pT1, pT2].map((p) => [[p.ptr, p.n], xiw])))
            .forEach((v, i) => { evs[EV[i]] = v; });

        // ---- ROUND 4 (:965-1057)
        tr = new Transcript(f);
        tr.scalar(xiSeed);
        for (const k of EV) tr.scalar(evs[k]);
        const alpha = tr.challenge();
        const C0p = key.sec(17, 0), C0n = 8 * n, S22 = S2.concat(S2p), neg = (v) => mod(-v, r);   // C0 is read where it lies in the key
        // eighteen evaluations at the opening roots, three waits
        const vals = evalAll(S0.map((x) => [[C0p, C0n], x]).concat(S1.map((x) => [[C1.ptr, C1.n], x]), S22.map((x) => [[C2.ptr, C2.n], x])));
        const R0 = lagrange(S0, vals.slice(0, 8), r), R1 = lagrange(S1, vals.slice(8, 12), r), R2 = lagrange(S22, vals.slice(12), r);
        // F = (C0 - R0) / ZT0 + alpha (C1 - R1) / ZT1 + alpha^2 (C2 - R2) / ZT2 (:1009-1042): each numerator one launch (zkmi_poly_lincomb_dev)
        const nF = Math.max(C0n, C1.n, C2.n), a2 = alpha * alpha % r;
        const F = I.lincomb(f, P(nF, false), [[C0p, C0n, null], [small(f, R0, track).ptr, R0.length, neg(1n)]]);
        call("zkmi_poly_div_by_zerofier_dev", f.cid, F.ptr, C0n, 8, mont(xi));                         // the division acts on C0's own length
        const f2 = I.lincomb(f, P(C1.n, false), [[C1.ptr, C1.n, alpha], [small(f, R1, track).ptr, R1.length, neg(alpha)]]);
        divZerofier(f2, 4, xi);
        const f3 = I.lincomb(f, P(C2.n, false), [[C2.ptr, C2.n, a2], [small(f, R2, track).ptr, R2.length, neg(a2)]]);
        divZerofier(f3, 3, xi); divZerofier(f3, 3, xiw);
        I.lincomb(f, F, [[F.ptr, nF, null], [f2.ptr, f2.n, null], [f3.ptr, f3.n, null]]);
        if (degree(F) >= 9 * n - 6) throw new Error("F Polynomial is not well calculated");
        pts.W1 = yield F;

        // ---- ROUND 5 (:1059-1180)
        tr = new Transcript(f);
        tr.scalar(alpha); tr.point(pts.W1);
        const y = tr.challenge();
        const prod = (xs) => xs.reduce((a, x) => a * mod(y - x, r) % r, 1n);
        const mulL0 = prod(S0), mulL1 = prod(S1), mulL2 = prod(S22);
        const preL0 = mulL1 * mulL2 % r, preL1 = alpha * mulL0 % r * mulL2 % r, preL2 = alpha * alpha % r * mulL0 % r * mulL1 % r;
        const toInv = [["denH1", mulL1], ["denH2", mulL2]];
        // L = preL0 (C0 - R0(y)) + preL1 (C1 - R1(y)) + preL2 (C2 - R2(y)) - ZT(y) F, times 1 / ZTS2(y) (:1061-1083): one launch. The scalar does not move the degree the
        // reference tests before it multiplies by it
        const inv2 = modinv(evalSmall(zerofier(S1.concat(S22), r), y, r), r), zty = evalSmall(zerofier(S0.concat(S1, S22), r), y, r);
        const Lp = I.lincomb(f, P(nF, false), [[C0p, C0n, preL0 * inv2 % r], [C1.ptr, C1.n, preL1 * inv2 % r], [C2.ptr, C2.n, preL2 * inv2 % r], [F.ptr, nF, neg(zty * inv2 % r)]],
                             neg((preL0 * evalSmall(R0, y, r) + preL1 * evalSmall(R1, y, r) + preL2 * evalSmall(R2, y, r)) % r * inv2 % r));
        if (degree(Lp) >= 9 * n) throw new Error("L Polynomial is not well calculated");
        try { divZerofier(Lp, 1, y); } catch (e) { throw new Error("Degree of L(X)/(ZTS2(y)(X-y)) remainder is not 0"); }
        if (degree(Lp) >= 9 * n - 1) throw new Error("Degree of L(X)/(ZTS2(y)(X-y)) is not correct");
        pts.W2 = yield Lp;

        // ---- getMontgomeryBatchedInverse (:1182-1287)
        toInv.push(["zh", mod(modpow(xi, BigInt(n), r) - 1n, r)]);
        for (const [name, roots] of [["LiS0", S0], ["LiS1", S1]]) {
            const ln = roots.length, den1 = BigInt(ln) * modpow(roots[0], BigInt(ln - 2), r) % r;
            for (let i = 0; i < ln; i++) toInv.push([`${name}_${i + 1}`, den1 * roots[((ln - 1) * i) % ln] % r * mod(y - roots[i], r) % r]);
        }
        let den1 = 3n * S2[0] % r * mod(xi - xiw, r) % r;
        for (let i = 0; i < 3; i++) toInv.push([`LiS2_${i + 1}`, den1 * (S2[2 * i % 3] * mod(y - S2[i], r) % r) % r]);
        den1 = 3n * S2p[0] % r * mod(xiw - xi, r) % r;
        for (let i = 0; i < 3; i++) toInv.push([`LiS2_${i + 4}`, den1 * (S2p[2 * i % 3] * mod(y - S2p[i], r) % r) % r]);
        let ww = 1n;
        for (let i = 0; i < Math.max(1, key.nPublic); i++) { toInv.push([`Li_${i + 1}`, BigInt(n) * mod(xi - ww, r) % r]); ww = ww * wv % r; }
        evs.inv = modinv(toInv.reduce((a, [, v]) => a * v % r, 1n), r);

        const proof = { polynomials: {}, evaluations: {} };                                            // src/proof.js:61-83
        for (const k of ["C1", "C2", "W1", "W2"]) proof.polynomials[k] = [pts[k][0].toString(), pts[k][1].toString(), "1"];
        for (const k of EV.concat(["inv"])) proof.evaluations[k] = evs[k].toString();
        proof.protocol = "fflonk";
        proof.curve = key.curveName;
        return { proof, publicSignals: pub.map((p) => p.toString()) };
    } finally {
        devFree(dWit); devFree(dInt);
    }
}

module.exports = { prove, proveAsync, FflonkKey };
