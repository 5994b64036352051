// snarkjs_amd/js/plonk_native.js — plonk.prove on the MI355X from Node.js (opt-in fused driver, SURVEY.md §8 f2).
//
// Same inputs and outputs as the reference driver (src/plonk_prove.js:47-888: zkey + wtns bytes in, {proof, publicSignals} out,
// same checks and error messages); every O(n) step runs in libzkmi on device-resident polynomials, reached through the addon's
// generic `call(name, ...)` binding of the C-ABI (napi/zkmi_napi.c).  This file is the JavaScript twin of snarkjs_amd/plonk.py
// (which the pytest parity suite drives): the reference's Polynomial / Evaluations objects become handles on device buffers and
// each of their methods one C entry point.  The host keeps what is O(1) in the reference too: the Keccak transcript
// (zkmi_keccak256), challenges, a handful of field operations (BigInt), and calculateAdditions (a sequential chain).
//
//   const { prove, proveAsync } = require("snarkjs_amd/js/plonk_native.js");
//   const { proof, publicSignals } = prove(zkeyBytes, wtnsBytes);          // synchronous; throws without a GPU (no fallback)
//   const res = await proveAsync(zkeyBytes, wtnsBytes, null, { device: 3 });   // like the reference's async plonk16Prove (src/plonk_prove.js:47): the
//                                                                              // waits run on a libuv pool thread, the event loop keeps turning
"use strict";
const path = require("path");
const crypto = require("crypto");
const addon = require(path.join(__dirname, "..", "napi", "zkmi_napi.node"));
const call = (name, ...a) => addon.call(name, ...a);
// The library binds ONE device per process (include/zkmi.h: zkmi_init). The first key decides: new PlonkKey(zkey, { device }) / prove(.., { device }),
// default device 0 (HIP_VISIBLE_DEVICES renumbers the visible ones from 0); a later key that asks for another device is refused — PLONK replicas
// run one process per GPU, like every other multi-GPU path here.
let boundDevice = null;
function bindDevice(device) {
    const d = device === undefined || device === null ? (boundDevice === null ? 0 : boundDevice) : device;
    if (!Number.isInteger(d) || d < 0) throw new Error(`plonk_native: bad device ${device}`);
    if (boundDevice !== null && boundDevice !== d) throw new Error(`plonk_native: this process is bound to device ${boundDevice}; device ${d} needs its own process`);
    addon.init(d);
    boundDevice = d;
    return d;
}

const R_BN = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const Q_BN = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
const R_BLS = 52435875175126190479447740508185965837690552500527637822603658699938581184513n;
const Q_BLS = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaabn;

const mod = (a, m) => { const x = a % m; return x < 0n ? x + m : x; };
function modinv(a, m) {
    let [r0, r1, s0, s1] = [mod(a, m), m, 1n, 0n];
    while (r1 !== 0n) { const q = r0 / r1; [r0, r1] = [r1, r0 - q * r1]; [s0, s1] = [s1, s0 - q * s1]; }
    if (r0 !== 1n) throw new Error("not invertible");
    return mod(s0, m);
}
function modpow(b, e, m) { let r = 1n; b = mod(b, m); while (e > 0n) { if (e & 1n) r = r * b % m; b = b * b % m; e >>= 1n; } return r; }
function toLE(v, n) { const o = new Uint8Array(n); for (let i = 0; i < n; i++) { o[i] = Number(v & 0xffn); v >>= 8n; } return o; }
function fromLE(b) { let v = 0n; for (let i = b.length - 1; i >= 0; i--) v = (v << 8n) | BigInt(b[i]); return v; }
function toBE(v, n) { return toLE(v, n).reverse(); }
function fromBE(b) { let v = 0n; for (let i = 0; i < b.length; i++) v = (v << 8n) | BigInt(b[i]); return v; }

class Field {
    constructor(cid) {
        this.cid = cid; this.r = cid === 0 ? R_BN : R_BLS; this.q = cid === 0 ? Q_BN : Q_BLS; this.n8q = cid === 0 ? 32 : 48;
        this.Rr = mod(1n << 256n, this.r); this.Rri = modinv(this.Rr, this.r);
        this.Rqi = modinv(mod(1n << BigInt(8 * this.n8q), this.q), this.q);
    }
    mont(v) { return toLE(mod(v, this.r) * this.Rr % this.r, 32); }                 // BigInt -> 32 Montgomery bytes
    unmont(b) { return fromLE(b) * this.Rri % this.r; }
    unmontQ(b) { return fromLE(b) * this.Rqi % this.q; }
    root(i) { const o = new Uint8Array(32); call("zkmi_fr_root", this.cid, i, o); return o; }
}

// ---- device memory ------------------------------------------------------------------------------------------------------
function devAlloc(bytes) {
    const out = new Uint8Array(8);
    call("zkmi_dev_alloc", Math.max(bytes, 32), out);
    return Number(new DataView(out.buffer).getBigUint64(0, true));
}
const devFree = (p) => { if (p) call("zkmi_dev_free", p); };
function devFrom(host) { const p = devAlloc(host.length); if (host.length) call("zkmi_memcpy_h2d", p, host, host.length); return p; }

// the reference's Polynomial / Evaluations: n Montgomery Fr elements in device memory
class Poly {
    constructor(f, n, zero = true) { this.f = f; this.n = n; this.ptr = devAlloc(n * 32); if (zero) call("zkmi_memset_dev", this.ptr, 0, n * 32); }
    at(i) { return this.ptr + 32 * i; }
    copyFrom(src, count, dstOff = 0) { call("zkmi_memcpy_d2d", this.at(dstOff), src, count * 32); return this; }
    get(i) { const o = new Uint8Array(32); call("zkmi_memcpy_d2h", o, this.at(i), 32); return this.f.unmont(o); }
    set(i, v) { call("zkmi_memcpy_h2d", this.at(i), this.f.mont(v), 32); }
    axpy(other, k = null, sub = false) { call("zkmi_poly_axpy_dev", this.f.cid, this.ptr, other.ptr, other.n, k === null ? null : this.f.mont(k), sub ? 1 : 0); }
    scale(k) { call("zkmi_poly_scale_dev", this.f.cid, this.ptr, this.n, this.f.mont(k)); }
    addScalar(v) { call("zkmi_poly_add_scalar_dev", this.f.cid, this.ptr, this.f.mont(v)); }
    evaluate(x) { const o = new Uint8Array(32); call("zkmi_poly_evaluate_dev", this.f.cid, this.ptr, this.n, this.f.mont(x), o); return this.f.unmont(o); }
    tailIsZero(start) { const z = new Int32Array(1); call("zkmi_poly_is_zero_dev", this.f.cid, this.at(start), this.n - start, z); return z[0] === 1; }
    blinded(factors) {                                                   // blindCoefficients (polynomial.js:68-93)
        const out = new Poly(this.f, this.n + factors.length).copyFrom(this.ptr, this.n);
        const fb = new Uint8Array(32 * factors.length);
        factors.forEach((x, i) => fb.set(this.f.mont(x), 32 * i));
        call("zkmi_poly_blind_dev", this.f.cid, out.ptr, this.n, fb, factors.length);
        return out;
    }
    ntt(inverse, out = null) { out = out || new Poly(this.f, this.n, false); call("zkmi_ntt_dev", this.f.cid, this.ptr, out.ptr, Math.log2(this.n), inverse ? 1 : 0, null, null); return out; }
    extendedEvals(ext) {                                                 // Evaluations.fromPolynomial: the zero padding is read by the first pass, never written
        const e = new Poly(this.f, this.n * ext, false);
        call("zkmi_ntt_padded_dev", this.f.cid, this.ptr, this.n, e.ptr, Math.log2(this.n * ext), 0);
        return e;
    }
    // The pattern of rounds 1 and 2 (plonk_prove.js:285-311, :441-455): coefficients = ifft(this) into a buffer with room for the blinding tail, Evaluations.fromPolynomial(.., 4)
    // with the zero padding READ instead of written (zkmi_ntt_padded_dev), blindCoefficients in place (zkmi_poly_blind_tail_dev) -> [blinded polynomial, 4n evaluations]
    ifftBlinded(factors) {
        const f = this.f, n = this.n, out = new Poly(f, n + factors.length, false), ev = new Poly(f, 4 * n, false);
        call("zkmi_ntt_dev", f.cid, this.ptr, out.ptr, Math.log2(n), 1, null, null);
        call("zkmi_ntt_padded_dev", f.cid, out.ptr, n, ev.ptr, Math.log2(4 * n), 0);
        const fb = new Uint8Array(32 * factors.length);
        factors.forEach((x, i) => fb.set(f.mont(x), 32 * i));
        call("zkmi_poly_blind_tail_dev", f.cid, out.ptr, n, fb, factors.length);
        return [out, ev];
    }
    free() { devFree(this.ptr); this.ptr = 0; }
}
// out[i] = sum_j k_j p_j[i] + (i == 0 ? constant : 0) in one launch (zkmi_poly_lincomb_dev). terms: [device pointer, length, k | null]; a zkmi_poly_term is 56 bytes:
// pointer and length as 64-bit little-endian integers, k (32 Montgomery bytes), has_k, reserved
function lincomb(f, out, terms, constant = null) {
    const buf = new Uint8Array(56 * terms.length), dv = new DataView(buf.buffer);
    terms.forEach(([ptr, len, k], j) => {
        dv.setBigUint64(56 * j, BigInt(ptr), true); dv.setBigUint64(56 * j + 8, BigInt(len), true);
        if (k !== null) { buf.set(f.mont(k), 56 * j + 16); dv.setUint32(56 * j + 48, 1, true); }
    });
    call("zkmi_poly_lincomb_dev", f.cid, out.ptr, out.n, buf, terms.length, constant === null ? null : f.mont(constant));
    return out;
}
// [p(x)] for [device pointer, length] pairs and points, one wait (zkmi_poly_evaluate_multi_dev)
function evaluateMany(f, polys, xs) {
    const cnt = polys.length, xb = new Uint8Array(32 * cnt), out = new Uint8Array(32 * cnt);
    xs.forEach((x, i) => xb.set(f.mont(x), 32 * i));
    call("zkmi_poly_evaluate_multi_dev", f.cid, new BigUint64Array(polys.map(([p]) => BigInt(p))), new BigUint64Array(polys.map(([, n]) => BigInt(n))), xb, cnt, out);
    return polys.map((_, i) => f.unmont(out.subarray(32 * i, 32 * i + 32)));
}

function readSections(data) {
    const dv = new DataView(data.buffer, data.byteOffset, data.byteLength), s = {};
    let off = 12;
    for (let i = 0, k = dv.getUint32(8, true); i < k; i++) {
        const t = dv.getUint32(off, true), ln = Number(dv.getBigUint64(off + 4, true));
        s[t] = [off + 12, ln];
        off += 12 + ln;
    }
    return { dv, s };
}

// A PLONK zkey resident on the device (src/zkey_utils.js:261-299)
class PlonkKey {
    constructor(zkey, options) {
        const data = zkey instanceof Uint8Array ? zkey : new Uint8Array(zkey);
        const { dv, s } = readSections(data);
        if (dv.getUint32(s[1][0], true) !== 2) throw new Error("zkey file is not plonk");                  // plonk_prove.js:60-62
        let off = s[2][0];
        const n8q = dv.getUint32(off, true), q = fromLE(data.subarray(off + 4, off + 4 + n8q)); off += 4 + n8q;
        const n8r = dv.getUint32(off, true); this.r = fromLE(data.subarray(off + 4, off + 4 + n8r)); off += 4 + n8r;
        if (q === Q_BN) { this.curveId = 0; this.curveName = "bn128"; } else if (q === Q_BLS) { this.curveId = 1; this.curveName = "bls12381"; } else throw new Error(`Curve not supported: ${q}`);
        const f = this.f = new Field(this.curveId);
        this.nVars = dv.getUint32(off, true); this.nPublic = dv.getUint32(off + 4, true); this.n = dv.getUint32(off + 8, true);
        this.nAdditions = dv.getUint32(off + 12, true); this.nConstraints = dv.getUint32(off + 16, true); off += 20;
        this.power = Math.log2(this.n);
        this.k1 = f.unmont(data.subarray(off, off + 32)); this.k2 = f.unmont(data.subarray(off + 32, off + 64)); off += 64;
        this.commit = {};
        for (const nm of ["Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"]) { this.commit[nm] = [f.unmontQ(data.subarray(off, off + n8q)), f.unmontQ(data.subarray(off + n8q, off + 2 * n8q))]; off += 2 * n8q; }
        this.device = bindDevice(options && options.device);
        if (s[3][1] < 72 * this.nAdditions) throw new Error("zkey additions section is shorter than its header says");
        this.dev = {};                                      // section 3 (additions) too, as it lies in the file: calculateAdditions runs on the device
        for (let t = 3; t <= 14; t++) if (s[t] && s[t][1]) this.dev[t] = devFrom(data.subarray(s[t][0], s[t][0] + s[t][1]));
        this.nPtau = s[14][1] / (2 * n8q);
        const h = new Uint8Array(8);
        call("zkmi_msm_table_build", this.curveId, 1, this.dev[14], this.nPtau, h);     // the SRS is static: window tables, built once
        this.ptauTable = Number(new DataView(h.buffer).getBigUint64(0, true));
    }
    sec(t, elemOff = 0) { return this.dev[t] + 32 * elemOff; }
    release() { for (const t of Object.keys(this.dev)) devFree(this.dev[t]); this.dev = {}; if (this.ptauTable) { call("zkmi_msm_table_release", this.ptauTable); this.ptauTable = 0; } }
}

class Transcript {                                          // src/Keccak256Transcript.js
    constructor(f) { this.f = f; this.parts = []; }
    reset() { this.parts = []; }
    point(p) { this.parts.push(toBE(p[0], this.f.n8q), toBE(p[1], this.f.n8q)); }
    scalar(v) { this.parts.push(toBE(v, 32)); }
    challenge() {
        if (!this.parts.length) throw new Error("Keccak256Transcript: No data to generate a transcript");
        const msg = Buffer.concat(this.parts.map((x) => Buffer.from(x))), out = new Uint8Array(32);
        call("zkmi_keccak256", new Uint8Array(msg.buffer, msg.byteOffset, msg.length), msg.length, out);
        return fromBE(out) % this.f.r;
    }
}

// Polynomial.multiExponentiation (polynomial.js:970-977) for the commitments of one round: batchFromMontgomery of the round's polynomials (one launch) and up to four MSMs
// against the resident SRS table are ENQUEUED by one call (zkmi_msm_table_multi_enqueue_mont_dev); the collect waits for them; toAffine on the host
function commitPoints(key, count, jac) {
    const f = key.f, out = [];
    for (let i = 0; i < count; i++) {
        const aff = new Uint8Array(2 * f.n8q);
        call("zkmi_to_affine", f.cid, 1, jac.slice(i * 3 * f.n8q, (i + 1) * 3 * f.n8q), aff);
        out.push([f.unmontQ(aff.subarray(0, f.n8q)), f.unmontQ(aff.subarray(f.n8q))]);
    }
    return out;
}
// two halves (r06): enqueue on the proof's own pipeline slot, collect at its next turn — in proveMany the other proof's next segment, its commitments included, is enqueued in
// between and runs underneath this round's latency-bound reduction tail
function commitEnqueue(key, polys) {
    addon.msmTableMultiEnqueueMontDev(key.ptauTable, polys.map((p) => p.ptr), polys.map((p) => p.n));
    return { polys };
}
const commitCollect = (key, st) => commitPoints(key, st.polys.length, addon.msmTableMultiCollect(key.ptauTable, st.polys.length));
const commit = (key, polys) => commitCollect(key, commitEnqueue(key, polys));
async function commitAsync(key, polys, slot) {
    const st = commitEnqueue(key, polys);
    await addon.synchronizeAsync(slot);                                  // the wait, on a libuv pool thread
    call("zkmi_pipeline_select", slot);
    return commitCollect(key, st);                                       // everything is finished: folds the window sums
}
// A proof is a generator (proveSteps): it yields right before each of its long blocking calls. `yield { commit: [polys] }` asks the driver for the
// commitments of a round and receives the points; a bare `yield` stands before a read-back that waits for everything queued so far.
const serve = (key, req) => (req && req.commit ? commit(key, req.commit) : undefined);

// plonk.prove(zkey, wtns[, blindingMont]): blindingMont = the 11 Fr.random() draws (:224-227) as 32-byte Montgomery values, for
// bit-exact reproduction of a reference proof; default = fresh randomness.
function prove(zkey, wtns, blindingMont = null, options = null) {
    const key = zkey instanceof PlonkKey ? zkey : new PlonkKey(zkey, options);
    const polys = [];
    const P = (n, zero = true) => { const p = new Poly(key.f, n, zero); polys.push(p); return p; };
    const track = (p) => { polys.push(p); return p; };
    try {
        const steps = proveSteps(key, (wtns instanceof Uint8Array || wtns instanceof PlonkWitness) ? wtns : new Uint8Array(wtns), blindingMont, P, track);
        for (let s = steps.next(); ; s = steps.next(serve(key, s.value))) if (s.done) return s.value;
    } finally {
        polys.forEach((p) => p.free());
        if (!(zkey instanceof PlonkKey)) key.release();
    }
}
// The reference's plonk16Prove is async (src/plonk_prove.js:47): here every wait of the round driver — the commitments of a round, the queued
// transforms before a read-back — runs on a libuv pool thread (addon.msmTableMultiDevAsync / synchronizeAsync); what is left on the main thread
// between two awaits only enqueues kernels (tens of microseconds per call). Same result as prove() for the same blinding values. Calls are
// serialised per process (one device context, pipeline slot 0): a second proveAsync waits for the first.
let asyncQueue = Promise.resolve();
function proveAsync(zkey, wtns, blindingMont = null, options = null) {
    const run = async () => {
        const key = zkey instanceof PlonkKey ? zkey : new PlonkKey(zkey, options);
        const polys = [];
        const P = (n, zero = true) => { const p = new Poly(key.f, n, zero); polys.push(p); return p; };
        const track = (p) => { polys.push(p); return p; };
        try {
            const steps = proveSteps(key, (wtns instanceof Uint8Array || wtns instanceof PlonkWitness) ? wtns : new Uint8Array(wtns), blindingMont, P, track);
            let s = steps.next();
            while (!s.done) {
                const req = s.value;
                const ans = req && req.commit ? await commitAsync(key, req.commit, 0) : await addon.synchronizeAsync(0);
                call("zkmi_pipeline_select", 0);
                s = steps.next(ans);
            }
            return s.value;
        } finally {
            polys.forEach((p) => p.free());
            if (!(zkey instanceof PlonkKey)) key.release();
        }
    };
    const p = asyncQueue.then(run, run);
    asyncQueue = p.catch(() => {});
    return p;
}

// Throughput mode: one proof per witness against one key, TWO in flight from this one thread. Every proof is a generator (proveSteps) that
// yields right before each of its long blocking calls (the commitment rounds, the divisibility check behind the T pipeline, the round-4
// evaluations); the driver switches the library's pipeline slot (zkmi_pipeline_select: own stream, scratch buffers, allocation pool) and lets
// the other proof enqueue up to ITS next blocking call first, so the GPU holds queued work of one proof while the host waits for the other.
// Results come back in input order and equal what prove() returns for the same blinding values.
function proveMany(zkey, wtnsList, blindingMonts = null, options = null) {
    const key = zkey instanceof PlonkKey ? zkey : new PlonkKey(zkey, options);
    const out = new Array(wtnsList.length), live = [], free = [0, 1];
    let nxt = 0;
    const finish = (ent) => { ent.polys.forEach((p) => p.free()); live.splice(live.indexOf(ent), 1); free.push(ent.slot); };
    try {
        while (nxt < wtnsList.length || live.length) {
            while (free.length && nxt < wtnsList.length) {
                const polys = [], w = wtnsList[nxt];
                const P = (n, zero = true) => { const p = new Poly(key.f, n, zero); polys.push(p); return p; };
                const track = (p) => { polys.push(p); return p; };
                live.push({ slot: free.shift(), idx: nxt, polys, pending: null, steps: proveSteps(key, (w instanceof Uint8Array || w instanceof PlonkWitness) ? w : new Uint8Array(w), blindingMonts ? blindingMonts[nxt] : null, P, track) });
                nxt++;
            }
            for (const ent of live.slice()) {
                call("zkmi_pipeline_select", ent.slot);
                // the blocking call this proof stopped in front of (a round's commitments were ENQUEUED when it stopped: collected here), then on to its next one
                const s = ent.steps.next(ent.pending ? commitCollect(key, ent.pending) : undefined);
                ent.pending = null;
                if (s.done) { out[ent.idx] = s.value; finish(ent); } else if (s.value && s.value.commit) ent.pending = commitEnqueue(key, s.value.commit);
            }
        }
    } finally {
        for (const ent of live.slice()) {                  // an error in one proof: drop the other one too, leave no queued work behind
            try { call("zkmi_pipeline_select", ent.slot); ent.steps.return(); call("zkmi_synchronize"); ent.polys.forEach((p) => p.free()); } catch (e) { /* already failing */ }
        }
        call("zkmi_pipeline_select", 0);
        if (!(zkey instanceof PlonkKey)) key.release();
    }
    return out;
}

// plonk.prove as a generator: `yield` stands right before every long blocking call; everything between two yields only enqueues work
function* proveSteps(key, wt, blindingMont, P, track) {
    const f = key.f, r = f.r, n = key.n, power = key.power;
    const own = !(wt instanceof PlonkWitness);
    const wres = own ? new PlonkWitness(key, wt) : wt;
    const { pub, dWit, nW } = wres;
    const b = [0n];
    for (let i = 0; i < 11; i++) b.push(blindingMont ? f.unmont(blindingMont[i]) : fromLE(crypto.randomBytes(64)) % r);
    try {
        const tr = new Transcript(f), pts = {}, evs = {};
        const wN = f.root(power), w4N = f.root(power + 2), w2 = f.root(2), mont = (v) => f.mont(v);

        // ---- ROUND 1 (:222-313)
        // calculateAdditions (:174-204): the internal signals, ONE launch on the device into this proof's own buffer (two proofs may be in flight)
        const internal = P(Math.max(key.nAdditions, 1), false), dInt = internal.ptr;
        if (key.nAdditions) call("zkmi_plonk_additions_dev", f.cid, key.sec(3), key.nAdditions, dWit, nW, dInt);
        const A = P(n, false), B = P(n, false), Cw = P(n, false);
        call("zkmi_plonk_gather_wires_mont_dev", f.cid, dWit, nW, dInt, key.nAdditions, key.sec(4), key.sec(5), key.sec(6), key.nConstraints, n, A.ptr, B.ptr, Cw.ptr);   // buffers + batchToMontgomery in one pass
        const [pA, eA] = A.ifftBlinded([b[2], b[1]]).map(track), [pB, eB] = B.ifftBlinded([b[4], b[3]]).map(track), [pC, eC] = Cw.ifftBlinded([b[6], b[5]]).map(track);
        [pts.A, pts.B, pts.C] = yield { commit: [pA, pB, pC] };

        // ---- ROUND 2 (:315-455)
        tr.reset();
        for (const nm of ["Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"]) tr.point(key.commit[nm]);
        for (let i = 0; i < key.nPublic; i++) tr.scalar(A.get(i));
        for (const nm of ["A", "B", "C"]) tr.point(pts[nm]);
        const beta = tr.challenge();
        tr.reset(); tr.scalar(beta);
        const gamma = tr.challenge();
        const Zb = P(n, false);
        call("zkmi_plonk_compute_z_enqueue", f.cid, A.ptr, B.ptr, Cw.ptr, key.sec(12, n), key.sec(12, 6 * n), key.sec(12, 11 * n), n, mont(beta), mont(gamma), mont(key.k1), mont(key.k2), wN, Zb.ptr);
        const [pZ, eZ] = Zb.ifftBlinded([b[9], b[8], b[7]]).map(track);
        [pts.Z] = yield { commit: [pZ] };
        if (Zb.get(0) !== 1n) throw new Error("Copy constraints does not match");                        // computeZ's check (:437-439), read behind the commitment's own wait

        // ---- ROUND 3 (:457-684)
        tr.reset(); tr.scalar(beta); tr.scalar(gamma); tr.point(pts.Z);
        const alpha = tr.challenge();
        const ev = new BigUint64Array([eA.ptr, eB.ptr, eC.ptr, eZ.ptr, key.sec(7, n), key.sec(8, n), key.sec(9, n), key.sec(10, n), key.sec(11, n),
                                       key.sec(12, n), key.sec(12, 6 * n), key.sec(12, 11 * n), key.sec(13), A.ptr].map(BigInt));      // zkmi_plonk_evals
        const T = P(4 * n, false), Tz = P(4 * n, false), blind = new Uint8Array(11 * 32);
        for (let i = 1; i <= 11; i++) blind.set(mont(b[i]), 32 * (i - 1));
        call("zkmi_plonk_compute_t_dev", f.cid, ev, n, key.nPublic, blind, mont(beta), mont(gamma), mont(alpha), mont(key.k1), mont(key.k2), wN, w4N, w2, T.ptr, Tz.ptr);
        const pT = T.ntt(true, T);
        call("zkmi_poly_div_zh_dev", f.cid, pT.ptr, 4 * n, n, 4);
        const pTz = Tz.ntt(true, Tz);
        pT.axpy(pTz);
        yield;
        if (!pT.tailIsZero(3 * n + 6)) throw new Error("T Polynomial is not well calculated");            // :645-647
        const T1 = P(n + 1, false), T2 = P(n + 1, false), T3 = P(n + 6, false);
        call("zkmi_plonk_split_t_dev", f.cid, pT.ptr, 4 * n, n, mont(b[10]), mont(b[11]), T1.ptr, T2.ptr, T3.ptr);                                     // :649-672 in one launch
        [pts.T1, pts.T2, pts.T3] = yield { commit: [T1, T2, T3] };

        // ---- ROUND 4 (:686-708)
        tr.reset(); tr.scalar(alpha);
        for (const nm of ["T1", "T2", "T3"]) tr.point(pts[nm]);
        const xi = tr.challenge(), wv = f.unmont(wN), xiw = xi * wv % r;
        const S1 = key.sec(12, 0), S2 = key.sec(12, 5 * n), S3 = key.sec(12, 10 * n);                      // the coefficient halves of the sigma section, read where they lie
        yield;
        [evs.eval_a, evs.eval_b, evs.eval_c, evs.eval_s1, evs.eval_s2, evs.eval_zw] =
            evaluateMany(f, [[pA.ptr, pA.n], [pB.ptr, pB.n], [pC.ptr, pC.n], [S1, n], [S2, n], [pZ.ptr, pZ.n]], [xi, xi, xi, xi, xi, xiw]);               // six evaluations, one wait

        // ---- ROUND 5 (:710-888)
        tr.reset(); tr.scalar(xi);
        for (const k of ["eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"]) tr.scalar(evs[k]);
        const v = [0n, tr.challenge()];
        for (let i = 2; i < 6; i++) v.push(v[i - 1] * v[1] % r);
        const xin = modpow(xi, BigInt(n), r), zh = mod(xin - 1n, r), N = BigInt(n);
        const Lg = [0n];
        let ww = 1n;
        for (let i = 1; i <= Math.max(1, key.nPublic); i++) { Lg.push(ww * zh % r * modinv(N * mod(xi - ww, r) % r, r) % r); ww = ww * wv % r; }
        const evalL1 = mod(xin - 1n, r) * modinv(N * mod(xi - 1n, r) % r, r) % r;
        let evalPi = 0n;
        pub.forEach((p, i) => { evalPi = mod(evalPi - p * Lg[i + 1], r); });
        const { eval_a: ea, eval_b: eb, eval_c: ec, eval_s1: es1, eval_s2: es2, eval_zw: ezw } = evs;
        const alpha2 = alpha * alpha % r, betaxi = beta * xi % r;
        const e2 = (ea + betaxi + gamma) * (eb + betaxi * key.k1 + gamma) % r * (ec + betaxi * key.k2 + gamma) % r * alpha % r;
        const e3 = (ea + beta * es1 + gamma) * (eb + beta * es2 + gamma) % r * ezw % r * alpha % r;
        const e4 = evalL1 * alpha2 % r;
        // The linearisation polynomial R (:769-838) and the opening numerator Wxi = R + v1 (A - a) + ... (:840-866) are ONE linear combination of fifteen resident polynomials:
        // a single launch (zkmi_poly_lincomb_dev) instead of 18 add / sub, 2 mulScalar and 5 copies of selector polynomials; exact arithmetic, same coefficients
        const zhN = mod(-zh, r), xin2 = xin * xin % r;
        const r0 = mod(evalPi - e3 * mod(ec + gamma, r) - e4, r);
        const Wxi = lincomb(f, P(n + 6, false), [
            [key.sec(7, 0), n, ea * eb % r], [key.sec(8, 0), n, ea], [key.sec(9, 0), n, eb], [key.sec(10, 0), n, ec], [key.sec(11, 0), n, null],
            [pZ.ptr, pZ.n, (e2 + e4) % r], [S3, n, mod(-(e3 * beta), r)],
            [T1.ptr, T1.n, zhN], [T2.ptr, T2.n, zhN * xin % r], [T3.ptr, T3.n, zhN * xin2 % r],
            [pA.ptr, pA.n, v[1]], [pB.ptr, pB.n, v[2]], [pC.ptr, pC.n, v[3]], [S1, n, v[4]], [S2, n, v[5]]],
            mod(r0 - (v[1] * ea + v[2] * eb + v[3] * ec + v[4] * es1 + v[5] * es2), r));
        call("zkmi_poly_div_by_zerofier_enqueue", f.cid, Wxi.ptr, Wxi.n, 1, mont(xi));
        const Wxiw = lincomb(f, P(pZ.n, false), [[pZ.ptr, pZ.n, null]], mod(-ezw, r));
        call("zkmi_poly_div_by_zerofier_enqueue", f.cid, Wxiw.ptr, Wxiw.n, 1, mont(xiw));
        [pts.Wxi, pts.Wxiw] = yield { commit: [Wxi, Wxiw] };
        if (!(Wxi.tailIsZero(Wxi.n - 1) && Wxiw.tailIsZero(Wxiw.n - 1))) throw new Error("Polynomial is not divisible");   // divByZerofier's test (polynomial.js:665-669), read behind the commitments' wait

        const proof = {};
        for (const nm of ["A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw"]) proof[nm] = [pts[nm][0].toString(), pts[nm][1].toString(), "1"];   // src/proof.js:61-83
        for (const k of ["eval_a", "eval_b", "eval_c", "eval_s1", "eval_s2", "eval_zw"]) proof[k] = evs[k].toString();
        proof.protocol = "plonk";
        proof.curve = key.curveName;
        return { proof, publicSignals: pub.map((p) => p.toString()) };
    } finally {
        if (own) wres.release();
    }
}

// A witness resident on the device for any number of proofs against `key` (the reference reads the .wtns file once per proof,
// src/plonk_prove.js:83-97): header checks, public signals, calculateAdditions (:174-204, sequential on the host), signals and internal signals
// uploaded once; read-only afterwards, so proofs on both pipeline slots share it.
class PlonkWitness {
    constructor(key, wtIn) {
        const wt = wtIn instanceof Uint8Array ? wtIn : new Uint8Array(wtIn);
        const f = key.f, r = f.r;
        const { dv, s: ws } = readSections(wt);
        const n8 = dv.getUint32(ws[1][0], true), wq = fromLE(wt.subarray(ws[1][0] + 4, ws[1][0] + 4 + n8)), nWitness = dv.getUint32(ws[1][0] + 4 + n8, true);
        if (key.r !== wq) throw new Error("Curve of the witness does not match the curve of the proving key");
        if (nWitness !== key.nVars - key.nAdditions) throw new Error(`Invalid witness length. Circuit: ${key.nVars}, witness: ${nWitness}, ${key.nAdditions}`);
        if (ws[2][1] < nWitness * 32 || ws[2][0] + nWitness * 32 > wt.length) throw new Error("Invalid witness length: the wtns data section is shorter than its header says");
        const wit = wt.slice(ws[2][0], ws[2][0] + nWitness * 32);
        this.pub = [];
        for (let i = 1; i <= key.nPublic; i++) this.pub.push(fromLE(wit.subarray(32 * i, 32 * i + 32)));
        wit.fill(0, 0, 32);                                                                                 // :94-96
        // calculateAdditions (src/plonk_prove.js:174-204) is per-proof work of the reference: it runs on the device at the start of every proof
        // (zkmi_plonk_additions_dev in proveSteps), not here
        this.nW = key.nVars - key.nAdditions;
        this.dWit = devFrom(wit);
    }
    release() { if (this.dWit) { devFree(this.dWit); this.dWit = 0; } }
}

module.exports = { prove, proveAsync, proveMany, PlonkKey, PlonkWitness,
                   _internals: { addon, call, bindDevice, Field, Poly, lincomb, evaluateMany, Transcript, readSections, devAlloc, devFree, devFrom, mod, modinv, modpow, toLE, fromLE, Q_BN, Q_BLS } };
