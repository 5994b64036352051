// oracle/gen_golden.js — TEST INFRASTRUCTURE ONLY.
//
// Runs the REAL reference (bundle evaluated by ref_shim.js) in this container and writes the golden vectors that
// tests/ compare both the C restatement (oracle/*.c) and the HIP path against.  The reference ships no MSM/NTT
// golden vectors and no golden proofs (SURVEY.md §8c), so these are "outputs of the reference itself run here".
// Regenerate with:   make -C oracle golden        (needs /root/reference; ≈1–2 min on 8 cores)
//
// Synthetic inputs are defined so that Python (tests/synth.py) regenerates them bit-identically:
//   word(seed,k) = fmix32(seed + k*0x9E3779B9)   (murmur3 finaliser), little-endian u32 stream;
//   field/scalar element i = words 8i..8i+7, top byte &= 0x1f  (uniform 253-bit, < r on both curves).
'use strict';
const fs = require('fs'), path = require('path'), crypto = require('crypto');
const snarkjs = require('./ref_shim.js');
const OUT = path.join(__dirname, '..', 'tests', 'golden');
const sha = b => crypto.createHash('sha256').update(b).digest('hex');
const hex = b => Buffer.from(b).toString('hex');

function fmix32(h) {
    h ^= h >>> 16; h = Math.imul(h, 0x85ebca6b); h ^= h >>> 13; h = Math.imul(h, 0xc2b2ae35); h ^= h >>> 16;
    return h >>> 0;
}
function synthElems(seed, n, maskTop = 0x1f) {          // n × 32 B
    const out = new Uint8Array(n * 32), dv = new DataView(out.buffer);
    for (let k = 0; k < n * 8; k++) dv.setUint32(4 * k, fmix32((seed + Math.imul(k, 0x9E3779B9)) >>> 0), true);
    for (let i = 0; i < n; i++) out[32 * i + 31] &= maskTop;
    return out;
}
function iota(n) {                                       // element i = integer (i+1), little-endian, as-is
    const x = new Uint8Array(n * 32);
    for (let i = 0; i < n; i++) { const v = i + 1; x[i * 32] = v & 255; x[i * 32 + 1] = (v >> 8) & 255; x[i * 32 + 2] = (v >> 16) & 255; }
    return x;
}
// witness-like mix (SURVEY §8d): 60 % {0,1}, 30 % 64-bit, 10 % full width — selector from word stream seed^0xabcdef
function synthWitnessLike(seed, n) {
    const full = synthElems(seed, n);
    for (let i = 0; i < n; i++) {
        const sel = fmix32(((seed ^ 0xabcdef) + Math.imul(i, 0x9E3779B9)) >>> 0) % 100;
        if (sel < 60) { const b = full[32 * i] & 1; full.fill(0, 32 * i, 32 * i + 32); full[32 * i] = b; }
        else if (sel < 90) full.fill(0, 32 * i + 8, 32 * i + 32);
    }
    return full;
}

async function geomBases(curve, G, n) {                  // P_i = 7·11^i·G, affine Montgomery (SURVEY Appendix C.1)
    const sG = G.F.n8 * 2, buf = new Uint8Array(n * sG), g = G.toAffine(G.g);
    for (let i = 0; i < n; i++) buf.set(g, i * sG);
    return await G.batchApplyKey(buf, curve.Fr.e(7), curve.Fr.e(11));
}

async function kernelVectors(name, tag) {
    const curve = await snarkjs.curves.getCurveFromName(name);
    const { Fr, G1, G2 } = curve, res = { curve: name, n8q: G1.F.n8, n8r: Fr.n8 };
    res.Fr_one = hex(Fr.one); res.Fq_one = hex(G1.F.one); res.s = Fr.s; res.nqr = Fr.toString(Fr.nqr);
    res.w_s = Fr.toString(Fr.w[Fr.s]); res.shift = Fr.toString(Fr.shift);
    res.w = []; for (let i = 0; i <= Fr.s; i++) res.w.push(hex(Fr.w[i]));
    res.r = Fr.p.toString(); res.q = G1.F.p.toString();
    res.G1_g = hex(G1.toAffine(G1.g)); res.G2_g = hex(G2.toAffine(G2.g));
    const save = (f, b) => fs.writeFileSync(path.join(OUT, `${tag}_${f}.bin`), b);

    // ---- n = 1024, iota input: raw files (SURVEY Appendix C.1) ----
    const n = 1024, x = iota(n);
    const v = res.n1024 = {};
    const put = (k, b, raw = true) => { v[k] = sha(b); if (raw) save(`n1024_${k}`, b); };
    put('fft', await Fr.fft(x)); put('ifft', await Fr.ifft(x));
    put('applykey_7_11', await Fr.batchApplyKey(x, Fr.e(7), Fr.e(11)));
    put('to_mont', await Fr.batchToMontgomery(x)); put('from_mont', await Fr.batchFromMontgomery(x));
    put('inverse', await Fr.batchInverse(x));
    const b1 = await geomBases(curve, G1, n), b2 = await geomBases(curve, G2, n);
    put('g1_bases', b1); put('g2_bases', b2);
    put('g1_msm_affine', G1.toAffine(await G1.multiExpAffine(b1, x)));
    put('g2_msm_affine', G2.toAffine(await G2.multiExpAffine(b2, x)));

    // ---- NTT over synthetic inputs, several sizes (hash only; inputs regenerated from the seed) ----
    res.ntt = {};
    for (const lg of [0, 1, 2, 3, 5, 8, 11, 12, 14, 16, 17]) {
        const xs = synthElems(0x1000 + lg, 1 << lg);
        const f = await Fr.fft(xs), fi = await Fr.ifft(xs);
        res.ntt[lg] = { seed: 0x1000 + lg, fft: sha(f), ifft: sha(fi), fft_first: hex(f.slice(0, 32)), ifft_last: hex(fi.slice(fi.length - 32)) };
    }
    // Groth16's coset chain  ifft → batchApplyKey(1, w[lg+1]) → fft   (src/groth16_prove.js:64-76)
    res.coset_chain = {};
    for (const lg of [4, 10, 13]) {
        const xs = synthElems(0x2000 + lg, 1 << lg);
        const a = await Fr.ifft(xs), b = await Fr.batchApplyKey(a, Fr.e(1), Fr.w[lg + 1]), c = await Fr.fft(b);
        res.coset_chain[lg] = { seed: 0x2000 + lg, out: sha(c) };
    }
    // batch ops on synthetic input incl. zero elements (batchInverse: 0 ↦ 0)
    {
        const xs = synthElems(0x3000, 4096); xs.fill(0, 32 * 5, 32 * 6); xs.fill(0, 32 * 4095, 32 * 4096);
        res.batch = { seed: 0x3000, n: 4096, zeroed: [5, 4095], inverse: sha(await Fr.batchInverse(xs)),
            to_mont: sha(await Fr.batchToMontgomery(xs)), from_mont: sha(await Fr.batchFromMontgomery(xs)),
            applykey_shift: sha(await Fr.batchApplyKey(xs, Fr.e(3), Fr.shift)) };
    }

    // ---- MSM: sizes, scalar widths, edge cases (results hashed after toAffine; zero = all-zero bytes) ----
    res.msm = {};
    const msmCase = async (key, G, bases, scalars, extra) => {
        const r = G.toAffine(await G.multiExpAffine(bases, scalars));
        res.msm[key] = Object.assign({ affine: hex(r) }, extra || {});
    };
    const NB = 1 << 14, B1 = await geomBases(curve, G1, NB), B2 = await geomBases(curve, G2, 1 << 12);
    res.g1_bases_16384 = sha(B1); res.g2_bases_4096 = sha(B2);
    const s1 = G1.F.n8 * 2, s2 = G2.F.n8 * 2;
    for (const k of [1, 2, 3, 63, 64, 65, 1000, 4096, 16384])
        await msmCase(`g1_uniform_${k}`, G1, B1.slice(0, k * s1), synthElems(0x4000 + k, k), { seed: 0x4000 + k, n: k });
    for (const k of [1, 2, 100, 4096])
        await msmCase(`g2_uniform_${k}`, G2, B2.slice(0, k * s2), synthElems(0x5000 + k, k), { seed: 0x5000 + k, n: k });
    await msmCase('g1_witnesslike_16384', G1, B1, synthWitnessLike(0x6000, NB), { seed: 0x6000, n: NB });
    await msmCase('g2_witnesslike_4096', G2, B2, synthWitnessLike(0x6001, 4096), { seed: 0x6001, n: 4096 });
    // scalars ≥ r are NOT reduced by the reference (full 256-bit, no top mask)
    await msmCase('g1_full256_2048', G1, B1.slice(0, 2048 * s1), synthElems(0x7000, 2048, 0xff), { seed: 0x7000, n: 2048, mask: 0xff });
    { const sc = new Uint8Array(64 * 32).fill(0xff); await msmCase('g1_allff_64', G1, B1.slice(0, 64 * s1), sc, { n: 64 }); }
    // 4-byte scalars (src/powersoftau_verify.js:371 uses them)
    { const w = synthElems(0x7100, 128); await msmCase('g1_scalar4B_1024', G1, B1.slice(0, 1024 * s1), w.slice(0, 4096), { seed: 0x7100, n: 1024, scalar_bytes: 4 }); }
    // zero scalars, zero (infinity = all-zero bytes) bases, repeated bases (forces P+P doubling inside a bucket), P + (−P)
    { const sc = synthElems(0x7200, 1024); for (let i = 0; i < 1024; i += 3) sc.fill(0, 32 * i, 32 * i + 32);
      const bz = B1.slice(0, 1024 * s1); for (let i = 1; i < 1024; i += 5) bz.fill(0, i * s1, (i + 1) * s1);
      await msmCase('g1_zeros_1024', G1, bz, sc, { seed: 0x7200, n: 1024, zero_scalar_stride: 3, zero_base_from1_stride: 5 }); }
    { const bz = new Uint8Array(512 * s1); for (let i = 0; i < 512; i++) bz.set(B1.slice((i & 3) * s1, ((i & 3) + 1) * s1), i * s1);
      const sc = new Uint8Array(512 * 32); for (let i = 0; i < 512; i++) { sc[32 * i] = 5; sc[32 * i + 2] = i >> 6; }
      await msmCase('g1_repeated_512', G1, bz, sc, { n: 512 }); }
    { const bz = new Uint8Array(2 * s1); bz.set(B1.slice(0, s1), 0); bz.set(G1.toAffine(G1.neg(B1.slice(0, s1))), s1);
      const sc = new Uint8Array(64); sc[0] = 9; sc[32] = 9; await msmCase('g1_cancel_2', G1, bz, sc, { n: 2 }); }
    await msmCase('g1_allzero_scalars_100', G1, B1.slice(0, 100 * s1), new Uint8Array(3200), { n: 100 });
    res.msm.g1_empty = { affine: hex(G1.toAffine(await G1.multiExpAffine(new Uint8Array(0), new Uint8Array(0)))) };
    fs.writeFileSync(path.join(OUT, `${tag}_kernel_vectors.json`), JSON.stringify(res, null, 1));
    console.log(tag, 'kernel vectors done');
    return curve;
}

// Fully seeded in-memory Groth16 ceremony → zkey, witness, proof (SURVEY.md Appendix C.3); also records r, s and the
// bytes crossing every bulk-op boundary inside groth16.prove so that single stages can be replayed.
async function groth16Golden() {
    const curve = await snarkjs.curves.getCurveFromName('bn128');
    const T = '/root/reference/test/groth16/';
    const mem = () => ({ type: 'mem' });
    const p0 = mem(), p1 = mem(), pf = mem(), z0 = mem(), z1 = mem(), w = mem();
    await snarkjs.powersOfTau.newAccumulator(curve, 11, p0);
    await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
    await snarkjs.powersOfTau.preparePhase2(p1, pf);
    await snarkjs.zKey.newZKey(new Uint8Array(fs.readFileSync(T + 'circuit.r1cs')), pf, z0);
    await snarkjs.zKey.contribute(z0, z1, 'p2_C1', 'pa_Entropy1');
    await snarkjs.wtns.calculate({ a: 11, b: 2 }, new Uint8Array(fs.readFileSync(T + 'circuit.wasm')), w);
    const rnd = [], calls = [], Fr = curve.Fr;
    const origRandom = Fr.random.bind(Fr); Fr.random = () => { const v = origRandom(); rnd.push(hex(v)); return v; };
    const wrap = (obj, nm, label) => { const o = obj[nm].bind(obj); obj[nm] = async (...a) => { const r = await o(...a);
        calls.push({ op: label, in_len: a[0].byteLength, in0: sha(a[0].slice(0, a[0].byteLength)), in1: (a[1] && a[1].byteLength !== undefined && a[1].byteLength > 32) ? sha(a[1].slice(0, a[1].byteLength)) : undefined, out: sha(r.slice(0, r.byteLength)) }); return r; }; return o; };
    const undo = [[Fr, 'fft', wrap(Fr, 'fft', 'Fr.fft')], [Fr, 'ifft', wrap(Fr, 'ifft', 'Fr.ifft')], [Fr, 'batchApplyKey', wrap(Fr, 'batchApplyKey', 'Fr.batchApplyKey')],
        [curve.G1, 'multiExpAffine', wrap(curve.G1, 'multiExpAffine', 'G1.multiExpAffine')], [curve.G2, 'multiExpAffine', wrap(curve.G2, 'multiExpAffine', 'G2.multiExpAffine')]];
    const { proof, publicSignals } = await snarkjs.groth16.prove(z1.data, w.data);
    for (const [o, nm, f] of undo) o[nm] = f; Fr.random = origRandom;
    const vk = await snarkjs.zKey.exportVerificationKey(z1.data);
    const ok = await snarkjs.groth16.verify(vk, publicSignals, proof);
    if (!ok) throw new Error('golden groth16 proof does not verify');
    fs.writeFileSync(path.join(OUT, 'groth16_bn128_n1024.zkey'), z1.data);
    fs.writeFileSync(path.join(OUT, 'groth16_bn128_n1024.wtns'), w.data);
    fs.writeFileSync(path.join(OUT, 'groth16_bn128_n1024.json'), JSON.stringify({
        zkey_sha256: sha(z1.data), wtns_sha256: sha(w.data), proof_sha256: sha(JSON.stringify(proof)),
        r_mont: rnd[0], s_mont: rnd[1], n_random_calls: rnd.length, proof, publicSignals, verified: ok, vk, calls }, null, 1));
    console.log('groth16 golden done: proof sha', sha(JSON.stringify(proof)), 'verify', ok);
}

// The reference has no BLS12-381 fixture or test of any kind (SURVEY.md 8c/8d). Recipe of SURVEY.md 8d: the Multiplier(n) chain of
// test/groth16/circuit.circom written directly in the r1cs binary format with prime = BLS12-381 r (x_0 = a*a + b, x_i = x_{i-1}^2 + b;
// public output c = x_{n-1}, public input a), its witness from a BigInt loop, then the same seeded in-memory ceremony as above on
// getCurveFromName("bls12381"): powersOfTau -> newZKey -> contribute -> groth16.prove -> verify.
function multiplierR1cs(r, n) {
    const le = (v, k) => { const o = Buffer.alloc(k); let x = BigInt(v); for (let i = 0; i < k; i++) { o[i] = Number(x & 255n); x >>= 8n; } return o; };
    const u32 = (v) => le(v, 4), u64 = (v) => le(v, 8);
    const nWires = n + 3;                                      // [1, c = x_{n-1}, a, b, x_0 .. x_{n-2}]
    const wireOf = (i) => (i == n - 1) ? 1 : 4 + i;            // wire of x_i
    const lc = (terms) => Buffer.concat([u32(terms.length)].concat(terms.map(([w, v]) => Buffer.concat([u32(w), le(v, 32)]))));
    const cons = [];
    for (let i = 0; i < n; i++) {
        const prev = i == 0 ? 2 : wireOf(i - 1);
        cons.push(lc([[prev, 1n]]), lc([[prev, 1n]]), lc([[3, r - 1n], [wireOf(i), 1n]]));      // x_{i-1} * x_{i-1} = x_i - b
    }
    const hdr = Buffer.concat([u32(32), le(r, 32), u32(nWires), u32(1), u32(1), u32(1), u64(nWires), u32(n)]);
    const cs = Buffer.concat(cons), map = Buffer.concat(Array.from({ length: nWires }, (_, i) => u64(i)));
    const sec = (t, b) => Buffer.concat([u32(t), u64(b.length), b]);
    return new Uint8Array(Buffer.concat([Buffer.from('r1cs'), u32(1), u32(3), sec(1, hdr), sec(2, cs), sec(3, map)]));
}
function multiplierWtns(r, n, a, b) {
    const le = (v, k) => { const o = Buffer.alloc(k); let x = BigInt(v); for (let i = 0; i < k; i++) { o[i] = Number(x & 255n); x >>= 8n; } return o; };
    const xs = [(a * a + b) % r];
    for (let i = 1; i < n; i++) xs.push((xs[i - 1] * xs[i - 1] + b) % r);
    const sig = [1n, xs[n - 1], a, b].concat(xs.slice(0, n - 1));
    const hdrS = Buffer.concat([le(32, 4), le(r, 32), le(sig.length, 4)]), dataS = Buffer.concat(sig.map((v) => le(v, 32)));
    return new Uint8Array(Buffer.concat([Buffer.from('wtns'), le(2, 4), le(2, 4), le(1, 4), le(hdrS.length, 8), hdrS, le(2, 4), le(dataS.length, 8), dataS]));
}
async function groth16GoldenBls() {
    const curve = await snarkjs.curves.getCurveFromName('bls12381');
    const r = curve.Fr.p, n = 1000;
    const mem = () => ({ type: 'mem' });
    const p0 = mem(), p1 = mem(), pf = mem(), z0 = mem(), z1 = mem();
    await snarkjs.powersOfTau.newAccumulator(curve, 11, p0);
    await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
    await snarkjs.powersOfTau.preparePhase2(p1, pf);
    await snarkjs.zKey.newZKey(multiplierR1cs(r, n), pf, z0);
    await snarkjs.zKey.contribute(z0, z1, 'p2_C1', 'pa_Entropy1');
    const w = { data: multiplierWtns(r, n, 11n, 2n) };
    const rnd = [], Fr = curve.Fr;
    const origRandom = Fr.random.bind(Fr); Fr.random = () => { const v = origRandom(); rnd.push(hex(v)); return v; };
    const { proof, publicSignals } = await snarkjs.groth16.prove(z1.data, w.data);
    Fr.random = origRandom;
    const vk = await snarkjs.zKey.exportVerificationKey(z1.data);
    const ok = await snarkjs.groth16.verify(vk, publicSignals, proof);
    if (!ok) throw new Error('golden BLS12-381 groth16 proof does not verify');
    fs.writeFileSync(path.join(OUT, 'groth16_bls12381_n1024.zkey'), z1.data);
    fs.writeFileSync(path.join(OUT, 'groth16_bls12381_n1024.wtns'), w.data);
    fs.writeFileSync(path.join(OUT, 'groth16_bls12381_n1024.json'), JSON.stringify({
        zkey_sha256: sha(z1.data), wtns_sha256: sha(w.data), proof_sha256: sha(JSON.stringify(proof)),
        r_mont: rnd[0], s_mont: rnd[1], n_random_calls: rnd.length, proof, publicSignals, verified: ok, vk }, null, 1));
    console.log('groth16 BLS12-381 golden done: zkey', z1.data.length, 'bytes, proof sha', sha(JSON.stringify(proof)), 'verify', ok);
}

// Group-element FFTs and G.batchApplyKey (SURVEY.md 8 f4, ceremony side): G.fft / G.ifft / G.lagrangeEvaluations over the first n points
// of the geometric base table and G.batchApplyKey with non-trivial first / inc; raw outputs for the small sizes, hashes otherwise.
async function groupVectors(name, tag) {
    const curve = await snarkjs.curves.getCurveFromName(name);
    const { Fr, G1, G2 } = curve, res = { curve: name };
    for (const [gn, G, n] of [['g1', G1, 256], ['g2', G2, 64]]) {
        const sG = G.F.n8 * 2, bases = (await geomBases(curve, G, n));
        const f = await G.fft(bases, 'affine', 'affine'), fi = await G.ifft(bases, 'affine', 'affine'), le = await G.lagrangeEvaluations(bases, 'affine', 'affine');
        const ak = await G.batchApplyKey(bases, Fr.e(3), Fr.e(5));
        fs.writeFileSync(path.join(OUT, `${tag}_gfft_${gn}_n${n}_fft.bin`), f);
        fs.writeFileSync(path.join(OUT, `${tag}_gfft_${gn}_n${n}_ifft.bin`), fi);
        // with a point at infinity and a repeated point inside
        const b2 = bases.slice(0, bases.byteLength); b2.fill(0, 5 * sG, 6 * sG); b2.set(bases.slice(0, sG), 9 * sG);
        res[gn] = { n, bases_sha: sha(bases), fft: sha(f), ifft: sha(fi), lagrange: sha(le), lagrange_equals_ifft: sha(le) === sha(fi), applykey_3_5: sha(ak),
                    fft_with_zero_and_repeat: sha(await G.fft(b2, 'affine', 'affine')) };
        for (const lg of [0, 1, 2, 5]) { const k = 1 << lg; res[gn]['fft_n' + k] = sha(await G.fft(bases.slice(0, k * sG), 'affine', 'affine')); res[gn]['ifft_n' + k] = sha(await G.ifft(bases.slice(0, k * sG), 'affine', 'affine')); }
    }
    fs.writeFileSync(path.join(OUT, `${tag}_group_vectors.json`), JSON.stringify(res, null, 1));
    console.log(tag, 'group vectors done');
}

// Point-format conversions of the ceremony files (SURVEY 8 f4: G.batchLEMtoU / batchUtoLEM / batchLEMtoC / batchCtoLEM, callers
// src/powersoftau_import.js:159, src/powersoftau_contribute.js:145,176, src/mpc_applykey.js:64-70, src/zkey_export_bellman.js:36-83):
// bases P_i = 7*11^i*G with a point at infinity inside; the U and C byte strings the reference writes, and its round trips.
async function convertVectors(name, tag) {
    const curve = await snarkjs.curves.getCurveFromName(name), res = { curve: name };
    for (const [gn, G, n] of [['g1', curve.G1, 96], ['g2', curve.G2, 48]]) {
        const sG = G.F.n8 * 2, bases = await geomBases(curve, G, n);
        bases.fill(0, 7 * sG, 8 * sG); bases.fill(0, (n - 1) * sG, n * sG);
        const U = await G.batchLEMtoU(bases), C = await G.batchLEMtoC(bases);
        const backU = await G.batchUtoLEM(U), backC = await G.batchCtoLEM(C);
        fs.writeFileSync(path.join(OUT, `${tag}_conv_${gn}_n${n}_lem.bin`), bases);
        fs.writeFileSync(path.join(OUT, `${tag}_conv_${gn}_n${n}_u.bin`), U);
        fs.writeFileSync(path.join(OUT, `${tag}_conv_${gn}_n${n}_c.bin`), C);
        res[gn] = { n, lem: sha(bases), u: sha(U), c: sha(C), u_roundtrip: sha(backU) === sha(bases), c_roundtrip: sha(backC) === sha(bases),
                    u_bytes: U.byteLength, c_bytes: C.byteLength };
    }
    fs.writeFileSync(path.join(OUT, `${tag}_conv_vectors.json`), JSON.stringify(res, null, 1));
    console.log(tag, 'conversion vectors done', JSON.stringify(res));
}

// Seeded PLONK fixtures: plonk.setup on two circuits of the reference's test tree with the same seeded ptau, then a seeded
// plonk.prove whose 11 blinding draws (src/plonk_prove.js:224-227) and Fiat-Shamir challenges are recorded.
async function plonkGolden() {
    const curve = await snarkjs.curves.getCurveFromName('bn128');
    const mem = () => ({ type: 'mem' });
    const p0 = mem(), p1 = mem(), pf = mem();
    await snarkjs.powersOfTau.newAccumulator(curve, 12, p0);
    await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
    await snarkjs.powersOfTau.preparePhase2(p1, pf);
    const cases = [['plonk_bn128_small', '/root/reference/test/plonk_circuit/', JSON.parse(fs.readFileSync('/root/reference/test/plonk_circuit/input.json'))],
                   ['plonk_bn128_n2048', '/root/reference/test/groth16/', { a: 11, b: 2 }]];
    for (const [tag, T, input] of cases) {
        const z = mem(), w = mem();
        await snarkjs.plonk.setup(new Uint8Array(fs.readFileSync(T + 'circuit.r1cs')), pf, z);
        await snarkjs.wtns.calculate(input, new Uint8Array(fs.readFileSync(T + 'circuit.wasm')), w);
        const rnd = [], Fr = curve.Fr, census = {};
        const origRandom = Fr.random.bind(Fr); Fr.random = () => { const v = origRandom(); rnd.push(hex(v)); return v; };
        const undo = [];
        for (const [obj, nm] of [[Fr, 'fft'], [Fr, 'ifft'], [Fr, 'batchToMontgomery'], [Fr, 'batchFromMontgomery'], [Fr, 'batchInverse'], [curve.G1, 'multiExpAffine']]) {
            const o = obj[nm]; undo.push([obj, nm, o]);
            obj[nm] = async function (...a) { census[nm] = (census[nm] || 0) + 1; return o.apply(obj, a); };
        }
        const { proof, publicSignals } = await snarkjs.plonk.prove(z.data, w.data);
        for (const [o, nm, f] of undo) o[nm] = f; Fr.random = origRandom;
        const vk = await snarkjs.zKey.exportVerificationKey(z.data);
        // the verifier's own intermediate values (src/plonk_verify.js:62-105 logs them at debug level): they pin the restatement
        // oracle/plonk_verify_oracle.py up to, but excluding, the final pairing
        const verify_trace = [];
        const vlog = { debug: (m) => verify_trace.push(m), info() {}, warn() {}, error() {} };
        const ok = await snarkjs.plonk.verify(vk, publicSignals, proof, vlog);
        if (!ok) throw new Error('golden plonk proof does not verify');
        fs.writeFileSync(path.join(OUT, `${tag}.zkey`), z.data);
        fs.writeFileSync(path.join(OUT, `${tag}.wtns`), w.data);
        fs.writeFileSync(path.join(OUT, `${tag}.json`), JSON.stringify({
            zkey_sha256: sha(z.data), wtns_sha256: sha(w.data), proof_sha256: sha(JSON.stringify(proof)), blinding_mont: rnd, proof, publicSignals,
            verified: ok, vk, verify_trace, census }, null, 1));
        console.log(tag, 'plonk golden done: zkey', z.data.length, 'bytes, proof sha', sha(JSON.stringify(proof)), 'verify', ok);
    }
}

// The reference proves PLONK on whatever curve the zkey names (src/plonk_prove.js:66-75, src/curves.js:36-53) but ships no BLS12-381 fixture:
// plonk.setup over a seeded BLS12-381 ptau on the Multiplier(n) r1cs of the recipe above (SURVEY.md 8d), a seeded plonk.prove with its 11
// blinding draws recorded, the reference's own verifier as the judge.
async function plonkGoldenBls() {
    const curve = await snarkjs.curves.getCurveFromName('bls12381');
    const r = curve.Fr.p, n = 40;
    const mem = () => ({ type: 'mem' });
    const p0 = mem(), p1 = mem(), pf = mem(), z = mem();
    await snarkjs.powersOfTau.newAccumulator(curve, 8, p0);
    await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
    await snarkjs.powersOfTau.preparePhase2(p1, pf);
    await snarkjs.plonk.setup(multiplierR1cs(r, n), pf, z);
    const w = { data: multiplierWtns(r, n, 11n, 2n) };
    const rnd = [], Fr = curve.Fr;
    const origRandom = Fr.random.bind(Fr); Fr.random = () => { const v = origRandom(); rnd.push(hex(v)); return v; };
    const { proof, publicSignals } = await snarkjs.plonk.prove(z.data, w.data);
    Fr.random = origRandom;
    const vk = await snarkjs.zKey.exportVerificationKey(z.data);
    const verify_trace = [];
    const vlog = { debug: (m) => verify_trace.push(m), info() {}, warn() {}, error() {} };
    const ok = await snarkjs.plonk.verify(vk, publicSignals, proof, vlog);
    if (!ok) throw new Error('golden BLS12-381 plonk proof does not verify');
    const tag = 'plonk_bls12381_small';
    fs.writeFileSync(path.join(OUT, `${tag}.zkey`), z.data);
    fs.writeFileSync(path.join(OUT, `${tag}.wtns`), w.data);
    fs.writeFileSync(path.join(OUT, `${tag}.json`), JSON.stringify({
        zkey_sha256: sha(z.data), wtns_sha256: sha(w.data), proof_sha256: sha(JSON.stringify(proof)), blinding_mont: rnd, proof, publicSignals, verified: ok, vk, verify_trace }, null, 1));
    console.log(tag, 'plonk golden done: zkey', z.data.length, 'bytes, proof sha', sha(JSON.stringify(proof)), 'verify', ok);
}

// Seeded FFLONK fixtures: fflonk.setup on two circuits of the reference's test tree with a seeded ptau (only tauG1/tauG2 are read,
// src/fflonk_setup.js:430-438), then a seeded fflonk.prove whose 9 blinding draws (src/fflonk_prove.js:321-324) are recorded.
async function fflonkGolden() {
    const curve = await snarkjs.curves.getCurveFromName('bn128');
    const mem = () => ({ type: 'mem' });
    const p0 = mem(), p1 = mem(), pf = mem();
    await snarkjs.powersOfTau.newAccumulator(curve, 12, p0);
    await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
    await snarkjs.powersOfTau.preparePhase2(p1, pf);
    const cases = [['fflonk_bn128_small', '/root/reference/test/plonk_circuit/', JSON.parse(fs.readFileSync('/root/reference/test/plonk_circuit/input.json'))],
                   ['fflonk_bn128_n256', '/root/reference/test/fflonk/', JSON.parse(fs.readFileSync('/root/reference/test/fflonk/witness.json'))]];
    for (const [tag, T, input] of cases) {
        const z = mem(), w = mem();
        await snarkjs.fflonk.setup(new Uint8Array(fs.readFileSync(T + 'circuit.r1cs')), pf, z);
        if (tag === 'fflonk_bn128_n256') {
            // test/fflonk/circuit.wasm does not belong to circuit.r1cs (it yields 14 signals); the witness of Multiplier(100)
            // (circuit.circom) is computed here directly: signals = [1, c, a, b, int[0..98]], c = int[99]
            const r = curve.Fr.p, a = BigInt(input.a), b = BigInt(input.b), ints = [(a * a + b) % r];
            for (let i = 1; i < 100; i++) ints.push((ints[i - 1] * ints[i - 1] + b) % r);
            const sig = [1n, ints[99], a, b].concat(ints.slice(0, 99));
            const le = (v, n) => { const o = new Uint8Array(n); for (let i = 0; i < n; i++) { o[i] = Number(v & 255n); v >>= 8n; } return o; };
            const u32 = (v) => le(BigInt(v), 4), u64 = (v) => le(BigInt(v), 8);
            const hdrS = Buffer.concat([u32(32), le(r, 32), u32(sig.length)]), dataS = Buffer.concat(sig.map((v) => le(v, 32)));
            w.data = new Uint8Array(Buffer.concat([Buffer.from('wtns'), u32(2), u32(2), u32(1), u64(hdrS.length), hdrS, u32(2), u64(dataS.length), dataS]));
        } else await snarkjs.wtns.calculate(input, new Uint8Array(fs.readFileSync(T + 'circuit.wasm')), w);
        const rnd = [], Fr = curve.Fr, census = {};
        const origRandom = Fr.random.bind(Fr); Fr.random = () => { const v = origRandom(); rnd.push(hex(v)); return v; };
        const undo = [];
        for (const [obj, nm] of [[Fr, 'fft'], [Fr, 'ifft'], [Fr, 'batchToMontgomery'], [Fr, 'batchFromMontgomery'], [Fr, 'batchInverse'], [curve.G1, 'multiExpAffine']]) {
            const o = obj[nm]; undo.push([obj, nm, o]);
            obj[nm] = async function (...a) { census[nm] = (census[nm] || 0) + 1; return o.apply(obj, a); };
        }
        const { proof, publicSignals } = await snarkjs.fflonk.prove(z.data, w.data);
        for (const [o, nm, f] of undo) o[nm] = f; Fr.random = origRandom;
        const vk = await snarkjs.zKey.exportVerificationKey(z.data);
        // the two G1 arguments of the verifier's final pairing (src/fflonk_verify.js:529-541: pairingEq(-A1, [1]_2, W2, [x]_2)) and its
        // challenges pin the restatement oracle/fflonk_verify_oracle.py up to, but excluding, the pairing itself
        const verify_trace = [], pairing_inputs = [];
        const origPairingEq = curve.pairingEq.bind(curve);
        curve.pairingEq = async (...a) => { for (const k of [0, 2]) { const o = curve.G1.toObject(curve.G1.toAffine(a[k])); pairing_inputs.push([o[0].toString(), o[1].toString()]); } return origPairingEq(...a); };
        const vlog = { debug() {}, info: (m) => { if (/challenges\./.test(m)) verify_trace.push(m); }, warn() {}, error() {} };
        const ok = await snarkjs.fflonk.verify(vk, publicSignals, proof, vlog);
        curve.pairingEq = origPairingEq;
        if (!ok) throw new Error('golden fflonk proof does not verify');
        fs.writeFileSync(path.join(OUT, `${tag}.zkey`), z.data);
        fs.writeFileSync(path.join(OUT, `${tag}.wtns`), w.data);
        fs.writeFileSync(path.join(OUT, `${tag}.json`), JSON.stringify({
            zkey_sha256: sha(z.data), wtns_sha256: sha(w.data), proof_sha256: sha(JSON.stringify(proof)), blinding_mont: rnd, proof, publicSignals,
            verified: ok, vk, verify_trace, pairing_inputs, census }, null, 1));
        console.log(tag, 'fflonk golden done: zkey', z.data.length, 'bytes, proof sha', sha(JSON.stringify(proof)), 'verify', ok);
    }
}

// FFLONK on BLS12-381 — a NEGATIVE probe. The reference's prover takes the curve from the zkey (src/fflonk_prove.js:51-110), but its setup
// hard-codes BN254 constants: computeW3 raises the BN254 generator 31624 to BN254's (r - 1) / 3, getOmegaCubicRoot starts from a literal cube root
// of BN254's Fr.w[28] (src/fflonk_setup.js:533-556). On BLS12-381 fflonk.setup therefore writes a key whose w3 / wr are not roots of unity there,
// and fflonk.prove on it throws "Polynomial is not divisible" for a satisfied circuit: the reference has no FFLONK on BLS12-381, so there is
// nothing to be bit-identical to. This records the reference's own verdict (tests/golden/fflonk_bls12381_unsupported.json); the device path
// refuses such keys up front ("Curve not supported", snarkjs_amd/fflonk.py, js/fflonk_native.js).
async function fflonkBlsProbe() {
    const curve = await snarkjs.curves.getCurveFromName('bls12381');
    const r = curve.Fr.p, n = 40;
    const mem = () => ({ type: 'mem' });
    const p0 = mem(), p1 = mem(), pf = mem(), z = mem();
    await snarkjs.powersOfTau.newAccumulator(curve, 10, p0);
    await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
    await snarkjs.powersOfTau.preparePhase2(p1, pf);
    await snarkjs.fflonk.setup(multiplierR1cs(r, n), pf, z);
    const w = { data: multiplierWtns(r, n, 11n, 2n) };
    let error = null;
    try { await snarkjs.fflonk.prove(z.data, w.data); } catch (e) { error = String(e && e.message || e); }
    // the constants the setup wrote (zkey section 2 tail is parsed by the reference's own reader)
    const zk = await snarkjs.zKey.exportVerificationKey(z.data);
    const Fr = curve.Fr, cube = (x) => Fr.mul(Fr.mul(x, x), x);
    const hdr = { protocol: zk.protocol, curve: zk.curve, power: zk.power, w3: zk.w3 || null, w3_cubed_is_one: zk.w3 ? Fr.eq(cube(Fr.e(zk.w3)), Fr.one) : null,
                  wr: zk.wr || null, wr_cubed_is_w_power: zk.wr ? Fr.eq(cube(Fr.e(zk.wr)), Fr.w[zk.power]) : null };
    if (error === null) throw new Error('the reference proved FFLONK on BLS12-381: the probe\'s premise no longer holds, generate a positive fixture instead');
    fs.writeFileSync(path.join(OUT, 'fflonk_bls12381_unsupported.json'), JSON.stringify({
        what: 'fflonk.setup + fflonk.prove of the reference bundle on the Multiplier(40) r1cs over a seeded BLS12-381 ptau (oracle/gen_golden.js: fflonkBlsProbe)',
        reference_prove_error: error, zkey_sha256: sha(z.data), header: hdr,
        why: 'src/fflonk_setup.js:533-556 hard-codes BN254 constants (generator 31624 and exponent (r_bn254 - 1) / 3 in computeW3; a literal cube root of BN254 Fr.w[28] in getOmegaCubicRoot)' }, null, 1));
    console.log('fflonk on bls12381: the reference fails with', JSON.stringify(error));
}

(async () => {
    fs.mkdirSync(OUT, { recursive: true });
    const what = process.argv[2] || 'all';
    if (what === 'all' || what === 'bn128') await kernelVectors('bn128', 'bn128');
    if (what === 'all' || what === 'bls12381') await kernelVectors('bls12381', 'bls12381');
    if (what === 'all' || what === 'group') { await groupVectors('bn128', 'bn128'); await groupVectors('bls12381', 'bls12381'); }
    if (what === 'all' || what === 'conv') { await convertVectors('bn128', 'bn128'); await convertVectors('bls12381', 'bls12381'); }
    if (what === 'all' || what === 'groth16') await groth16Golden();
    if (what === 'all' || what === 'groth16bls') await groth16GoldenBls();
    if (what === 'all' || what === 'plonk') await plonkGolden();
    if (what === 'all' || what === 'plonkbls') await plonkGoldenBls();
    if (what === 'all' || what === 'fflonk') await fflonkGolden();
    if (what === 'all' || what === 'fflonkbls') await fflonkBlsProbe();
    process.exit(0);
})().catch(e => { console.error(e); process.exit(1); });
