// oracle/gen_replay.js — TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).
//
// Records every bulk call the REAL snarkjs makes through the eight curve methods that snarkjs_amd/js/register.js replaces
// (G1/G2.multiExpAffine, Fr.fft/ifft/batchApplyKey/batchToMontgomery/batchFromMontgomery/batchInverse), while it runs
//   * the seeded groth16.prove of tests/golden/groth16_bn128_n1024.{zkey,wtns}   (src/groth16_prove.js:64-101), and
//   * the seeded plonk.prove   of tests/golden/plonk_bn128_n2048.{zkey,wtns}     (src/plonk_prove.js:247-313 and on),
//   * a power-8 ceremony (powersOfTau new / contribute / preparePhase2) followed by plonk.setup and zKey.newZKey (setup-side callers),
// with the reference's own WASM implementation doing the work. Per call: method, argument containers (Uint8Array or
// ffjavascript BigBuffer) and bytes, result container and bytes (MSM results after toAffine: the Jacobian representative is
// implementation-defined). tests/js/register_replay.js feeds the same calls through register() + the REAL N-API addon on the
// GPU and compares — the reference bundle itself cannot travel to the GPU box.
//
// Bytes are stored once: a buffer that is a slice of the zkey / wtns fixture is recorded as {file, off, len}; one that equals an
// earlier result as {out: callIndex}; anything else goes to replay_<tag>.bin.
// Run:  make -C oracle replay     (≈20 s)
'use strict';
const fs = require('fs'), path = require('path'), crypto = require('crypto');
process.env.SINGLE = '1';
const snarkjs = require('./ref_shim.js');
const OUT = path.join(__dirname, '..', 'tests', 'golden');
const sha = b => crypto.createHash('sha256').update(b).digest('hex');
const hexb = s => new Uint8Array(Buffer.from(s, 'hex'));
const isBig = b => b && !(b instanceof Uint8Array) && Array.isArray(b.buffers);
const flat = b => (b instanceof Uint8Array) ? b : b.slice(0, b.byteLength);      // a BigBuffer of at most one page slices to a Uint8Array
const kind = b => (b instanceof Uint8Array) ? 'u8' : (isBig(b) ? 'big' : typeof b);

async function record(tag, curve, files, run) {
    const { Fr, G1, G2 } = curve;
    const calls = [], blobs = [], outs = new Map(), inputs = new Map();
    let blobLen = 0;
    const refIn = (buf) => {
        const b = Buffer.from(buf.buffer, buf.byteOffset, buf.byteLength), h = sha(b);
        if (b.length <= 64) return { hex: b.toString('hex') };                       // single elements (first / inc of batchApplyKey)
        if (outs.has(h)) return { out: outs.get(h), len: b.length, sha256: h };       // produced by an earlier recorded call
        if (inputs.has(h)) return Object.assign({}, inputs.get(h));
        for (const [name, data] of Object.entries(files)) { const at = data.indexOf(b); if (at >= 0) return { file: name, off: at, len: b.length, sha256: h }; }
        const r = { blob: blobLen, len: b.length, sha256: h };
        blobs.push(b); blobLen += b.length;
        inputs.set(h, r);
        return Object.assign({}, r);
    };
    const undo = [];
    const wrap = (obj, oname, nm, isMsm) => {
        const o = obj[nm]; undo.push([obj, nm, o]);
        obj[nm] = async function (...a) {
            const rec = { m: `${oname}.${nm}`, args: [] };
            for (const x of a) {
                if (x instanceof Uint8Array || isBig(x)) rec.args.push(Object.assign({ c: kind(x) }, refIn(flat(x))));
                else if (typeof x === 'string' && (x === 'affine' || x === 'jacobian')) rec.args.push({ s: x });               // inType / outType of the group FFTs
                else if (x === undefined || x === null || typeof x === 'string' || typeof x === 'object') rec.args.push(null);   // logger / log text
                else rec.args.push({ v: String(x) });
            }
            const r = await o.apply(obj, a);
            const rb = flat(r);
            rec.res = { c: kind(r), len: rb.byteLength, sha256: sha(rb) };
            if (isMsm) rec.res.affine = Buffer.from(obj.toAffine(rb)).toString('hex');
            else if (!outs.has(rec.res.sha256)) outs.set(rec.res.sha256, calls.length);
            calls.push(rec);
            return r;
        };
    };
    wrap(G1, 'G1', 'multiExpAffine', true); wrap(G2, 'G2', 'multiExpAffine', true);
    for (const nm of ['fft', 'ifft', 'batchApplyKey', 'batchToMontgomery', 'batchFromMontgomery', 'batchInverse']) wrap(Fr, 'Fr', nm, false);
    // ceremony / setup side (SURVEY 8 f3, f4): group FFTs, G.batchApplyKey and the point-format conversions
    for (const [G, gn] of [[G1, 'G1'], [G2, 'G2']]) for (const nm of ['fft', 'ifft', 'batchApplyKey', 'batchLEMtoU', 'batchUtoLEM', 'batchLEMtoC', 'batchCtoLEM']) wrap(G, gn, nm, false);
    const proof = await run();
    for (const [o, nm, f] of undo) o[nm] = f;
    return { tag, calls, blobs, proof_sha256: proof && proof.proof ? sha(JSON.stringify(proof.proof)) : null };
}

(async () => {
    const curve = await snarkjs.curves.getCurveFromName('bn128');
    const Fr = curve.Fr, realRandom = Fr.random;
    const out = { curve: 'bn128', n8q: 32, n8r: 32, G1_zero_len: curve.G1.zero.length, runs: [] };
    const allBlobs = [];
    let base = 0;
    const rd = (f) => fs.readFileSync(path.join(OUT, f));
    {
        const g = JSON.parse(rd('groth16_bn128_n1024.json'));
        const files = { 'groth16_bn128_n1024.zkey': rd('groth16_bn128_n1024.zkey'), 'groth16_bn128_n1024.wtns': rd('groth16_bn128_n1024.wtns') };
        const draws = [hexb(g.r_mont), hexb(g.s_mont)];
        Fr.random = () => draws.shift();
        const r = await record('groth16_bn128_n1024', curve, files, () => snarkjs.groth16.prove(new Uint8Array(files['groth16_bn128_n1024.zkey']), new Uint8Array(files['groth16_bn128_n1024.wtns'])));
        Fr.random = realRandom;
        if (r.proof_sha256 !== g.proof_sha256) throw new Error('recorded groth16 proof differs from the golden one');
        for (const c of r.calls) for (const a of c.args) if (a && a.blob !== undefined) a.blob += base;
        for (const b of r.blobs) { allBlobs.push(b); base += b.length; }
        out.runs.push({ tag: r.tag, proof_sha256: r.proof_sha256, calls: r.calls });
    }
    {
        const g = JSON.parse(rd('plonk_bn128_n2048.json'));
        const files = { 'plonk_bn128_n2048.zkey': rd('plonk_bn128_n2048.zkey'), 'plonk_bn128_n2048.wtns': rd('plonk_bn128_n2048.wtns') };
        const draws = g.blinding_mont.map(hexb);
        Fr.random = () => draws.shift();
        const r = await record('plonk_bn128_n2048', curve, files, () => snarkjs.plonk.prove(new Uint8Array(files['plonk_bn128_n2048.zkey']), new Uint8Array(files['plonk_bn128_n2048.wtns'])));
        Fr.random = realRandom;
        if (r.proof_sha256 !== g.proof_sha256) throw new Error('recorded plonk proof differs from the golden one');
        for (const c of r.calls) for (const a of c.args) if (a && a.blob !== undefined) a.blob += base;
        for (const b of r.blobs) { allBlobs.push(b); base += b.length; }
        out.runs.push({ tag: r.tag, proof_sha256: r.proof_sha256, calls: r.calls });
    }
    {   // setup side: a power-8 ceremony (new -> contribute -> preparePhase2), then plonk.setup and groth16 zKey.newZKey of the reference's
        // small PLONK test circuit over it (src/powersoftau_*.js, src/plonk_setup.js:323-403, src/zkey_new.js)
        const mem = () => ({ type: 'mem' });
        const r1cs = new Uint8Array(fs.readFileSync(path.join(snarkjs.refRoot, 'test/plonk_circuit/circuit.r1cs')));
        const r = await record('setup_bn128_p8', curve, {}, async () => {
            const p0 = mem(), p1 = mem(), pf = mem(), zp = mem(), zg = mem();
            await snarkjs.powersOfTau.newAccumulator(curve, 8, p0);
            await snarkjs.powersOfTau.contribute(p0, p1, 'C1', 'Entropy1');
            await snarkjs.powersOfTau.preparePhase2(p1, pf);
            await snarkjs.plonk.setup(r1cs, pf, zp);
            await snarkjs.zKey.newZKey(r1cs, pf, zg);
            return null;
        });
        for (const c of r.calls) for (const a of c.args) if (a && a.blob !== undefined) a.blob += base;
        for (const b of r.blobs) { allBlobs.push(b); base += b.length; }
        out.runs.push({ tag: r.tag, proof_sha256: null, calls: r.calls });
    }
    const blob = Buffer.concat(allBlobs);
    out.blob_sha256 = sha(blob);
    fs.writeFileSync(path.join(OUT, 'replay_bn128.bin'), blob);
    fs.writeFileSync(path.join(OUT, 'replay_bn128.json'), JSON.stringify(out));
    for (const r of out.runs) {
        const census = {};
        for (const c of r.calls) census[c.m] = (census[c.m] || 0) + 1;
        console.log(r.tag, JSON.stringify(census));
    }
    console.log('replay blob', blob.length, 'bytes');
    process.exit(0);
})().catch(e => { console.error(e); process.exit(1); });
