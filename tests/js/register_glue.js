// tests/js/register_glue.js — CPU-only check of snarkjs_amd/js/register.js against the REAL reference bundle.
// Needs the reference bundle (/root/reference in the build container, or the staged oracle/_ref/).  The HIP addon is replaced by a MOCK whose entry points call the
// reference's own (saved) WASM functions, so what is tested is exactly the glue: which calls snarkjs makes through the
// patched surface, container types (Uint8Array vs BigBuffer), argument conventions and that a seeded groth16 /
// plonk proof through the patched curve is byte-identical to the unpatched one.
// Run:  node --harmony-optional-chaining --harmony-nullish tests/js/register_glue.js
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
process.env.SINGLE = "1";
const snarkjs = require(path.join(__dirname, "..", "..", "oracle", "ref_shim.js"));
const { register, unregister } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "register.js"));
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
const GOLD = path.join(__dirname, "..", "golden");
let fails = 0;
function check(name, ok) { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); }

function cat(p) { if (p instanceof Uint8Array) return p; const n = p.reduce((a, b) => a + b.length, 0), o = new Uint8Array(n); let k = 0; for (const b of p) { o.set(b, k); k += b.length; } return o; }
function spread(src, dst) { if (dst instanceof Uint8Array) { dst.set(src); return; } let k = 0; for (const b of dst) { b.set(src.subarray(k, k + b.length)); k += b.length; } }

(async () => {
    const curve = await snarkjs.curves.getCurveFromName("bn128");
    const saved = { g1: curve.G1.multiExpAffine.bind(curve.G1), g2: curve.G2.multiExpAffine.bind(curve.G2), fft: curve.Fr.fft.bind(curve.Fr), ifft: curve.Fr.ifft.bind(curve.Fr),
                    gfft: { 1: [curve.G1.fft.bind(curve.G1), curve.G1.ifft.bind(curve.G1)], 2: [curve.G2.fft.bind(curve.G2), curve.G2.ifft.bind(curve.G2)] },
                    gak: { 1: curve.G1.batchApplyKey.bind(curve.G1), 2: curve.G2.batchApplyKey.bind(curve.G2) },
                    conv: { 1: ["batchLEMtoU", "batchUtoLEM", "batchLEMtoC", "batchCtoLEM"].map((nm) => curve.G1[nm].bind(curve.G1)), 2: ["batchLEMtoU", "batchUtoLEM", "batchLEMtoC", "batchCtoLEM"].map((nm) => curve.G2[nm].bind(curve.G2)) },
                    ak: curve.Fr.batchApplyKey.bind(curve.Fr), tm: curve.Fr.batchToMontgomery.bind(curve.Fr), fm: curve.Fr.batchFromMontgomery.bind(curve.Fr), inv: curve.Fr.batchInverse.bind(curve.Fr) };
    const calls = {};
    const note = (k) => { calls[k] = (calls[k] || 0) + 1; };
    // The mock is synchronous like the real addon, so it may only use the reference's SYNC element ops; bulk results are
    // produced by running the saved async WASM functions beforehand is not possible — instead the mock records the
    // request and the glue test drives it through `pending` promises resolved before the glue returns.
    // (the promise that fills an output rides on the output container as __p)
    const mock = {
        init() { note("init"); },
        msm(cid, group, bases, scalars, n, sb, key) { note("msm" + group); const out = new Uint8Array(96 * group); out.__p = (group == 1 ? saved.g1 : saved.g2)(cat(bases), cat(scalars)).then((r) => out.set(r)); return out; },
        ntt(cid, inp, out, logn, inverse) { note(inverse ? "ifft" : "fft"); out.__p = (inverse ? saved.ifft : saved.fft)(cat(inp)).then((r) => spread(r, out)); },
        groupFft(cid, group, inp, out, logn, inverse) { note("gfft" + group); out.__p = saved.gfft[group][inverse ? 1 : 0](cat(inp), "affine", "affine").then((r) => spread(r, out)); },
        groupApplyKey(cid, group, inp, out, n, first, inc) { note("gak" + group); out.__p = saved.gak[group](cat(inp), first, inc).then((r) => spread(r, out)); },
        groupConvert(cid, group, kind, inp, out, n) { note("conv" + group + kind); out.__p = saved.conv[group][kind](cat(inp)).then((r) => spread(r, out)); },
        applyKey(cid, inp, out, n, first, inc) { note("applyKey"); out.__p = saved.ak(cat(inp), first, inc).then((r) => spread(r, out)); },
        frBatch(cid, op, inp, out, n) { note("batch" + op); out.__p = [saved.tm, saved.fm, saved.inv][op](cat(inp)).then((r) => spread(r, out)); },
    };
    register(curve, { addon: mock });
    // the mock fills its outputs asynchronously: wrap every patched method so that it awaits that work
    for (const [obj, names] of [[curve.G1, ["multiExpAffine", "fft", "ifft", "batchApplyKey", "batchLEMtoU", "batchUtoLEM", "batchLEMtoC", "batchCtoLEM"]], [curve.G2, ["multiExpAffine", "fft", "ifft", "batchApplyKey", "batchLEMtoU", "batchUtoLEM", "batchLEMtoC", "batchCtoLEM"]], [curve.Fr, ["fft", "ifft", "batchApplyKey", "batchToMontgomery", "batchFromMontgomery", "batchInverse"]]]) {
        for (const nm of names) {
            const f = obj[nm];
            obj[nm] = async function () { const r = await f.apply(this, arguments); const c = (r instanceof Uint8Array) ? r : (r && r.buffers); if (c && c.__p) { await c.__p; delete c.__p; } return r; };
        }
    }

    // 1. container rule: Uint8Array in -> Uint8Array out, BigBuffer in -> BigBuffer out
    const x = new Uint8Array(1024 * 32); for (let i = 0; i < 1024; i++) x[32 * i] = i + 1;
    const y = await curve.Fr.fft(x);
    check("fft Uint8Array -> Uint8Array, bytes equal reference", y instanceof Uint8Array && sha(y) === sha(await saved.fft(x)));
    let threw = false;
    try { await curve.Fr.fft(new Uint8Array(96)); } catch (e) { threw = e.message === "fft must be multiple of 2"; }
    check("fft error message", threw);
    threw = false;
    try { await curve.G1.multiExpAffine(new Uint8Array(128), new Uint8Array(63)); } catch (e) { threw = e.message === "Scalar size does not match"; }
    check("multiExpAffine error message", threw);
    check("multiExpAffine empty -> G1.zero", (await curve.G1.multiExpAffine(new Uint8Array(0), new Uint8Array(0))) === curve.G1.zero);

    // 1b. ceremony-side group operations (SURVEY.md 8 f4): affine -> affine forms go to the addon, everything else to the WASM original
    {
        const gen = curve.G1.toAffine(curve.G1.g), pts = new Uint8Array(64 * 64);
        for (let i = 0; i < 64; i++) pts.set(gen, i * 64);
        const bases = await saved.gak[1](pts, curve.Fr.e(7), curve.Fr.e(11));
        for (const k of Object.keys(calls)) delete calls[k];
        check("G1.ifft affine -> affine through register.js", sha(await curve.G1.ifft(bases, "affine", "affine")) === sha(await saved.gfft[1][1](bases, "affine", "affine")) && calls.gfft1 === 1);
        check("G1.lagrangeEvaluations reaches the patched G1.ifft", sha(await curve.G1.lagrangeEvaluations(bases, "affine", "affine")) === sha(await saved.gfft[1][1](bases, "affine", "affine")) && calls.gfft1 === 2);
        check("G1.batchApplyKey through register.js", sha(await curve.G1.batchApplyKey(bases, curve.Fr.e(3), curve.Fr.e(5))) === sha(await saved.gak[1](bases, curve.Fr.e(3), curve.Fr.e(5))) && calls.gak1 === 1);
        const U = await curve.G1.batchLEMtoU(bases), Cc = await curve.G1.batchLEMtoC(bases);
        check("G1.batchLEMtoU / batchLEMtoC through register.js", U instanceof Uint8Array && sha(U) === sha(await saved.conv[1][0](bases)) && Cc.byteLength === 64 * 32 && sha(Cc) === sha(await saved.conv[1][2](bases)) && calls.conv10 === 1 && calls.conv12 === 1);
        check("G1.batchUtoLEM / batchCtoLEM through register.js", sha(await curve.G1.batchUtoLEM(U)) === sha(bases) && sha(await curve.G1.batchCtoLEM(Cc)) === sha(bases) && calls.conv11 === 1 && calls.conv13 === 1);
        let bad = false;
        try { await curve.G1.batchLEMtoU(new Uint8Array(65)); } catch (e) { bad = e.message === "Invalid buffer size"; }
        check("batchLEMtoU error message", bad);
        const before = calls.gfft1;
        const jac = await curve.G1.fft(bases, "affine", "jacobian");
        check("G1.fft affine -> jacobian falls through to the WASM original", jac.byteLength === 64 * 96 && calls.gfft1 === before);
    }

    // 2. seeded Groth16 proof through the patched surface == SURVEY.md Appendix C.3 / golden fixture
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.json")));
    const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.zkey")));
    const wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.wtns")));
    const hexb = (s) => new Uint8Array(Buffer.from(s, "hex"));
    const draws = [hexb(g.r_mont), hexb(g.s_mont)];
    const realRandom = curve.Fr.random;
    curve.Fr.random = () => draws.shift();
    for (const k of Object.keys(calls)) delete calls[k];
    const res = await snarkjs.groth16.prove(zkey, wtns);
    curve.Fr.random = realRandom;
    check("groth16.prove through register.js == reference proof", sha(JSON.stringify(res.proof)) === g.proof_sha256);
    check("bulk-op census (SURVEY.md Appendix B): 4 G1 + 1 G2 MSM, 3 ifft, 3 fft, 3 applyKey",
          calls.msm1 === 4 && calls.msm2 === 1 && calls.ifft === 3 && calls.fft === 3 && calls.applyKey === 3);

    // 3. PLONK through the patched surface still verifies (exercises batchToMontgomery / FromMontgomery / Inverse + BigBuffer ffts)
    const pz = path.join(snarkjs.refRoot, "test/circuit2/circuit.zkey"), pw = path.join(snarkjs.refRoot, "test/circuit2/witness.wtns");
    if (fs.existsSync(pz)) {
        for (const k of Object.keys(calls)) delete calls[k];
        const zk = new Uint8Array(fs.readFileSync(pz)), wt = new Uint8Array(fs.readFileSync(pw));
        const pr = await snarkjs.plonk.prove(zk, wt);
        const vk = await snarkjs.zKey.exportVerificationKey(zk);
        check("plonk.prove through register.js verifies", await snarkjs.plonk.verify(vk, pr.publicSignals, pr.proof));
        check("plonk census: 9 MSM, batchInverse, batchFromMontgomery used", calls.msm1 === 9 && calls.batch2 >= 1 && calls.batch1 >= 9 && calls.batch0 === 3);
    }
    // 4. setup-side callers (SURVEY.md 8 f3): plonk.setup / fflonk.setup reach the bulk ops through the same patched methods
    //    (src/plonk_setup.js:322-330,395-403; src/fflonk_setup.js via Polynomial.to4T / multiExponentiation) — the zkey written
    //    through the patched surface must equal the unpatched one byte for byte. A small seeded ptau is made with the originals.
    const r1cs = path.join(snarkjs.refRoot, "test/plonk_circuit/circuit.r1cs");
    if (fs.existsSync(r1cs)) {
        const mem = () => ({ type: "mem" });
        const mkPtau = async () => { const p0 = mem(), p1 = mem(), pf = mem(); await snarkjs.powersOfTau.newAccumulator(curve, 7, p0); await snarkjs.powersOfTau.contribute(p0, p1, "C1", "Entropy1"); await snarkjs.powersOfTau.preparePhase2(p1, pf); return pf; };
        const rb = new Uint8Array(fs.readFileSync(r1cs));
        for (const k of Object.keys(calls)) delete calls[k];
        const ptau = await mkPtau();                       // ceremony code also runs through the patched curve (G1/G2 group FFTs stay WASM)
        const zPatched = mem(), fPatched = mem();
        await snarkjs.plonk.setup(rb, ptau, zPatched);
        const setupCalls = Object.assign({}, calls);
        await snarkjs.fflonk.setup(rb, ptau, fPatched);
        unregister(curve);
        const zPlain = mem(), fPlain = mem();
        await snarkjs.plonk.setup(rb, ptau, zPlain);
        await snarkjs.fflonk.setup(rb, ptau, fPlain);
        check("plonk.setup through register.js writes the same zkey", sha(zPatched.data) === sha(zPlain.data));
        check("fflonk.setup through register.js writes the same zkey", sha(fPatched.data) === sha(fPlain.data));
        check("plonk.setup census: 8 MSMs (Qm..S3), ifft + fft(4n) per selector / sigma / Lagrange", setupCalls.msm1 === 8 && setupCalls.ifft >= 9 && setupCalls.fft >= 9);
        register(curve, { addon: mock });
    }
    unregister(curve);
    check("unregister restores the WASM entry points", curve.__zkmi === undefined);
    // options.minPoints (SURVEY.md 8b: small calls stay on the curve's own entry points): below the threshold nothing reaches the addon, at the
    // threshold everything does; the default (0) sends every buffer-form call to the device
    {
        register(curve, { addon: mock, minPoints: 64 });
        const before = Object.assign({}, calls);
        const b4 = new Uint8Array(4 * 64), s4 = new Uint8Array(4 * 32), x16 = new Uint8Array(16 * 32), x64 = new Uint8Array(64 * 32);
        for (let i = 0; i < 4; i++) s4[i * 32] = i + 1;                 // bases: four points at infinity (all-zero bytes)
        for (let i = 0; i < 16; i++) x16[i * 32] = i + 1;
        for (let i = 0; i < 64; i++) x64[i * 32] = i + 1;
        const small = await curve.G1.multiExpAffine(b4, s4), f16 = await curve.Fr.fft(x16);
        check("minPoints: a 4-point multiExpAffine and a 16-point fft stay on the WASM originals", (calls.msm1 || 0) === (before.msm1 || 0) && (calls.fft || 0) === (before.fft || 0) &&
              sha(f16) === sha(await saved.fft(x16)) && sha(small) === sha(await saved.g1(b4, s4)));
        const f64 = await curve.Fr.fft(x64);
        if (f64.__p) { await f64.__p; delete f64.__p; }               // the mock fills its output asynchronously
        check("minPoints: a 64-point fft goes to the addon", (calls.fft || 0) === (before.fft || 0) + 1 && sha(f64) === sha(await saved.fft(x64)));
        unregister(curve);
    }
    // the permission bits handed to zkmi_msm (include/zkmi.h): default = ZKMI_BASES_CACHE alone (full content hash on every call: the result follows
    // the bytes passed), immutableBases: true adds ZKMI_BASES_IMMUTABLE (the caller's promise), cacheBases: false / small calls pass 0
    {
        const keys = [];
        const spy = Object.assign({}, mock, { msm(cid, group, bases, scalars, n, sb, key) { keys.push(key); return mock.msm(cid, group, bases, scalars, n, sb, key); } });
        const nb = 4096, bb = new Uint8Array(nb * 64), sb = new Uint8Array(nb * 32);
        for (const [opts, want] of [[{}, 1], [{ immutableBases: true }, 3], [{ cacheBases: false }, 0], [{ immutableBases: true, cacheMinPoints: 1 << 20 }, 0]]) {
            register(curve, Object.assign({ addon: spy }, opts));
            const r = await curve.G1.multiExpAffine(bb, sb);
            if (r.__p) { await r.__p; delete r.__p; }
            unregister(curve);
            check(`register(${JSON.stringify(opts)}) passes base_cache_key ${want}`, keys[keys.length - 1] === want);
        }
    }
    // a transform of more than 2^Fr.s elements (the reference's "extended" FFT, min.js:1@216148) is forwarded to the saved WASM entry point, not refused:
    // checked by lowering Fr.s for the duration of one call (a real 2^29-element buffer is 16 GiB) and watching which side is called
    {
        let origCalls = 0;
        const realFft = curve.Fr.fft;
        curve.Fr.fft = async function () { origCalls++; return new Uint8Array(arguments[0].byteLength); };       // stands in for the WASM original
        register(curve, { addon: mock });
        const before = calls.fft || 0, sKeep = curve.Fr.s;
        curve.Fr.s = 3;
        const out = await curve.Fr.fft(new Uint8Array(16 * 32));
        curve.Fr.s = sKeep;
        check("Fr.fft of 2^(s+1) elements goes to the saved original, not to the addon", origCalls === 1 && (calls.fft || 0) === before && out.byteLength === 16 * 32);
        unregister(curve);
        curve.Fr.fft = realFft;
    }
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.log("ERROR", e); process.exit(2); });
