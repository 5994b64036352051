// tests/js/register_replay.js — GPU test of the drop-in path in ONE piece short of the bundle itself: snarkjs_amd/js/register.js
// applied to a curve object, driving the REAL N-API addon (zkmi_napi.node -> libzkmi.so), fed with exactly the bulk calls the real
// snarkjs makes during its seeded groth16.prove (n = 1024) and plonk.prove (n = 2048), and during a power-8 ceremony (powersOfTau new /
// contribute / preparePhase2: G.batchApplyKey, G.ifft, batchLEMtoU / LEMtoC) followed by plonk.setup and zKey.newZKey (the setup-side
// callers, src/plonk_setup.js:323-403): tests/golden/replay_bn128.{json,bin},
// recorded from the reference bundle by oracle/gen_replay.js (the bundle cannot travel to the GPU box). Checked per call: result
// container type (Uint8Array vs BigBuffer, the rule downstream snarkjs code depends on), byte-exact Fr results, MSM results as
// affine points; chained inputs use OUR earlier outputs. A second pass allows the resident-base cache for every MSM, which sends
// PLONK's nine PTau.slice(0, k) calls (five distinct k) through the content-addressed prefix logic of zkmi_msm.
// Run:  node tests/js/register_replay.js      (exit code 0 = all green)
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const { register, unregister, loadAddon } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "register.js"));
const GOLD = path.join(__dirname, "..", "golden");
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
let fails = 0;
function check(name, ok) { if (!ok) { fails++; console.log("FAIL", name); } }

// test double of ffjavascript's BigBuffer (min.js:1@183423): pages of at most PAGE bytes, slice() of a range inside one page is
// a Uint8Array. A small page size makes the fixtures span several pages, as 2^24-sized buffers do with the real 1 GiB pages.
const PAGE = 1000 * 32;
class BigBuffer {
    constructor(size) {
        this.buffers = []; this.byteLength = size;
        for (let i = 0; i < size; i += PAGE) this.buffers.push(new Uint8Array(Math.min(PAGE, size - i)));
    }
    slice(fr, to) {
        if (to === undefined) to = this.byteLength;
        if (fr === undefined) fr = 0;
        const len = to - fr, first = Math.floor(fr / PAGE), last = Math.floor((fr + len - 1) / PAGE);
        if (first == last || len == 0) return new Uint8Array(this.buffers[first].buffer, this.buffers[first].byteOffset + fr % PAGE, len);
        const out = new Uint8Array(len);
        let p = first, o = fr % PAGE, r = len, k = 0;
        while (r > 0) { const l = Math.min(PAGE - o, r); out.set(this.buffers[p].subarray(o, o + l), k); k += l; r -= l; p++; o = 0; }
        return out;
    }
    set(buff, offset) {
        if (offset === undefined) offset = 0;
        let p = Math.floor(offset / PAGE), o = offset % PAGE, r = buff.byteLength, k = 0;
        while (r > 0) { const l = Math.min(PAGE - o, r); this.buffers[p].set(buff.slice(k, k + l), o); k += l; r -= l; p++; o = 0; }
    }
}
const flat = (b) => (b instanceof Uint8Array) ? b : b.slice(0, b.byteLength);
const kind = (b) => (b instanceof Uint8Array) ? "u8" : ((b && Array.isArray(b.buffers)) ? "big" : typeof b);

const rec = JSON.parse(fs.readFileSync(path.join(GOLD, "replay_bn128.json")));
const blob = fs.readFileSync(path.join(GOLD, "replay_bn128.bin"));
check("blob file intact", sha(blob) === rec.blob_sha256);
const addon = loadAddon();
addon.init(0);

function makeCurve() {           // the part of ffjavascript's curve object register.js touches
    const nope = (nm) => async function () { throw new Error(nm + ": the WASM original must not be reached in this test"); };
    const Fr = { n8: 32, s: 28, e: (x) => x };
    for (const nm of ["fft", "ifft", "batchApplyKey", "batchToMontgomery", "batchFromMontgomery", "batchInverse"]) Fr[nm] = nope("Fr." + nm);
    const group = (gn, n8, zlen) => {
        const G = { F: { n8 }, zero: new Uint8Array(zlen) };
        for (const nm of ["multiExpAffine", "fft", "ifft", "batchApplyKey", "batchLEMtoU", "batchUtoLEM", "batchLEMtoC", "batchCtoLEM"]) G[nm] = nope(gn + "." + nm);
        return G;
    };
    return { name: "bn128", Fr, G1: group("G1", 32, 96), G2: group("G2", 64, 192) };
}

async function replay(passName, options) {
    const curve = register(makeCurve(), Object.assign({ addon }, options));
    for (const run of rec.runs) {
        const files = {}, outs = [];
        const bytesOf = (a) => {
            if (a.hex !== undefined) return new Uint8Array(Buffer.from(a.hex, "hex"));
            let b;
            if (a.out !== undefined) b = outs[a.out];
            else if (a.file !== undefined) { files[a.file] = files[a.file] || fs.readFileSync(path.join(GOLD, a.file)); b = files[a.file].subarray(a.off, a.off + a.len); }
            else b = blob.subarray(a.blob, a.blob + a.len);
            if (sha(b) !== a.sha256) throw new Error("fixture bytes do not match their recorded hash");
            return new Uint8Array(b);                               // private copy: inputs must never be mutated (checked below)
        };
        let i = 0;
        for (const c of run.calls) {
            const [oname, mname] = c.m.split(".");
            const args = c.args.map((a) => {
                if (a === null) return undefined;
                if (a.v !== undefined) return Number(a.v);
                if (a.s !== undefined) return a.s;                       // inType / outType of the group FFTs
                const b = bytesOf(a);
                if (a.c === "big") { const bb = new BigBuffer(b.byteLength); bb.set(b, 0); return bb; }
                return b;
            });
            const before = args.map((x) => (x instanceof Uint8Array || (x && x.buffers)) ? sha(flat(x)) : null);
            const res = await curve[oname][mname](...args);
            const label = `${passName} ${run.tag} call ${i} ${c.m}`;
            check(label + " container " + c.res.c, kind(res) === c.res.c);
            const rb = flat(res);
            check(label + " length", rb.byteLength === c.res.len);
            if (c.res.affine !== undefined) {
                const group = oname === "G1" ? 1 : 2;
                check(label + " point", Buffer.from(addon.toAffine(0, group, rb)).toString("hex") === c.res.affine);
            } else check(label + " bytes", sha(rb) === c.res.sha256);
            args.forEach((x, k) => { if (before[k]) check(label + " input " + k + " untouched", sha(flat(x)) === before[k]); });
            outs.push(new Uint8Array(rb));
            i++;
        }
        console.log(`ok   ${passName}: ${run.tag}: ${run.calls.length} bulk calls replayed through register.js + the real addon`);
    }
    unregister(curve);
}

(async () => {
    await replay("plain", { cacheBases: false });
    addon.releaseBases(0);
    await replay("resident-bases pass 1", { cacheMinPoints: 1, async: false });   // first sight of every base buffer; blocking addon calls (the other passes use msmAsync / nttAsync)
    await replay("resident-bases pass 2", { cacheMinPoints: 1 });          // tables get built, prefixes re-use them
    await replay("resident-bases pass 3", { cacheMinPoints: 1 });          // everything served from resident tables
    addon.releaseBases(0);
    // error conventions of the patched surface (reference messages)
    const curve = register(makeCurve(), { addon });
    let threw = false;
    try { await curve.Fr.fft(new Uint8Array(96)); } catch (e) { threw = e.message === "fft must be multiple of 2"; }
    check("fft error message", threw);
    threw = false;
    try { await curve.G1.multiExpAffine(new Uint8Array(128), new Uint8Array(63)); } catch (e) { threw = e.message === "Scalar size does not match"; }
    check("multiExpAffine error message", threw);
    check("multiExpAffine empty -> G1.zero", (await curve.G1.multiExpAffine(new Uint8Array(0), new Uint8Array(0))) === curve.G1.zero);
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.log("ERROR", e); process.exit(2); });
