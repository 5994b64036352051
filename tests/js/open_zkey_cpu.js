// tests/js/open_zkey_cpu.js — CPU: js/groth16_native.js reads a zkey the way the reference does (section table, then sections BY OFFSET: src/groth16_prove.js:29-33,
// 57-59, 84-100) from bytes, a path, a fastfile descriptor and a BigBuffer-backed memory file, into pages of any size and — for a key shard — with gaps where nothing was
// read: every variant must describe the same bytes as the flat parse. No device, no addon call.
"use strict";
const fs = require("fs"), path = require("path");
const { openZkey, descFromSections, toPages, parseWtns } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "groth16_native.js"));
const GOLD = path.join(__dirname, "..", "golden");
let fails = 0;
const check = (name, ok) => { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); };
const SEC = ["coeffs", "A", "B1", "B2", "C", "H"];
// materialise a section (Uint8Array | pages with gaps) into one array + a mask of the bytes that were provided
function flat(x) {
    if (x instanceof Uint8Array) return { data: x, have: new Uint8Array(x.length).fill(1) };
    const len = x.reduce((a, e) => a + (typeof e === "number" ? e : e.length), 0), data = new Uint8Array(len), have = new Uint8Array(len);
    let o = 0;
    for (const e of x) { if (typeof e === "number") o += e; else { data.set(e, o); have.fill(1, o, o + e.length); o += e.length; } }
    return { data, have };
}
for (const tag of ["groth16_bn128_n1024", "groth16_bls12381_n1024"]) {
    const p = path.join(GOLD, tag + ".zkey"), bytes = new Uint8Array(fs.readFileSync(p));
    const ref = openZkey(bytes);
    check(`${tag}: header`, ref.nVars === 1003 && ref.nPublic === 2 && ref.domainSize === 1024 && ref.n8r === 32 && SEC.every((k) => ref.desc[k] instanceof Uint8Array));
    const big = { buffers: [bytes.subarray(0, 70000), bytes.subarray(70000)], byteLength: bytes.length,
                  slice(a, b) { const o = new Uint8Array(b - a); for (let i = a; i < b; i++) o[i - a] = bytes[i]; return o; } };       // the shape of a BigBuffer
    for (const [nm, src, opts] of [["path, 64 KiB pages", p, { pageBytes: 65536 }], ["fastfile descriptor, 4140-byte pages", { type: "file", fileName: p }, { pageBytes: 4140 }],
                                   ["mem descriptor", { type: "mem", data: bytes }, { pageBytes: 1 << 14 }], ["BigBuffer-backed mem file", { type: "mem", data: big }, { pageBytes: 50000 }]]) {
        const zk = openZkey(src, opts);
        const same = SEC.every((k) => { const f = flat(zk.desc[k]); return f.have.every((v) => v === 1) && Buffer.from(f.data).equals(Buffer.from(ref.desc[k])); });
        const hdr = ["alpha1", "beta1", "beta2", "delta1", "delta2"].every((k) => Buffer.from(zk.desc[k]).equals(Buffer.from(ref.desc[k])));
        check(`${tag}: ${nm} == flat parse (${SEC.map((k) => Array.isArray(zk.desc[k]) ? zk.desc[k].length : 1).join("/")} pages)`, same && hdr && zk.nVars === ref.nVars);
    }
    // a shard: only its byte ranges are provided, and they hold the right bytes
    const sh = { vLo: 251, vHi: 502, hLo: 256, hHi: 512 }, g1 = 2 * ref.n8q, fc = ref.nPublic + 1;
    const zs = openZkey(p, { shard: sh, pageBytes: 8192 });
    const need = { A: [sh.vLo * g1, sh.vHi * g1], B1: [sh.vLo * g1, sh.vHi * g1], B2: [sh.vLo * 2 * g1, sh.vHi * 2 * g1], C: [(sh.vLo - fc) * g1, (sh.vHi - fc) * g1], H: [sh.hLo * g1, sh.hHi * g1] };
    let ok = flat(zs.desc.coeffs).have.every((v) => v === 1);
    for (const k of Object.keys(need)) {
        const f = flat(zs.desc[k]), [lo, hi] = need[k];
        ok = ok && f.data.length === ref.desc[k].length && f.have.every((v, i) => v === ((i >= lo && i < hi) ? 1 : 0)) &&
             Buffer.from(f.data.subarray(lo, hi)).equals(Buffer.from(ref.desc[k].subarray(lo, hi)));
    }
    check(`${tag}: a shard reads exactly its own byte ranges (gaps elsewhere), the coefficient section whole`, ok);
    check(`${tag}: headerOnly reads no bulk section`, openZkey(p, { headerOnly: true }).desc === null);
    // sections a caller read itself with readSection: Uint8Array stays, BigBuffer -> its pages
    const d = descFromSections(ref, { 4: ref.desc.coeffs, 5: { buffers: [ref.desc.A.subarray(0, 100), ref.desc.A.subarray(100)], byteLength: ref.desc.A.length }, 6: ref.desc.B1, 7: ref.desc.B2, 8: ref.desc.C, 9: ref.desc.H });
    check(`${tag}: descFromSections turns a BigBuffer into its pages`, Array.isArray(d.A) && d.A.length === 2 && d.coeffs === ref.desc.coeffs && toPages(ref.desc.H) === ref.desc.H);
    const w1 = parseWtns(path.join(GOLD, tag + ".wtns"), ref), w2 = parseWtns(new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".wtns"))), ref);
    check(`${tag}: parseWtns from a path == from bytes`, Buffer.from(w1).equals(Buffer.from(w2)) && w1.length === 1003 * 32);
}
let threw = false;
try { openZkey(path.join(GOLD, "plonk_bn128_small.zkey")); } catch (e) { threw = /not groth16|Missing section/.test(e.message); }
check("a PLONK zkey is refused", threw);
threw = false;
try { openZkey(new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.wtns")))); } catch (e) { threw = /Invalid File format/.test(e.message); }
check("wrong magic: Invalid File format", threw);
console.log(fails ? `${fails} FAILED` : "ALL OK");
process.exit(fails ? 1 : 0);
