// tests/js/addon_golden.js — GPU parity of the Node boundary: the N-API addon (zkmi_napi.node -> libzkmi.so) against the
// golden vectors generated from the reference (tests/golden/, oracle/gen_golden.js).  No snarkjs needed at run time.
// Run:  node tests/js/addon_golden.js      (exit code 0 = all green)
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const addon = require(path.join(__dirname, "..", "..", "snarkjs_amd", "napi", "zkmi_napi.node"));
const GOLD = path.join(__dirname, "..", "golden");
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
const eq = (a, b) => Buffer.compare(Buffer.from(a.buffer, a.byteOffset, a.byteLength), Buffer.from(b.buffer, b.byteOffset, b.byteLength)) === 0;
let fails = 0;
function check(name, ok) { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); }

function iota(n) {
    const x = new Uint8Array(n * 32);
    for (let i = 0; i < n; i++) { const v = i + 1; x[i * 32] = v & 255; x[i * 32 + 1] = (v >> 8) & 255; x[i * 32 + 2] = (v >> 16) & 255; }
    return x;
}
const raw = (tag, k) => new Uint8Array(fs.readFileSync(path.join(GOLD, `${tag}_n1024_${k}.bin`)));
function frE(r, v) {            // Montgomery form of a small integer: v * 2^256 mod r, 32 bytes LE
    let x = (BigInt(v) << 256n) % r;
    const o = new Uint8Array(32);
    for (let i = 0; i < 32; i++) { o[i] = Number(x & 255n); x >>= 8n; }
    return o;
}

addon.init(0);
for (const [tag, cid] of [["bn128", 0], ["bls12381", 1]]) {
    const d = JSON.parse(fs.readFileSync(path.join(GOLD, `${tag}_kernel_vectors.json`)));
    const r = BigInt(d.r), x = iota(1024), n8q = d.n8q;
    const out = () => new Uint8Array(1024 * 32);
    let o = out(); addon.ntt(cid, x, o, 10, 0, null, null); check(`${tag} fft`, eq(o, raw(tag, "fft")) && sha(o) === d.n1024.fft);
    o = out(); addon.ntt(cid, x, o, 10, 1, null, null); check(`${tag} ifft`, eq(o, raw(tag, "ifft")));
    o = out(); addon.applyKey(cid, x, o, 1024, frE(r, 7), frE(r, 11)); check(`${tag} batchApplyKey`, eq(o, raw(tag, "applykey_7_11")));
    o = out(); addon.frBatch(cid, 0, x, o, 1024); check(`${tag} batchToMontgomery`, eq(o, raw(tag, "to_mont")));
    o = out(); addon.frBatch(cid, 1, x, o, 1024); check(`${tag} batchFromMontgomery`, eq(o, raw(tag, "from_mont")));
    o = out(); addon.frBatch(cid, 2, x, o, 1024); check(`${tag} batchInverse`, eq(o, raw(tag, "inverse")));
    // paged (BigBuffer-style) input and output, split at an element that is not a power of two
    const o1 = new Uint8Array(300 * 32), o2 = new Uint8Array(724 * 32);
    addon.ntt(cid, [x.subarray(0, 500 * 32), x.subarray(500 * 32)], [o1, o2], 10, 0, null, null);
    check(`${tag} fft paged`, eq(Buffer.concat([o1, o2]), raw(tag, "fft")));
    for (const g of [1, 2]) {
        const jac = addon.msm(cid, g, raw(tag, `g${g}_bases`), x, 1024, 32, 0);
        check(`${tag} G${g}.multiExpAffine`, jac.length === 3 * g * n8q && eq(addon.toAffine(cid, g, jac), raw(tag, `g${g}_msm_affine`)));
        // resident bases: base_cache_key is a set of PERMISSION BITS (include/zkmi.h: ZKMI_BASES_CACHE = 1, ZKMI_BASES_IMMUTABLE = 2), not an identity.
        // First sight remembers the hashes, second sight builds the window table, third call uses it; with the promise bit (3) the re-check is sampled
        for (const key of [1, 3]) {
            const js = [0, 1, 2].map(() => addon.msm(cid, g, raw(tag, `g${g}_bases`), x, 1024, 32, key));
            check(`${tag} G${g}.multiExpAffine (resident tables, cache key ${key})`, js.every((j) => eq(addon.toAffine(cid, g, j), raw(tag, `g${g}_msm_affine`))));
        }
        addon.releaseBases(1);
        // a legacy caller that passes an arbitrary "identity" (r04 semantics: any non-zero value) is refused, not silently given another mode
        let threw = false;
        try { addon.msm(cid, g, raw(tag, `g${g}_bases`), x, 1024, 32, 1000 + 10 * cid + g); } catch (e) { threw = /permission bits/.test(e.message); }
        check(`${tag} G${g}: a cache key with unknown bits fails loudly`, threw);
    }
    // ceremony side (SURVEY.md 8 f4): group-element FFTs and G.batchApplyKey against the reference's vectors
    {
        const gv = JSON.parse(fs.readFileSync(path.join(GOLD, `${tag}_group_vectors.json`)));
        for (const [gn, g, n] of [["g1", 1, 256], ["g2", 2, 64]]) {
            const sG = 2 * g * n8q, bases = raw(tag, `g${g}_bases`).subarray(0, n * sG);
            const lg = Math.round(Math.log2(n));
            let o2 = new Uint8Array(n * sG); addon.groupFft(cid, g, bases, o2, lg, 0);
            check(`${tag} G${g}.fft`, sha(o2) === gv[gn].fft && eq(o2, new Uint8Array(fs.readFileSync(path.join(GOLD, `${tag}_gfft_${gn}_n${n}_fft.bin`)))));
            o2 = new Uint8Array(n * sG); addon.groupFft(cid, g, [bases.subarray(0, 10 * sG), bases.subarray(10 * sG)], o2, lg, 1);
            check(`${tag} G${g}.ifft (paged input)`, sha(o2) === gv[gn].ifft);
            o2 = new Uint8Array(n * sG); addon.groupApplyKey(cid, g, bases, o2, n, frE(r, 3), frE(r, 5));
            check(`${tag} G${g}.batchApplyKey`, sha(o2) === gv[gn].applykey_3_5);
        }
    }
    // point-format conversions of the ceremony files against the reference's vectors (oracle/gen_golden.js convertVectors)
    {
        const cv = JSON.parse(fs.readFileSync(path.join(GOLD, `${tag}_conv_vectors.json`)));
        for (const [gn, g] of [["g1", 1], ["g2", 2]]) {
            const n = cv[gn].n, sG = 2 * g * n8q, rd = (k) => new Uint8Array(fs.readFileSync(path.join(GOLD, `${tag}_conv_${gn}_n${n}_${k}.bin`)));
            const lem = rd("lem"), U = rd("u"), Cc = rd("c");
            let o2 = new Uint8Array(n * sG); addon.groupConvert(cid, g, 0, lem, o2, n);
            check(`${tag} G${g}.batchLEMtoU`, eq(o2, U));
            o2 = new Uint8Array(n * sG); addon.groupConvert(cid, g, 1, [U.subarray(0, 3 * sG), U.subarray(3 * sG)], o2, n);
            check(`${tag} G${g}.batchUtoLEM (paged input)`, eq(o2, lem));
            o2 = new Uint8Array(n * sG / 2); addon.groupConvert(cid, g, 2, lem, o2, n);
            check(`${tag} G${g}.batchLEMtoC`, eq(o2, Cc));
            o2 = new Uint8Array(n * sG); addon.groupConvert(cid, g, 3, Cc, o2, n);
            check(`${tag} G${g}.batchCtoLEM`, eq(o2, lem));
        }
    }
    let threw = false;
    try { addon.msm(cid, 1, raw(tag, "g1_bases"), x.subarray(0, 1024 * 32 - 1), 1024, 32, 0); } catch (e) { threw = /Scalar size does not match/.test(e.message); }
    check(`${tag} scalar size error`, threw);
}
// fused Groth16 prover against the reference's seeded proof (SURVEY.md Appendix C.3)
(async () => {
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.json")));
    const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.zkey")));
    const wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.wtns")));
    const sections = (data) => {
        const dv = new DataView(data.buffer, data.byteOffset, data.byteLength), out = {};
        let off = 12;
        for (let i = 0, n = dv.getUint32(8, true); i < n; i++) {
            const t = dv.getUint32(off, true), len = Number(dv.getBigUint64(off + 4, true));
            off += 12; out[t] = data.subarray(off, off + len); off += len;
        }
        return out;
    };
    const zs = sections(zkey), ws = sections(wtns), h = zs[2], hv = new DataView(h.buffer, h.byteOffset, h.byteLength);
    let o = 8 + 32 + 32;
    const nVars = hv.getUint32(o, true), nPublic = hv.getUint32(o + 4, true), domainSize = hv.getUint32(o + 8, true);
    o += 12;
    const pt = (k) => { const v = h.subarray(o, o + k * 32); o += k * 32; return v; };
    const alpha1 = pt(2), beta1 = pt(2), beta2 = pt(4); pt(4); const delta1 = pt(2), delta2 = pt(4);
    const hexb = (s) => new Uint8Array(Buffer.from(s, "hex"));
    const res = addon.groth16Prove({ curve: 0, nVars, nPublic, domainSize, coeffs: zs[4], A: zs[5], B1: zs[6], B2: zs[7], C: zs[8], H: zs[9], alpha1, beta1, beta2, delta1, delta2 },
                                   7, ws[2], hexb(g.r_mont), hexb(g.s_mont));
    const q = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;
    const inv = (a, m) => { let [x0, x1, b] = [1n, 0n, m]; a %= m; while (b) { const t = a / b; [a, b] = [b, a - t * b]; [x0, x1] = [x1, x0 - t * x1]; } return ((x0 % m) + m) % m; };
    const rinv = inv(1n << 256n, q);
    const coord = (b, i) => { let v = 0n; for (let k = 31; k >= 0; k--) v = (v << 8n) | BigInt(b[32 * i + k]); return (v * rinv % q).toString(); };
    const proof = { pi_a: [coord(res.pi_a, 0), coord(res.pi_a, 1), "1"],
                    pi_b: [[coord(res.pi_b, 0), coord(res.pi_b, 1)], [coord(res.pi_b, 2), coord(res.pi_b, 3)], ["1", "0"]],
                    pi_c: [coord(res.pi_c, 0), coord(res.pi_c, 1), "1"], protocol: "groth16", curve: "bn128" };
    check("groth16Prove == reference proof (sha256 of JSON)", sha(JSON.stringify(proof)) === g.proof_sha256);
    const again = addon.groth16Prove(0, 7, ws[2], hexb(g.r_mont), hexb(g.s_mont));       // resident key
    check("groth16Prove with resident key", eq(again.pi_a, res.pi_a) && eq(again.pi_b, res.pi_b) && eq(again.pi_c, res.pi_c));
    // asynchronous variants (napi_create_async_work): same results, the event loop keeps turning while they run, errors reject
    await (async () => {
        let ticks = 0;
        const timer = setInterval(() => { ticks++; }, 0);
        const x = iota(1024), bases = raw("bn128", "g1_bases"), o1 = new Uint8Array(1024 * 32);
        const pr = addon.groth16ProveAsync(0, 7, ws[2], hexb(g.r_mont), hexb(g.s_mont)), pm = addon.msmAsync(0, 1, bases, x, 1024, 32, 0),
              pn = addon.nttAsync(0, x, o1, 10, 0, null, null);
        check("async calls return promises", pr instanceof Promise && pm instanceof Promise && pn instanceof Promise);
        const [ar, am] = await Promise.all([pr, pm, pn]);
        check("groth16ProveAsync == groth16Prove", eq(ar.pi_a, res.pi_a) && eq(ar.pi_b, res.pi_b) && eq(ar.pi_c, res.pi_c));
        check("msmAsync == reference", eq(addon.toAffine(0, 1, am), raw("bn128", "g1_msm_affine")));
        check("nttAsync == reference", eq(o1, raw("bn128", "fft")));
        const big = new Uint8Array(32 << 20), ob = new Uint8Array(32 << 20);
        for (let i = 0; i < (1 << 20); i++) big[32 * i] = i & 255;
        const t0 = ticks;
        await Promise.all([0, 1, 2, 3].map(() => addon.nttAsync(0, big, ob, 20, 0, null, null)));    // >= 5 ms of library time in total
        check(`event loop turned while async calls ran (${ticks - t0} timer ticks)`, ticks - t0 >= 1);
        let rejected = false;
        try { await addon.msmAsync(0, 1, bases, x.subarray(0, 1024 * 32 - 1), 1024, 32, 0); } catch (e) { rejected = /Scalar size does not match/.test(e.message); }
        check("msmAsync rejects with the library's message", rejected);
        clearInterval(timer);
    })();
    addon.groth16Release(7);
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.error(e); process.exit(1); });
