// snarkjs_amd/js/register.js — makes unmodified snarkjs run its prover hot path on the MI355X.
//
// snarkjs reaches all bulk arithmetic through ONE object per curve, returned by getCurveFromName / getCurveFromQ /
// getCurveFromR (reference src/curves.js:9-53) and cached in globalThis.curve_bn128 / curve_bls12381 (bundle
// build/snarkjs.min.js:1@240638).  register(curve) replaces the bulk entry points of that object
//     curve.G1.multiExpAffine, curve.G2.multiExpAffine                       (min.js:1@214996)
//     curve.Fr.fft, curve.Fr.ifft                                             (min.js:1@215859)
//     curve.Fr.batchApplyKey / batchToMontgomery / batchFromMontgomery / batchInverse (min.js:1@211529, @188677)
// with calls into the HIP library through the N-API addon (../napi/zkmi_napi.node -> libzkmi.so, include/zkmi.h),
// keeping the reference's argument meaning, container rules (Uint8Array in -> Uint8Array out, BigBuffer in ->
// BigBuffer out, inputs never mutated) and error messages.  Everything else on the curve object (element-level
// Fr/G1/G2 ops, pairing, curve.tm, ceremony-only group FFTs) is left untouched.
//
// There is no silent fallback: without a HIP device register() throws.
"use strict";
const path = require("path");

const CURVE_ID = { bn128: 0, bls12381: 1 };
const OP = { TO_MONTGOMERY: 0, FROM_MONTGOMERY: 1, INVERSE: 2 };

function loadAddon() {
    return require(path.join(__dirname, "..", "napi", "zkmi_napi.node"));
}
function isBig(b) { return b && !(b instanceof Uint8Array) && Array.isArray(b.buffers) && typeof b.byteLength === "number"; }
function isBuf(b) { return (b instanceof Uint8Array) || isBig(b); }
// Uint8Array -> itself; BigBuffer -> its page list (ffjavascript BigBuffer.buffers)
function pagesOf(b) { return (b instanceof Uint8Array) ? b : b.buffers; }
// Result containers follow the reference function by function (downstream code tests `instanceof BigBuffer`, e.g.
// src/groth16_prove.js:361): batchToMontgomery / batchFromMontgomery / batchApplyKey return the input's own type
// (min.js:1 helper before @185893, @211529); fft / ifft / batchInverse first take `buff.slice(0, byteLength)`, which for a
// BigBuffer of at most one page is a plain Uint8Array (BigBuffer.slice, min.js:1@183423), and return THAT type.
const PAGE_SIZE = 1 << 30;
function allocLike(b, byteLength) { return new b.constructor(byteLength); }
function allocLikeSliced(b, byteLength) {
    return (b instanceof Uint8Array || b.byteLength <= PAGE_SIZE) ? new Uint8Array(byteLength) : new b.constructor(byteLength);
}
// Resident bases: the last argument of addon.msm only ALLOWS the library to keep a base buffer's pre-computed window tables on the
// device. Identity is established by the library itself from the full content of the buffer (include/zkmi.h: zkmi_msm) — never
// by a fingerprint computed here — and a table is only built the second time the same bytes are seen, under an LRU byte budget.
const CACHE_ALLOWED = 1;                                    // ZKMI_BASES_CACHE
// options.immutableBases: the caller's PROMISE that a base buffer it passes again (same memory, same length) still holds the same bytes — zkey
// sections and SRS slices that snarkjs reads once and never writes to. Only then the library re-checks a resident buffer by sample instead of
// hashing all of it on every call (include/zkmi.h: ZKMI_BASES_IMMUTABLE; saves ~0.6 ms of a 2^20-point call). Default false: the result of
// every call follows the bytes passed, as the reference's does, even if a buffer was edited in place between two calls.
const BASES_IMMUTABLE = 2;                                  // ZKMI_BASES_IMMUTABLE
function log2(n) { let l = 0; while ((1 << (l + 1)) <= n && l < 40) l++; return l; }

function register(curve, options) {
    options = options || {};
    const addon = options.addon || loadAddon();
    if (curve.__zkmi) return curve;
    const cid = CURVE_ID[curve.name];
    if (cid === undefined) throw new Error(`Curve not supported: ${curve.name}`);
    addon.init(options.device === undefined ? 0 : options.device);      // throws when no HIP device is visible
    const Fr = curve.Fr;
    const orig = {};
    const cacheBases = options.cacheBases !== false;      // keep base tables resident between calls (static zkey sections)
    const cacheMinPoints = options.cacheMinPoints === undefined ? 4096 : options.cacheMinPoints;
    const cacheBits = CACHE_ALLOWED | (options.immutableBases === true ? BASES_IMMUTABLE : 0);
    // options.async !== false: multiExpAffine / fft / ifft run on a libuv pool thread (addon.msmAsync / nttAsync, napi_create_async_work)
    // and the Node event loop keeps turning meanwhile; `async: false` keeps the blocking calls.
    const useAsync = options.async !== false && typeof addon.msmAsync === "function" && typeof addon.nttAsync === "function";
    // options.minPoints (default 0: every buffer-form call runs on the device): calls on FEWER elements go to the curve's own saved entry points
    // — the reference's WASM, the behaviour SURVEY.md 8b describes for a drop-in ("falls back to the saved originals when n is small"). A call on
    // 4 points costs the device path about a dozen launches (bench.py: wall_through_napi.msm_small_ms / ntt_small_ms); a host that issues
    // many tiny calls (a verifier's nPublic-point MSM in a loop) sets e.g. minPoints: 1024. Never applied silently: the default is 0.
    const minPoints = options.minPoints === undefined ? 0 : options.minPoints;

    for (const [gname, group] of [["G1", 1], ["G2", 2]]) {
        const G = curve[gname];
        orig[gname] = { multiExpAffine: G.multiExpAffine };
        G.multiExpAffine = async function (buffBases, buffScalars, logger, logText) {
            if (!isBuf(buffBases)) {
                if (logger) logger.error(`${logText} _multiExpChunk buffBases is not Uint8Array`);
                throw new Error(`${logText} _multiExpChunk buffBases is not Uint8Array`);
            }
            if (!isBuf(buffScalars)) {
                if (logger) logger.error(`${logText} _multiExpChunk buffScalars is not Uint8Array`);
                throw new Error(`${logText} _multiExpChunk buffScalars is not Uint8Array`);
            }
            const sGIn = G.F.n8 * 2;
            const nPoints = Math.floor(buffBases.byteLength / sGIn);
            if (nPoints == 0) return G.zero;
            const sScalar = Math.floor(buffScalars.byteLength / nPoints);
            if (sScalar * nPoints != buffScalars.byteLength) throw new Error("Scalar size does not match");
            if (nPoints < minPoints) return orig[gname].multiExpAffine.apply(G, arguments);
            if (logger) logger.debug(`Multiexp start: ${logText}: 0/${nPoints}`);
            const key = (cacheBases && nPoints >= cacheMinPoints) ? cacheBits : 0;
            const res = useAsync ? await addon.msmAsync(cid, group, pagesOf(buffBases), pagesOf(buffScalars), nPoints, sScalar, key)
                                 : addon.msm(cid, group, pagesOf(buffBases), pagesOf(buffScalars), nPoints, sScalar, key);
            if (logger) logger.debug(`Multiexp end: ${logText}: 0/${nPoints}`);
            return res;                                                  // Jacobian, Montgomery, 3*F.n8 bytes
        };
    }

    // Ceremony side (SURVEY.md 8 f4): G.fft / G.ifft over affine points (src/powersoftau_preparephase2.js:87 reaches G.ifft through
    // G.lagrangeEvaluations, which the reference implements on top of this very property for 2^k <= 2^Fr.s points) and
    // G.batchApplyKey (src/mpc_applykey.js:44-70). Only the affine -> affine forms the reference's callers use are taken over; any other
    // inType / outType combination, the array-of-elements form and sizes above 2^28 fall through to the WASM original.
    // The point-format conversions G.batchLEMtoU / batchUtoLEM / batchLEMtoC / batchCtoLEM of the ceremony files are taken over as well.
    // options.ceremony === false leaves all of these methods alone.
    if (options.ceremony !== false) {
        for (const [gname, group] of [["G1", 1], ["G2", 2]]) {
            const G = curve[gname], sG = G.F.n8 * 2;
            Object.assign(orig[gname], { fft: G.fft, ifft: G.ifft, batchApplyKey: G.batchApplyKey });
            const gfft = (inverse, origFn) => async function (buff, inType, outType, logger, loggerTxt) {
                inType = inType || "affine"; outType = outType || "affine";
                if (!isBuf(buff) || inType !== "affine" || outType !== "affine") return origFn.apply(G, arguments);
                const n = buff.byteLength / sG, bits = log2(n);
                if ((1 << bits) != n) throw new Error("fft must be multiple of 2");
                if (bits > 28 || bits > Fr.s) return origFn.apply(G, arguments);
                const out = allocLikeSliced(buff, buff.byteLength);
                addon.groupFft(cid, group, pagesOf(buff), pagesOf(out), bits, inverse ? 1 : 0);
                return out;
            };
            if (typeof G.fft === "function") { G.fft = gfft(false, orig[gname].fft); G.ifft = gfft(true, orig[gname].ifft); }
            // Point-format conversions of the ceremony files (helper `Ia`, min.js:1 before @185893: output has the input's own container
            // type, "Invalid buffer size" when the length is not a whole number of points): LEM <-> U, LEM <-> C.
            for (const [name, kind, sIn, sOut] of [["batchLEMtoU", 0, sG, sG], ["batchUtoLEM", 1, sG, sG], ["batchLEMtoC", 2, sG, sG / 2], ["batchCtoLEM", 3, sG / 2, sG]]) {
                if (typeof G[name] !== "function") continue;
                orig[gname][name] = G[name];
                G[name] = async function (buff) {
                    if (!isBuf(buff)) return orig[gname][name].apply(G, arguments);
                    const n = Math.floor(buff.byteLength / sIn);
                    if (n * sIn !== buff.byteLength) throw new Error("Invalid buffer size");
                    const out = allocLike(buff, n * sOut);
                    addon.groupConvert(cid, group, kind, pagesOf(buff), pagesOf(out), n);
                    return out;
                };
            }
            if (typeof G.batchApplyKey === "function") {
                G.batchApplyKey = async function (buff, first, inc, inType, outType) {
                    inType = inType || "affine"; outType = outType || "affine";
                    if (!isBuf(buff) || inType !== "affine" || outType !== "affine") return orig[gname].batchApplyKey.apply(G, arguments);
                    const n = Math.floor(buff.byteLength / sG);
                    const out = allocLike(buff, n * sG);
                    addon.groupApplyKey(cid, group, pagesOf(buff), pagesOf(out), n, Fr.e(first), Fr.e(inc));
                    return out;
                };
            }
        }
    }

    orig.Fr = { fft: Fr.fft, ifft: Fr.ifft, batchApplyKey: Fr.batchApplyKey, batchToMontgomery: Fr.batchToMontgomery,
                batchFromMontgomery: Fr.batchFromMontgomery, batchInverse: Fr.batchInverse };

    async function ntt(buff, inverse, origFn, args) {
        if (Array.isArray(buff)) return origFn.apply(Fr, args);          // array-of-elements form: not on the prove path
        if (!isBuf(buff)) throw new Error("fft: buffer is not Uint8Array or BigBuffer");
        const n = buff.byteLength / Fr.n8;
        const bits = log2(n);
        if ((1 << bits) != n) throw new Error("fft must be multiple of 2");
        // n = 2^(Fr.s+1): the reference's "extended" transform over the quadratic extension's root (min.js:1@216148) is not built on the device;
        // like the group FFTs above, such a call goes to the saved WASM entry point instead of failing
        if (n < minPoints || bits > Fr.s) return origFn.apply(Fr, args);
        const out = allocLikeSliced(buff, buff.byteLength);
        if (useAsync) await addon.nttAsync(cid, pagesOf(buff), pagesOf(out), bits, inverse ? 1 : 0, null, null);
        else addon.ntt(cid, pagesOf(buff), pagesOf(out), bits, inverse ? 1 : 0, null, null);
        return out;
    }
    Fr.fft = function (buff, inType, outType, logger, loggerTxt) { return ntt(buff, false, orig.Fr.fft, arguments); };
    Fr.ifft = function (buff, inType, outType, logger, loggerTxt) { return ntt(buff, true, orig.Fr.ifft, arguments); };

    Fr.batchApplyKey = async function (buff, first, inc, inType, outType) {
        if (!isBuf(buff) || buff.byteLength / Fr.n8 < minPoints) return orig.Fr.batchApplyKey.apply(Fr, arguments);
        const out = allocLike(buff, buff.byteLength);
        addon.applyKey(cid, pagesOf(buff), pagesOf(out), Math.floor(buff.byteLength / Fr.n8), Fr.e(first), Fr.e(inc));
        return out;
    };
    function batch(op, name) {
        return async function (buff) {
            if (!isBuf(buff)) return orig.Fr[name].apply(Fr, arguments);
            if (buff.byteLength % Fr.n8) throw new Error("Invalid buffer size");
            if (buff.byteLength / Fr.n8 < minPoints) return orig.Fr[name].apply(Fr, arguments);
            const out = (op == OP.INVERSE ? allocLikeSliced : allocLike)(buff, buff.byteLength);
            addon.frBatch(cid, op, pagesOf(buff), pagesOf(out), Math.floor(buff.byteLength / Fr.n8));
            return out;
        };
    }
    Fr.batchToMontgomery = batch(OP.TO_MONTGOMERY, "batchToMontgomery");
    Fr.batchFromMontgomery = batch(OP.FROM_MONTGOMERY, "batchFromMontgomery");
    Fr.batchInverse = batch(OP.INVERSE, "batchInverse");

    curve.__zkmi = { addon, cid, orig };
    return curve;
}

// Undo register(): restore the reference WASM entry points (used by A/B parity tests).
function unregister(curve) {
    if (!curve.__zkmi) return curve;
    const o = curve.__zkmi.orig;
    Object.assign(curve.G1, o.G1);
    Object.assign(curve.G2, o.G2);
    Object.assign(curve.Fr, o.Fr);
    delete curve.__zkmi;
    return curve;
}

// Patch both curves through snarkjs's own getter so that the cached instances are the ones patched.
// options.fused (r06): ALSO put the fused provers behind snarkjs's own prover entry points (installFused below).
async function registerAll(snarkjs, options) {
    const out = {};
    for (const name of ["bn128", "bls12381"]) out[name] = register(await snarkjs.curves.getCurveFromName(name), options);
    if (options && options.fused) out.fused = installFused(snarkjs, options);
    return out;
}

// ---- the fused provers behind snarkjs.groth16 / plonk / fflonk (opt-in: registerAll(snarkjs, { fused: true })) ------------------------------------
// Patching the curve object leaves the reference's own JavaScript around the bulk calls: at 2^20 constraints snarkjs.groth16.prove then takes 2.6 s, 0.08 s of
// it on the device (buildABC1, src/groth16_prove.js:147-187, is a single-threaded loop of per-element WASM calls). The prover functions themselves are reached
// through the module object — `snarkjs.groth16` is a frozen namespace, but the PROPERTY snarkjs.groth16 is writable — so a caller of
// snarkjs.groth16.prove / fullProve, snarkjs.plonk.prove / fullProve, snarkjs.fflonk.prove / fullProve gets the fused device provers (js/groth16_native.js,
// plonk_native.js, fflonk_native.js) with the reference's signature, inputs (paths, bytes, fastfile descriptors), blinding draws (curve.Fr.random, in the
// reference's order) and outputs: same proof for the same draws (tests/js/unmodified_gpu.js step 6). Keys stay resident per zkey (path or buffer identity).
// What is NOT redirected: snarkjs's own CLI (its action table calls the prover functions directly, not through the namespace) and
// `{ singleThread: true }` calls, which keep their private WASM curve. uninstallFused(snarkjs) restores the namespaces and frees the keys.
function installFused(snarkjs, options) {
    if (snarkjs.__zkmiFused) return snarkjs.__zkmiFused;
    const fs = require("fs");
    const { makeProver } = require("./groth16_native.js");
    const plonkN = require("./plonk_native.js"), fflonkN = require("./fflonk_native.js");
    const orig = { groth16: snarkjs.groth16, plonk: snarkjs.plonk, fflonk: snarkjs.fflonk };
    const prover = makeProver(snarkjs, options);
    const keys = new Map();                                  // PLONK / FFLONK keys resident per zkey source
    const idOf = (src) => (typeof src === "string" ? "file:" + src : (src && src.type === "file" ? "file:" + src.fileName : (src && src.type === "mem" ? src.data : src)));
    const bytesOf = (src) => {
        if (typeof src === "string") return new Uint8Array(fs.readFileSync(src));
        if (src instanceof Uint8Array) return src;
        if (src && src.type === "file") return new Uint8Array(fs.readFileSync(src.fileName));
        if (src && src.type === "mem") return (src.data instanceof Uint8Array) ? src.data : src.data.slice(0, src.data.byteLength);
        throw new Error("expected a path, the file's bytes or a fastfile descriptor");
    };
    const residentKey = (Cls, src) => {
        const id = idOf(src);
        let k = keys.get(id);
        if (!k) { k = new Cls(bytesOf(src), options); keys.set(id, k); }
        return k;
    };
    const single = (o) => !!(o && o.singleThread);
    async function polyProve(mod, Cls, nDraws, zkey, wtns) {
        const key = residentKey(Cls, zkey);
        const curve = await snarkjs.curves.getCurveFromName(key.curveName);
        const draws = [];
        for (let i = 0; i < nDraws; i++) draws.push(Uint8Array.from(curve.Fr.random()));     // src/plonk_prove.js:222-227 / src/fflonk_prove.js:321-324: all draws first, in order
        return mod.proveAsync(key, bytesOf(wtns), draws);
    }
    const fullOf = (prove) => async function (input, wasmFile, zkey, logger, wtnsCalcOptions, proverOptions) {      // src/groth16_fullprove.js:24-33 and its twins
        const wtns = { type: "mem" };
        await snarkjs.wtns.calculate(input, wasmFile, wtns, wtnsCalcOptions);
        return prove(zkey, wtns, logger, proverOptions);
    };
    const g16 = (zkey, wtns, logger, o) => (single(o) ? orig.groth16.prove(zkey, wtns, logger, o) : prover.prove(zkey, wtns));
    const pl = (zkey, wtns, logger, o) => (single(o) ? orig.plonk.prove(zkey, wtns, logger, o) : polyProve(plonkN, plonkN.PlonkKey, 11, zkey, wtns));
    const ff = (zkey, wtns, logger, o) => (single(o) ? orig.fflonk.prove(zkey, wtns, logger, o) : polyProve(fflonkN, fflonkN.FflonkKey, 9, zkey, wtns));
    snarkjs.groth16 = Object.freeze(Object.assign({}, orig.groth16, { prove: g16, fullProve: fullOf(g16) }));
    snarkjs.plonk = Object.freeze(Object.assign({}, orig.plonk, { prove: pl, fullProve: fullOf(pl) }));
    snarkjs.fflonk = Object.freeze(Object.assign({}, orig.fflonk, { prove: ff, fullProve: fullOf(ff) }));
    snarkjs.__zkmiFused = { orig, prover, keys };
    return snarkjs.__zkmiFused;
}
async function uninstallFused(snarkjs) {
    const st = snarkjs.__zkmiFused;
    if (!st) return;
    snarkjs.groth16 = st.orig.groth16; snarkjs.plonk = st.orig.plonk; snarkjs.fflonk = st.orig.fflonk;
    delete snarkjs.__zkmiFused;
    for (const k of st.keys.values()) { try { k.release(); } catch (e) { /* already released */ } }
    st.keys.clear();
    await st.prover.release();
}

module.exports = { register, unregister, registerAll, installFused, uninstallFused, loadAddon };
