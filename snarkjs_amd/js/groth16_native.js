// snarkjs_amd/js/groth16_native.js — opt-in fused prover: the whole of groth16.prove (reference
// src/groth16_prove.js:28-144) between "sections read" and "proof points" in ONE device call (zkmi_groth16_prove),
// removing the single-threaded JS loops buildABC1 (:147-187) and the per-bulk-op host round trips.
// Same inputs and outputs as snarkjs.groth16.prove; snarkjs itself (its bundled binfile reader, Fr.random, G1.toObject)
// does everything else.
//
//   const prover = makeProver(snarkjs);                       // snarkjs: the module a caller already uses (curves, Fr.random, toObject)
//   const { proof, publicSignals } = await prover.prove(zkeyBytes, wtnsBytes);
//   const results = await prover.proveMany(zkeyBytes, [wtns0, wtns1, ...]);      // throughput mode: two proofs in flight on the GPU
//   prover.release();
"use strict";
const { loadAddon } = require("./register.js");

function sections(data, magic) {                           // @iden3/binfileutils container (SURVEY.md §2 #17)
    const dv = new DataView(data.buffer, data.byteOffset, data.byteLength);
    for (let i = 0; i < 4; i++) if (data[i] != magic.charCodeAt(i)) throw new Error(data.length + ": Invalid File format");
    const n = dv.getUint32(8, true);
    const out = {};
    let off = 12;
    for (let i = 0; i < n; i++) {
        const t = dv.getUint32(off, true);
        const len = Number(dv.getBigUint64(off + 4, true));
        off += 12;
        (out[t] = out[t] || []).push(data.subarray(off, off + len));
        off += len;
    }
    return out;
}

// zkey header + sections -> the descriptor the addon takes (src/zkey_utils.js:229-259); throws the reference's messages
function parseZkey(zkeyBytes) {
    const zs = sections(zkeyBytes, "zkey");
    if (new DataView(zs[1][0].buffer, zs[1][0].byteOffset, 4).getUint32(0, true) != 1) throw new Error("zkey file is not groth16");
    const hdr = zs[2][0], hv = new DataView(hdr.buffer, hdr.byteOffset, hdr.byteLength);
    const n8q = hv.getUint32(0, true), n8r = hv.getUint32(4 + n8q, true);
    let o = 8 + n8q + n8r;
    const nVars = hv.getUint32(o, true), nPublic = hv.getUint32(o + 4, true), domainSize = hv.getUint32(o + 8, true);
    o += 12;
    const pt = (k) => { const v = hdr.subarray(o, o + k * n8q); o += k * n8q; return v; };
    const alpha1 = pt(2), beta1 = pt(2), beta2 = pt(4); pt(4); const delta1 = pt(2), delta2 = pt(4);
    for (const [sec, cnt, g] of [[5, nVars, 2], [6, nVars, 2], [7, nVars, 4], [8, nVars - nPublic - 1, 2], [9, domainSize, 2]])
        if (!zs[sec] || zs[sec][0].byteLength < cnt * g * n8q) throw new Error(`zkey section ${sec} is shorter than its header requires`);
    const curveName = n8q == 32 ? "bn128" : "bls12381";
    return { n8q, n8r, nVars, nPublic, domainSize, curveName, curveId: n8q == 32 ? 0 : 1,
             desc: { curve: n8q == 32 ? 0 : 1, nVars, nPublic, domainSize, coeffs: zs[4][0], A: zs[5][0], B1: zs[6][0], B2: zs[7][0], C: zs[8][0], H: zs[9][0],
                     alpha1, beta1, beta2, delta1, delta2 } };
}
// wtns sections -> the witness buffer (src/wtns_utils.js:62-72), checked against the circuit like src/groth16_prove.js:45-47
function parseWtns(wtnsBytes, zk) {
    const ws = sections(wtnsBytes, "wtns");
    const wh = new DataView(ws[1][0].buffer, ws[1][0].byteOffset, ws[1][0].byteLength);
    const nWitness = wh.getUint32(4 + wh.getUint32(0, true), true);
    if (nWitness != zk.nVars) throw new Error(`Invalid witness length. Circuit: ${zk.nVars}, witness: ${nWitness}`);
    const witness = ws[2][0];
    if (witness.byteLength != zk.nVars * zk.n8r) throw new Error(`Invalid witness length. Circuit: ${zk.nVars}, witness: ${witness.byteLength / zk.n8r}`);
    return witness;
}

// Resident Groth16 keys live in a process-global native map: the key numbers are allocated module-wide, so that two provers in
// one process can never address each other's zkey (the library additionally refuses a descriptor that does not match the
// resident key's circuit).
let nextKey = 1;

function makeProver(snarkjs, options) {
    options = options || {};
    const addon = options.addon || loadAddon();
    addon.init(options.device === undefined ? 0 : options.device);
    const useAsync = options.async !== false;
    // zkey Uint8Array -> Promise of its resident key. The promise is stored BEFORE the load is awaited: concurrent prove() calls on a
    // zkey that is not resident yet share ONE load (two loads would leave a key — GBs of window tables at 2^20 — on the device that
    // release() never frees); a load that fails is forgotten and whatever it left on the device is released.
    const resident = new Map();

    function ensureKey(zkeyBytes) {
        let p = resident.get(zkeyBytes);
        if (p) return p;
        p = (async () => {
            const zk = parseZkey(zkeyBytes);
            zk.key = nextKey++;
            zk.curve = await snarkjs.curves.getCurveFromName(zk.curveName);
            try {
                if (useAsync && typeof addon.groth16LoadAsync === "function") await addon.groth16LoadAsync(zk.desc, zk.key);
                else await addon.groth16Load(zk.desc, zk.key);
            } catch (e) {
                resident.delete(zkeyBytes);
                try { addon.groth16Release(zk.key); } catch (e2) { /* nothing was loaded */ }
                throw e;
            }
            return zk;
        })();
        resident.set(zkeyBytes, p);
        return p;
    }

    function finishProof(zk, witness, res) {
        const curve = zk.curve;
        const str = (x) => Array.isArray(x) ? x.map(str) : x.toString();
        const publicSignals = [];
        for (let i = 1; i <= zk.nPublic; i++) {
            let v = 0n;
            for (let b = zk.n8r - 1; b >= 0; b--) v = (v << 8n) | BigInt(witness[i * zk.n8r + b]);
            publicSignals.push(v.toString());
        }
        return { proof: { pi_a: str(curve.G1.toObject(res.pi_a)), pi_b: str(curve.G2.toObject(res.pi_b)), pi_c: str(curve.G1.toObject(res.pi_c)), protocol: "groth16", curve: curve.name },
                 publicSignals };
    }

    // one proof; the event loop keeps turning while it runs (options.async !== false: the call runs on a libuv pool thread)
    async function prove(zkeyBytes, wtnsBytes) {
        const zk = await ensureKey(zkeyBytes);
        const witness = parseWtns(wtnsBytes, zk);
        const r = zk.curve.Fr.random(), s = zk.curve.Fr.random();           // src/groth16_prove.js:103-104
        const res = (useAsync && typeof addon.groth16ProveAsync === "function") ? await addon.groth16ProveAsync(zk.curveId, zk.key, witness, r, s)
                                                                                : await addon.groth16Prove(zk.curveId, zk.key, witness, r, s);
        return finishProof(zk, witness, res);
    }

    // Throughput mode: one proof per witness, TWO in flight (zkmi_groth16_submit / _collect): the witness of proof k+1 crosses PCIe on its
    // slot's stream and its kernels are enqueued while proof k computes; the latency-bound tail of proof k (bucket reductions, result copies,
    // host folds) runs underneath the front of proof k+1. Results come back in input order; blinding draws are taken in proof order.
    async function proveMany(zkeyBytes, wtnsList) {
        const zk = await ensureKey(zkeyBytes);
        const submit = (useAsync && typeof addon.groth16SubmitAsync === "function") ? addon.groth16SubmitAsync : addon.groth16Submit;
        const collect = (useAsync && typeof addon.groth16CollectAsync === "function") ? addon.groth16CollectAsync : addon.groth16Collect;
        const out = new Array(wtnsList.length), pending = [];
        try {
            for (let i = 0; i < wtnsList.length; i++) {
                const witness = parseWtns(wtnsList[i], zk);
                const r = zk.curve.Fr.random(), s = zk.curve.Fr.random();
                await submit(zk.key, witness, i & 1);
                pending.push({ i, witness, r, s });
                if (pending.length == 2) { const p = pending.shift(); out[p.i] = finishProof(zk, p.witness, await collect(zk.curveId, zk.key, p.i & 1, p.r, p.s)); }
            }
            while (pending.length) { const p = pending.shift(); out[p.i] = finishProof(zk, p.witness, await collect(zk.curveId, zk.key, p.i & 1, p.r, p.s)); }
        } catch (e) {
            // leave no proof in flight behind an error: drain the slots (their results are discarded: any blinding values will do)
            const z = new Uint8Array(32);
            for (const p of pending) { try { await collect(zk.curveId, zk.key, p.i & 1, z, z); } catch (e2) { /* already failed */ } }
            throw e;
        }
        return out;
    }

    async function release() {
        const all = Array.from(resident.values());
        resident.clear();
        for (const p of all) { try { const zk = await p; addon.groth16Release(zk.key); } catch (e) { /* a failed load holds nothing */ } }
    }
    return { prove, proveMany, release };
}

module.exports = { makeProver, parseZkey, parseWtns, sections };
