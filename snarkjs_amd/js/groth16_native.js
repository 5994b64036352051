// snarkjs_amd/js/groth16_native.js — opt-in fused prover: the whole of groth16.prove (reference
// src/groth16_prove.js:28-144) between "sections read" and "proof points" in ONE device call (zkmi_groth16_prove),
// removing the single-threaded JS loops buildABC1 (:147-187) and the per-bulk-op host round trips.
// Same inputs and outputs as snarkjs.groth16.prove; snarkjs itself (its bundled binfile reader, Fr.random, G1.toObject)
// does everything else.
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

// Resident Groth16 keys live in a process-global native map: the key numbers are allocated module-wide, so that two provers in
// one process can never address each other's zkey (the library additionally refuses a descriptor that does not match the
// resident key's circuit).
let nextKey = 1;

// prover = makeProver(snarkjs, options) ; proof = await prover.prove(zkeyBytes, wtnsBytes)
function makeProver(snarkjs, options) {
    options = options || {};
    const addon = options.addon || loadAddon();
    addon.init(options.device === undefined ? 0 : options.device);
    const resident = new Map();                              // zkey Uint8Array -> cache key (base tables stay on the device)

    async function prove(zkeyBytes, wtnsBytes) {
        const zs = sections(zkeyBytes, "zkey"), ws = sections(wtnsBytes, "wtns");
        const hv = new DataView(zs[2][0].buffer, zs[2][0].byteOffset, zs[2][0].byteLength);
        if (new DataView(zs[1][0].buffer, zs[1][0].byteOffset, 4).getUint32(0, true) != 1) throw new Error("zkey file is not groth16");
        const n8q = hv.getUint32(0, true), n8r = hv.getUint32(4 + n8q, true);
        let o = 8 + n8q + n8r;
        const nVars = hv.getUint32(o, true), nPublic = hv.getUint32(o + 4, true), domainSize = hv.getUint32(o + 8, true);
        o += 12;
        const hdr = zs[2][0];
        const pt = (k) => { const v = hdr.subarray(o, o + k * n8q); o += k * n8q; return v; };
        const alpha1 = pt(2), beta1 = pt(2), beta2 = pt(4); pt(4); const delta1 = pt(2), delta2 = pt(4);
        const curve = await snarkjs.curves.getCurveFromName(n8q == 32 ? "bn128" : "bls12381");
        const wh = new DataView(ws[1][0].buffer, ws[1][0].byteOffset, ws[1][0].byteLength);
        const nWitness = wh.getUint32(4 + wh.getUint32(0, true), true);
        if (nWitness != nVars) throw new Error(`Invalid witness length. Circuit: ${nVars}, witness: ${nWitness}`);
        const witness = ws[2][0];
        if (witness.byteLength != nVars * n8r) throw new Error(`Invalid witness length. Circuit: ${nVars}, witness: ${witness.byteLength / n8r}`);
        for (const [sec, cnt, g] of [[5, nVars, 2], [6, nVars, 2], [7, nVars, 4], [8, nVars - nPublic - 1, 2], [9, domainSize, 2]])
            if (!zs[sec] || zs[sec][0].byteLength < cnt * g * n8q) throw new Error(`zkey section ${sec} is shorter than its header requires`);
        let key = resident.get(zkeyBytes), desc = curve.name == "bn128" ? 0 : 1;
        const fresh = !key;
        if (fresh) {
            key = nextKey++;
            desc = { curve: desc, nVars, nPublic, domainSize, coeffs: zs[4][0], A: zs[5][0], B1: zs[6][0], B2: zs[7][0], C: zs[8][0], H: zs[9][0],
                     alpha1, beta1, beta2, delta1, delta2 };
        }
        const r = curve.Fr.random(), s = curve.Fr.random();           // src/groth16_prove.js:103-104
        // options.async !== false: the proof runs on a libuv pool thread (addon.groth16ProveAsync) and the event loop keeps turning
        const res = (options.async !== false && typeof addon.groth16ProveAsync === "function") ? await addon.groth16ProveAsync(desc, key, witness, r, s)
                                                                                               : addon.groth16Prove(desc, key, witness, r, s);
        if (fresh) resident.set(zkeyBytes, key);                      // only a key that actually loaded is remembered
        const proof = {
            pi_a: curve.G1.toObject(res.pi_a), pi_b: curve.G2.toObject(res.pi_b), pi_c: curve.G1.toObject(res.pi_c),
            protocol: "groth16", curve: curve.name,
        };
        const publicSignals = [];
        for (let i = 1; i <= nPublic; i++) {
            let v = 0n;
            for (let b = n8r - 1; b >= 0; b--) v = (v << 8n) | BigInt(witness[i * n8r + b]);
            publicSignals.push(v.toString());
        }
        const str = (x) => Array.isArray(x) ? x.map(str) : x.toString();
        return { proof: { pi_a: str(proof.pi_a), pi_b: str(proof.pi_b), pi_c: str(proof.pi_c), protocol: proof.protocol, curve: proof.curve }, publicSignals };
    }
    function release() { for (const k of resident.values()) addon.groth16Release(k); resident.clear(); }
    return { prove, release };
}

module.exports = { makeProver };
