// tools/napi_wall.js — wall time THROUGH the N-API boundary (SURVEY.md 8d timing protocol: "hipEvent device time and wall time
// through the N-API call", bases resident and cold). Driven by bench.py on rank 0 (outside the timed region):
//   node tools/napi_wall.js <zkey> <wtns> <reps>
// Measures, with host buffers in and host results out (H2D of inputs and D2H of results inside the timed call):
//   groth16  addon.groth16Prove: cold = first call (zkey sections H2D + window-table build + proof), warm = key resident, only the
//            witness (32 B x nVars) crosses PCIe per proof
//            pipelined = two proofs in flight (addon.groth16Submit / groth16Collect), the witness upload of proof k+1 under proof k
//   msm      addon.msm on the A section (nVars points): cold (bases uploaded every call, cache not allowed) / resident (3rd+ call)
//   ntt      addon.ntt of domainSize elements (32 B x n in, 32 B x n out)
//   small    addon.msm / addon.ntt at 4, 64 and 1024 elements (what the drop-in boundary costs below the sizes the device is built for)
//   sharded  one proof over two worker processes (js/groth16_shards.js), chain outputs exchanged device to device ("peer") and through shared
//            host memory ("shm")
// Prints ONE JSON line.
"use strict";
const fs = require("fs"), path = require("path");
const addon = require(path.join(__dirname, "..", "snarkjs_amd", "napi", "zkmi_napi.node"));
const [zkeyPath, wtnsPath, repsArg, drawsHex] = process.argv.slice(2);
const reps = parseInt(repsArg || "5");
const now = () => Number(process.hrtime.bigint()) / 1e6;
const med = (a) => { const s = a.slice().sort((x, y) => x - y); return s[s.length >> 1]; };
addon.init(0);
// r06: the key is opened the way js/groth16_native.js opens it — sections read by offset from the file descriptor into pages of <= 1 GiB — so that
// this tool (and a Node host) can take a 2^24-constraint key (9.4 GB; sections of 1 - 2 GB: beyond one Node buffer)
const { openZkey, parseWtns } = require(path.join(__dirname, "..", "snarkjs_amd", "js", "groth16_native.js"));
let tOpen = now();
const zk = openZkey(zkeyPath);
tOpen = now() - tOpen;
const desc = zk.desc, cid = zk.curveId, n8q = zk.n8q, nVars = zk.nVars, nPublic = zk.nPublic, domainSize = zk.domainSize;
const wtns = new Uint8Array(fs.readFileSync(wtnsPath));
const witness = parseWtns(wtns, zk);
const BIG = fs.statSync(zkeyPath).size >= 2147483647 || !!process.env.ZKMI_NAPI_WALL_BIG;
const zs = { 5: desc.A };
let r = new Uint8Array(32), s = new Uint8Array(32); r[0] = 3; s[0] = 5;
if (drawsHex) [r, s] = drawsHex.split(",").map((h) => new Uint8Array(Buffer.from(h, "hex")));        // the caller's blinding draws: its own proof for them must equal ours
const out = { n_vars: nVars, domain: domainSize, reps, zkey_bytes: fs.statSync(zkeyPath).size, zkey_open_ms: +tOpen.toFixed(1),
              zkey_pages: ["coeffs", "A", "B1", "B2", "C", "H"].map((k) => (Array.isArray(desc[k]) ? desc[k].length : 1)) };
{
    const key = 424242;
    let t0 = now();
    addon.groth16Prove(desc, key, witness, r, s);
    out.groth16_cold_ms = +(now() - t0).toFixed(3);
    const t = [];
    for (let i = 0; i < reps; i++) { t0 = now(); addon.groth16Prove(cid, key, witness, r, s); t.push(now() - t0); }
    out.groth16_warm_ms = +med(t).toFixed(3);
    out.groth16_warm_h2d_bytes = witness.byteLength;
    // throughput mode through Node (js/groth16_native.js: proveMany): two proofs in flight, the witness of proof k+1 crosses PCIe on its slot's
    // stream while proof k computes (zkmi_groth16_submit / _collect); per-proof wall time over `reps * 4` proofs
    {
        const N = Math.max(8, reps * 4);
        for (let i = 0; i < 2; i++) addon.groth16Submit(key, witness, i & 1);            // size the second slot's buffers outside the timed region (as bench.py does)
        for (let i = 0; i < 2; i++) addon.groth16Collect(cid, key, i & 1, r, s);
        t0 = now();
        for (let i = 0; i < N; i++) {
            addon.groth16Submit(key, witness, i & 1);
            if (i) addon.groth16Collect(cid, key, (i - 1) & 1, r, s);
        }
        const last = addon.groth16Collect(cid, key, (N - 1) & 1, r, s);
        out.groth16_pipelined_ms = +((now() - t0) / N).toFixed(3);
        const ref = addon.groth16Prove(cid, key, witness, r, s);
        out.proof_sha256 = require("crypto").createHash("sha256").update(Buffer.concat([Buffer.from(ref.pi_a), Buffer.from(ref.pi_b), Buffer.from(ref.pi_c)])).digest("hex");
        out.groth16_pipelined_equals_serial = Buffer.from(last.pi_a).equals(Buffer.from(ref.pi_a)) && Buffer.from(last.pi_b).equals(Buffer.from(ref.pi_b)) && Buffer.from(last.pi_c).equals(Buffer.from(ref.pi_c));
    }
    addon.groth16Release(key);
}
if (BIG) { console.log(JSON.stringify(out)); process.exit(0); }      // a key beyond one buffer: the fused prover's figures only (the MSM / NTT / shard legs are measured at 2^20)
{
    const bases = zs[5], scalars = witness;
    const t = [], tr = [];
    for (let i = 0; i < reps; i++) { const t0 = now(); addon.msm(cid, 1, bases, scalars, nVars, 32, 0); t.push(now() - t0); }
    addon.msm(cid, 1, bases, scalars, nVars, 32, 1); addon.msm(cid, 1, bases, scalars, nVars, 32, 1);      // 1st sight, 2nd sight (table build)
    for (let i = 0; i < reps; i++) { const t0 = now(); addon.msm(cid, 1, bases, scalars, nVars, 32, 1); tr.push(now() - t0); }
    const ti = [];
    addon.msm(cid, 1, bases, scalars, nVars, 32, 3);                                                        // first sight under the promise: full check
    for (let i = 0; i < reps; i++) { const t0 = now(); addon.msm(cid, 1, bases, scalars, nVars, 32, 3); ti.push(now() - t0); }
    addon.releaseBases(0);
    out.g1_msm_cold_ms = +med(t).toFixed(3);                 // bases + scalars H2D every call
    out.g1_msm_resident_ms = +med(tr).toFixed(3);            // the default: scalars H2D + FULL content hash of the bases on the host, every call (ZKMI_BASES_CACHE)
    out.g1_msm_resident_immutable_ms = +med(ti).toFixed(3);  // opt-in: the caller promises not to edit the buffer, re-check by sample (ZKMI_BASES_IMMUTABLE)
    out.g1_msm_h2d_bytes_cold = bases.byteLength + scalars.byteLength;
}
{
    const n = domainSize, lg = Math.round(Math.log2(n));
    const x = new Uint8Array(n * 32), y = new Uint8Array(n * 32);
    x.set(witness.subarray(0, Math.min(witness.byteLength, x.byteLength)));
    const t = [];
    for (let i = 0; i < reps + 1; i++) { const t0 = now(); addon.ntt(cid, x, y, lg, 0, null, null); t.push(now() - t0); }
    out.ntt_ms = +med(t.slice(1)).toFixed(3);
    out.ntt_bytes_over_pcie = 2 * n * 32;
}
// small inputs at the drop-in boundary (VERDICT r03 weak #2: what does a 3-point G1.multiExpAffine — the verifier's nPublic-point MSM, PLONK's tiny
// Lagrange transforms — cost through sort + accumulate + reduce?): host buffers in and out, median of 20 calls each
{
    const sm = {}, sn = {};
    for (const k of [4, 64, 1024]) {
        const bases = (Array.isArray(zs[5]) ? zs[5][0] : zs[5]).subarray(0, k * 2 * n8q), scalars = witness.subarray(32, 32 + k * 32);
        let t = [];
        for (let i = 0; i < 22; i++) { const t0 = now(); addon.msm(cid, 1, bases, scalars, k, 32, 0); t.push(now() - t0); }
        sm[k] = +med(t.slice(2)).toFixed(4);
        const lg = Math.round(Math.log2(k)), x = new Uint8Array(k * 32), y = new Uint8Array(k * 32);
        x.set(witness.subarray(0, k * 32));
        t = [];
        for (let i = 0; i < 22; i++) { const t0 = now(); addon.ntt(cid, x, y, lg, 0, null, null); t.push(now() - t0); }
        sn[k] = +med(t.slice(2)).toFixed(4);
    }
    out.msm_small_ms = sm;                                   // G1.multiExpAffine of 4 / 64 / 1024 points, bases not cached
    out.ntt_small_ms = sn;                                   // Fr.fft of 4 / 64 / 1024 elements
}
// ONE proof over two worker PROCESSES (js/groth16_shards.js), both on device 0 when the box has one GPU: the protocol and the peer copies are
// the multi-GPU ones, the placement is not. exchange "peer" = zkmi_ipc_* + zkmi_peer_copy (device to device), "shm" = through pinned host pages.
async function sharded() {
    const { ShardedProver } = require(path.join(__dirname, "..", "snarkjs_amd", "js", "groth16_shards.js"));
    const ndev = addon.deviceCount();
    for (const exchange of ["peer", "shm"]) {
        let sp = null;
        try {
            sp = new ShardedProver({ world: 2, zkeyPath, exchange, devices: ndev >= 2 ? [0, 1] : [0, 0] });
            await sp.ready();
            const t = [];
            let tl = null;
            for (let i = 0; i < reps + 1; i++) { const t0 = now(); const res = await sp.prove(wtns, { r, s }); t.push(now() - t0); tl = res.timeline_ms; }
            out[exchange === "peer" ? "groth16_sharded_2proc_ms" : "groth16_sharded_2proc_shm_ms"] = +med(t.slice(1)).toFixed(3);
            if (exchange === "peer") { out.groth16_sharded_2proc_timeline_ms = tl; out.groth16_sharded_2proc_devices = ndev >= 2 ? [0, 1] : [0, 0]; }
        } catch (e) { out[`groth16_sharded_2proc_${exchange}_error`] = String(e && e.message || e).slice(0, 300); }
        if (sp) { try { await sp.close(); } catch (e) { /* best effort */ } }
    }
}
if (process.env.ZKMI_NAPI_WALL_NO_SHARDS) { console.log(JSON.stringify(out)); process.exit(0); }
sharded().then(() => { console.log(JSON.stringify(out)); process.exit(0); }, (e) => { out.groth16_sharded_error = String(e); console.log(JSON.stringify(out)); process.exit(0); });
