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

// ---- reading a zkey the way the reference does: section by section, by offset (src/groth16_prove.js:29-33, :57-59, :84-100) -----------------
// groth16Prove opens the file with binFileUtils.readBinFile and pulls each section with readSection, which hands back a Uint8Array below 2^30
// bytes and a BigBuffer (pages of <= 1 GiB) from there on: sections 4 - 9 of a 2^24-constraint key are 1 - 2 GB each and the whole file is
// 9.4 GB, more than one Node buffer can hold (2 GiB - 1 under Node 12). A source is
//   a Uint8Array with the file's bytes | a path | { type: "file", fileName } | { type: "mem", data } (fastfile's descriptors; data may be a BigBuffer).
function readerOf(src) {
    if (typeof src === "string") src = { type: "file", fileName: src };
    if (src instanceof Uint8Array) src = { type: "mem", data: src };
    if (src && src.type === "mem" && src.data) {
        const d = src.data;
        if (d instanceof Uint8Array) return { size: d.byteLength, read: (pos, len) => d.subarray(pos, pos + len), close() {} };
        if (Array.isArray(d.buffers) && typeof d.slice === "function")                 // ffjavascript BigBuffer (min.js:1@183423)
            return { size: d.byteLength, read: (pos, len) => { const x = d.slice(pos, pos + len); if (x.buffers) throw new Error("zkey: page larger than a BigBuffer page"); return x; }, close() {} };
    }
    if (src && src.type === "file" && src.fileName) {
        const fs = require("fs");
        const fd = fs.openSync(src.fileName, "r");
        return { size: fs.fstatSync(fd).size, close() { fs.closeSync(fd); },
                 read(pos, len) {
                     const b = Buffer.allocUnsafe(len);
                     for (let got = 0; got < len;) { const k = fs.readSync(fd, b, got, len - got, pos + got); if (k <= 0) throw new Error(src.fileName + ": Invalid File format (truncated)"); got += k; }
                     return new Uint8Array(b.buffer, b.byteOffset, len);
                 } };
    }
    throw new Error("zkey: expected a Uint8Array, a path, { type: \"file\", fileName } or { type: \"mem\", data }");
}
// section table of an @iden3/binfileutils container: type -> [{ pos, len }]
function sectionTable(rd, magic) {
    const h = rd.read(0, 12);
    for (let i = 0; i < 4; i++) if (h[i] != magic.charCodeAt(i)) throw new Error(rd.size + ": Invalid File format");
    const n = new DataView(h.buffer, h.byteOffset, 12).getUint32(8, true), tab = {};
    let pos = 12;
    for (let i = 0; i < n; i++) {
        if (pos + 12 > rd.size) throw new Error(rd.size + ": Invalid File format");
        const sh = rd.read(pos, 12), dv = new DataView(sh.buffer, sh.byteOffset, 12);
        const t = dv.getUint32(0, true), len = Number(dv.getBigUint64(4, true));
        (tab[t] = tab[t] || []).push({ pos: pos + 12, len });
        pos += 12 + len;
    }
    return tab;
}
const BIG_PAGE = 1 << 30;                                 // ffjavascript's BigBuffer page; readSection returns a plain Uint8Array below it
// one section as the addon takes it: a Uint8Array, or an array of pages; with `need` = [lo, hi) only that byte range is read and the rest of the
// section is passed as gaps (numbers) — a shard worker reads 1/world of the base sections
function sectionPages(rd, s, pageBytes, need) {
    if (!need && s.len <= pageBytes) return rd.read(s.pos, s.len);
    const [lo, hi] = need || [0, s.len], out = [];
    if (lo > 0) out.push(lo);
    for (let o = lo; o < hi; o += pageBytes) out.push(rd.read(s.pos + o, Math.min(pageBytes, hi - o)));
    if (hi < s.len) out.push(s.len - hi);
    if (!out.length) out.push(new Uint8Array(0));
    return out;
}
// what a caller that has ALREADY read the sections with the reference's readSection passes on: Uint8Array stays, BigBuffer -> its pages
const toPages = (x) => (x && Array.isArray(x.buffers)) ? x.buffers : x;

// zkey header + sections -> the descriptor the addon takes (src/zkey_utils.js:229-259); throws the reference's messages.
// opts.pageBytes: page size of the bulk sections (default 2^30); opts.shard = { vLo, vHi, hLo, hHi }: read only this shard's ranges of sections 5 - 9;
// opts.headerOnly: no bulk section is read (desc = null).
function openZkey(src, opts) {
    opts = opts || {};
    const rd = readerOf(src);
    try {
        const tab = sectionTable(rd, "zkey");
        for (const id of [1, 2, 4, 5, 6, 7, 8, 9]) if (!tab[id]) throw new Error(`zkey: Missing section ${id}`);
        const s1 = rd.read(tab[1][0].pos, 4);
        if (new DataView(s1.buffer, s1.byteOffset, 4).getUint32(0, true) != 1) throw new Error("zkey file is not groth16");
        const hdr = Uint8Array.from(rd.read(tab[2][0].pos, tab[2][0].len)), hv = new DataView(hdr.buffer, hdr.byteOffset, hdr.byteLength);
        const n8q = hv.getUint32(0, true), n8r = hv.getUint32(4 + n8q, true);
        let o = 8 + n8q + n8r;
        const nVars = hv.getUint32(o, true), nPublic = hv.getUint32(o + 4, true), domainSize = hv.getUint32(o + 8, true);
        o += 12;
        const pt = (k) => { const v = hdr.subarray(o, o + k * n8q); o += k * n8q; return v; };
        const alpha1 = pt(2), beta1 = pt(2), beta2 = pt(4); pt(4); const delta1 = pt(2), delta2 = pt(4);
        for (const [sec, cnt, g] of [[5, nVars, 2], [6, nVars, 2], [7, nVars, 4], [8, nVars - nPublic - 1, 2], [9, domainSize, 2]])
            if (tab[sec][0].len < cnt * g * n8q) throw new Error(`zkey section ${sec} is shorter than its header requires`);
        const curveName = n8q == 32 ? "bn128" : "bls12381";
        const zk = { n8q, n8r, nVars, nPublic, domainSize, curveName, curveId: n8q == 32 ? 0 : 1, desc: null };
        if (opts.headerOnly) return zk;
        const page = opts.pageBytes || BIG_PAGE, sh = opts.shard || null, g1 = 2 * n8q, fc = nPublic + 1;
        const need = sh ? { 5: [sh.vLo * g1, sh.vHi * g1], 6: [sh.vLo * g1, sh.vHi * g1], 7: [sh.vLo * 2 * g1, sh.vHi * 2 * g1],
                            8: [Math.max(0, sh.vLo - fc) * g1, Math.max(0, sh.vHi - fc) * g1], 9: [sh.hLo * g1, sh.hHi * g1] } : {};
        const sec = (id) => sectionPages(rd, tab[id][0], page, need[id]);
        zk.desc = { curve: zk.curveId, nVars, nPublic, domainSize, coeffs: sec(4), A: sec(5), B1: sec(6), B2: sec(7), C: sec(8), H: sec(9), alpha1, beta1, beta2, delta1, delta2 };
        return zk;
    } finally { rd.close(); }
}
const parseZkey = (zkeyBytes, opts) => openZkey(zkeyBytes, opts);
// the same descriptor from sections a caller read itself with the reference's binFileUtils.readSection (Uint8Array | BigBuffer each)
function descFromSections(zk, secs) {
    return { curve: zk.curveId, nVars: zk.nVars, nPublic: zk.nPublic, domainSize: zk.domainSize, coeffs: toPages(secs[4]), A: toPages(secs[5]), B1: toPages(secs[6]),
             B2: toPages(secs[7]), C: toPages(secs[8]), H: toPages(secs[9]), alpha1: secs.alpha1, beta1: secs.beta1, beta2: secs.beta2, delta1: secs.delta1, delta2: secs.delta2 };
}
// wtns sections -> the witness buffer (src/wtns_utils.js:62-72), checked against the circuit like src/groth16_prove.js:45-47
function parseWtns(wtnsBytes, zk) {
    if (typeof wtnsBytes === "string") wtnsBytes = new Uint8Array(require("fs").readFileSync(wtnsBytes));
    else if (wtnsBytes && wtnsBytes.type === "file") wtnsBytes = new Uint8Array(require("fs").readFileSync(wtnsBytes.fileName));
    else if (wtnsBytes && wtnsBytes.type === "mem") wtnsBytes = wtnsBytes.data;
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

// ---- the process-wide two-slot pipeline (r06) --------------------------------------------------------------------------------------------------
// The library has two pipeline slots per process (zkmi_groth16_submit / _collect): a proof enqueued in one slot computes while the latency-bound tail of the proof in
// the other slot (bucket reductions, result copies, host folds) finishes. Until r05 only proveMany used them; a server that calls prove() once per request got one
// proof at a time however many requests were waiting. Now every proof of every prover of this process goes through ONE queue and one pump: requests are submitted in
// arrival order into alternating slots, never more than two in flight, the older one collected before a third is submitted — the order proveMany always used
// (submit0 submit1 collect0 submit0 collect1 ...) — so concurrent prove() calls pipeline by themselves (10.8 -> 9.5 ms per proof at 2^20) and proveMany is a loop over
// the same queue. A lone request with nothing in flight takes the one-call path. A job that fails rejects its own promise only; the slot it held is free again.
const pipeQueue = [];
let pumping = false;
function pipelined(job) {
    return new Promise((resolve, reject) => {
        pipeQueue.push(Object.assign(job, { resolve, reject }));
        if (!pumping) { pumping = true; Promise.resolve().then(pump); }        // started behind the current turn: requests made in the same turn are all in the queue when it looks
    });
}
async function pump() {
    const flight = [];                                     // submitted, not collected: oldest first
    try {
        while (pipeQueue.length || flight.length) {
            if (pipeQueue.length == 1 && !flight.length && pipeQueue[0].single) {
                // a lone request with nothing in flight: the one-call path (zkmi_groth16_prove) — measured 4 ms faster per isolated proof than submit + collect from Node
                // (11.1 against 15.2 ms at 2^20); requests that arrive meanwhile wait in the queue and pipeline from the next turn on
                const job = pipeQueue.shift();
                try { job.resolve(await job.single(job.curveId, job.key, job.witness, job.r, job.s)); } catch (e) { job.reject(e); }
                continue;
            }
            if (pipeQueue.length && flight.length < 2) {
                const job = pipeQueue.shift(), slot = flight.length ? 1 - flight[0].slot : 0;
                try { await job.submit(job.key, job.witness, slot); flight.push({ job, slot }); } catch (e) { job.reject(e); }
                continue;
            }
            const { job, slot } = flight.shift();
            try { job.resolve(await job.collect(job.curveId, job.key, slot, job.r, job.s)); } catch (e) { job.reject(e); }
        }
    } finally { pumping = false; }
}

function makeProver(snarkjs, options) {
    options = options || {};
    const addon = options.addon || loadAddon();
    addon.init(options.device === undefined ? 0 : options.device);
    const useAsync = options.async !== false;
    // zkey source (Uint8Array | path | fastfile descriptor: readerOf above) -> Promise of its resident key. The promise is stored BEFORE the load is awaited: concurrent prove() calls on a
    // zkey that is not resident yet share ONE load (two loads would leave a key — GBs of window tables at 2^20 — on the device that
    // release() never frees); a load that fails is forgotten and whatever it left on the device is released.
    const resident = new Map();

    const keyOf = (src) => (src && typeof src === "object" && !(src instanceof Uint8Array) && src.type === "file") ? "file:" + src.fileName : (src && src.type === "mem" ? src.data : src);
    function ensureKey(zkeySrc) {
        const zkeyBytes = keyOf(zkeySrc);
        let p = resident.get(zkeyBytes);
        if (p) return p;
        p = (async () => {
            const zk = openZkey(zkeySrc, { pageBytes: options.pageBytes });     // sections read by offset, in pages: any key size (2^24: 9.4 GB)
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
            zk.desc = null;                                       // the host copy of the sections is not needed once the key is resident
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

    const submitFn = (useAsync && typeof addon.groth16SubmitAsync === "function") ? addon.groth16SubmitAsync : addon.groth16Submit;
    const collectFn = (useAsync && typeof addon.groth16CollectAsync === "function") ? addon.groth16CollectAsync : addon.groth16Collect;
    const canPipe = typeof submitFn === "function" && typeof collectFn === "function" && options.pipeline !== false;
    // witness parsed and blinding values drawn HERE, in call order (src/groth16_prove.js:103-104); the device part goes through the process-wide pipeline above
    function start(zk, wtnsBytes) {
        const witness = parseWtns(wtnsBytes, zk);
        const r = zk.curve.Fr.random(), s = zk.curve.Fr.random();
        const single = (useAsync && typeof addon.groth16ProveAsync === "function") ? addon.groth16ProveAsync : (typeof addon.groth16Prove === "function" ? addon.groth16Prove : null);
        const res = canPipe ? pipelined({ key: zk.key, curveId: zk.curveId, witness, r, s, submit: submitFn, collect: collectFn, single })
                            : Promise.resolve((useAsync && typeof addon.groth16ProveAsync === "function") ? addon.groth16ProveAsync(zk.curveId, zk.key, witness, r, s)
                                                                                                     : addon.groth16Prove(zk.curveId, zk.key, witness, r, s));
        return res.then((pts) => finishProof(zk, witness, pts));
    }
    // one proof; the event loop keeps turning while it runs (options.async !== false: the calls run on a libuv pool thread). Concurrent calls — of this prover or any other
    // in the process — share the two pipeline slots: two proofs in flight as soon as two requests are waiting ({ pipeline: false }: one blocking call per proof, as before r06).
    // zkeyBytes / wtnsBytes: the files' bytes, paths, or fastfile descriptors — what snarkjs.groth16.prove takes (src/groth16_prove.js:28)
    async function prove(zkeyBytes, wtnsBytes) {
        const zk = await ensureKey(zkeyBytes);
        return start(zk, wtnsBytes);
    }

    // Throughput mode: one proof per witness, TWO in flight (zkmi_groth16_submit / _collect): the witness of proof k+1 crosses PCIe on its
    // slot's stream and its kernels are enqueued while proof k computes; the latency-bound tail of proof k (bucket reductions, result copies,
    // host folds) runs underneath the front of proof k+1. Results come back in input order; blinding draws are taken in proof order. r06: a loop over the same
    // queue prove() uses; a witness that does not parse fails the call before anything of it is submitted, the proofs already queued still run to their end.
    async function proveMany(zkeyBytes, wtnsList) {
        const zk = await ensureKey(zkeyBytes);
        const running = [];
        let failed = null;
        for (const w of wtnsList) {
            try { running.push(start(zk, w)); } catch (e) { failed = e; break; }
        }
        const settled = await Promise.all(running.map((p) => p.then((v) => ({ v }), (e) => ({ e }))));       // leave no proof in flight behind an error
        if (failed) throw failed;
        for (const x of settled) if (x.e) throw x.e;
        return settled.map((x) => x.v);
    }

    async function release() {
        const all = Array.from(resident.values());
        resident.clear();
        for (const p of all) { try { const zk = await p; addon.groth16Release(zk.key); } catch (e) { /* a failed load holds nothing */ } }
    }
    return { prove, proveMany, release };
}

module.exports = { makeProver, parseZkey, openZkey, descFromSections, toPages, parseWtns, sections };
