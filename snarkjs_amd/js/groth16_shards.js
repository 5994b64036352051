// snarkjs_amd/js/groth16_shards.js — ONE Groth16 proof over several GPUs from Node.js (BASELINE configs[2]; north star: "host side stays
// Node.js"). The Node twin of snarkjs_amd/distributed.py: one worker PROCESS per GPU (child_process.fork — the library binds one device per
// process), the same split of the work. The bulk exchange runs GPU to GPU through the C-ABI's peer layer (include/zkmi.h: zkmi_ipc_export /
// zkmi_ipc_open / zkmi_peer_copy — xGMI between two GPUs, HBM inside one) where distributed.py uses RCCL send / recv.
//
// What the reference does with its workers (ffjavascript engine_multiexp, build/snarkjs.min.js:1@214651): cut a multiExp into contiguous index
// chunks, one per worker, add the chunk results on the host. Here a chunk is the base-index range of a GPU's key shard, and the transforms are
// split too:
//   0. once: the owner of chain c (rank c % world) exports the device buffer that will hold the chain's output; every other worker opens the
//      handle (exchange "peer", the default when the addon has ipcExport). exchange "shm" (r03; kept for hosts without peer access) copies
//      the outputs through page-locked POSIX shared memory instead: owner GPU -> host pages -> the other GPUs over PCIe;
//   1. the parent writes the witness into a 0600 shared-memory region (never into a file); every worker uploads it; the owner of chain c runs
//      buildABC + iNTT -> coset -> NTT of chain c (src/groth16_prove.js:64-76: A, B, C are independent until joinABC) into its exported buffer;
//   2. meanwhile every worker runs the witness-side half of its shard's MSMs (A, B1, B2, C need the witness only: zkmi_groth16_sums_w_dev);
//   3. when the three chains are complete each worker PULLS its slice [h_lo, h_hi) of them device to device (zkmi_peer_copy), joins it into its
//      H-MSM scalars (zkmi_groth16_join_abc_dev) and runs the H half (zkmi_groth16_sums_h_dev): 7 x 3 x n8q bytes of partial sums per worker;
//   4. worker 0 adds the sums of all workers point by point in rank order (zkmi_point_add) and applies blinding + toAffine
//      (zkmi_groth16_finish, src/groth16_prove.js:103-132).
// Control messages travel over the fork IPC channel (a few hundred bytes each). A chain buffer is overwritten only by the next proof, which
// the parent starts after this one has delivered every worker's sums — i.e. after every pull has completed.
// prove() calls are serialised (the workers hold ONE witness buffer, one set of chain buffers and one pipeline slot); after a worker error the
// parent resets every worker (zkmi_groth16_reset) before the next proof; a worker that dies rejects every pending and future proof.
//
//   const { ShardedProver } = require("snarkjs_amd/js/groth16_shards.js");
//   const sp = new ShardedProver({ world: 8, zkeyPath });          // forks the workers, every worker loads its key shard
//   await sp.ready();
//   const { proof, publicSignals } = await sp.prove(wtnsBytes, { r, s });      // r, s: Montgomery Fr bytes (default: fresh random draws)
//   await sp.close();
"use strict";
const path = require("path"), fs = require("fs"), crypto = require("crypto");
const { fork } = require("child_process");
const { openZkey, parseWtns } = require("./groth16_native.js");

const R = { 0: 21888242871839275222246405745257275088548364400416034343698204186575808495617n, 1: 52435875175126190479447740508185965837690552500527637822603658699938581184513n };
const Q = { 0: 21888242871839275222246405745257275088696311157297823662689037894645226208583n,
            1: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaabn };
const fromLE = (b) => { let v = 0n; for (let i = b.length - 1; i >= 0; i--) v = (v << 8n) | BigInt(b[i]); return v; };
const toLE = (v, n) => { const o = new Uint8Array(n); for (let i = 0; i < n; i++) { o[i] = Number(v & 0xffn); v >>= 8n; } return o; };
function modinv(a, m) { let [r0, r1, s0, s1] = [((a % m) + m) % m, m, 1n, 0n]; while (r1 !== 0n) { const q = r0 / r1; [r0, r1] = [r1, r0 - q * r1]; [s0, s1] = [s1, s0 - q * s1]; } return ((s0 % m) + m) % m; }
// contiguous slice [lo, hi) of n terms owned by `rank` (ceil split, like the reference's chunking; = distributed.py: shard_range)
function shardRange(n, rank, world) { const per = Math.ceil(n / world), lo = Math.min(n, rank * per); return [lo, Math.min(n, lo + per)]; }
const chainOwner = (c, world) => c % world;
// affine Montgomery bytes -> the decimal strings of G1.toObject / G2.toObject (projective [x, y, "1"])
function pointToObject(cid, group, bytes) {
    const q = Q[cid], n8q = cid === 0 ? 32 : 48, ri = modinv((1n << BigInt(8 * n8q)) % q, q);
    const el = (k) => (fromLE(bytes.subarray(k * n8q, (k + 1) * n8q)) * ri % q).toString();
    const zero = bytes.every((x) => x === 0);
    if (group === 1) return zero ? ["0", "1", "0"] : [el(0), el(1), "1"];
    return zero ? [["0", "0"], ["1", "0"], ["0", "0"]] : [[el(0), el(1)], [el(2), el(3)], ["1", "0"]];
}
// a uniform Fr element in Montgomery form (the reference draws with its ChaCha: curve.Fr.random(), src/groth16_prove.js:103-104)
function randomFrMont(cid) { const r = R[cid]; return toLE((fromLE(crypto.randomBytes(64)) % r) * ((1n << 256n) % r) % r, 32); }

// ---- worker process -----------------------------------------------------------------------------------------------------------------------
async function workerMain() {
    const cfg = JSON.parse(process.env.ZKMI_SHARD_CFG);
    const addon = require(cfg.addonPath);
    const { rank, world } = cfg;
    const peer = cfg.exchange === "peer";
    await addon.init(cfg.devices[rank]);
    // the header first, then ONLY this shard's byte ranges of the five base sections (1 / world of them; the rest is passed as gaps) and the whole
    // coefficient section, by offset from the file: a 2^24 key (9.4 GB) never exists in one buffer, and 8 workers read it once between them
    const hd = openZkey(cfg.zkeyPath, { headerOnly: true });
    const n = hd.domainSize, m = hd.nVars, cid = hd.curveId, key = 1;
    const [vLo, vHi] = shardRange(m, rank, world), [hLo, hHi] = shardRange(n, rank, world);
    let zk = openZkey(cfg.zkeyPath, { shard: { vLo, vHi, hLo, hHi }, pageBytes: cfg.pageBytes || undefined });
    await addon.groth16LoadShard(zk.desc, key, vLo, vHi, hLo, hHi);
    zk = null;
    const owned = [0, 1, 2].filter((c) => chainOwner(c, world) === rank);
    const shmW = addon.shmMap(`${cfg.shmPrefix}_w`, m * 32, false);                  // the witness: 0600 shared memory written by the parent
    const shm = peer ? null : [0, 1, 2].map((c) => addon.shmMap(`${cfg.shmPrefix}_c${c}`, n * 32, false));
    const dW = await addon.devAlloc(m * 32);
    const dChain = {};
    for (const c of owned) dChain[c] = await addon.devAlloc(n * 32);
    const cnt = hHi - hLo;
    const dSl = [], dH = await addon.devAlloc(Math.max(cnt, 1) * 32);
    for (let c = 0; c < 3; c++) dSl.push(await addon.devAlloc(Math.max(cnt, 1) * 32));
    const src = [0, 0, 0];                                      // peer mode: where chain c's output can be read from in THIS process
    const opened = [];
    const handles = {};
    const send = (m) => { if (process.connected) process.send(m, () => { /* a parent that went away is handled by 'disconnect' */ }); };
    if (peer) {
        // a host without HSA IPC (some containers) or with mixed HIP runtimes fails here or in "peers" below: reported without an id, the parent
        // then restarts every worker with exchange "shm" when the exchange was not asked for explicitly
        try { for (const c of owned) { handles[c] = Buffer.from(await addon.ipcExport(dChain[c])).toString("base64"); src[c] = dChain[c]; } }
        catch (e) { send({ ev: "error", rank, phase: "peer", message: `ipcExport: ${String(e && e.message || e)}` }); return; }
    }
    send({ ev: "ready", rank, handles });
    let busy = Promise.resolve();                              // one command at a time: an async handler yields at every await
    const handle = async (msg) => {
        try {
            if (msg.cmd === "peers") {                          // every chain owner's handle: open the ones that live elsewhere
                try {
                    for (let c = 0; c < 3; c++) if (!owned.includes(c)) { src[c] = await addon.ipcOpen(new Uint8Array(Buffer.from(msg.handles[c], "base64"))); opened.push(src[c]); }
                    // one small pull from every mapping now: peer access that does not work must show up in the handshake, not in the first proof
                    for (let c = 0; c < 3; c++) if (!owned.includes(c)) await addon.peerCopy(dSl[0], src[c], 32);
                } catch (e) { send({ ev: "error", rank, phase: "peer", message: `peer handshake: ${String(e && e.message || e)}` }); return; }
                send({ ev: "peers_ok", rank });
            } else if (msg.cmd === "prove") {
                await addon.memcpyH2D(dW, shmW);
                if (owned.length) {                             // transforms first: their output leaves early
                    const ptr = (c) => (dChain[c] === undefined ? 0 : dChain[c]);
                    await addon.groth16ChainsDev(key, dW, owned.reduce((a, c) => a | (1 << c), 0), ptr(0), ptr(1), ptr(2));     // complete on return
                    for (const c of owned) { if (!peer) await addon.memcpyD2H(shm[c], dChain[c]); send({ ev: "chain", id: msg.id, chain: c }); }
                }
                await addon.groth16SumsWDev(key, dW);           // witness-side MSMs: enqueued, run while the other chains finish and travel
                send({ ev: "w", id: msg.id, rank });
            } else if (msg.cmd === "slices") {
                if (cnt) {
                    for (let c = 0; c < 3; c++) {
                        // device to device (xGMI, or HBM on a shared device), on the library's copy stream: not queued behind the witness-side
                        // accumulations already on the library stream; the fence below orders the join after the three copies
                        if (peer) await (addon.peerCopyAsync ? addon.peerCopyAsync : addon.peerCopy)(dSl[c], src[c] + 32 * hLo, 32 * cnt);
                        else await addon.memcpyH2D(dSl[c], shm[c].subarray(32 * hLo, 32 * hHi));
                    }
                    if (peer && addon.peerFence) await addon.peerFence();
                    await addon.joinABCDev(cid, dSl[0], dSl[1], dSl[2], dH, cnt);
                }
                const sums = await addon.groth16SumsHDev(cid, key, dW, dH);
                send({ ev: "sums", id: msg.id, rank, sums: Buffer.from(sums).toString("base64") });
            } else if (msg.cmd === "finish") {                 // worker 0: fold in rank order + blinding + toAffine
                const q = cid === 0 ? 32 : 48, j1 = 3 * q;
                const parts = msg.sums.map((b) => new Uint8Array(Buffer.from(b, "base64")));
                const total = new Uint8Array(7 * j1);
                for (const [a, b, grp] of [[0, j1, 1], [j1, 2 * j1, 1], [2 * j1, 4 * j1, 2], [4 * j1, 5 * j1, 1], [5 * j1, 6 * j1, 1]]) {
                    let acc = new Uint8Array(b - a);
                    for (const p of parts) acc = await addon.pointAdd(cid, grp, acc, p.subarray(a, b));
                    total.set(acc, a);
                }
                const res = await addon.groth16Finish(cid, key, total, new Uint8Array(Buffer.from(msg.r, "base64")), new Uint8Array(Buffer.from(msg.s, "base64")));
                send({ ev: "proof", id: msg.id, pi_a: Buffer.from(res.pi_a).toString("base64"), pi_b: Buffer.from(res.pi_b).toString("base64"), pi_c: Buffer.from(res.pi_c).toString("base64") });
            } else if (msg.cmd === "reset") {                  // after a failed proof anywhere: forget half-enqueued work (Work.w_enqueued / in_flight)
                if (addon.groth16Reset) await addon.groth16Reset(key);
                send({ ev: "reset_ok", id: msg.id, rank });
            } else if (msg.cmd === "exit") {
                for (const p of opened) { try { await addon.ipcClose(p); } catch (e) { /* going away anyway */ } }
                try { await addon.groth16Release(key); } catch (e) { /* going away anyway */ }
                process.exit(0);
            }
        } catch (e) { send({ ev: "error", id: msg.id, rank, message: String(e && e.message || e) }); }
    };
    process.on("message", (msg) => { busy = busy.then(() => handle(msg)); });
    process.on("disconnect", () => process.exit(0));           // the parent is gone: do not linger with a GPU bound
}

// ---- parent ---------------------------------------------------------------------------------------------------------------------------------
class ShardedProver {
    constructor(opts) {
        this.world = opts.world;
        this.zkeyPath = opts.zkeyPath;
        this.addonPath = opts.addonPath || path.join(__dirname, "..", "napi", "zkmi_napi.node");
        const addon = this.addon = require(this.addonPath);
        const zk = this.zk = openZkey(this.zkeyPath, { headerOnly: true });
        // worker rank -> HIP device ordinal: opts.devices, else ZKMI_SHARD_DEVICES="0,1,2,..." in the environment, else rank modulo the devices this
        // process can see (HIP_VISIBLE_DEVICES renumbers the visible ones from 0: the identity over deviceCount() honours it; more workers than
        // devices share them round robin and say so)
        const envMap = process.env.ZKMI_SHARD_DEVICES ? process.env.ZKMI_SHARD_DEVICES.split(",").map((x) => parseInt(x, 10)) : null;
        let devices = opts.devices || envMap;
        const visible = typeof addon.deviceCount === "function" ? addon.deviceCount() : 0;
        if (!devices) { devices = []; for (let r = 0; r < this.world; r++) devices.push(visible > 0 ? r % visible : r); }
        if (devices.length < this.world || devices.some((d) => !Number.isInteger(d) || d < 0)) throw new Error(`ShardedProver: device map ${JSON.stringify(devices)} does not cover ${this.world} workers`);
        if (visible > 0 && devices.some((d) => d >= visible)) throw new Error(`ShardedProver: device map ${JSON.stringify(devices)} names a device beyond the ${visible} visible to this process (HIP_VISIBLE_DEVICES=${process.env.HIP_VISIBLE_DEVICES || ""})`);
        this.devices = devices.slice(0, this.world);
        // "peer": chain outputs move GPU to GPU (zkmi_ipc_* / zkmi_peer_copy); "shm": through page-locked shared host memory (PCIe both ways).
        // Not given: "peer" when the addon has the entry points, and — should the peer handshake fail on this host (no HSA IPC in the container,
        // processes on different HIP runtimes, no peer access between two GPUs) — every worker is restarted with "shm"; the path taken is in
        // this.exchange and on every proof, the reason in this.exchangeFallback. An exchange asked for explicitly is never replaced.
        const auto = !opts.exchange;
        this.exchange = opts.exchange || (typeof addon.ipcExport === "function" ? "peer" : "shm");
        if (this.exchange !== "peer" && this.exchange !== "shm") throw new Error(`ShardedProver: unknown exchange "${this.exchange}"`);
        this.exchangeFallback = null;
        this.log = opts.log || ((m) => console.error(m));
        this.opts = opts;
        this.shmPrefix = `/zkmi_${process.pid}_${crypto.randomBytes(4).toString("hex")}`;
        // shared regions (created 0600 here, mapped by every worker, unlinked at close()): the witness; in "shm" mode the three chain outputs too
        this.shmNames = [`${this.shmPrefix}_w`];
        this.witnessRegion = addon.shmMap(this.shmNames[0], zk.nVars * 32, true);
        this.regions = [];
        this.waiters = new Map();
        this.nextId = 1;
        this.workers = [];
        this.gen = 0;                                           // generation of the worker set: events of a set that was torn down are ignored
        this.dead = null;                                       // Error once a worker has gone away
        this.needReset = false;
        this.queue = Promise.resolve();                         // prove() calls run one at a time
        this._ready = this._start().catch(async (err) => {
            if (!(auto && this.exchange === "peer" && err && err.peerPhase)) throw err;
            this.log(`ShardedProver: peer exchange unavailable (${err.message}); restarting the workers with exchange "shm"`);
            this.exchangeFallback = err.message;
            await this._stopWorkers();
            this.exchange = "shm";
            this.dead = null;
            return this._start();
        });
        this._ready.catch(() => {});                            // surfaced through ready() / prove()
    }
    // fork one worker per rank with the current exchange; resolves when every worker has loaded its shard (and, "peer", opened and probed its peers)
    _start() {
        const opts = this.opts, addon = this.addon, zk = this.zk, gen = ++this.gen;
        if (this.exchange === "shm" && !this.regions.length) for (let c = 0; c < 3; c++) { this.shmNames.push(`${this.shmPrefix}_c${c}`); this.regions.push(addon.shmMap(this.shmNames[this.shmNames.length - 1], zk.domainSize * 32, true)); }
        this.workers = [];
        return new Promise((resolve, reject) => {
            let up = 0, peersOk = 0;
            const handles = {};
            for (let rank = 0; rank < this.world; rank++) {
                const cfg = { rank, world: this.world, zkeyPath: this.zkeyPath, addonPath: this.addonPath, shmPrefix: this.shmPrefix, devices: this.devices, exchange: this.exchange, pageBytes: opts.pageBytes || null };
                const w = fork(__filename, ["--zkmi-shard-worker"], { env: Object.assign({}, process.env, { ZKMI_SHARD_CFG: JSON.stringify(cfg) }), execArgv: opts.execArgv || process.execArgv });
                w.on("message", (msg) => {
                    if (gen !== this.gen) return;
                    if (msg.ev === "ready") {
                        Object.assign(handles, msg.handles || {});
                        if (++up === this.world) { if (this.exchange === "peer") for (const x of this.workers) x.send({ cmd: "peers", handles }); else resolve(); }
                        return;
                    }
                    if (msg.ev === "peers_ok") { if (++peersOk === this.world) resolve(); return; }
                    if (msg.ev === "error" && !msg.id) { const e = new Error(`shard worker ${msg.rank}: ${msg.message}`); e.peerPhase = msg.phase === "peer"; reject(e); return; }
                    const wt = this.waiters.get(msg.id);
                    if (wt) wt(msg);
                });
                w.on("error", () => { /* a send to a worker that has just gone away: the 'exit' handler below reports it */ });
                w.on("exit", (code, signal) => {
                    if (this.closing || gen !== this.gen) return;
                    const err = new Error(`shard worker ${rank} exited (code ${code}, signal ${signal})`);
                    this.dead = this.dead || err;
                    reject(err);                                // no-op once resolved
                    for (const wt of Array.from(this.waiters.values())) wt({ ev: "error", rank, message: err.message, fatal: true });
                });
                this.workers.push(w);
            }
        });
    }
    // tear the current worker set down (peer -> shm restart): its late events are ignored from here on
    async _stopWorkers() {
        this.gen++;
        const old = this.workers;
        this.workers = [];
        for (const w of old) { if (w.connected) { try { w.send({ cmd: "exit" }); } catch (e) { /* already gone */ } } }
        await Promise.all(old.map((w) => new Promise((res) => {
            if (w.exitCode !== null || w.signalCode !== null) { res(); return; }
            const t = setTimeout(() => { try { w.kill("SIGKILL"); } catch (e) { /* gone */ } }, 10000);
            w.on("exit", () => { clearTimeout(t); res(); });
        })));
    }
    ready() { return this._ready; }
    // wtns: Uint8Array with the .wtns file's bytes, or a path. Calls are serialised: the returned promise settles in call order.
    prove(wtns, opts) {
        const run = () => this._proveOne(wtns, opts || {});
        const p = this.queue.then(run, run);
        this.queue = p.catch(() => {});
        return p;
    }
    _broadcastAndWait(cmd, ev) {
        const id = this.nextId++;
        return new Promise((resolve, reject) => {
            let got = 0;
            this.waiters.set(id, (msg) => {
                if (msg.ev === "error") { this.waiters.delete(id); reject(new Error(`shard worker ${msg.rank}: ${msg.message}`)); return; }
                if (msg.ev === ev && ++got === this.world) { this.waiters.delete(id); resolve(); }
            });
            for (const w of this.workers) if (w.connected) w.send({ cmd, id });
        });
    }
    async _proveOne(wtns, opts) {
        await this._ready;
        if (this.dead) throw this.dead;
        if (this.needReset) { await this._broadcastAndWait("reset", "reset_ok"); this.needReset = false; }
        const zk = this.zk, cid = zk.curveId, id = this.nextId++;
        const witness = parseWtns(typeof wtns === "string" ? new Uint8Array(fs.readFileSync(wtns)) : wtns, zk);
        this.witnessRegion.set(witness);                       // the private inputs never touch the file system
        const r = opts.r || randomFrMont(cid), s = opts.s || randomFrMont(cid);
        const t0 = process.hrtime.bigint(), at = {};
        try {
            const res = await new Promise((resolve, reject) => {
                let chains = 0, ws = 0;
                const sums = new Array(this.world).fill(null);
                const order = [];
                this.waiters.set(id, (msg) => {
                    if (msg.ev === "error") { reject(new Error(`shard worker ${msg.rank}: ${msg.message}`)); return; }
                    order.push(msg.ev);
                    at[msg.ev] = Number(process.hrtime.bigint() - t0) / 1e6;       // last arrival of each kind, ms after the proof was started
                    if (msg.ev === "chain") chains++;
                    if (msg.ev === "w") ws++;
                    // every chain complete and every witness-side half enqueued: the slices may be pulled
                    if ((msg.ev === "chain" || msg.ev === "w") && chains === 3 && ws === this.world) for (const w of this.workers) if (w.connected) w.send({ cmd: "slices", id });
                    if (msg.ev === "sums") {
                        sums[msg.rank] = msg.sums;
                        if (sums.every((x) => x !== null)) this.workers[0].send({ cmd: "finish", id, sums, r: Buffer.from(r).toString("base64"), s: Buffer.from(s).toString("base64") });
                    }
                    if (msg.ev === "proof") resolve({ msg, order });
                });
                for (const w of this.workers) if (w.connected) w.send({ cmd: "prove", id });
            });
            const b = (x) => new Uint8Array(Buffer.from(x, "base64"));
            const publicSignals = [];
            for (let i = 1; i <= zk.nPublic; i++) publicSignals.push(fromLE(witness.subarray(i * zk.n8r, (i + 1) * zk.n8r)).toString());
            return { proof: { pi_a: pointToObject(cid, 1, b(res.msg.pi_a)), pi_b: pointToObject(cid, 2, b(res.msg.pi_b)), pi_c: pointToObject(cid, 1, b(res.msg.pi_c)), protocol: "groth16", curve: zk.curveName },
                     publicSignals, events: res.order, exchange: this.exchange, exchangeFallback: this.exchangeFallback, devices: this.devices, timeline_ms: at };
        } catch (e) {
            this.needReset = !this.dead;                       // some workers may sit between the two halves of this proof
            throw e;
        } finally {
            this.waiters.delete(id);
            this.witnessRegion.fill(0);
        }
    }
    async close() {
        this.closing = true;
        for (const w of this.workers) { if (w.connected) { try { w.send({ cmd: "exit" }); } catch (e) { /* already gone */ } } }
        await Promise.all(this.workers.map((w) => new Promise((res) => { if (w.exitCode !== null || w.signalCode !== null) res(); else w.on("exit", res); })));
        for (const nm of this.shmNames) this.addon.shmUnlink(nm);
        this.regions = []; this.witnessRegion = null;
    }
}

if (process.argv.includes("--zkmi-shard-worker")) workerMain().catch((e) => { try { process.send({ ev: "error", rank: -1, message: String(e && e.message || e) }); } catch (e2) { /* no channel */ } process.exit(1); });
else module.exports = { ShardedProver, shardRange, chainOwner, pointToObject, randomFrMont };
