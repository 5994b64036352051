// tests/js/shards_mock.js — the multi-process shard driver snarkjs_amd/js/groth16_shards.js on CPU: `world` worker PROCESSES, each with
// tests/js/ref_backend.js in place of the addon (the reference's own curve does the arithmetic; build container only), the chain outputs
// exchanged through real POSIX shared memory mapped by the real addon. A proof of the reference's seeded Groth16 fixture assembled from
// 2 and from 3 key shards must equal the reference's own proof, and the protocol must run in the overlapped order (no slice is read before
// all three chain regions are complete; every worker's witness-side half is enqueued before its H half).
// Run:  node --harmony-optional-chaining --harmony-nullish tests/js/shards_mock.js
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const { ShardedProver, shardRange } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "groth16_shards.js"));
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
const GOLD = path.join(__dirname, "..", "golden");
let fails = 0;
const check = (name, ok) => { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); };
const hexb = (s) => new Uint8Array(Buffer.from(s, "hex"));

(async () => {
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.json")));
    const wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.wtns")));
    for (const n of [0, 1, 7, 1024, 1003]) for (const w of [1, 2, 3, 8]) {
        const sp = []; for (let r = 0; r < w; r++) sp.push(shardRange(n, r, w));
        if (sp[0][0] !== 0 || sp[w - 1][1] !== n || sp.some((x, i) => i && x[0] !== sp[i - 1][1])) check(`shardRange(${n}, ${w}) covers everything`, false);
    }
    for (const [world, exchange] of [[2, "peer"], [3, "peer"], [2, "shm"]]) {
        const sp = new ShardedProver({ world, zkeyPath: path.join(GOLD, "groth16_bn128_n1024.zkey"), addonPath: path.join(__dirname, "ref_backend.js"), exchange,
                                       execArgv: ["--harmony-optional-chaining", "--harmony-nullish"] });
        await sp.ready();
        const res = await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) });
        check(`world ${world} (${exchange}): sharded proof == reference proof`, sha(JSON.stringify(res.proof)) === g.proof_sha256 && res.exchange === exchange);
        check(`world ${world} (${exchange}): no witness file in the temporary directory`, !fs.readdirSync(require("os").tmpdir()).some((f) => f.startsWith(`zkmi_${process.pid}_`)));
        const ev = res.events, firstSums = ev.indexOf("sums");
        check(`world ${world}: 3 chains + ${world} witness-side halves before the first H half (${ev.join(",")})`,
              ev.filter((e) => e === "chain").length === 3 && ev.filter((e) => e === "w").length === world && ev.slice(0, firstSums).filter((e) => e === "chain" || e === "w").length === 3 + world);
        const res2 = await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) });             // the workers and their key shards stay up
        check(`world ${world}: second proof over the same workers`, sha(JSON.stringify(res2.proof)) === g.proof_sha256);
        // two prove() calls issued together are serialised (one witness buffer, one set of chain buffers, one pipeline slot per worker)
        const both = await Promise.all([sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) }), sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) })]);
        check(`world ${world}: concurrent prove() calls are serialised and both correct`, both.every((x) => sha(JSON.stringify(x.proof)) === g.proof_sha256));
        await sp.close();
    }
    // error path: worker 1 fails once between the two halves of a proof; the proof is rejected, every worker is reset, the next proof is correct
    {
        process.env.ZKMI_MOCK_FAIL = "1:groth16SumsHDev";
        const sp = new ShardedProver({ world: 2, zkeyPath: path.join(GOLD, "groth16_bn128_n1024.zkey"), addonPath: path.join(__dirname, "ref_backend.js"),
                                       execArgv: ["--harmony-optional-chaining", "--harmony-nullish"] });
        delete process.env.ZKMI_MOCK_FAIL;
        await sp.ready();
        let msg = "";
        try { await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) }); } catch (e) { msg = String(e.message); }
        check(`a failing worker rejects the proof (${msg})`, /injected failure/.test(msg));
        const res = await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) });
        check("the proof after a failed one is correct (workers reset)", sha(JSON.stringify(res.proof)) === g.proof_sha256);
        // a worker that dies: pending and later proofs reject instead of hanging
        const pending = sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) }).then(() => "resolved", (e) => String(e.message));
        sp.workers[1].kill("SIGKILL");
        const out = await pending;
        let later = "";
        try { await sp.prove(wtns, {}); } catch (e) { later = String(e.message); }
        check(`a dead worker rejects the pending proof (${out}) and every later one (${later})`, /exited|resolved/.test(out) && /exited/.test(later));
        await sp.close();
    }
    // peer exchange that does not work on this host (no HSA IPC, mixed runtimes): with the exchange NOT asked for explicitly the parent restarts the
    // workers with "shm" and says so; asked for explicitly, ready() rejects instead of silently taking another path
    for (const failing of ["1:ipcOpen", "0:ipcExport", "1:peerCopy"]) {
        process.env.ZKMI_MOCK_FAIL = failing;
        const logs = [];
        const sp = new ShardedProver({ world: 2, zkeyPath: path.join(GOLD, "groth16_bn128_n1024.zkey"), addonPath: path.join(__dirname, "ref_backend.js"), log: (m) => logs.push(m),
                                       execArgv: ["--harmony-optional-chaining", "--harmony-nullish"] });
        await sp.ready();
        delete process.env.ZKMI_MOCK_FAIL;
        const res = await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) });
        check(`peer handshake fails (${failing}): workers restarted with "shm", proof correct, path reported (${res.exchangeFallback})`,
              sha(JSON.stringify(res.proof)) === g.proof_sha256 && res.exchange === "shm" && /injected failure/.test(res.exchangeFallback || "") && logs.length === 1);
        await sp.close();
    }
    {
        process.env.ZKMI_MOCK_FAIL = "1:ipcOpen";
        const sp = new ShardedProver({ world: 2, zkeyPath: path.join(GOLD, "groth16_bn128_n1024.zkey"), addonPath: path.join(__dirname, "ref_backend.js"), exchange: "peer",
                                       execArgv: ["--harmony-optional-chaining", "--harmony-nullish"] });
        delete process.env.ZKMI_MOCK_FAIL;
        let msg = "";
        try { await sp.ready(); } catch (e) { msg = String(e.message); }
        check(`an explicitly requested peer exchange that fails rejects ready() (${msg})`, /injected failure/.test(msg));
        await sp.close();
    }
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.log("ERROR", e); process.exit(2); });
