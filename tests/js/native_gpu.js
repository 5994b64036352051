// tests/js/native_gpu.js — snarkjs_amd/js/groth16_native.js (makeProver: prove, throughput mode) and snarkjs_amd/js/groth16_shards.js (one
// proof over several worker processes) with the REAL addon on a GPU. The reference bundle cannot travel to the GPU box, so `snarkjs` is a
// stub that provides exactly what makeProver takes from it (curves.getCurveFromName -> {name, Fr.random, G1/G2.toObject}); the blinding draws
// are the recorded ones of the reference's seeded proof, and sha256(JSON.stringify(proof)) must equal the reference's (tests/golden).
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const { makeProver } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "groth16_native.js"));
const { ShardedProver, pointToObject } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "groth16_shards.js"));
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
const GOLD = path.join(__dirname, "..", "golden");
let fails = 0;
const check = (name, ok) => { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); };
const hexb = (s) => new Uint8Array(Buffer.from(s, "hex"));

function stubSnarkjs(cid, name, draws) {
    const curve = { name, Fr: { random: () => draws.shift() }, G1: { toObject: (b) => pointToObject(cid, 1, b) }, G2: { toObject: (b) => pointToObject(cid, 2, b) } };
    return { curves: { getCurveFromName: async () => curve } };
}

(async () => {
    for (const [tag, cid, name] of [["groth16_bn128_n1024", 0, "bn128"], ["groth16_bls12381_n1024", 1, "bls12381"]]) {
        const g = JSON.parse(fs.readFileSync(path.join(GOLD, tag + ".json")));
        const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".zkey"))), wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".wtns")));
        const draws = [];
        const seed = (k) => { for (let i = 0; i < k; i++) draws.push(hexb(g.r_mont), hexb(g.s_mont)); };
        const prover = makeProver(stubSnarkjs(cid, name, draws));
        seed(1);
        const res = await prover.prove(zkey, wtns);
        check(`${name}: makeProver.prove (real addon) == reference proof`, sha(JSON.stringify(res.proof)) === g.proof_sha256 && JSON.stringify(res.publicSignals) === JSON.stringify(g.publicSignals));
        seed(2);
        const both = await Promise.all([prover.prove(zkey, wtns), prover.prove(zkey, wtns)]);
        check(`${name}: two overlapping proofs`, both.every((x) => sha(JSON.stringify(x.proof)) === g.proof_sha256));
        seed(7);
        const many = await prover.proveMany(zkey, [wtns, wtns, wtns, wtns, wtns, wtns, wtns]);
        check(`${name}: proveMany, two proofs in flight, seven proofs`, many.length === 7 && many.every((x) => sha(JSON.stringify(x.proof)) === g.proof_sha256));
        let threw = false;
        seed(1);
        try { await prover.proveMany(zkey, [wtns, zkey]); } catch (e) { threw = /Invalid File format/.test(e.message); }
        draws.length = 0; seed(1);
        const after = await prover.proveMany(zkey, [wtns]);
        check(`${name}: an error inside proveMany leaves no proof in flight`, threw && sha(JSON.stringify(after[0].proof)) === g.proof_sha256);
        await prover.release();
        // r06: the key opened BY OFFSET FROM THE FILE, bulk sections handed to the FUSED load as 64 KiB pages (a 2^24 key's sections arrive like this,
        // as 1 GiB pages: src/groth16_prove.js:57-59 readSection -> BigBuffer), also with an odd page size that cuts records and points
        for (const pageBytes of [65536, 4096 + 44]) {
            const paged = makeProver(stubSnarkjs(cid, name, draws), { pageBytes });
            draws.length = 0; seed(2);
            const zkeyPath = path.join(GOLD, tag + ".zkey");
            const r1 = await paged.prove(zkeyPath, path.join(GOLD, tag + ".wtns"));
            const r2 = await paged.prove({ type: "file", fileName: zkeyPath }, { type: "mem", data: wtns });
            check(`${name}: fused load from a file descriptor in ${pageBytes}-byte pages == reference proof (path and fastfile descriptor share one resident key)`,
                  [r1, r2].every((x) => sha(JSON.stringify(x.proof)) === g.proof_sha256));
            await paged.release();
        }
        {   // sections a caller read itself with the reference's readSection: Uint8Array or BigBuffer (.buffers) each
            const { openZkey, descFromSections, toPages } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "groth16_native.js"));
            const addon = require(path.join(__dirname, "..", "..", "snarkjs_amd", "napi", "zkmi_napi.node"));
            const zk = openZkey(zkey, { pageBytes: 1 << 15 });
            const big = (x) => Array.isArray(x) ? { buffers: x, byteLength: x.reduce((a, b) => a + b.length, 0) } : x;       // the shape of a BigBuffer
            const d = descFromSections(zk, { 4: big(zk.desc.coeffs), 5: big(zk.desc.A), 6: big(zk.desc.B1), 7: big(zk.desc.B2), 8: big(zk.desc.C), 9: big(zk.desc.H),
                                             alpha1: zk.desc.alpha1, beta1: zk.desc.beta1, beta2: zk.desc.beta2, delta1: zk.desc.delta1, delta2: zk.desc.delta2 });
            const w = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "groth16_native.js")).parseWtns(wtns, zk);
            const res = addon.groth16Prove(d, 0, w, hexb(g.r_mont), hexb(g.s_mont));          // key 0: load, prove, release
            const str = (x) => Array.isArray(x) ? x.map(str) : x.toString();
            const proof = { pi_a: str(pointToObject(cid, 1, res.pi_a)), pi_b: str(pointToObject(cid, 2, res.pi_b)), pi_c: str(pointToObject(cid, 1, res.pi_c)), protocol: "groth16", curve: name };
            check(`${name}: groth16Prove with BigBuffer-shaped sections (key 0: load, prove, release)`, sha(JSON.stringify(proof)) === g.proof_sha256 && Array.isArray(toPages(big(zk.desc.B2))));
        }
        // the same proof from key shards held by separate worker processes. On a one-GPU box all of them sit on device 0: what is tested is the
        // PROTOCOL (ownership, order, hipIpc export / open between processes, device-to-device pulls of the slices), not the placement — with
        // two or more devices visible the workers spread over them and the pulls cross xGMI. Both exchanges: "peer" (zkmi_ipc_* + zkmi_peer_copy,
        // the default) and "shm" (page-locked shared host memory).
        const nDev = require(path.join(__dirname, "..", "..", "snarkjs_amd", "napi", "zkmi_napi.node")).deviceCount();
        for (const [world, exchange] of [[2, "peer"], [3, "peer"], [2, "shm"]]) {
            const devices = Array.from({ length: world }, (_, k) => k % nDev);
            const sp = new ShardedProver({ world, zkeyPath: path.join(GOLD, tag + ".zkey"), devices, exchange });
            await sp.ready();
            const r1 = await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) });
            const r2 = await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) });
            const both = await Promise.all([sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) }), sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) })]);
            check(`${name}: ${world} shard processes on devices [${devices}] (${exchange}) == reference proof (twice, then two calls at once)`,
                  r1.exchange === exchange && [r1, r2, both[0], both[1]].every((x) => sha(JSON.stringify(x.proof)) === g.proof_sha256));
            await sp.close();
        }
        {   // r06: workers that read ONLY their slice of the base sections (4 KiB pages, the rest as gaps), placed by the environment's device map
            process.env.ZKMI_SHARD_DEVICES = Array.from({ length: 3 }, (_, k) => (2 - k) % nDev).join(",");      // not the identity where there are devices
            const sp = new ShardedProver({ world: 3, zkeyPath: path.join(GOLD, tag + ".zkey"), pageBytes: 4096 });
            delete process.env.ZKMI_SHARD_DEVICES;
            await sp.ready();
            const r1 = await sp.prove(wtns, { r: hexb(g.r_mont), s: hexb(g.s_mont) });
            check(`${name}: 3 shard processes, slices read by offset in 4 KiB pages, ZKMI_SHARD_DEVICES -> [${r1.devices}]`,
                  sha(JSON.stringify(r1.proof)) === g.proof_sha256 && JSON.stringify(r1.devices) === JSON.stringify([2 % nDev, 1 % nDev, 0]));
            await sp.close();
            let threw = false;
            try { new ShardedProver({ world: 2, zkeyPath: path.join(GOLD, tag + ".zkey"), devices: [0, nDev] }); } catch (e) { threw = /beyond the/.test(e.message); }
            check(`${name}: a device map that names an invisible device is refused before any worker starts`, threw);
        }
    }
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.log("ERROR", e); process.exit(2); });
