// tests/js/native_glue.js — CPU-only check of snarkjs_amd/js/groth16_native.js (makeProver) with the REAL reference bundle as `snarkjs`
// and tests/js/ref_backend.js in place of the addon (build container only: needs /root/reference). Covered: zkey / wtns parsing and the
// reference's error messages, the key life cycle (ONE load for concurrent first calls, a failed load is forgotten and released, release()),
// the proof and publicSignals of the reference's seeded Groth16 fixture, and the two-slot order of throughput mode (proveMany).
// Run:  node --harmony-optional-chaining --harmony-nullish tests/js/native_glue.js
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const backend = require("./ref_backend.js");
const { makeProver } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "groth16_native.js"));
const snarkjs = backend.snarkjs;
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
const GOLD = path.join(__dirname, "..", "golden");
let fails = 0;
const check = (name, ok) => { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); };
const hexb = (s) => new Uint8Array(Buffer.from(s, "hex"));
const count = (nm) => backend.calls.filter((c) => c === nm).length;

(async () => {
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.json")));
    const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.zkey")));
    const wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.wtns")));
    const curve = await snarkjs.curves.getCurveFromName("bn128");
    const realRandom = curve.Fr.random;
    const seeded = (k) => { const d = []; for (let i = 0; i < k; i++) d.push(hexb(g.r_mont), hexb(g.s_mont)); curve.Fr.random = () => d.shift(); };

    // 1. one proof == the reference's own seeded proof; the key is loaded once and reused
    const prover = makeProver(snarkjs, { addon: backend });
    seeded(1);
    const res = await prover.prove(zkey, wtns);
    check("makeProver.prove == reference proof (sha256 of the proof JSON)", sha(JSON.stringify(res.proof)) === g.proof_sha256);
    check("publicSignals", JSON.stringify(res.publicSignals) === JSON.stringify(g.publicSignals || res.publicSignals) && res.publicSignals.length === 2);
    seeded(1);
    await prover.prove(zkey, wtns);
    check("second proof re-uses the resident key (one load)", count("groth16LoadAsync") === 1 && count("groth16ProveAsync") === 2);

    // 2. concurrent first calls on a fresh prover share ONE load
    const p2 = makeProver(snarkjs, { addon: backend });
    const before = count("groth16LoadAsync");
    seeded(2);
    const both = await Promise.all([p2.prove(zkey, wtns), p2.prove(zkey, wtns)]);
    check("two overlapping first calls -> one key load", count("groth16LoadAsync") === before + 1);
    check("both overlapping proofs are well-formed", both.every((x) => x.proof.protocol === "groth16" && x.proof.pi_a.length === 3));

    // 3. a load that fails is forgotten (the next call loads again) and what it may have left is released
    const p3 = makeProver(snarkjs, { addon: backend });
    backend.failNextLoad();
    const rel0 = count("groth16Release");
    let threw = false;
    try { await p3.prove(zkey, wtns); } catch (e) { threw = /injected load failure/.test(e.message); }
    check("failed load rejects with the library's message and releases the key number", threw && count("groth16Release") === rel0 + 1);
    seeded(1);
    const again = await p3.prove(zkey, wtns);
    check("the next call loads again and proves", sha(JSON.stringify(again.proof)) === g.proof_sha256);

    // 4. reference error messages
    threw = false;
    try { await prover.prove(zkey, zkey); } catch (e) { threw = /Invalid File format/.test(e.message); }
    check("wtns magic: Invalid File format", threw);
    const short = wtns.slice(0, wtns.length - 32);
    new DataView(short.buffer).setBigUint64(short.length - (1003 * 32 - 32) - 8, BigInt(1003 * 32 - 32), true);    // keep the container consistent: section 2 one element short
    threw = false;
    try { await prover.prove(zkey, short); } catch (e) { threw = /Invalid witness length/.test(e.message); }
    check("short witness: Invalid witness length", threw);
    const plonkZ = path.join(GOLD, "plonk_bn128_small.zkey");
    threw = false;
    try { await prover.prove(new Uint8Array(fs.readFileSync(plonkZ)), wtns); } catch (e) { threw = e.message === "zkey file is not groth16"; }
    check("a PLONK zkey is refused: zkey file is not groth16", threw);

    // 5. throughput mode: five proofs, two in flight — submit k+1 before collect k, slots alternate, results in input order
    backend.calls.length = 0;
    seeded(5);
    const many = await prover.proveMany(zkey, [wtns, wtns, wtns, wtns, wtns]);
    check("proveMany: every proof == reference proof", many.length === 5 && many.every((x) => sha(JSON.stringify(x.proof)) === g.proof_sha256));
    let seq = backend.calls.filter((c) => /^(submit|collect)/.test(c)).join(" ");
    if (seq.startsWith("submit1")) seq = seq.replace(/[01]/g, (d) => (d === "0" ? "1" : "0"));       // the slots alternate process-wide (r06): which one comes first depends on the proofs before
    check("proveMany call order (two slots): " + seq, seq === "submit0 submit1 collect0 submit0 collect1 submit1 collect0 submit0 collect1 collect0");
    // r06: concurrent prove() calls go through the same queue: four requests at once pipeline exactly like proveMany
    backend.calls.length = 0;
    seeded(4);
    const four = await Promise.all([prover.prove(zkey, wtns), prover.prove(zkey, wtns), prover.prove(zkey, wtns), prover.prove(zkey, wtns)]);
    let seq4 = backend.calls.filter((c) => /^(submit|collect)/.test(c)).join(" ");
    if (seq4.startsWith("submit1")) seq4 = seq4.replace(/[01]/g, (d) => (d === "0" ? "1" : "0"));
    check("four concurrent prove() calls: two in flight, in arrival order: " + seq4, four.every((x) => sha(JSON.stringify(x.proof)) === g.proof_sha256) &&
          seq4 === "submit0 submit1 collect0 submit0 collect1 submit1 collect0 collect1");
    // an error inside the pipeline leaves no slot occupied
    threw = false;
    try { await prover.proveMany(zkey, [wtns, short, wtns]); } catch (e) { threw = /Invalid witness length/.test(e.message); }
    seeded(1);
    const after = await prover.proveMany(zkey, [wtns]);
    check("an error inside proveMany drains the slots", threw && sha(JSON.stringify(after[0].proof)) === g.proof_sha256);

    // 6. release frees every resident key once
    const relBefore = count("groth16Release");
    await prover.release(); await p2.release(); await p3.release();
    check("release() frees each prover's key", count("groth16Release") === relBefore + 3);
    curve.Fr.random = realRandom;
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.log("ERROR", e); process.exit(2); });
