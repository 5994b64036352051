// tests/js/plonk_native_golden.js — GPU: the Node-side fused PLONK prover (snarkjs_amd/js/plonk_native.js over the addon's generic
// C-ABI binding) reproduces the reference's seeded proofs (tests/golden/plonk_bn128_*.json) bit for bit.
// Run:  node tests/js/plonk_native_golden.js
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const { prove, proveAsync, proveMany, PlonkKey, PlonkWitness } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "plonk_native.js"));
const GOLD = path.join(__dirname, "..", "golden");
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
let fails = 0;
function check(name, ok) { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); }
const hexb = (s) => new Uint8Array(Buffer.from(s, "hex"));

for (const tag of ["plonk_bn128_small", "plonk_bn128_n2048", "plonk_bls12381_small"]) {
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, tag + ".json")));
    const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".zkey"))), wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".wtns")));
    const res = prove(zkey, wtns, g.blinding_mont.map(hexb));
    check(tag + ": proof == reference proof (sha256 of the JSON)", sha(JSON.stringify(res.proof)) === g.proof_sha256);
    check(tag + ": publicSignals", JSON.stringify(res.publicSignals) === JSON.stringify(g.publicSignals));
    const key = new PlonkKey(zkey);                       // resident key: two more proofs, fresh blinding
    const p2 = prove(key, wtns), p3 = prove(key, wtns);
    key.release();
    check(tag + ": fresh blinding changes the proof, not the public signals",
          JSON.stringify(p2.proof) !== JSON.stringify(p3.proof) && JSON.stringify(p2.publicSignals) === JSON.stringify(g.publicSignals) && p2.proof.A[0] !== g.proof.A[0]);
    let threw = false;
    try { prove(zkey, wtns.subarray(0, wtns.length - 32)); } catch (e) { threw = /Invalid witness length/.test(e.message); }
    check(tag + ": truncated witness is rejected", threw);
}
// throughput mode (two generator proofs on the library's two pipeline slots): the same proofs as prove(), golden first; an error in the middle
// fails the call, leaves slot 0 active and the library usable
for (const tag of ["plonk_bn128_n2048", "plonk_bls12381_small"]) {
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, tag + ".json")));
    const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".zkey"))), wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".wtns")));
    const key = new PlonkKey(zkey);
    const blinds = [g.blinding_mont.map(hexb)];
    for (let k = 1; k < 5; k++) { const b = []; for (let i = 0; i < 11; i++) b.push(key.f.mont(BigInt(7000 + 131 * k + 17 * i))); blinds.push(b); }
    const serial = blinds.map((b) => prove(key, wtns, b));
    const many = proveMany(key, blinds.map(() => wtns), blinds);
    check(tag + ": proveMany == prove, proof by proof (5 proofs, two in flight)", JSON.stringify(many) === JSON.stringify(serial) && sha(JSON.stringify(many[0].proof)) === g.proof_sha256);
    const wres = new PlonkWitness(key, wtns);               // resident witness shared by the proofs of both slots
    check(tag + ": proveMany over a resident witness", JSON.stringify(proveMany(key, blinds.map(() => wres), blinds)) === JSON.stringify(serial) &&
          JSON.stringify(prove(key, wres, blinds[2])) === JSON.stringify(serial[2]));
    wres.release();
    const bad = wtns.slice(); bad[bad.length - 32] ^= 1;
    let msg = "";
    try { proveMany(key, [wtns, bad, wtns, wtns], blinds.slice(0, 4)); } catch (e) { msg = e.message; }
    check(tag + ": a bad witness in the middle fails the call with the reference's message", /Copy constraints does not match|not divisible|not well calculated/.test(msg));
    check(tag + ": the library is usable afterwards", JSON.stringify(proveMany(key, [wtns, wtns, wtns], blinds.slice(0, 3))) === JSON.stringify(serial.slice(0, 3)) &&
          JSON.stringify(prove(key, wtns, blinds[1])) === JSON.stringify(serial[1]));
    key.release();
}
let threw = false;
try { prove(new Uint8Array(fs.readFileSync(path.join(GOLD, "groth16_bn128_n1024.zkey"))), new Uint8Array(64)); } catch (e) { threw = e.message === "zkey file is not plonk"; }
check("a Groth16 zkey is rejected with the reference's message", threw);
// r06: proveAsync (the reference's plonk16Prove is async, src/plonk_prove.js:47): same proof; the event loop turns while the commitments are computed
// (a timer keeps firing); two calls at once are serialised; errors reject; the device option (one device per process)
(async () => {
    for (const tag of ["plonk_bn128_n2048", "plonk_bls12381_small"]) {
        const g = JSON.parse(fs.readFileSync(path.join(GOLD, tag + ".json")));
        const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".zkey"))), wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".wtns")));
        let ticks = 0;
        const timer = setInterval(() => { ticks++; }, 0);
        const blind = g.blinding_mont.map(hexb);
        const [a, b] = await Promise.all([proveAsync(zkey, wtns, blind, { device: 0 }), proveAsync(zkey, wtns, blind)]);
        clearInterval(timer);
        check(tag + `: proveAsync == reference proof, twice at once; the event loop turned ${ticks} times meanwhile`,
              sha(JSON.stringify(a.proof)) === g.proof_sha256 && sha(JSON.stringify(b.proof)) === g.proof_sha256 && ticks >= 4);
        let msg = "";
        try { await proveAsync(zkey, wtns.subarray(0, wtns.length - 32)); } catch (e) { msg = e.message; }
        check(tag + ": proveAsync rejects with the reference's message", /Invalid witness length/.test(msg));
        const again = await proveAsync(zkey, wtns, blind);
        check(tag + ": and works afterwards", sha(JSON.stringify(again.proof)) === g.proof_sha256);
    }
    let msg = "";
    try { new PlonkKey(new Uint8Array(fs.readFileSync(path.join(GOLD, "plonk_bn128_small.zkey"))), { device: 1 }); } catch (e) { msg = e.message; }
    check("a key for another device than the one this process is bound to is refused", /bound to device 0/.test(msg));
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.log("ERROR", e); process.exit(2); });
