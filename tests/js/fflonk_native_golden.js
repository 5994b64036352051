// tests/js/fflonk_native_golden.js — GPU: the Node-side fused FFLONK prover (snarkjs_amd/js/fflonk_native.js) reproduces the
// reference's seeded proofs (tests/golden/fflonk_bn128_*.json) bit for bit.   Run:  node tests/js/fflonk_native_golden.js
"use strict";
const fs = require("fs"), path = require("path"), crypto = require("crypto");
const { prove, proveAsync, FflonkKey } = require(path.join(__dirname, "..", "..", "snarkjs_amd", "js", "fflonk_native.js"));
const GOLD = path.join(__dirname, "..", "golden");
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex");
let fails = 0;
function check(name, ok) { if (!ok) { fails++; console.log("FAIL", name); } else console.log("ok  ", name); }
const hexb = (s) => new Uint8Array(Buffer.from(s, "hex"));

for (const tag of ["fflonk_bn128_small", "fflonk_bn128_n256"]) {
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, tag + ".json")));
    const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".zkey"))), wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".wtns")));
    const res = prove(zkey, wtns, g.blinding_mont.map(hexb));
    check(tag + ": proof == reference proof (sha256 of the JSON)", sha(JSON.stringify(res.proof)) === g.proof_sha256);
    check(tag + ": publicSignals", JSON.stringify(res.publicSignals) === JSON.stringify(g.publicSignals));
    const key = new FflonkKey(zkey);
    const p2 = prove(key, wtns), p3 = prove(key, wtns);
    key.release();
    check(tag + ": fresh blinding changes the proof, not the public signals",
          JSON.stringify(p2.proof) !== JSON.stringify(p3.proof) && JSON.stringify(p2.publicSignals) === JSON.stringify(g.publicSignals));
    let threw = false;
    try { prove(zkey, wtns.subarray(0, wtns.length - 32)); } catch (e) { threw = /Invalid witness length/.test(e.message); }
    check(tag + ": truncated witness is rejected", threw);
}
let threw = false;
try { prove(new Uint8Array(fs.readFileSync(path.join(GOLD, "plonk_bn128_small.zkey"))), new Uint8Array(64)); } catch (e) { threw = e.message === "zkey file is not fflonk"; }
check("a PLONK zkey is rejected with the reference's message", threw);
// r06: proveAsync (fflonkProve is async in the reference, src/fflonk_prove.js:51)
(async () => {
    const tag = "fflonk_bn128_n256";
    const g = JSON.parse(fs.readFileSync(path.join(GOLD, tag + ".json")));
    const zkey = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".zkey"))), wtns = new Uint8Array(fs.readFileSync(path.join(GOLD, tag + ".wtns")));
    let ticks = 0;
    const timer = setInterval(() => { ticks++; }, 0);
    const [a, b] = await Promise.all([proveAsync(zkey, wtns, g.blinding_mont.map(hexb), { device: 0 }), proveAsync(zkey, wtns, g.blinding_mont.map(hexb))]);
    clearInterval(timer);
    check(tag + `: proveAsync == reference proof, twice at once; the event loop turned ${ticks} times meanwhile`,
          sha(JSON.stringify(a.proof)) === g.proof_sha256 && sha(JSON.stringify(b.proof)) === g.proof_sha256 && ticks >= 4);
    let msg = "";
    try { await proveAsync(zkey, wtns.subarray(0, wtns.length - 32)); } catch (e) { msg = e.message; }
    check(tag + ": proveAsync rejects with the reference's message", /Invalid witness length/.test(msg));
    console.log(fails ? `${fails} FAILED` : "ALL OK");
    process.exit(fails ? 1 : 0);
})().catch((e) => { console.log("ERROR", e); process.exit(2); });
