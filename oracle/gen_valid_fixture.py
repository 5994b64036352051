#!/usr/bin/env python
"""oracle/gen_valid_fixture.py — TEST INFRASTRUCTURE ONLY (build container; needs /root/reference and Node).

Pins the valid-key synthesiser (tests/synth_valid_groth16.py) and the verifier restatement (oracle/groth16_verify_oracle.py) to the
REFERENCE: a small synthetic valid key (n = 64) is written to tests/golden/, then the real snarkjs bundle
  * exports its verification key (zKey.exportVerificationKey),
  * proves on it with the seeded r, s (groth16.prove) and
  * verifies that proof (groth16.verify)  ->  must be true,
and the results are recorded in tests/golden/groth16_valid_synth_n64.json. Run:  make -C oracle valid
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth_valid_groth16 as SV  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
JS = r"""
const fs=require('fs');
process.env.SINGLE='1';
const snarkjs=require(process.argv[2]);
(async()=>{
  const zkey=new Uint8Array(fs.readFileSync(process.argv[3])), wtns=new Uint8Array(fs.readFileSync(process.argv[4]));
  const curve=await snarkjs.curves.getCurveFromName('bn128');
  const hex=b=>Buffer.from(b).toString('hex');
  const rnd=[], Fr=curve.Fr, orig=Fr.random.bind(Fr);
  Fr.random=()=>{const v=orig(); rnd.push(hex(v)); return v;};
  const vk=await snarkjs.zKey.exportVerificationKey(zkey);
  const {proof, publicSignals}=await snarkjs.groth16.prove(zkey, wtns);
  Fr.random=orig;
  const ok=await snarkjs.groth16.verify(vk, publicSignals, proof);
  const bad=JSON.parse(JSON.stringify(proof)); bad.pi_c=proof.pi_a;
  const okBad=await snarkjs.groth16.verify(vk, publicSignals, bad);
  console.log(JSON.stringify({vk, proof, publicSignals, r_mont:rnd[0], s_mont:rnd[1], verified:ok, tampered_verified:okBad}));
  process.exit(0);
})().catch(e=>{console.error(e);process.exit(1)});
"""


def main():
    lg = 6
    zkey, wtns, info = SV.make("bn128", lg, use_device=False)
    zf, wf = os.path.join(GOLD, "groth16_valid_synth_n64.zkey"), os.path.join(GOLD, "groth16_valid_synth_n64.wtns")
    open(zf, "wb").write(zkey)
    open(wf, "wb").write(wtns)
    js = "/tmp/valid_fixture.js"
    open(js, "w").write(JS)
    r = subprocess.run(["node", "--harmony-optional-chaining", "--harmony-nullish", js, os.path.join(ROOT, "oracle", "ref_shim.js"), zf, wf],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(r.stderr[-3000:])
        raise SystemExit(1)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["verified"] is True and d["tampered_verified"] is False, d
    sha = lambda b: hashlib.sha256(b).hexdigest()
    out = {"what": "synthetic VALID Groth16 key (tests/synth_valid_groth16.py, trapdoor TRAPDOOR, lg 6) checked by the reference: its own exported vk, its own seeded proof, "
                   "its own verifier", "zkey_sha256": sha(zkey), "wtns_sha256": sha(wtns),
           "proof_sha256": sha(json.dumps(d["proof"], separators=(",", ":")).encode()), **d}
    json.dump(out, open(os.path.join(GOLD, "groth16_valid_synth_n64.json"), "w"), indent=1)
    print("reference verify:", d["verified"], "| tampered:", d["tampered_verified"], "| vk equals the synthesiser's:", d["vk"]["IC"] == info["vk"]["IC"] and d["vk"]["vk_alpha_1"] == info["vk"]["vk_alpha_1"])


if __name__ == "__main__":
    main()
