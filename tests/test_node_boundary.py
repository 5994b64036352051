"""The Node.js side of the drop-in boundary: N-API addon (snarkjs_amd/napi) + register.js glue (snarkjs_amd/js).

 * not gpu: the addon loads under this container's Node, exports every entry point and FAILS LOUDLY without a device;
   with /root/reference present (build container only) register.js is exercised against the real snarkjs bundle with a
   mock addon: a seeded groth16.prove through the patched curve reproduces the reference's proof bit for bit.
 * gpu: tests/js/addon_golden.js drives the real addon against the golden vectors.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "snarkjs_amd", "napi", "zkmi_napi.node")
NODE = shutil.which("node")
need_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is missing")
FLAGS = ["--harmony-optional-chaining", "--harmony-nullish"]


@need_node
def test_addon_exports_and_no_silent_fallback():
    js = ("const a=require(%r);const want=['init','deviceCount','version','msm','releaseBases','ntt','frBatch','applyKey','joinABC','toAffine',"
          "'groth16Prove','groth16Release','call','groupFft','groupApplyKey'];for(const k of want) if(typeof a[k]!=='function'){console.log('missing',k);process.exit(3)}"
          "if(a.deviceCount()==0){try{a.init(0);console.log('init did not throw');process.exit(4)}catch(e){if(!/no HIP device/.test(e.message)){console.log(e.message);process.exit(5)}}}"
          "console.log('ok')") % ADDON
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@need_node
@pytest.mark.skipif(not os.path.exists("/root/reference/build/snarkjs.min.js"), reason="reference bundle not present (GPU box)")
def test_register_glue_against_reference_bundle():
    r = subprocess.run([NODE] + FLAGS + [os.path.join(ROOT, "tests", "js", "register_glue.js")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@need_node
def test_addon_against_golden_vectors_on_gpu():
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "addon_golden.js")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@need_node
def test_register_js_replay_with_real_addon_on_gpu():
    """register.js + the REAL addon on the GPU, fed with the bulk calls the real snarkjs makes in its seeded groth16.prove (n = 1024)
    and plonk.prove (n = 2048), recorded from the reference bundle (oracle/gen_replay.js -> tests/golden/replay_bn128.*)."""
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "register_replay.js")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@need_node
def test_node_fused_plonk_prover_on_gpu():
    """snarkjs_amd/js/plonk_native.js (device-resident PLONK prover driven from Node) == the reference's seeded proofs"""
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "plonk_native_golden.js")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@need_node
def test_node_fused_fflonk_prover_on_gpu():
    """snarkjs_amd/js/fflonk_native.js == the reference's seeded FFLONK proofs"""
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "fflonk_native_golden.js")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@need_node
def test_node_fused_plonk_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    js = ("const {prove}=require(%r);const fs=require('fs');try{prove(new Uint8Array(fs.readFileSync(%r)),new Uint8Array(fs.readFileSync(%r)));console.log('no throw');process.exit(3)}"
          "catch(e){if(!/no HIP device/.test(e.message)){console.log(e.message);process.exit(4)}console.log('ok')}") % (
        os.path.join(ROOT, "snarkjs_amd", "js", "plonk_native.js"), os.path.join(ROOT, "tests", "golden", "plonk_bn128_small.zkey"),
        os.path.join(ROOT, "tests", "golden", "plonk_bn128_small.wtns"))
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
