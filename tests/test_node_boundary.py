"""The Node.js side of the drop-in boundary: N-API addon (snarkjs_amd/napi) + register.js glue (snarkjs_amd/js).

 * not gpu: the addon loads under this container's Node, exports every entry point and FAILS LOUDLY without a device;
   with /root/reference present (build container only) register.js is exercised against the real snarkjs bundle with a
   mock addon: a seeded groth16.prove through the patched curve reproduces the reference's proof bit for bit.
 * gpu: tests/js/addon_golden.js drives the real addon against the golden vectors.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "snarkjs_amd", "napi", "zkmi_napi.node")
NODE = shutil.which("node")
# the reference's bundle: /root/reference in the build container, or its staged copy oracle/_ref/ (make -C oracle _ref; git-ignored, travels with gpurun)
BUNDLE = next((p for p in (os.path.join(ROOT, "oracle", "_ref", "build", "snarkjs.min.js"), "/root/reference/build/snarkjs.min.js") if os.path.exists(p)), None)
need_bundle = pytest.mark.skipif(BUNDLE is None, reason="reference bundle not present: run `make -C oracle _ref` where /root/reference exists")
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
def test_zkey_opened_by_offset_in_pages_and_shard_gaps():
    """js/groth16_native.js: openZkey from bytes / a path / fastfile descriptors / a BigBuffer-backed memory file, any page size, a shard's byte ranges with gaps
    elsewhere — all describe the same bytes as the flat parse (r06: how a 2^24 key reaches the fused Groth16 boundary from Node)"""
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "open_zkey_cpu.js")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@need_node
@need_bundle
def test_register_glue_against_reference_bundle():
    r = subprocess.run([NODE] + FLAGS + [os.path.join(ROOT, "tests", "js", "register_glue.js")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@need_node
@need_bundle
def test_make_prover_against_reference_bundle():
    """js/groth16_native.js (makeProver: parsing, key life cycle, throughput-mode order) with the real bundle as `snarkjs` and a
    reference-backed stand-in for the addon (tests/js/ref_backend.js)"""
    r = subprocess.run([NODE] + FLAGS + [os.path.join(ROOT, "tests", "js", "native_glue.js")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@need_node
@need_bundle
def test_node_shard_driver_processes_on_cpu():
    """js/groth16_shards.js: 2 and 3 worker PROCESSES, the exchange through POSIX shared memory mapped by the real addon, the arithmetic by the
    reference's own curve: the sharded proof equals the reference's proof and the protocol runs in the overlapped order"""
    r = subprocess.run([NODE] + FLAGS + [os.path.join(ROOT, "tests", "js", "shards_mock.js")], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@need_node
def test_addon_checked_call_table():
    """addon.call reaches only the entry points of its binding table and checks every argument's kind and minimum length before the call;
    zkmi_keccak256 / zkmi_fr_root need no device"""
    js = ("const a=require(%r);const eq=(x,y)=>Buffer.from(x).toString('hex')===y;"
          "const o=new Uint8Array(32);a.call('zkmi_keccak256',new Uint8Array(0),0,o);"
          "if(!eq(o,'c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470')){console.log('keccak',Buffer.from(o).toString('hex'));process.exit(3)}"
          "const w=new Uint8Array(32);a.call('zkmi_fr_root',0,1,w);"
          "const bad=(f,re)=>{try{f();return false}catch(e){return re.test(e.message)}};"
          "if(!bad(()=>a.call('zkmi_init',0),/binding table/))process.exit(4);"
          "if(!bad(()=>a.call('system',0),/binding table/))process.exit(5);"
          "if(!bad(()=>a.call('zkmi_fr_root',0,1,new Uint8Array(31)),/shorter/))process.exit(6);"
          "if(!bad(()=>a.call('zkmi_fr_root',0,1),/too few/))process.exit(7);"
          "if(!bad(()=>a.call('zkmi_fr_root',0,1,w,1),/too many/))process.exit(8);"
          "if(!bad(()=>a.call('zkmi_keccak256',new Uint8Array(4),5,o),/length argument/))process.exit(9);"
          "if(!bad(()=>a.call('zkmi_fr_root','x',1,w),/bad argument/))process.exit(10);"
          "const want=['groth16Load','groth16LoadAsync','groth16LoadShard','groth16Submit','groth16SubmitAsync','groth16Collect','groth16CollectAsync','devAlloc','devFree','memcpyH2D','memcpyD2H',"
          "'groth16ChainsDev','groth16SumsWDev','groth16SumsHDev','groth16SumsDev','groth16Finish','joinABCDev','pointAdd','shmMap','shmUnlink'];"
          "for(const k of want) if(typeof a[k]!=='function'){console.log('missing',k);process.exit(11)}"
          "const z=new Uint8Array(96);if(a.pointAdd(0,1,z,z).length!==96)process.exit(12);"
          "const m=a.shmMap('/zkmi_test_'+process.pid,4096,true);m[5]=7;const m2=a.shmMap('/zkmi_test_'+process.pid,4096,false);if(m2[5]!==7)process.exit(13);a.shmUnlink('/zkmi_test_'+process.pid);"
          "console.log('ok')") % ADDON
    r = subprocess.run([NODE, "-e", js], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@need_node
@need_bundle
def test_unmodified_flow_with_mock_addon():
    """tests/js/unmodified_gpu.js with a stand-in for the addon (the reference's own WASM behind the addon's entry points): the logic of the
    patched-vs-unpatched comparison itself, on a GPU-less box"""
    r = subprocess.run([NODE] + FLAGS + [os.path.join(ROOT, "tests", "js", "unmodified_gpu.js")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ZKMI_MOCK_ADDON="1"))
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@need_node
@need_bundle
def test_same_box_reference_tool_reproduces_the_golden_proof():
    """tools/ref_wasm_same_box.js (bench.py's cpu_baseline leg: the reference's own prover under the worker shim, blinding draws handed to
    curve.Fr.random) on the reference-generated fixture: the proof hash it reports is the golden one, the reference's verifier accepts"""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "groth16_bn128_n1024.json")))
    zk, wt = (os.path.join(ROOT, "tests", "golden", "groth16_bn128_n1024." + e) for e in ("zkey", "wtns"))
    r = subprocess.run([NODE] + FLAGS + [os.path.join(ROOT, "tools", "ref_wasm_same_box.js"), "groth16", zk, wt, g["r_mont"] + "," + g["s_mont"]], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, VERIFY="1", NTHREADS="4"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["proof_json_sha256"] == g["proof_sha256"] and d["draws_used"] == 2 and d["verified"] is True and d["threads"] == 4


@pytest.mark.gpu
@need_node
def test_unmodified_snarkjs_with_real_addon_on_gpu():
    """UNMODIFIED snarkjs (the reference's bundle from oracle/_ref) + register.js + the REAL addon in one process: groth16 / plonk / fflonk prove and
    fullProve on both curves, a power-8 ceremony and the three setups, each patched and unpatched with the same draws: proofs and key bytes
    identical, the reference's verifier accepts (north_star: "run unmodified and emit proofs bit-identical to the WASM path")."""
    if BUNDLE is None:
        pytest.skip("NOT CHECKED ON THIS BOX: oracle/_ref is absent — `make -C oracle _ref` (or __graft_entry__.build()) stages the reference's bundle in the build "
                    "container and gpurun ships it; without it the unmodified-snarkjs claim is unverified here (profiles/r05_unmodified_snarkjs_gpu.log holds the last run)")
    r = subprocess.run([NODE] + FLAGS + [os.path.join(ROOT, "tests", "js", "unmodified_gpu.js")], capture_output=True, text=True, timeout=1500)
    sys.stdout.write(r.stdout[-6000:])
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@need_node
def test_node_make_prover_and_shard_processes_on_gpu():
    """js/groth16_native.js (prove, proveMany: two proofs in flight) and js/groth16_shards.js (2 and 3 worker processes, shared-memory exchange)
    with the REAL addon: the reference's seeded proofs on both curves"""
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", "native_gpu.js")], capture_output=True, text=True, timeout=900)
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
