"""GPU parity tests: the HIP path (through the C-ABI, include/zkmi.h) against the CPU oracle and the committed
golden vectors generated from the reference.  Bit-exact for all byte outputs; MSM results compared as group
elements after toAffine (SURVEY.md §8c parity procedure).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import synth
from test_oracle_golden import load, raw, msm_inputs, _Q

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURVES = ["bn128", "bls12381"]
sha = lambda b: hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def zk():
    import torch
    torch.cuda.init()                  # before the library takes device memory: the sharded tests exchange torch CUDA tensors
    import snarkjs_amd
    from snarkjs_amd import zkmi
    zkmi.init(0)
    return snarkjs_amd


def curve_of(zk, name):
    return zk.get_curve_from_name(name)


@pytest.mark.parametrize("name", CURVES)
def test_n1024_golden(zk, golden_dir, name):
    d, c, cv = load(golden_dir, name), O.CURVE_ID[name], curve_of(zk, name)
    x = synth.iota(1024)
    got = {
        "fft": cv.Fr.fft(x), "ifft": cv.Fr.ifft(x),
        "applykey_7_11": cv.Fr.batchApplyKey(x, O.fr_e(c, 7), O.fr_e(c, 11)),
        "to_mont": cv.Fr.batchToMontgomery(x), "from_mont": cv.Fr.batchFromMontgomery(x),
        "inverse": cv.Fr.batchInverse(x),
    }
    for k, v in got.items():
        assert np.array_equal(v, raw(golden_dir, name, k)), k
        assert sha(v) == d["n1024"][k]
    for g, G in ((1, cv.G1), (2, cv.G2)):
        jac = G.multiExpAffine(raw(golden_dir, name, f"g{g}_bases"), x)
        assert np.array_equal(O.to_affine(c, g, jac), raw(golden_dir, name, f"g{g}_msm_affine"))
        assert np.array_equal(G.toAffine(jac), raw(golden_dir, name, f"g{g}_msm_affine"))


@pytest.mark.parametrize("name", CURVES)
def test_ntt_all_sizes_vs_oracle(zk, name):
    c, cv = O.CURVE_ID[name], curve_of(zk, name)
    for lg in range(0, 19):
        x = synth.elems(0x1000 + lg, 1 << lg)
        assert np.array_equal(cv.Fr.fft(x), O.ntt(c, x)), f"fft 2^{lg}"
        assert np.array_equal(cv.Fr.ifft(x), O.ntt(c, x, True)), f"ifft 2^{lg}"


@pytest.mark.parametrize("force", ["1", "0"])
def test_ntt29_passes_opt_in_parity(force):
    """Both forms of the NTT passes forced for EVERY size (the default picks the 9 x 29-bit passes of csrc/ntt29.cuh up to 2^22 and the saturated
    32-bit ones of ntt.cuh above, r04): the same bytes as the oracle for every size, both curves, forward / inverse / fused pre-scale, in a
    process of its own (ZKMI_NTT29 is read once)."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path[:0] = [%r, %r]\n"
        "import oracle_lib as O, synth, snarkjs_amd\n"
        "for name in ('bn128', 'bls12381'):\n"
        "    c, cv = O.CURVE_ID[name], snarkjs_amd.get_curve_from_name(name)\n"
        "    for lg in list(range(0, 19)) + [20]:\n"
        "        x = synth.elems(0x2900 + lg, 1 << lg)\n"
        "        if lg <= 18:\n"
        "            assert np.array_equal(cv.Fr.fft(x), O.ntt(c, x)), ('fft', name, lg)\n"
        "            assert np.array_equal(cv.Fr.ifft(x), O.ntt(c, x, True)), ('ifft', name, lg)\n"
        "        else:\n"
        "            assert np.array_equal(cv.Fr.ifft(cv.Fr.fft(x)), x), ('roundtrip', name, lg)\n"
        "    x = synth.elems(0x2999, 1 << 13)\n"
        "    first, inc = O.fr_e(c, 7), O.fr_e(c, 11)\n"
        "    assert np.array_equal(cv.Fr.fft(cv.Fr.batchApplyKey(x, first, inc)), O.ntt(c, O.apply_key(c, x, first, inc)))\n"
        "print('ntt29 ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, ZKMI_NTT29=force))
    assert r.returncode == 0 and "ntt29 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.skipif(os.environ.get("ZKMI_TEST_NESTED") == "1", reason="already the nested run")
@pytest.mark.parametrize("env", [{"ZKMI_R29_REDUCE_G2": "0", "ZKMI_R29_REDUCE": "0"}, {"ZKMI_ACC29_BLOCK": "64"}, {"ZKMI_R29_G2": "0"}, {"ZKMI_COMPACT_CODE": "31", "ZKMI_G2_SPLIT": "0"},
                                 {"ZKMI_COMPACT_CODE": "0"}, {"ZKMI_G2_SPLIT": "0"}],
                         ids=["generic-rowcol-both-groups", "accum-64-thread-blocks", "generic-g2-accumulation", "compact-code-kernels", "inlined-kernels",
                              "g2-lds-parked-accumulators"])
def test_non_default_kernel_variants_parity(env):
    """The A/B switches select kernels that the default configuration no longer runs where a window table is resident (the generic 32-bit row /
    column sums of both groups, the generic Fq2 accumulation, other block shapes) or picks per box (the Compact instantiations with called
    products against the inlined ones: decided by a probe of the box unless ZKMI_COMPACT_CODE fixes it): the proof and MSM parity tests that
    reach those kernels are run again in a process with the switch flipped (the switches are read once per process)."""
    import subprocess
    import sys
    sel = "synthetic_vs_oracle or valid_key_proof_verifies or msm_resident_tables or two_proofs_in_flight"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=dict(os.environ, ZKMI_TEST_NESTED="1", **env))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", CURVES)
def test_ntt_golden_hashes(zk, golden_dir, name):
    d, cv = load(golden_dir, name), curve_of(zk, name)
    for lg, v in d["ntt"].items():
        x = synth.elems(v["seed"], 1 << int(lg))
        assert sha(cv.Fr.fft(x)) == v["fft"] and sha(cv.Fr.ifft(x)) == v["ifft"], lg


@pytest.mark.parametrize("name", CURVES)
def test_coset_chain_and_batch_golden(zk, golden_dir, name):
    d, c, cv = load(golden_dir, name), O.CURVE_ID[name], curve_of(zk, name)
    for lg, v in d["coset_chain"].items():
        lg = int(lg)
        x = synth.elems(v["seed"], 1 << lg)
        y = cv.Fr.fft(cv.Fr.batchApplyKey(cv.Fr.ifft(x), O.fr_one(c), O.fr_w(c, lg + 1)))
        assert sha(y) == v["out"], lg
    b = d["batch"]
    x = synth.elems(b["seed"], b["n"]).reshape(-1, 32)
    for z in b["zeroed"]:
        x[z] = 0
    x = x.reshape(-1)
    assert sha(cv.Fr.batchInverse(x)) == b["inverse"]
    assert sha(cv.Fr.batchToMontgomery(x)) == b["to_mont"]
    assert sha(cv.Fr.batchFromMontgomery(x)) == b["from_mont"]
    assert sha(cv.Fr.batchApplyKey(x, O.fr_e(c, 3), O.fr_e(c, 25))) == b["applykey_shift"]


@pytest.mark.parametrize("name", CURVES)
def test_fused_prescale_ntt(zk, name):
    """zkmi_ntt_dev with prescale == batchApplyKey followed by fft (the Groth16 coset step, src/groth16_prove.js:66-76)."""
    from snarkjs_amd import zkmi
    c = O.CURVE_ID[name]
    for lg in (0, 1, 3, 8, 11, 12, 15, 17):
        n = 1 << lg
        x = synth.elems(0x77 + lg, n)
        first, inc = O.fr_e(c, 5), O.fr_w(c, lg + 1)
        want = O.ntt(c, O.apply_key(c, x, first, inc))
        din, dout = zkmi.DeviceBuffer.from_host(x), zkmi.DeviceBuffer(n * 32)
        zkmi.check(zkmi.lib().zkmi_ntt_dev(c, din.ptr, dout.ptr, lg, 0, zkmi.ptr(first), zkmi.ptr(inc)))
        assert np.array_equal(dout.to_host(), want), lg
        # in-place inverse
        zkmi.check(zkmi.lib().zkmi_ntt_dev(c, dout.ptr, dout.ptr, lg, 1, None, None))
        assert np.array_equal(dout.to_host(), O.apply_key(c, x, first, inc)), lg


@pytest.mark.parametrize("name", CURVES)
def test_ntt_large_roundtrip(zk, name):
    """Size-independent property at BASELINE sizes: ifft(fft(x)) == x bytes; linearity spot check via x -> 2x."""
    cv = curve_of(zk, name)
    for lg in (20, 22, 24):
        x = synth.elems(0xABC + lg, 1 << lg)
        X = cv.Fr.fft(x)
        assert np.array_equal(cv.Fr.ifft(X), x), lg
        if lg == 20:   # DC term: X[0] = sum_j x[j] mod r (residues add linearly), independent host computation
            r = int(load(os.path.join(os.path.dirname(__file__), "golden"), name)["r"])
            v = x.reshape(-1, 32).view("<u8").astype(object)
            tot = sum(int(v[:, k].sum()) << (64 * k) for k in range(4)) % r
            assert int.from_bytes(bytes(X[:32]), "little") == tot


@pytest.mark.parametrize("name", CURVES)
def test_msm_golden_cases(zk, golden_dir, name):
    d, c, cv = load(golden_dir, name), O.CURVE_ID[name], curve_of(zk, name)
    _Q[c] = d["q"]
    B1, B2 = O.geom_bases(c, 1, 1 << 14), O.geom_bases(c, 2, 1 << 12)
    for key, v in d["msm"].items():
        g, bases, scalars, n, sb = msm_inputs(c, key, v, B1, B2)
        G = cv.G1 if g == 1 else cv.G2
        jac = G.multiExpAffine(bases, scalars)
        assert bytes(O.to_affine(c, g, jac)).hex() == v["affine"], key


@pytest.mark.parametrize("name", CURVES)
def test_msm_window_sweep(zk, name):
    """Every window width gives the same group element (different bucket layouts / reduction depths)."""
    from snarkjs_amd import zkmi
    c, cv = O.CURVE_ID[name], curve_of(zk, name)
    n = 3000
    B1 = O.geom_bases(c, 1, n)
    sc = synth.witness_like(0x4242, n)
    want = O.to_affine(c, 1, O.msm(c, 1, B1, sc, n))
    try:
        for cbits in (1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 16):
            zkmi.check(zkmi.lib().zkmi_msm_set_window_bits(cbits))
            assert np.array_equal(O.to_affine(c, 1, cv.G1.multiExpAffine(B1, sc)), want), cbits
    finally:
        zkmi.lib().zkmi_msm_set_window_bits(0)


def test_msm_edge_cases(zk):
    c, cv = O.BN128, curve_of(zk, "bn128")
    assert not cv.G1.multiExpAffine(b"", b"").any()
    B = O.geom_bases(c, 1, 8)
    assert not cv.G1.multiExpAffine(B, np.zeros(8 * 32, np.uint8)).any()
    with pytest.raises(ValueError):
        cv.G1.multiExpAffine(B, np.zeros(8 * 32 - 1, np.uint8))
    with pytest.raises(ValueError):
        cv.Fr.fft(np.zeros(3 * 32, np.uint8))
    # P + (-P) and P + P through the same bucket
    q = int(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn128_kernel_vectors.json")))["q"])
    P = B[:64].copy()
    y = int.from_bytes(bytes(P[32:]), "little")
    negP = np.concatenate([P[:32], np.frombuffer(((q - y) % q).to_bytes(32, "little"), np.uint8)])
    sc = np.zeros(64, np.uint8); sc[0] = 9; sc[32] = 9
    assert not O.to_affine(c, 1, cv.G1.multiExpAffine(np.concatenate([P, negP]), sc)).any()
    jac = cv.G1.multiExpAffine(np.concatenate([P, P]), sc)
    k = np.zeros(32, np.uint8); k[0] = 18
    assert np.array_equal(O.to_affine(c, 1, jac), O.to_affine(c, 1, O.msm(c, 1, P, k, 1)))
    # paged ("BigBuffer") inputs split at an arbitrary boundary give the same result
    n = 1000
    B1, sc = O.geom_bases(c, 1, n), synth.elems(0x31337, n)
    want = O.to_affine(c, 1, O.msm(c, 1, B1, sc, n))
    jac = cv.G1.multiExpAffine([B1[:64 * 300], B1[64 * 300:]], [sc[:32 * 700], sc[32 * 700:]])
    assert np.array_equal(O.to_affine(c, 1, jac), want)
    x = synth.elems(0x777, 1024)
    out = cv.Fr.fft([x[:32 * 300], x[32 * 300:]])          # <= one page: the reference returns a flat Uint8Array here
    assert isinstance(out, np.ndarray) and np.array_equal(out, O.ntt(c, x))
    out = cv.Fr.batchToMontgomery([x[:32 * 300], x[32 * 300:]])   # ... and the input's own container type here
    assert isinstance(out, list) and np.array_equal(np.concatenate(out), O.to_mont(c, x))


def test_msm_base_cache_is_content_addressed(zk):
    """The resident-base cache behind zkmi_msm (drop-in path: zkey sections / SRS slices stay on the device) must never serve a
    stale table: it is keyed by the full content of the base buffer. Two buffers that differ in ONE interior point give different
    (correct) results; a table is built on the second sight only; an MSM over a prefix of a resident buffer (PLONK's
    PTau.slice(0, k)) re-uses its table; paged input hashes like flat input; the byte budget evicts."""
    import ctypes as C
    from snarkjs_amd import zkmi
    L = zkmi.lib()
    c, cv = O.BN128, curve_of(zk, "bn128")
    n = 20000                                             # 1.28 MB of bases: 20 chunks of 64 KiB, the last one partial

    def stats():
        a, b, s_ = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        zkmi.check(L.zkmi_base_cache_stats(C.byref(a), C.byref(b), C.byref(s_)))
        return a.value, b.value, s_.value
    zkmi.check(L.zkmi_release_bases(0))
    B = O.geom_bases(c, 1, n)
    sc = synth.elems(0xCAC4E, n)
    want = O.to_affine(c, 1, O.msm(c, 1, B, sc, n))
    aff = lambda j: O.to_affine(c, 1, j)
    assert np.array_equal(aff(cv.G1.multiExpAffine(B, sc, cache_key=1)), want) and stats() == (0, 0, 1)        # 1st sight: remembered only
    assert np.array_equal(aff(cv.G1.multiExpAffine(B, sc, cache_key=1)), want) and stats()[0] == 1             # 2nd sight: table built
    assert np.array_equal(aff(cv.G1.multiExpAffine(B, sc, cache_key=1)), want) and stats()[0] == 1             # 3rd: table used
    # one interior point replaced (same length, same first / middle / last bytes): must NOT hit the resident table
    B2 = B.copy()
    i = 7777
    B2[i * 64:(i + 1) * 64] = B[(i + 1) * 64:(i + 2) * 64]
    want2 = O.to_affine(c, 1, O.msm(c, 1, B2, sc, n))
    assert not np.array_equal(want, want2)
    for _ in range(3):
        assert np.array_equal(aff(cv.G1.multiExpAffine(B2, sc, cache_key=1)), want2)
    assert stats()[0] == 2
    assert np.array_equal(aff(cv.G1.multiExpAffine(B, sc, cache_key=1)), want)
    # prefixes of a resident buffer re-use its table (chunk-aligned and not), paged input included
    for k in (n - 3, 16384, 5000):
        wk = O.to_affine(c, 1, O.msm(c, 1, B[:k * 64], sc[:k * 32], k))
        before = stats()
        assert np.array_equal(aff(cv.G1.multiExpAffine(B[:k * 64].copy(), sc[:k * 32], cache_key=1)), wk)
        assert np.array_equal(aff(cv.G1.multiExpAffine([B[:64 * 1234].copy(), B[64 * 1234:k * 64].copy()], sc[:k * 32], cache_key=1)), wk)
        assert stats() == before                          # no new table, no new entry
    # a prefix of the MODIFIED buffer that includes the modified point must come out as the modified result
    k = 9000
    wk2 = O.to_affine(c, 1, O.msm(c, 1, B2[:k * 64], sc[:k * 32], k))
    assert np.array_equal(aff(cv.G1.multiExpAffine(B2[:k * 64].copy(), sc[:k * 32], cache_key=1)), wk2)
    # The default (cache_key = 1 = ZKMI_BASES_CACHE): the full content hash on EVERY call, so the result always follows the bytes passed — a resident
    # buffer edited IN PLACE gives the edited buffer's result on the very next call, and the original's when the edit is undone (the reference's
    # contract: the result is a function of the bytes, min.js:1@213360).
    Bd = B.copy()
    for _ in range(3):
        assert np.array_equal(aff(cv.G1.multiExpAffine(Bd, sc, cache_key=1)), want)
    for _ in range(3):
        Bd[i * 64:(i + 1) * 64] = B[(i + 1) * 64:(i + 2) * 64]                               # in place: the content of B2 at Bd's address
        for _ in range(2):
            assert np.array_equal(aff(cv.G1.multiExpAffine(Bd, sc, cache_key=1)), want2)     # never the stale table's result
        Bd[i * 64:(i + 1) * 64] = B[i * 64:(i + 1) * 64]
        assert np.array_equal(aff(cv.G1.multiExpAffine(Bd, sc, cache_key=1)), want)
    # Opt-in (cache_key = 3: ZKMI_BASES_CACHE | ZKMI_BASES_IMMUTABLE, the caller's promise not to edit): a buffer that comes back at the SAME address
    # and length as one already checked byte for byte against a resident table is re-checked by sample (first, last and 30 pseudo-random chunks;
    # here 19 whole chunks: practically all of them) and in full on every 32nd sight (include/zkmi.h: zkmi_msm). A caller that breaks its promise
    # is noticed at the latest by that full check, and from then on the results are the edited buffer's.
    Bm = B.copy()
    for _ in range(3):
        assert np.array_equal(aff(cv.G1.multiExpAffine(Bm, sc, cache_key=3)), want)        # full check on first sight of this address, then samples
    Bm[i * 64:(i + 1) * 64] = B[(i + 1) * 64:(i + 2) * 64]                                   # in place: the content of B2 at Bm's address
    seen_new = False
    for _ in range(40):
        got = aff(cv.G1.multiExpAffine(Bm, sc, cache_key=3))
        if np.array_equal(got, want2):
            seen_new = True
        else:
            assert not seen_new and np.array_equal(got, want)                                # stale only BEFORE the edit was noticed, never after
    assert seen_new
    # the promise is per call: the same edited-in-place buffer passed WITHOUT it is hashed in full at once
    Bm[i * 64:(i + 1) * 64] = B[i * 64:(i + 1) * 64]
    assert np.array_equal(aff(cv.G1.multiExpAffine(Bm, sc, cache_key=1)), want)
    # without the permission bit nothing is cached
    zkmi.check(L.zkmi_release_bases(0))
    for _ in range(3):
        assert np.array_equal(aff(cv.G1.multiExpAffine(B, sc)), want)
    assert stats() == (0, 0, 0)


@pytest.mark.parametrize("name,group,lg", [("bn128", 1, 16), ("bn128", 1, 20), ("bn128", 2, 16), ("bls12381", 1, 16), ("bls12381", 2, 14)])
def test_msm_closed_form_large(zk, name, group, lg):
    """bases P_i = 7·11^i·G generated on the device, uniform 253-bit scalars;
    result must equal (sum s_i·7·11^i mod r)·G — an O(n) host computation independent of any MSM."""
    from snarkjs_amd import zkmi
    c = O.CURVE_ID[name]
    n = 1 << lg
    r = int(json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"{name}_kernel_vectors.json")))["r"])
    q8 = O.n8q(c)
    pb = 2 * group * q8
    d_b = zkmi.DeviceBuffer(n * pb)
    zkmi.check(zkmi.lib().zkmi_gen_geometric_bases_dev(c, group, n, 7, 11, d_b.ptr))
    # the generated table equals the oracle's (prefix check)
    m = min(n, 512)
    assert np.array_equal(d_b.to_host(m * pb), O.geom_bases(c, group, m))
    sc = synth.elems(0x5EED + lg, n)
    d_s = zkmi.DeviceBuffer.from_host(sc)
    out = np.zeros(3 * group * q8, np.uint8)
    zkmi.check(zkmi.lib().zkmi_msm_dev(c, group, d_b.ptr, d_s.ptr, n, 32, zkmi.ptr(out)))
    s = sc.reshape(n, 32).view("<u8").astype(object)      # n x 4 python ints
    k, f = 0, 7
    vals = [int(s[i, 0]) | int(s[i, 1]) << 64 | int(s[i, 2]) << 128 | int(s[i, 3]) << 192 for i in range(n)]
    for v in vals:
        k = (k + v * f) % r
        f = f * 11 % r
    want = O.to_affine(c, group, O.generator_mul(c, group, k))
    assert np.array_equal(O.to_affine(c, group, out), want)


@pytest.mark.parametrize("name", CURVES)
def test_join_abc(zk, name):
    c, cv = O.CURVE_ID[name], curve_of(zk, name)
    n = 5000
    a, b, cc = (synth.elems(s, n) for s in (1, 2, 3))
    assert np.array_equal(cv.joinABC(a, b, cc), O.join_abc(c, a, b, cc))


@pytest.mark.parametrize("tag,sha_proof", [("groth16_bn128_n1024", "08797809c8de2c2053a925b1772af3e04c41f8c9b4c41f5b71ae5135435da44d"),
                                           ("groth16_bls12381_n1024", "955b9f3652e544aac16a90fd1ce6b8660701a7eb20e71fa7681ced22f0acd5eb")])
def test_groth16_golden_proof(zk, golden_dir, tag, sha_proof):
    """The seeded Groth16 proofs generated by the reference itself (oracle/gen_golden.js): BN254 = SURVEY.md Appendix C.3, BLS12-381 =
    the Multiplier(1000) r1cs of SURVEY.md 8d over the BLS12-381 scalar field (the reference's own verifier accepted both). The fused
    device prover must emit the same proof JSON (sha256)."""
    from snarkjs_amd import groth16
    with open(os.path.join(golden_dir, tag + ".json")) as f:
        g = json.load(f)
    zkey = open(os.path.join(golden_dir, tag + ".zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, tag + ".wtns"), "rb").read()
    res = groth16.prove(zkey, wtns, r_mont=bytes.fromhex(g["r_mont"]), s_mont=bytes.fromhex(g["s_mont"]))
    assert res["proof"] == g["proof"]
    assert res["publicSignals"] == g["publicSignals"]
    assert sha(groth16.proof_to_json(res["proof"]).encode()) == g["proof_sha256"] == sha_proof
    with pytest.raises(ValueError):
        groth16.prove(zkey, wtns[:-32])


def test_groth16_two_proofs_in_flight(zk):
    """zkmi_groth16_submit_dev / _collect: two proofs in flight (different witnesses) in the two pipeline slots give exactly the serial
    proofs, in any collect order, repeatedly; a slot cannot be submitted twice without a collect."""
    import synth_zkey
    from snarkjs_amd import groth16, binfile, zkmi
    c = O.BN128
    zkey, wtns = synth_zkey.make("bn128", 15, seed=0x91E, b_zero_every=0)
    pk = groth16.ProvingKey(zkey)
    w0 = binfile.read_wtns(wtns)["witness"].copy()
    w1 = w0.copy()
    w1[32 * 7:32 * 8] = synth.elems(5, 1)                  # a different witness for the second slot
    r_m, s_m = O.fr_e(c, 21), O.fr_e(c, 34)
    want0 = [bytes(x) for x in pk.prove_raw(w0, r_m, s_m)]
    want1 = [bytes(x) for x in pk.prove_raw(w1, r_m, s_m)]
    assert want0 != want1
    d0, d1 = zkmi.DeviceBuffer.from_host(w0), zkmi.DeviceBuffer.from_host(w1)
    for rnd in range(6):
        pk.submit(d0.ptr, 0)
        pk.submit(d1.ptr, 1)
        with pytest.raises(Exception):
            pk.submit(d0.ptr, 0)
        order = (0, 1) if rnd % 2 == 0 else (1, 0)
        got = {sl: [bytes(x) for x in pk.collect(sl, r_m, s_m)] for sl in order}
        assert got[0] == want0 and got[1] == want1
    # steady-state pipeline: submit i+1 before collecting i
    seq = [d0, d1, d1, d0, d1, d0, d0]
    wants = [want0, want1, want1, want0, want1, want0, want0]
    outs = []
    for i, d in enumerate(seq):
        pk.submit(d.ptr, i & 1)
        if i:
            outs.append([bytes(x) for x in pk.collect((i - 1) & 1, r_m, s_m)])
    outs.append([bytes(x) for x in pk.collect((len(seq) - 1) & 1, r_m, s_m)])
    assert outs == wants
    with pytest.raises(Exception):
        pk.collect(0, r_m, s_m)                           # nothing in flight
    assert [bytes(x) for x in pk.prove_raw(w0, r_m, s_m)] == want0
    d0.free(); d1.free()
    pk.release()


@pytest.mark.parametrize("name,lg", [("bn128", 6), ("bn128", 16), ("bn128", 20), ("bls12381", 8), ("bls12381", 16)])
def test_groth16_valid_key_proof_verifies(zk, golden_dir, name, lg):
    """proofs on a synthetic VALID key (tests/synth_valid_groth16.py, known trapdoor; the small BN254 instance is the one the
    reference itself exported / proved / verified) VERIFY under the pinned verifier restatement (oracle/groth16_verify_oracle.py: BN254 and
    BLS12-381 pairings), at 2^16 on both curves and at BASELINE configs[1]'s full size 2^20; a wrong public input or a swapped proof point is
    rejected. bn128 lg = 6: the device proof equals the reference's seeded proof bit for bit."""
    import copy
    import sys
    import synth_valid_groth16 as SV
    from snarkjs_amd import groth16
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import groth16_verify_oracle as V
    zkey, wtns, info = SV.make(name, lg, use_device=True)
    if name == "bn128" and lg == 6:
        g = json.load(open(os.path.join(golden_dir, "groth16_valid_synth_n64.json")))
        assert sha(zkey) == g["zkey_sha256"]                   # device-generated points == the oracle's (and the reference accepted them)
        res = groth16.prove(zkey, wtns, r_mont=bytes.fromhex(g["r_mont"]), s_mont=bytes.fromhex(g["s_mont"]))
        assert res["proof"] == g["proof"] and res["publicSignals"] == g["publicSignals"]
    else:
        res = groth16.prove(zkey, wtns)                        # fresh random r, s
    assert V.groth16_verify(info["vk"], res["publicSignals"], res["proof"]) is True
    pub = list(res["publicSignals"])
    pub[1] = str(int(pub[1]) + 1)
    assert V.groth16_verify(info["vk"], pub, res["proof"]) is False
    bad = copy.deepcopy(res["proof"])
    bad["pi_a"], bad["pi_c"] = bad["pi_c"], bad["pi_a"]
    assert V.groth16_verify(info["vk"], res["publicSignals"], bad) is False


@pytest.mark.parametrize("name", CURVES)
def test_group_fft_and_apply_key(zk, golden_dir, name):
    """SURVEY.md 8 f4 (ceremony side): G1/G2.fft / ifft / lagrangeEvaluations / batchApplyKey on the device vs the reference's golden vectors
    (oracle/gen_golden.js groupVectors) bit for bit, vs the C restatement at 2^10, round trip at 2^14."""
    c, cv = O.CURVE_ID[name], curve_of(zk, name)
    g = json.load(open(os.path.join(golden_dir, f"{name}_group_vectors.json")))
    q8 = O.n8q(c)
    for gn, group, G in (("g1", 1, cv.G1), ("g2", 2, cv.G2)):
        v, pb = g[gn], 2 * group * q8
        n = v["n"]
        bases = O.geom_bases(c, group, n)
        assert sha(G.fft(bases)) == v["fft"] and sha(G.ifft(bases)) == v["ifft"] and sha(G.lagrangeEvaluations(bases)) == v["lagrange"]
        assert sha(G.batchApplyKey(bases, O.fr_e(c, 3), O.fr_e(c, 5))) == v["applykey_3_5"]
        b2 = bases.copy()
        b2[5 * pb:6 * pb] = 0
        b2[9 * pb:10 * pb] = bases[:pb]
        assert sha(G.fft(b2)) == v["fft_with_zero_and_repeat"]
        for k in (1, 2, 4, 32):
            assert sha(G.fft(bases[:k * pb])) == v[f"fft_n{k}"] and sha(G.ifft(bases[:k * pb])) == v[f"ifft_n{k}"]
        with pytest.raises(ValueError):
            G.fft(bases[:3 * pb])
        # paged ("BigBuffer") input
        assert sha(G.fft([bases[:100 * pb // 2], bases[100 * pb // 2:]])) == v["fft"]
    m = 1 << (10 if name == "bn128" else 8)
    B = O.geom_bases(c, 1, m)
    assert np.array_equal(cv.G1.fft(B), O.group_fft(c, 1, B)) and np.array_equal(cv.G1.ifft(B), O.group_fft(c, 1, B, inverse=True))
    f, inc = synth.elems(0x61, 1), synth.elems(0x62, 1)
    assert np.array_equal(cv.G1.batchApplyKey(B, O.to_mont(c, f), O.to_mont(c, inc)), O.group_apply_key(c, 1, B, O.to_mont(c, f), O.to_mont(c, inc)))
    big = O.geom_bases(c, 1, 64)
    big = np.tile(big, (1 << 14) // 64)
    big = cv.G1.batchApplyKey(big, O.fr_e(c, 2), O.fr_e(c, 3))            # 2^14 distinct points
    assert np.array_equal(cv.G1.ifft(cv.G1.fft(big)), big)


@pytest.mark.parametrize("name", CURVES)
def test_point_format_conversions(zk, golden_dir, name):
    """G.batchLEMtoU / batchUtoLEM / batchLEMtoC / batchCtoLEM on the device vs the reference's byte strings (oracle/gen_golden.js
    convertVectors: points at infinity inside), vs the C restatement on 2^12 points, compress -> decompress round trip at 2^16 (G1) / 2^14 (G2), a
    compressed x off the curve refused, "Invalid buffer size" on ragged input."""
    from snarkjs_amd import zkmi
    c, cv = O.CURVE_ID[name], curve_of(zk, name)
    g = json.load(open(os.path.join(golden_dir, f"{name}_conv_vectors.json")))
    q8 = O.n8q(c)
    for gn, group, G in (("g1", 1, cv.G1), ("g2", 2, cv.G2)):
        v, pb = g[gn], 2 * group * q8
        rd = lambda k: np.frombuffer(open(os.path.join(golden_dir, f"{name}_conv_{gn}_n{v['n']}_{k}.bin"), "rb").read(), np.uint8)
        lem, U, Cc = rd("lem"), rd("u"), rd("c")
        assert bytes(G.batchLEMtoU(lem)) == bytes(U) and bytes(G.batchLEMtoC(lem)) == bytes(Cc)
        assert bytes(G.batchUtoLEM(U)) == bytes(lem) and bytes(G.batchCtoLEM(Cc)) == bytes(lem)
        assert bytes(G.batchLEMtoU([lem[:5 * pb], lem[5 * pb:]])) == bytes(U)            # paged ("BigBuffer") input
        assert G.batchLEMtoU(lem[:0]).size == 0
        with pytest.raises(ValueError):
            G.batchLEMtoU(lem[:pb + 1])
        m = 1 << 12
        B = O.geom_bases(c, group, 64)
        B = G.batchApplyKey(np.tile(B, m // 64), O.fr_e(c, 5), O.fr_e(c, 7))              # 2^12 distinct points
        B[3 * pb:4 * pb] = 0
        for kind, fn in (("LEMtoU", G.batchLEMtoU), ("LEMtoC", G.batchLEMtoC)):
            assert np.array_equal(fn(B), O.group_convert(c, group, kind, B)), kind
        assert np.array_equal(G.batchUtoLEM(O.group_convert(c, group, "LEMtoU", B)), B)
        assert np.array_equal(G.batchCtoLEM(O.group_convert(c, group, "LEMtoC", B)), B)
        big = 1 << (16 if group == 1 else 14)
        P = G.batchApplyKey(np.tile(B[:64 * pb], big // 64), O.fr_e(c, 11), O.fr_e(c, 13))
        comp = G.batchLEMtoC(P)
        assert comp.size == P.size // 2 and np.array_equal(G.batchCtoLEM(comp), P) and np.array_equal(G.batchUtoLEM(G.batchLEMtoU(P)), P)
        bad, refused = Cc.copy(), False
        for delta in range(1, 40):
            bad[group * q8 - 1] = (int(Cc[group * q8 - 1]) + delta) & 0xff
            try:
                G.batchCtoLEM(bad)
            except zkmi.ZkmiError as e:
                refused = "not on the curve" in str(e)
                break
        assert refused


def test_groth16_malformed_inputs_fail_cleanly(zk, golden_dir):
    """Truncated sections / witness and key-number re-use must fail with an error, never read past the caller's buffers or prove
    against another circuit's key (include/zkmi.h: zkmi_groth16_zkey *_len fields, zkmi_groth16_prove witness_len)."""
    import ctypes as C
    from snarkjs_amd import groth16, zkmi, binfile
    L = zkmi.lib()
    zkey = open(os.path.join(golden_dir, "groth16_bn128_n1024.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, "groth16_bn128_n1024.wtns"), "rb").read()
    w = binfile.read_wtns(wtns)["witness"]
    pk = groth16.ProvingKey(zkey)
    r_m, s_m = O.fr_e(0, 3), O.fr_e(0, 5)
    good = [bytes(x) for x in pk.prove_raw(w, r_m, s_m)]
    out = [np.zeros(64, np.uint8), np.zeros(128, np.uint8), np.zeros(64, np.uint8)]
    short = zkmi.u8(w[:-32])
    rc = L.zkmi_groth16_prove(None, pk.key, zkmi.ptr(short), short.size, zkmi.ptr(r_m), zkmi.ptr(s_m), *(zkmi.ptr(o) for o in out))
    assert rc != 0 and b"Invalid witness length" in L.zkmi_last_error()
    # a descriptor of ANOTHER circuit next to the resident key number is refused (it used to prove against the resident key)
    import synth_zkey
    other, _ = synth_zkey.make("bn128", 8, seed=5)
    pk2 = groth16.ProvingKey(other)
    w8 = zkmi.u8(w)
    rc = L.zkmi_groth16_prove(C.byref(pk2.desc), pk.key, zkmi.ptr(w8), w8.size, zkmi.ptr(r_m), zkmi.ptr(s_m), *(zkmi.ptr(o) for o in out))
    assert rc != 0 and b"different circuit" in L.zkmi_last_error()
    pk2.release()
    # a section shorter than the header requires
    d = pk.desc
    bad = zkmi.Groth16Zkey.from_buffer_copy(bytes(d))
    bad.bases_h_len = d.bases_h_len - 64
    assert L.zkmi_groth16_load(C.byref(bad), 0x7777) != 0 and b"shorter" in L.zkmi_last_error()
    assert [bytes(x) for x in pk.prove_raw(w, r_m, s_m)] == good          # the resident key is intact
    pk.release()


@pytest.mark.parametrize("name,lg", [("bn128", 12), ("bn128", 16), ("bls12381", 12)])
def test_groth16_synthetic_vs_oracle(zk, name, lg):
    """Synthetic zkey/wtns of SURVEY.md §8d ('fast' variant): device prover == CPU oracle, point for point."""
    import synth_zkey
    from snarkjs_amd import groth16, binfile
    c = O.CURVE_ID[name]
    zkey, wtns = synth_zkey.make(name, lg, seed=0xC0FFEE + lg)
    r_m, s_m = O.fr_e(c, 0x1234567), O.fr_e(c, 0x7654321)
    pk = groth16.ProvingKey(zkey)
    w = binfile.read_wtns(wtns)["witness"]
    got = pk.prove_raw(w, r_m, s_m)
    want = O.groth16_prove(c, binfile.read_groth16_zkey(zkey), w, r_m, s_m)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # proving twice with the resident key gives the same bytes (no state leaks between proofs)
    again = pk.prove_raw(w, r_m, s_m)
    for a, b in zip(got, again):
        assert np.array_equal(a, b)
    pk.release()


@pytest.mark.parametrize("name,group,lg", [("bn128", 1, 14), ("bn128", 2, 12), ("bls12381", 1, 12), ("bls12381", 2, 11)])
def test_msm_resident_tables(zk, name, group, lg):
    """zkmi_msm_table_*: pre-computed window tables; MSMs over a PREFIX of the resident bases (PLONK commits with PTau[0:k])."""
    import ctypes as C
    from snarkjs_amd import zkmi
    c, L = O.CURVE_ID[name], zkmi.lib()
    n = 1 << lg
    q8 = O.n8q(c)
    bases = O.geom_bases(c, group, n)
    d_b = zkmi.DeviceBuffer.from_host(bases)
    h = C.c_uint64(0)
    zkmi.check(L.zkmi_msm_table_build(c, group, d_b.ptr, n, C.byref(h)))
    for k, sb, seed in ((n, 32, 1), (n - 5, 32, 2), (1, 32, 3), (777, 4, 4)):
        sc = synth.witness_like(0x7AB1E + seed, k) if sb == 32 else synth.elems(9, (k * sb + 31) // 32 + 1)[:k * sb]
        d_s = zkmi.DeviceBuffer.from_host(sc)
        out = np.zeros(3 * group * q8, np.uint8)
        zkmi.check(L.zkmi_msm_table_dev(h, d_s.ptr, k, sb, zkmi.ptr(out)))
        want = O.to_affine(c, group, O.msm(c, group, bases[:k * 2 * group * q8], sc, k, sb))
        assert np.array_equal(O.to_affine(c, group, out), want), (k, sb)
    assert L.zkmi_msm_table_dev(h, d_s.ptr, n + 1, 32, zkmi.ptr(out)) != 0
    zkmi.check(L.zkmi_msm_table_release(h))


@pytest.mark.parametrize("name,group", [("bn128", 1), ("bn128", 2), ("bls12381", 1), ("bls12381", 2)])
def test_msm_resident_tables_special_cases(zk, name, group):
    """Resident window tables (the unsaturated-limb accumulation kernels of msm29.cuh: 9 x 29 bits on BN254, 14 x 28 bits on BLS12-381) on inputs that force the special cases of the
    mixed addition inside one bucket: the same base many times with the same small scalar (P + P doubling branch), a base next to its
    negation (P - P -> infinity and back), points at infinity in the base array, zero scalars, scalars with every digit at the maximum."""
    import ctypes as C
    from snarkjs_amd import zkmi
    c, L = O.CURVE_ID[name], zkmi.lib()
    q8 = O.n8q(c)
    pb = 2 * group * q8
    q = int(json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"{name}_kernel_vectors.json")))["q"])
    n = 4096
    G = O.geom_bases(c, group, 16).reshape(16, pb)
    bases = np.zeros((n, pb), np.uint8)
    for i in range(n):
        bases[i] = G[i % 5]                                   # heavy repetition: equal points meet in one bucket
    neg = G[1].copy()                                          # -G[1]: negate y (Fq, or both Fq2 components)
    for k in range(group):
        y = int.from_bytes(bytes(G[1][(group + k) * q8:(group + k + 1) * q8]), "little")
        neg[(group + k) * q8:(group + k + 1) * q8] = np.frombuffer(((q - y) % q).to_bytes(q8, "little"), np.uint8)
    bases[7::11] = neg
    bases[3::17] = 0                                           # points at infinity
    bases = bases.reshape(-1)
    cases = []
    sc = np.zeros((n, 32), np.uint8); sc[:, 0] = 5                                   # one digit, same bucket for everything
    cases.append(sc.copy())
    sc = np.zeros((n, 32), np.uint8); sc[:, 0] = (np.arange(n) % 3).astype(np.uint8)   # zeros, ones, twos
    cases.append(sc.copy())
    sc = synth.elems(0xC0DE, n).reshape(n, 32).copy(); sc[::2] = sc[1::2]              # pairs of equal full-width scalars on (often) equal bases
    cases.append(sc.copy())
    sc = np.full((n, 32), 0xFF, np.uint8); sc[:, 31] = 0x1F                            # every signed digit at its extreme
    cases.append(sc.copy())
    d_b = zkmi.DeviceBuffer.from_host(bases)
    h = C.c_uint64(0)
    zkmi.check(L.zkmi_msm_table_build(c, group, d_b.ptr, n, C.byref(h)))
    for idx, sc in enumerate(cases):
        sc = sc.reshape(-1)
        d_s = zkmi.DeviceBuffer.from_host(sc)
        out = np.zeros(3 * group * q8, np.uint8)
        zkmi.check(L.zkmi_msm_table_dev(h, d_s.ptr, n, 32, zkmi.ptr(out)))
        want = O.to_affine(c, group, O.msm(c, group, bases, sc, n))
        assert np.array_equal(O.to_affine(c, group, out), want), idx
        d_s.free()
    zkmi.check(L.zkmi_msm_table_release(h))
    d_b.free()


_groth16_closed_form = O.groth16_closed_form          # tests/oracle_lib.py (bench.py's configs[2] line checks its 2^24 proof with it in-run)


@pytest.mark.parametrize("name,lg,b_zero_every", [("bn128", 20, 3), ("bn128", 20, 0), ("bls12381", 20, 3), ("bls12381", 20, 0), ("bn128", 24, 0)])
def test_groth16_full_size_closed_form(zk, name, lg, b_zero_every):
    """BASELINE configs[1], [4] (2^20 constraints, sparse and dense B sections) and configs[2] (2^24 constraints, single device AND
    8 key shards folded as the ranks would after the all_gather) at their full sizes, checked through a size-independent property:
    every proof point has a closed-form discrete log (see _groth16_closed_form), bit-exact affine bytes."""
    import synth_zkey
    from snarkjs_amd import groth16, binfile
    from snarkjs_amd import distributed as D
    c = O.CURVE_ID[name]
    n_public = 2
    zkey, wtns = synth_zkey.make(name, lg, seed=0xBEEF + lg, n_public=n_public, b_zero_every=b_zero_every)
    zk_ = binfile.read_groth16_zkey(zkey)
    w = binfile.read_wtns(wtns)["witness"]
    rr, ss = 0x1234567, 0x7654321
    r_m, s_m = O.fr_e(c, rr), O.fr_e(c, ss)
    pk = groth16.ProvingKey(zkey)
    got = [pk.prove_raw(w, r_m, s_m)]
    pk.release()
    if lg >= 24:                                               # configs[2]: the same proof from 8 base-index-range shards,
        world, keys = 8, []                                    # chain-parallel transforms + slice exchange (distributed.py), ranks simulated here

        def make(rank):
            keys.append(groth16.ProvingKey(zkey, shard=(rank, world)))
            return D.DeviceShard(keys[-1], w)
        got.append(D.groth16_prove_sharded_local(make, world, r_m, s_m))
        for k in keys:
            k.release()
    del zkey
    a, b, cc = _groth16_closed_form(c, name, zk_, w, lg, n_public, rr, ss, b_zero_every)
    want = (O.to_affine(c, 1, O.generator_mul(c, 1, a)), O.to_affine(c, 2, O.generator_mul(c, 2, b)), O.to_affine(c, 1, O.generator_mul(c, 1, cc)))
    for pi_a, pi_b, pi_c in got:
        assert np.array_equal(pi_a, want[0])
        assert np.array_equal(pi_b, want[1])
        assert np.array_equal(pi_c, want[2])


@pytest.mark.parametrize("dist", ["equal", "zero", "ones", "witness", "two_values", "max"])
@pytest.mark.parametrize("tables", [False, True])
def test_msm_sort_skewed_distributions(zk, dist, tables):
    """The LDS radix partition of the digit sort (msm.cuh: k_rsort_*) under distributions that send most entries to a handful of
    buckets (real witnesses are mostly 0/1), with and without window tables; closed-form check on the geometric bases, n = 2^18."""
    import ctypes as C
    from snarkjs_amd import zkmi
    c, group, lg = 0, 1, 18
    n = 1 << lg
    r = synth_r = int(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bn128_kernel_vectors.json")))["r"])
    L = zkmi.lib()
    d_b = zkmi.DeviceBuffer(n * 64)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(c, group, n, 7, 11, d_b.ptr))
    full = synth.elems(0xD157, n).reshape(n, 32).copy()
    if dist == "equal":
        full[:] = full[0]
    elif dist == "zero":
        full[:] = 0
    elif dist == "ones":
        full[:] = 0; full[:, 0] = 1
    elif dist == "witness":
        full = synth.witness_like(0xD158, n).reshape(n, 32).copy()
    elif dist == "two_values":
        full[::2] = full[0]; full[1::2] = full[1]
    elif dist == "max":
        full[:] = 0xFF                                     # 2^256 - 1: scalars >= r are not reduced
    sc = np.ascontiguousarray(full.reshape(-1))
    d_s = zkmi.DeviceBuffer.from_host(sc)
    out = np.zeros(96, np.uint8)
    if tables:
        h = C.c_uint64(0)
        zkmi.check(L.zkmi_msm_table_build(c, group, d_b.ptr, n, C.byref(h)))
        zkmi.check(L.zkmi_msm_table_dev(h, d_s.ptr, n, 32, zkmi.ptr(out)))
        zkmi.check(L.zkmi_msm_table_release(h))
    else:
        zkmi.check(L.zkmi_msm_dev(c, group, d_b.ptr, d_s.ptr, n, 32, zkmi.ptr(out)))
    s = full.view("<u8").astype(object)
    k, f = 0, 7
    for i in range(n):
        v = int(s[i, 0]) | int(s[i, 1]) << 64 | int(s[i, 2]) << 128 | int(s[i, 3]) << 192
        k = (k + v * f) % r
        f = f * 11 % r
    want = O.to_affine(c, group, O.generator_mul(c, group, k))
    assert np.array_equal(O.to_affine(c, group, out), want)


@pytest.mark.parametrize("name,sb,lg", [("bn128", 4, 17), ("bn128", 48, 15), ("bls12381", 4, 16), ("bn128", 33, 15)])
def test_msm_scalar_widths_large(zk, name, sb, lg):
    """Scalar widths other than 32 bytes on the large-input (LDS radix) sort path: 4-byte scalars (src/powersoftau_verify.js:371),
    48-byte and odd-width scalars; values >= r are not reduced by the reference, the group does it. Closed form on geometric bases."""
    from snarkjs_amd import zkmi
    c, L = O.CURVE_ID[name], zkmi.lib()
    n = 1 << lg
    r = int(json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"{name}_kernel_vectors.json")))["r"])
    q8 = O.n8q(c)
    d_b = zkmi.DeviceBuffer(n * 2 * q8)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(c, 1, n, 7, 11, d_b.ptr))
    raw = synth.elems(0xABC + sb, (n * sb + 31) // 32 + 1)[:n * sb].copy()
    raw[sb - 1::sb] |= 0x80                                      # top bit set: exercises the carry out of the last signed window
    d_s = zkmi.DeviceBuffer.from_host(raw)
    out = np.zeros(3 * q8, np.uint8)
    zkmi.check(L.zkmi_msm_dev(c, 1, d_b.ptr, d_s.ptr, n, sb, zkmi.ptr(out)))
    k, f = 0, 7
    rb = raw.tobytes()
    for i in range(n):
        k = (k + int.from_bytes(rb[i * sb:(i + 1) * sb], "little") * f) % r
        f = f * 11 % r
    assert np.array_equal(O.to_affine(c, 1, out), O.to_affine(c, 1, O.generator_mul(c, 1, k)))


def test_soak_many_proofs_same_key(zk):
    """200 Groth16 proofs + 40 PLONK proofs back to back on resident keys: identical bytes every time (event / constant-ring / plan
    buffer reuse across proofs) and no growth of device memory after the first proofs."""
    import ctypes
    import synth_plonk
    import synth_zkey
    from snarkjs_amd import groth16, binfile, plonk
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
        return fr.value
    zkey, wtns = synth_zkey.make("bn128", 14, seed=77)
    pk = groth16.ProvingKey(zkey)
    w = binfile.read_wtns(wtns)["witness"]
    r_m, s_m = O.fr_e(0, 11), O.fr_e(0, 13)
    first = [bytes(x) for x in pk.prove_raw(w, r_m, s_m)]
    pk.prove_raw(w, r_m, s_m)
    free0 = free_bytes()
    for _ in range(200):
        assert [bytes(x) for x in pk.prove_raw(w, r_m, s_m)] == first
    pzkey, pwtns = synth_plonk.make("bn128", 12, seed=5)
    key = plonk.PlonkKey(pzkey)
    f = plonk._Field(0)
    blind = [bytes(f.mont(900 + i)) for i in range(11)]
    p0 = plonk.prove(key, pwtns, blinding_mont=blind)
    for _ in range(40):
        assert plonk.prove(key, pwtns, blinding_mont=blind) == p0
    key.release()
    pk.release()
    assert free_bytes() >= free0 - (64 << 20)


@pytest.mark.parametrize("name,lg,world,bze", [("bn128", 12, 3, 3), ("bn128", 16, 8, 3), ("bls12381", 12, 2, 3), ("bn128", 10, 1, 3), ("bn128", 16, 4, 0), ("bls12381", 14, 3, 0)])
def test_groth16_sharded_equals_single_device(zk, name, lg, world, bze):
    """BASELINE configs[2] (MSMs sharded across the GPUs of a node by base-index range): the `world` key shards are loaded one after
    the other on this one GPU, their partial MSM sums are folded as the ranks would after the all_gather, and the finished proof
    must equal the single-device proof bit for bit (and the CPU oracle's at the small sizes)."""
    import synth_zkey
    from snarkjs_amd import groth16, binfile
    from snarkjs_amd import distributed as D
    c = O.CURVE_ID[name]
    zkey, wtns = synth_zkey.make(name, lg, seed=0x5A4D + lg, n_public=3, b_zero_every=bze)      # bze = 0: dense B1 / B2 sections
    w = binfile.read_wtns(wtns)["witness"]
    r_m, s_m = O.fr_e(c, 0xAAA1), O.fr_e(c, 0xBBB2)
    full = groth16.ProvingKey(zkey)
    want = [bytes(x) for x in full.prove_raw(w, r_m, s_m)]
    # sums + finish on the full key is the same thing as prove
    assert [bytes(x) for x in full.finish_raw(full.sums_raw(w), r_m, s_m)] == want
    full.release()
    parts = []
    for rank in range(world):
        pk = groth16.ProvingKey(zkey, shard=(rank, world))
        parts.append(pk.sums_raw(w))
        if world > 1:
            with pytest.raises(Exception):
                pk.prove_raw(w, r_m, s_m)                  # a shard cannot finish a proof on its own
        last = pk if rank == world - 1 else pk.release()
    got = [bytes(x) for x in last.finish_raw(D.fold_groth16_sums(c, parts), r_m, s_m)]
    last.release()
    assert got == want
    # chain-parallel flow (what distributed.groth16_prove_sharded runs over RCCL): chain c on rank c % world, slices of the chain
    # outputs exchanged, every rank joins ITS slice and runs its shard's MSMs with those H scalars — all ranks simulated on this GPU
    keys = []

    def make(rank):
        keys.append(groth16.ProvingKey(zkey, shard=(rank, world)))
        return D.DeviceShard(keys[-1], w)
    got2 = [bytes(x) for x in D.groth16_prove_sharded_local(make, world, r_m, s_m)]
    for k in keys:
        k.release()
    assert got2 == want
    if lg <= 12:
        ref = O.groth16_prove(c, binfile.read_groth16_zkey(zkey), w, r_m, s_m)
        assert got == [bytes(x) for x in ref]
