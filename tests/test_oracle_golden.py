"""Pins the C restatement (oracle/zk_oracle.c) against golden vectors produced by the REAL reference
(oracle/gen_golden.js running /root/reference/build/snarkjs.min.js).  CPU only.

Covers SURVEY.md §8a rows a1-a9 at the byte level (NTT / batch ops) or as group elements after toAffine (MSM),
and the seeded Groth16 proof of Appendix C.3 (proof JSON hash).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import binfile
import oracle_lib as O
import synth

CURVES = ["bn128", "bls12381"]
sha = lambda b: hashlib.sha256(bytes(b)).hexdigest()


def load(golden_dir, name):
    with open(os.path.join(golden_dir, f"{name}_kernel_vectors.json")) as f:
        return json.load(f)


def raw(golden_dir, name, key):
    return np.fromfile(os.path.join(golden_dir, f"{name}_n1024_{key}.bin"), dtype=np.uint8)


@pytest.mark.parametrize("name", CURVES)
def test_constants(golden_dir, name):
    d, c = load(golden_dir, name), O.CURVE_ID[name]
    assert bytes(O.fr_one(c)).hex() == d["Fr_one"]
    assert O.lib().orc_two_adicity(c) == d["s"]
    for i in range(d["s"] + 1):
        assert bytes(O.fr_w(c, i)).hex() == d["w"][i]
    assert bytes(O.generator(c, 1)).hex() == d["G1_g"]
    assert bytes(O.generator(c, 2)).hex() == d["G2_g"]
    assert O.n8q(c) == d["n8q"]


@pytest.mark.parametrize("name", CURVES)
def test_n1024_bytes(golden_dir, name):
    d, c = load(golden_dir, name), O.CURVE_ID[name]
    x = synth.iota(1024)
    got = {
        "fft": O.ntt(c, x), "ifft": O.ntt(c, x, True),
        "applykey_7_11": O.apply_key(c, x, O.fr_e(c, 7), O.fr_e(c, 11)),
        "to_mont": O.to_mont(c, x), "from_mont": O.from_mont(c, x), "inverse": O.batch_inverse(c, x),
        "g1_bases": O.geom_bases(c, 1, 1024), "g2_bases": O.geom_bases(c, 2, 1024),
    }
    for k, v in got.items():
        assert np.array_equal(v, raw(golden_dir, name, k)), k
        assert sha(v) == d["n1024"][k]
    for g in (1, 2):
        jac = O.msm(c, g, got[f"g{g}_bases"], x, 1024)
        assert np.array_equal(O.to_affine(c, g, jac), raw(golden_dir, name, f"g{g}_msm_affine"))


@pytest.mark.parametrize("name", CURVES)
def test_ntt_sizes(golden_dir, name):
    d, c = load(golden_dir, name), O.CURVE_ID[name]
    for lg, v in d["ntt"].items():
        x = synth.elems(v["seed"], 1 << int(lg))
        f, fi = O.ntt(c, x), O.ntt(c, x, True)
        assert sha(f) == v["fft"] and sha(fi) == v["ifft"], lg
        assert np.array_equal(O.ntt(c, fi), x)


@pytest.mark.parametrize("name", CURVES)
def test_coset_chain_and_batch(golden_dir, name):
    d, c = load(golden_dir, name), O.CURVE_ID[name]
    for lg, v in d["coset_chain"].items():
        lg = int(lg)
        x = synth.elems(v["seed"], 1 << lg)
        y = O.ntt(c, O.apply_key(c, O.ntt(c, x, True), O.fr_one(c), O.fr_w(c, lg + 1)))
        assert sha(y) == v["out"], lg
    b = d["batch"]
    x = synth.elems(b["seed"], b["n"]).reshape(-1, 32)
    for z in b["zeroed"]:
        x[z] = 0
    x = x.reshape(-1)
    assert sha(O.batch_inverse(c, x)) == b["inverse"]
    assert sha(O.to_mont(c, x)) == b["to_mont"]
    assert sha(O.from_mont(c, x)) == b["from_mont"]
    assert sha(O.apply_key(c, x, O.fr_e(c, 3), O.fr_e(c, 25))) == b["applykey_shift"]   # Fr.shift = nqr^2 = 25


def msm_inputs(c, key, v, B1, B2):
    """Rebuild the inputs of one golden MSM case (mirror of oracle/gen_golden.js kernelVectors)."""
    q = O.n8q(c)
    s1, s2 = 2 * q, 4 * q
    n = v.get("n", 0)
    if key.startswith("g1_uniform") or key == "g1_full256_2048":
        return 1, B1[: n * s1], synth.elems(v["seed"], n, v.get("mask", 0x1F)), n, 32
    if key.startswith("g2_uniform"):
        return 2, B2[: n * s2], synth.elems(v["seed"], n), n, 32
    if key == "g1_witnesslike_16384":
        return 1, B1, synth.witness_like(v["seed"], n), n, 32
    if key == "g2_witnesslike_4096":
        return 2, B2, synth.witness_like(v["seed"], n), n, 32
    if key == "g1_allff_64":
        return 1, B1[: 64 * s1], np.full(64 * 32, 0xFF, np.uint8), 64, 32
    if key == "g1_scalar4B_1024":
        return 1, B1[: 1024 * s1], synth.elems(v["seed"], 128)[:4096], 1024, 4
    if key == "g1_zeros_1024":
        sc = synth.elems(v["seed"], 1024).reshape(1024, 32).copy()
        sc[0::3] = 0
        bz = B1[: 1024 * s1].reshape(1024, s1).copy()
        bz[1::5] = 0
        return 1, bz.reshape(-1), sc.reshape(-1), 1024, 32
    if key == "g1_repeated_512":
        idx = np.arange(512) & 3
        bz = B1.reshape(-1, s1)[idx].reshape(-1).copy()
        sc = np.zeros((512, 32), np.uint8)
        sc[:, 0] = 5
        sc[:, 2] = np.arange(512) >> 6
        return 1, bz, sc.reshape(-1), 512, 32
    if key == "g1_cancel_2":
        P = B1[:s1].copy()
        qmod = int(load.__globals__["_Q"][c])
        y = int.from_bytes(bytes(P[q:]), "little")
        negy = np.frombuffer(((qmod - y) % qmod).to_bytes(q, "little"), np.uint8)
        bz = np.concatenate([P, P[:q], negy])
        sc = np.zeros(64, np.uint8)
        sc[0] = 9
        sc[32] = 9
        return 1, bz, sc, 2, 32
    if key == "g1_allzero_scalars_100":
        return 1, B1[: 100 * s1], np.zeros(3200, np.uint8), 100, 32
    if key == "g1_empty":
        return 1, np.zeros(0, np.uint8), np.zeros(0, np.uint8), 0, 32
    raise KeyError(key)


_Q = {}


@pytest.mark.parametrize("name", CURVES)
def test_msm_cases(golden_dir, name):
    d, c = load(golden_dir, name), O.CURVE_ID[name]
    _Q[c] = d["q"]
    B1, B2 = O.geom_bases(c, 1, 1 << 14), O.geom_bases(c, 2, 1 << 12)
    assert sha(B1) == d["g1_bases_16384"] and sha(B2) == d["g2_bases_4096"]
    for key, v in d["msm"].items():
        g, bases, scalars, n, sb = msm_inputs(c, key, v, B1, B2)
        jac = O.msm(c, g, bases, scalars, n, sb)
        assert bytes(O.to_affine(c, g, jac)).hex() == v["affine"], key
        if n <= 100:   # independent double-and-add cross-check on the small cases
            assert bytes(O.to_affine(c, g, O.msm(c, g, bases, scalars, n, sb, naive=True))).hex() == v["affine"], key


def test_msm_closed_form():
    """sum s_i·7·11^i mod r · G   equals the Pippenger restatement."""
    c, n = O.BN128, 3000
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    sc = synth.elems(0x9999, n)
    k, f = 0, 7
    for i in range(n):
        k = (k + synth.to_int(sc[32 * i:32 * i + 32]) * f) % r
        f = f * 11 % r
    want = O.to_affine(c, 1, O.generator_mul(c, 1, k))
    got = O.to_affine(c, 1, O.msm(c, 1, O.geom_bases(c, 1, n), sc, n))
    assert np.array_equal(want, got)


def test_groth16_golden_proof(golden_dir):
    """Seeded Groth16 proof: the C restatement reproduces the reference's proof JSON hash."""
    with open(os.path.join(golden_dir, "groth16_bn128_n1024.json")) as f:
        g = json.load(f)
    zkey = open(os.path.join(golden_dir, "groth16_bn128_n1024.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, "groth16_bn128_n1024.wtns"), "rb").read()
    assert sha(zkey) == g["zkey_sha256"] == "10c89c8325ab5dd0abb8f39bd02aa32a19d18b4e288c5b33a226c91fca9983f2"
    assert sha(wtns) == g["wtns_sha256"]
    zk, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)
    assert w["nWitness"] == zk["nVars"]
    c = O.BN128
    pa, pb, pc = O.groth16_prove(c, zk, w["witness"], bytes.fromhex(g["r_mont"]), bytes.fromhex(g["s_mont"]))
    proof, js = binfile.proof_json("bn128", 32, O.fq_from_mont(c, pa), O.fq_from_mont(c, pb), O.fq_from_mont(c, pc))
    assert proof == g["proof"]
    assert sha(js.encode()) == g["proof_sha256"] == "08797809c8de2c2053a925b1772af3e04c41f8c9b4c41f5b71ae5135435da44d"
    # stage-level replay: the bytes the reference passed across each bulk-op boundary inside groth16.prove
    A, B, Cc = O.build_abc(c, zk["coeffs"], w["witness"], zk["nVars"], zk["domainSize"])
    calls = [x for x in g["calls"] if x["op"] == "Fr.ifft"]
    assert [sha(A), sha(B), sha(Cc)] == [x["in0"] for x in calls]


def test_groth16_golden_proof_bls12381(golden_dir):
    """BLS12-381 Groth16 proof generated (and verified) by the reference itself on the Multiplier(1000) r1cs written over the
    BLS12-381 scalar field (oracle/gen_golden.js groth16GoldenBls, SURVEY.md 8d recipe): pins the 381-bit limb path of the C
    restatement — the reference has no BLS12-381 test of its own."""
    with open(os.path.join(golden_dir, "groth16_bls12381_n1024.json")) as f:
        g = json.load(f)
    zkey = open(os.path.join(golden_dir, "groth16_bls12381_n1024.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, "groth16_bls12381_n1024.wtns"), "rb").read()
    assert sha(zkey) == g["zkey_sha256"] and sha(wtns) == g["wtns_sha256"] and g["verified"] is True
    zk, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)
    assert w["nWitness"] == zk["nVars"] == 1003 and zk["domainSize"] == 1024 and zk["n8q"] == 48
    c = O.BLS12381
    pa, pb, pc = O.groth16_prove(c, zk, w["witness"], bytes.fromhex(g["r_mont"]), bytes.fromhex(g["s_mont"]))
    proof, js = binfile.proof_json("bls12381", 48, O.fq_from_mont(c, pa), O.fq_from_mont(c, pb), O.fq_from_mont(c, pc))
    assert proof == g["proof"]
    assert sha(js.encode()) == g["proof_sha256"] == "955b9f3652e544aac16a90fd1ce6b8660701a7eb20e71fa7681ced22f0acd5eb"


def _verify_mod():
    import sys
    od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if od not in sys.path:
        sys.path.insert(0, od)
    import groth16_verify_oracle as V
    return V


def test_groth16_verifier_restatement_pinned(golden_dir):
    """oracle/groth16_verify_oracle.py (src/groth16_verify.js:26-87 + plain-Python optimal ate pairings on BN254 and BLS12-381) against the
    reference's own verdicts: the proofs its verifier accepted are accepted, tampered ones are rejected; bilinearity of the pairings."""
    import copy
    V = _verify_mod()
    for tag in ("groth16_bn128_n1024", "groth16_valid_synth_n64", "groth16_bls12381_n1024"):
        g = json.load(open(os.path.join(golden_dir, tag + ".json")))
        assert g["verified"] is True
        R = V.CURVES[g["vk"].get("curve", "bn128")].R
        assert V.groth16_verify(g["vk"], g["publicSignals"], g["proof"]) is True
        bad = copy.deepcopy(g["proof"])
        bad["pi_c"] = g["proof"]["pi_a"]
        assert V.groth16_verify(g["vk"], g["publicSignals"], bad) is False
        pub = list(g["publicSignals"])
        pub[-1] = str((int(pub[-1]) + 1) % R)
        assert V.groth16_verify(g["vk"], pub, g["proof"]) is False
        assert V.groth16_verify(g["vk"], [str(R)] + pub[1:], g["proof"]) is False            # public input not < r (:37-42)
        cv = V.CURVES[g["vk"].get("curve", "bn128")]                                          # bilinearity: e(7 P, Q) = e(P, Q)^7 != 1
        g1, g2 = V._g1(g["vk"]["vk_alpha_1"]), V._g2(g["vk"]["vk_beta_2"])
        e1 = cv.final_exp(cv.miller_loop(g2, cv.g1_mul(g1, 7)))
        assert e1 == cv.f12_pow(cv.final_exp(cv.miller_loop(g2, g1)), 7) and e1 != cv.F12_ONE


def test_valid_key_synthesiser_bls12381_verifies():
    """tests/synth_valid_groth16.py on BLS12-381 (no reference-accepted instance of it is committed: the synthesiser is pinned on BN254 below,
    the BLS12-381 verifier on the reference's own BLS proof above): the C restatement's proof on the synthetic key verifies, a wrong public
    input does not."""
    import synth_valid_groth16 as SV
    V = _verify_mod()
    c = O.CURVE_ID["bls12381"]
    zkey, wtns, info = SV.make("bls12381", 6, use_device=False)
    zk, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)
    pa, pb, pc = O.groth16_prove(c, zk, w["witness"], bytes(O.fr_e(c, 3)), bytes(O.fr_e(c, 5)))
    proof, _ = binfile.proof_json("bls12381", 48, O.fq_from_mont(c, pa), O.fq_from_mont(c, pb), O.fq_from_mont(c, pc))
    pub = [str(int.from_bytes(bytes(w["witness"][32 * i:32 * i + 32]), "little")) for i in range(1, 3)]
    assert V.groth16_verify(info["vk"], pub, proof) is True
    assert V.groth16_verify(info["vk"], [pub[0], str(int(pub[1]) + 1)], proof) is False


def test_valid_key_synthesiser_pinned(golden_dir):
    """tests/synth_valid_groth16.py regenerates, byte for byte, the small valid key that the REFERENCE accepted (its exported vk, its
    seeded proof and its verifier: oracle/gen_valid_fixture.py); the C restatement's proof on it equals the reference's."""
    import synth_valid_groth16 as SV
    g = json.load(open(os.path.join(golden_dir, "groth16_valid_synth_n64.json")))
    zkey, wtns, info = SV.make("bn128", 6, use_device=False)
    assert sha(zkey) == g["zkey_sha256"] == sha(open(os.path.join(golden_dir, "groth16_valid_synth_n64.zkey"), "rb").read())
    assert sha(wtns) == g["wtns_sha256"]
    assert info["vk"]["IC"] == g["vk"]["IC"] and all(info["vk"][k] == g["vk"][k] for k in ("vk_alpha_1", "vk_beta_2", "vk_gamma_2", "vk_delta_2"))
    zk, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)
    pa, pb, pc = O.groth16_prove(O.BN128, zk, w["witness"], bytes.fromhex(g["r_mont"]), bytes.fromhex(g["s_mont"]))
    proof, js = binfile.proof_json("bn128", 32, O.fq_from_mont(O.BN128, pa), O.fq_from_mont(O.BN128, pb), O.fq_from_mont(O.BN128, pc))
    assert proof == g["proof"] and sha(js.encode()) == g["proof_sha256"]


@pytest.mark.parametrize("name", ["bn128", "bls12381"])
def test_group_fft_and_apply_key_golden(golden_dir, name):
    """G.fft / G.ifft / G.lagrangeEvaluations / G.batchApplyKey of the reference (oracle/gen_golden.js groupVectors) vs the C
    restatement, incl. a point at infinity and a repeated point inside the input, sizes 1 .. 256."""
    c = O.CURVE_ID[name]
    g = json.load(open(os.path.join(golden_dir, f"{name}_group_vectors.json")))
    q8 = O.n8q(c)
    for gn, group in (("g1", 1), ("g2", 2)):
        v, pb = g[gn], 2 * group * q8
        n = v["n"]
        bases = O.geom_bases(c, group, n)
        assert sha(bases) == v["bases_sha"]
        f, fi = O.group_fft(c, group, bases), O.group_fft(c, group, bases, inverse=True)
        assert sha(f) == v["fft"] and sha(fi) == v["ifft"] == v["lagrange"]
        assert bytes(f) == open(os.path.join(golden_dir, f"{name}_gfft_{gn}_n{n}_fft.bin"), "rb").read()
        assert bytes(fi) == open(os.path.join(golden_dir, f"{name}_gfft_{gn}_n{n}_ifft.bin"), "rb").read()
        assert sha(O.group_apply_key(c, group, bases, O.fr_e(c, 3), O.fr_e(c, 5))) == v["applykey_3_5"]
        b2 = bases.copy()
        b2[5 * pb:6 * pb] = 0
        b2[9 * pb:10 * pb] = bases[:pb]
        assert sha(O.group_fft(c, group, b2)) == v["fft_with_zero_and_repeat"]
        for k in (1, 2, 4, 32):
            assert sha(O.group_fft(c, group, bases[:k * pb])) == v[f"fft_n{k}"]
            assert sha(O.group_fft(c, group, bases[:k * pb], inverse=True)) == v[f"ifft_n{k}"]


@pytest.mark.parametrize("name", ["bn128", "bls12381"])
def test_point_format_conversions_golden(golden_dir, name):
    """G.batchLEMtoU / batchUtoLEM / batchLEMtoC / batchCtoLEM of the reference (oracle/gen_golden.js convertVectors: 96 G1 /
    48 G2 points with two points at infinity inside) vs the C restatement, both directions; a compressed x off the curve is refused."""
    c = O.CURVE_ID[name]
    g = json.load(open(os.path.join(golden_dir, f"{name}_conv_vectors.json")))
    for gn, group in (("g1", 1), ("g2", 2)):
        v = g[gn]
        rd = lambda k: np.frombuffer(open(os.path.join(golden_dir, f"{name}_conv_{gn}_n{v['n']}_{k}.bin"), "rb").read(), np.uint8)
        lem, U, Cc = rd("lem"), rd("u"), rd("c")
        assert sha(lem) == v["lem"] and sha(U) == v["u"] and sha(Cc) == v["c"] and v["u_roundtrip"] and v["c_roundtrip"]
        assert bytes(O.group_convert(c, group, "LEMtoU", lem)) == bytes(U)
        assert bytes(O.group_convert(c, group, "LEMtoC", lem)) == bytes(Cc)
        assert bytes(O.group_convert(c, group, "UtoLEM", U)) == bytes(lem)
        assert bytes(O.group_convert(c, group, "CtoLEM", Cc)) == bytes(lem)
        bad, found = Cc.copy(), False
        for delta in range(1, 40):                       # about half of all x have no point: one of these must be refused
            bad[group * O.n8q(c) - 1] = (int(Cc[group * O.n8q(c) - 1]) + delta) & 0xff
            try:
                O.group_convert(c, group, "CtoLEM", bad)
            except ValueError:
                found = True
                break
        assert found
