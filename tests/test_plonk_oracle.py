"""Pins the PLONK restatement (oracle/plonk_oracle.py) to the seeded proofs produced by the REAL reference
(oracle/gen_golden.js plonk -> tests/golden/plonk_bn128_*.{zkey,wtns,json}).  CPU only."""
import hashlib
import json
import os

import pytest

import plonk_oracle as P


def load(golden_dir, tag):
    with open(os.path.join(golden_dir, f"{tag}.json")) as f:
        g = json.load(f)
    zkey = open(os.path.join(golden_dir, f"{tag}.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, f"{tag}.wtns"), "rb").read()
    assert hashlib.sha256(zkey).hexdigest() == g["zkey_sha256"] and hashlib.sha256(wtns).hexdigest() == g["wtns_sha256"]
    return g, zkey, wtns


def test_keccak256_known_answers():
    assert P.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert P.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # rate-boundary paddings (135, 136, 137 bytes) are self-consistent with the multi-block path
    assert len({P.keccak256(b"x" * k) for k in (135, 136, 137)}) == 3


@pytest.mark.parametrize("tag", ["plonk_bn128_small", "plonk_bn128_n2048", "plonk_bls12381_small"])
def test_plonk_golden_proof(golden_dir, tag):
    g, zkey, wtns = load(golden_dir, tag)
    proof, public = P.plonk_prove(zkey, wtns, [bytes.fromhex(x) for x in g["blinding_mont"]])
    assert public == g["publicSignals"]
    assert proof == g["proof"]
    js = json.dumps(proof, separators=(",", ":"))
    assert hashlib.sha256(js.encode()).hexdigest() == g["proof_sha256"]


@pytest.mark.parametrize("tag", ["fflonk_bn128_small", "fflonk_bn128_n256"])
def test_fflonk_golden_proof(golden_dir, tag):
    """The FFLONK restatement (oracle/fflonk_oracle.py) reproduces the reference's seeded proofs."""
    import fflonk_oracle as FF
    g, zkey, wtns = load(golden_dir, tag)
    proof, public = FF.fflonk_prove(zkey, wtns, [bytes.fromhex(x) for x in g["blinding_mont"]])
    assert public == g["publicSignals"]
    assert proof == g["proof"]
    assert hashlib.sha256(json.dumps(proof, separators=(",", ":")).encode()).hexdigest() == g["proof_sha256"]


@pytest.mark.parametrize("tag", ["plonk_bn128_small", "plonk_bn128_n2048"])
def test_plonk_verifier_trace(golden_dir, tag):
    """The verifier restatement (oracle/plonk_verify_oracle.py) reproduces every intermediate value the reference's plonk.verify
    logs for its own seeded proofs (challenges, L_i(xi), PI(xi), r0 and the points D, F, E)."""
    import re
    import plonk_verify_oracle as V
    g, zkey, wtns = load(golden_dir, tag)
    val = V.verifier_values(g["vk"], g["publicSignals"], g["proof"])
    want = {}
    for line in g["verify_trace"]:
        m = re.match(r"(\w+)(?:\(xi\))?[:=] ?(.*)", line)
        want.setdefault(m.group(1), []).append(m.group(2))
    sc = lambda s: int(s, 16)
    pt = lambda s: tuple(int(x, 16) for x in re.findall(r"[0-9a-f]+", s))
    for k in ("beta", "gamma", "alpha", "xi", "u", "r0"):
        assert val[k] == sc(want[k][0]), k
    assert val["v"][1:] == [sc(x) for x in want["v"]]
    assert val["L"][1:] == [sc(want[f"L{i}"][0]) for i in range(1, len(val["L"]))]
    assert val["pi"] == sc(want["PI"][0])
    for k in ("D", "F", "E"):
        assert val[k] == pt(want[k][0]), k
    # the verification key the reference exported == the one read from the zkey header
    vk = V.vk_from_zkey(zkey)
    for k, v in vk.items():
        assert g["vk"][k] == v or str(g["vk"][k]) == str(v), k


@pytest.mark.parametrize("tag", ["fflonk_bn128_small", "fflonk_bn128_n256"])
def test_fflonk_verifier_trace(golden_dir, tag):
    """The FFLONK verifier restatement reproduces the challenges the reference's fflonk.verify logs and the two G1 points it hands
    to the final pairing (-A1 and W2) for the reference's own seeded proofs."""
    import re
    import fflonk_verify_oracle as V
    g, zkey, wtns = load(golden_dir, tag)
    val = V.verifier_values(g["vk"], g["publicSignals"], g["proof"])
    for line in g["verify_trace"]:
        m = re.search(r"challenges\.(\w+):\s+(\d+)", line)
        assert val[m.group(1)] == int(m.group(2)), m.group(1)
    neg_a1, b1 = (tuple(int(x) for x in p) for p in g["pairing_inputs"])
    assert val["A1"] == (neg_a1[0], (V.Ctx().q - neg_a1[1]) % V.Ctx().q)
    assert val["B1"] == b1
    vk = V.vk_from_zkey(zkey)
    for k, v in vk.items():
        assert str(g["vk"][k]) == str(v) or g["vk"][k] == v, k
