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


@pytest.mark.parametrize("tag", ["plonk_bn128_small", "plonk_bn128_n2048", "plonk_bls12381_small"])
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


def test_prove_many_schedules_two_coroutines_over_the_pipeline_slots(monkeypatch):
    """plonk.prove_many (host logic, no GPU): proofs are coroutines that yield before their blocking calls; the driver keeps at most two alive,
    selects the proof's pipeline slot before every step, returns results in input order, and on an error closes the other proof and goes back to
    slot 0."""
    from snarkjs_amd import plonk, zkmi

    class FakeLib:
        def __init__(self):
            self.active, self.log = 0, []

        def zkmi_pipeline_select(self, slot):
            self.active = slot
            return 0

        def zkmi_synchronize(self):
            self.log.append(("sync", self.active))
            return 0

    fake = FakeLib()
    monkeypatch.setattr(zkmi, "lib", lambda: fake)
    alive, peak, trace = set(), [0], []

    def steps(key, wt, logger, options, blind):
        alive.add(wt["id"])
        peak[0] = max(peak[0], len(alive))
        try:
            for k in range(wt["n"]):
                trace.append((wt["id"], fake.active))
                if wt.get("fail_at") == k:
                    raise ValueError("Copy constraints does not match")
                yield
            return {"proof": wt["id"], "blind": blind}
        finally:
            alive.discard(wt["id"])

    monkeypatch.setattr(plonk, "_prove_steps", steps)
    key = object.__new__(plonk.PlonkKey)
    wts = [{"id": i, "n": n} for i, n in enumerate([3, 7, 2, 5, 4])]
    out = plonk.prove_many(key, wts, blinding_monts=[10, 11, 12, 13, 14])
    assert [o["proof"] for o in out] == [0, 1, 2, 3, 4] and [o["blind"] for o in out] == [10, 11, 12, 13, 14]
    assert peak[0] == 2 and not alive and fake.active == 0
    slot_of = {}
    for pid, slot in trace:                                # every step of a proof runs on the slot it started on
        assert slot_of.setdefault(pid, slot) == slot
    assert sorted(set(slot_of.values())) == [0, 1]
    # the two live proofs alternate step by step
    first_two = [pid for pid, _ in trace if pid in (0, 1)][:6]
    assert first_two == [0, 1, 0, 1, 0, 1]
    # one proof fails: the other is closed, the slot goes back to 0, later calls work
    trace.clear()
    with pytest.raises(ValueError, match="Copy constraints"):
        plonk.prove_many(key, [{"id": 0, "n": 6}, {"id": 1, "n": 6, "fail_at": 2}, {"id": 2, "n": 3}])
    assert not alive and fake.active == 0 and 2 not in [pid for pid, _ in trace]
    assert [o["proof"] for o in plonk.prove_many(key, [{"id": 7, "n": 1}], in_flight=1)] == [7]


def test_mul4_karatsuba_form_equals_the_reference_expansion():
    """csrc/plonk.hip MulZ::mul4 computes (a + a'Z)(b + b'Z)(c + c'Z)(d + d'Z) as the product of two quadratics with Karatsuba on the pairs and on
    the quadratics (15 field multiplications); the oracle keeps the reference's literal expansion (mul_z.js:103-148, 27 multiplications). The two
    are the same field elements for every input — checked here on random values and on the degenerate ones, for the four residues of the point
    index, both scalar fields. (The kernel itself is held to the oracle on the GPU by test_compute_z_and_t_stages_vs_oracle and the golden proofs.)"""
    import random
    for r in (P.Ctx().r, P.Ctx(48).r if hasattr(P.Ctx(48), "r") else P.Ctx().r):
        rng = random.Random(r & 0xFFFF)
        Z = [[rng.randrange(r) for _ in range(4)] for _ in range(3)]

        def kernel_form(a, b, c, d, ap, bp, cp, dp, p):
            A0, B0 = a * b % r, c * d % r
            rr = A0 * B0 % r
            if not p:
                u, v = (a * bp + ap * b) % r, (c * dp + cp * d) % r
                return rr, (u * B0 + A0 * v) % r
            A2, B2 = ap * bp % r, cp * dp % r
            A1 = ((a + ap) * (b + bp) - A0 - A2) % r
            B1 = ((c + cp) * (d + dp) - B0 - B2) % r
            P1, P2 = A1 * B1 % r, A2 * B2 % r
            P01, P02, P12 = (A0 + A1) * (B0 + B1) % r, (A0 + A2) * (B0 + B2) % r, (A1 + A2) * (B1 + B2) % r
            z1, z2, z3 = (P01 - rr - P1) % r, (P02 - rr - P2 + P1) % r, (P12 - P1 - P2) % r
            return rr, (z1 + Z[0][p] * z2 + Z[1][p] * z3 + Z[2][p] * P2) % r

        cases = [[rng.randrange(r) for _ in range(8)] for _ in range(200)]
        cases += [[0] * 8, [r - 1] * 8, [1, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, r - 1, 1, r - 1, 1], [rng.randrange(r)] * 8]
        for vals in cases:
            for p in range(4):
                assert kernel_form(*vals, p) == P.mul4(*vals, p, Z, r)
