"""oracle/plonk_verify_oracle.py — CPU restatement of the snarkjs PLONK verifier.        *** TEST INFRASTRUCTURE ONLY ***

Never imported by the product path (verification is out of scope): tests use it as a size-independent
property check of full-size proofs — "the proof of the 2^20-constraint synthetic circuit verifies".

Restates src/plonk_verify.js:28-421 (snarkjs 0.7.6): calculatechallenges :207-271, calculateLagrangeEvaluations :273-296,
calculatePI :298-307, calculateR0 :309-331, calculateD :333-375, calculateF :377-387, calculateE :389-403, isValidPairing :405-421.

The final pairing e(-A1, [tau]_2) * e(B1, [1]_2) == 1 is equivalent to  B1 == tau * A1  in G1.  The synthetic keys of
tests/synth_plonk.py are built from a KNOWN toy tau, so the check is done in G1 with the C oracle's group arithmetic and needs no
pairing.  Parity is PINNED up to that last step: tests/test_plonk_oracle.py::test_plonk_verifier_trace reproduces every value the
reference verifier logs (beta, gamma, alpha, xi, v1..v5, u, L_i(xi), PI(xi), r0, D, F, E) on the reference's own seeded proofs
(tests/golden/plonk_bn128_*.json: verify_trace, written by oracle/gen_golden.js).
"""
import numpy as np

import oracle_lib as O
from plonk_oracle import Ctx, Transcript


class G1:
    """affine points as (x, y) ints in normal form, None = infinity; arithmetic through the C oracle's MSM (orc_msm)"""

    def __init__(self, cx):
        self.cx = cx

    def enc(self, p):
        n8 = self.cx.n8
        if p is None:
            return bytes(2 * n8)
        q, Rq = self.cx.q, pow(2, 8 * n8, self.cx.q)
        return (p[0] * Rq % q).to_bytes(n8, "little") + (p[1] * Rq % q).to_bytes(n8, "little")

    def lincomb(self, terms):
        """sum k_i * P_i for [(k_i, P_i)]"""
        r = self.cx.r
        bases = np.frombuffer(b"".join(self.enc(p) for _, p in terms), np.uint8)
        sc = np.frombuffer(b"".join((k % r).to_bytes(32, "little") for k, _ in terms), np.uint8)
        cid, n8 = self.cx.cid, self.cx.n8
        aff = O.to_affine(cid, 1, O.msm(cid, 1, bases, sc, len(terms)))
        if not aff.any():
            return None
        return (int.from_bytes(bytes(aff[:n8]), "little") * self.cx.Rqi % self.cx.q, int.from_bytes(bytes(aff[n8:]), "little") * self.cx.Rqi % self.cx.q)

    def generator(self):
        """G1.g of the curve (ffjavascript: bn128 (1, 2); bls12381 the standard generator)"""
        if self.cx.n8 == 32:
            return (1, 2)
        return (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
                0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)


def _pt(obj):
    return None if (int(obj[0]), int(obj[1])) == (0, 1) and int(obj[2]) == 0 else (int(obj[0]), int(obj[1]))


def _cx_of(vk):
    """the curve context a verification key belongs to (the reference: getCurveFromName(vk.curve), src/plonk_verify.js:38)"""
    return Ctx(48 if str(vk.get("curve", "bn128")) == "bls12381" else 32)


def verifier_values(vk, public_signals, proof):
    """Everything the verifier computes before the pairing: dict with the challenges, L, pi, r0 and the points D, F, E, A1, B1."""
    cx = _cx_of(vk)
    r, g = cx.r, G1(cx)
    P = {k: _pt(proof[k]) for k in ("A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw")}
    ev = {k: int(proof["eval_" + k]) % r for k in ("a", "b", "c", "s1", "s2", "zw")}
    V = {k: _pt(vk[k]) for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3")}
    k1, k2, power = int(vk["k1"]), int(vk["k2"]), int(vk["power"])
    pub = [int(x) % r for x in public_signals]
    if len(pub) != int(vk["nPublic"]):
        raise ValueError("Invalid number of public inputs")
    # challenges (:207-271)
    tr = Transcript(cx)
    add_point = tr.add_point
    tr.add_point = lambda p: add_point(p or (0, 0))          # G1.toRprUncompressed of the point at infinity = 64 zero bytes (probe)
    for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        tr.add_point(V[k])
    for x in pub:
        tr.add_scalar(x)
    for k in ("A", "B", "C"):
        tr.add_point(P[k])
    beta = tr.challenge()
    tr.reset(); tr.add_scalar(beta)
    gamma = tr.challenge()
    tr.reset(); tr.add_scalar(beta); tr.add_scalar(gamma); tr.add_point(P["Z"])
    alpha = tr.challenge()
    tr.reset(); tr.add_scalar(alpha)
    for k in ("T1", "T2", "T3"):
        tr.add_point(P[k])
    xi = tr.challenge()
    tr.reset(); tr.add_scalar(xi)
    for k in ("a", "b", "c", "s1", "s2", "zw"):
        tr.add_scalar(ev[k])
    v = [0, tr.challenge()]
    for i in range(2, 6):
        v.append(v[i - 1] * v[1] % r)
    tr.reset(); tr.add_point(P["Wxi"]); tr.add_point(P["Wxiw"])
    u = tr.challenge()
    # Lagrange evaluations, PI, r0 (:273-331)
    n = 1 << power
    xin = pow(xi, n, r)
    zh = (xin - 1) % r
    L, w = [0], 1
    for i in range(1, max(1, len(pub)) + 1):
        L.append(w * zh % r * pow(n * (xi - w) % r, -1, r) % r)
        w = w * cx.w[power] % r
    pi = 0
    for i, x in enumerate(pub):
        pi = (pi - x * L[i + 1]) % r
    e3 = (ev["a"] + beta * ev["s1"] + gamma) * (ev["b"] + beta * ev["s2"] + gamma) % r * (ev["c"] + gamma) % r * ev["zw"] % r * alpha % r
    r0 = (pi - L[1] * alpha * alpha - e3) % r
    # D, F, E (:333-403)
    betaxi = beta * xi % r
    d2a = (ev["a"] + betaxi + gamma) * (ev["b"] + betaxi * k1 + gamma) % r * (ev["c"] + betaxi * k2 + gamma) % r * alpha % r
    d2 = (d2a + L[1] * alpha * alpha + u) % r
    d3 = (ev["a"] + beta * ev["s1"] + gamma) * (ev["b"] + beta * ev["s2"] + gamma) % r * (alpha * beta % r * ev["zw"] % r) % r
    D = g.lincomb([(ev["a"] * ev["b"], V["Qm"]), (ev["a"], V["Ql"]), (ev["b"], V["Qr"]), (ev["c"], V["Qo"]), (1, V["Qc"]), (d2, P["Z"]), (-d3, V["S3"]),
                   (-zh, P["T1"]), (-zh * xin, P["T2"]), (-zh * xin % r * xin, P["T3"])])
    F = g.lincomb([(1, D), (v[1], P["A"]), (v[2], P["B"]), (v[3], P["C"]), (v[4], V["S1"]), (v[5], V["S2"])])
    e = (-r0 + v[1] * ev["a"] + v[2] * ev["b"] + v[3] * ev["c"] + v[4] * ev["s1"] + v[5] * ev["s2"] + u * ev["zw"]) % r
    E = g.lincomb([(e, g.generator())])
    # isValidPairing (:405-421), the two G1 arguments
    A1 = g.lincomb([(1, P["Wxi"]), (u, P["Wxiw"])])
    B1 = g.lincomb([(xi, P["Wxi"]), (u * xi % r * cx.w[power], P["Wxiw"]), (1, F), (-1, E)])
    return dict(beta=beta, gamma=gamma, alpha=alpha, xi=xi, v=v, u=u, L=L, pi=pi, r0=r0, D=D, F=F, E=E, A1=A1, B1=B1)


def verify_known_tau(vk, public_signals, proof, tau):
    """plonk.verify with e(-A1, [tau]_2) e(B1, [1]_2) == 1 evaluated as B1 == tau * A1 (valid only for a key whose SRS is [tau^i] G)"""
    cx = _cx_of(vk)
    val = verifier_values(vk, public_signals, proof)
    return val["B1"] == G1(cx).lincomb([(tau, val["A1"])])


def vk_from_zkey(zkey_bytes):
    """zKey.exportVerificationKey for a PLONK key (src/zkey_export_verificationkey.js:66-104): the fields the verifier reads"""
    import struct
    from plonk_oracle import read_plonk_zkey, read_sections
    n8q = struct.unpack_from("<I", zkey_bytes, read_sections(zkey_bytes)[2][0])[0]        # the curve comes from the key's prime (src/curves.js:36-53)
    cx = Ctx(n8q)
    zk = read_plonk_zkey(zkey_bytes, cx)
    vk = {"nPublic": zk["nPublic"], "power": zk["power"], "k1": str(zk["k1"]), "k2": str(zk["k2"]), "curve": cx.name}
    for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        vk[k] = ["0", "1", "0"] if zk[k] == (0, 0) else [str(zk[k][0]), str(zk[k][1]), "1"]     # G1.toObject of the point at infinity
    return vk
