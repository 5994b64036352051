"""oracle/fflonk_verify_oracle.py — CPU restatement of the snarkjs FFLONK verifier.        *** TEST INFRASTRUCTURE ONLY ***

Never imported by the product path: tests use it as a size-independent property check — "the device proof of the synthetic
2^16 / 2^18-constraint circuit verifies".

Restates src/fflonk_verify.js:27-134 (snarkjs 0.7.6): computeChallenges :196-319, computeLagrangeEvaluations :321-345, calculatePI
:347-356, computeR0 :358-388, computeR1 :390-426, computeR2 :428-486, computeF :488-521, computeE :523-531, computeJ :533-537,
isValidPairing :539-551, computeLagrangeLiSi / computeLagrangeLiS2 :554-597.

The final check e(-A1, [1]_2) e(W2, [tau]_2) == 1 is equivalent to A1 == tau * W2 in G1; tests/synth_plonk.make_fflonk builds its SRS
from a KNOWN toy tau, so no pairing is needed.  Parity is PINNED up to that last step: tests/test_plonk_oracle.py::
test_fflonk_verifier_trace reproduces the challenges the reference verifier logs and the two G1 points it hands to pairingEq
(tests/golden/fflonk_bn128_*.json: verify_trace, pairing_inputs, written by oracle/gen_golden.js).
"""
from plonk_oracle import Ctx, Transcript
from plonk_verify_oracle import G1, _pt


def _li_si(roots, x, xi, r):
    """computeLagrangeLiSi (:554-571)"""
    ln = len(roots)
    num = (pow(x, ln, r) - xi) % r
    den1 = ln * pow(roots[0], ln - 2, r) % r
    return [num * pow(den1 * roots[((ln - 1) * i) % ln] % r * ((x - roots[i]) % r) % r, -1, r) % r for i in range(ln)]


def _li_s2(roots, v, xi0, xi1, r):
    """computeLagrangeLiS2 (:573-597)"""
    ln = len(roots[0])
    n = ln * len(roots)
    num = (pow(v, n, r) - (xi0 + xi1) * pow(v, ln, r) + xi0 * xi1) % r
    out = []
    for k, (a, b) in enumerate(((xi0, xi1), (xi1, xi0))):
        den1 = ln * roots[k][0] % r * ((a - b) % r) % r
        for i in range(ln):
            den = den1 * (roots[k][(ln - 1) * i % ln] * ((v - roots[k][i]) % r) % r) % r
            out.append(num * pow(den, -1, r) % r)
    return out


def verifier_values(vk, public_signals, proof):
    cx = Ctx()
    r, g = cx.r, G1(cx)
    P = {k: _pt(proof["polynomials"][k]) for k in ("C1", "C2", "W1", "W2")}
    ev = {k: int(v) % r for k, v in proof["evaluations"].items()}
    C0 = _pt(vk["C0"])
    k1, k2, power = int(vk["k1"]), int(vk["k2"]), int(vk["power"])
    w3, w4, w8, wr = (int(vk[k]) for k in ("w3", "w4", "w8", "wr"))
    pub = [int(x) % r for x in public_signals]
    if len(pub) != int(vk["nPublic"]):
        raise ValueError("Number of public signals does not match with vk")
    # challenges and roots (:196-319)
    tr = Transcript(cx)
    tr.add_point(C0)
    for x in pub:
        tr.add_scalar(x)
    tr.add_point(P["C1"])
    beta = tr.challenge()
    tr.reset(); tr.add_scalar(beta)
    gamma = tr.challenge()
    tr.reset(); tr.add_scalar(gamma); tr.add_point(P["C2"])
    xi_seed = tr.challenge()
    xs2 = xi_seed * xi_seed % r
    h0 = xs2 * xi_seed % r
    S0 = [h0 * pow(w8, i, r) % r for i in range(8)]
    h1 = h0 * h0 % r
    S1 = [h1 * pow(w4, i, r) % r for i in range(4)]
    h2 = h1 * xs2 % r
    S2 = [h2 * pow(w3, i, r) % r for i in range(3)]
    h3 = h2 * wr % r
    S2p = [h3 * pow(w3, i, r) % r for i in range(3)]
    xi = h2 * h2 % r * h2 % r
    xiw = xi * cx.w[power] % r
    n = 1 << power
    xin = pow(xi, n, r)
    tr.reset(); tr.add_scalar(xi_seed)
    for k in ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3", "a", "b", "c", "z", "zw", "t1w", "t2w"):
        tr.add_scalar(ev[k])
    alpha = tr.challenge()
    tr.reset(); tr.add_scalar(alpha); tr.add_point(P["W1"])
    y = tr.challenge()
    zh = (xin - 1) % r
    invzh = pow(zh, -1, r)
    # Lagrange evaluations and PI (:321-356)
    L, w = [0], 1
    for i in range(max(1, len(pub))):
        L.append(w * zh % r * pow(n * (xi - w) % r, -1, r) % r)
        w = w * cx.w[power] % r
    pi = 0
    for i, x in enumerate(pub):
        pi = (pi - x * L[i + 1]) % r
    # r0, r1, r2 (:358-486)
    li0 = _li_si(S0, y, xi, r)
    r0 = 0
    for i in range(8):
        h = S0[i]
        c0 = sum(ev[k] * pow(h, j, r) for j, k in enumerate(("ql", "qr", "qo", "qm", "qc", "s1", "s2", "s3"))) % r
        r0 = (r0 + c0 * li0[i]) % r
    t0 = (ev["ql"] * ev["a"] + ev["qr"] * ev["b"] + ev["qm"] * ev["a"] % r * ev["b"] + ev["qo"] * ev["c"] + ev["qc"] + pi) % r * invzh % r
    li1 = _li_si(S1, y, xi, r)
    r1 = 0
    for i in range(4):
        h = S1[i]
        c1 = (ev["a"] + h * ev["b"] + h * h % r * ev["c"] + h * h % r * h % r * t0) % r
        r1 = (r1 + c1 * li1[i]) % r
    t1 = (ev["z"] - 1) * L[1] % r * invzh % r
    betaxi = beta * xi % r
    t21 = (ev["a"] + betaxi + gamma) * (ev["b"] + betaxi * k1 + gamma) % r * (ev["c"] + betaxi * k2 + gamma) % r * ev["z"] % r
    t22 = (ev["a"] + beta * ev["s1"] + gamma) * (ev["b"] + beta * ev["s2"] + gamma) % r * (ev["c"] + beta * ev["s3"] + gamma) % r * ev["zw"] % r
    t2 = (t21 - t22) * invzh % r
    li2 = _li_s2([S2, S2p], y, xi, xiw, r)
    r2 = 0
    for i in range(3):
        c2 = (ev["z"] + S2[i] * t1 + S2[i] * S2[i] % r * t2) % r
        r2 = (r2 + c2 * li2[i]) % r
    for i in range(3):
        c2 = (ev["zw"] + S2p[i] * ev["t1w"] + S2p[i] * S2p[i] % r * ev["t2w"]) % r
        r2 = (r2 + c2 * li2[i + 3]) % r
    # F, E, J and the pairing arguments (:488-551)
    mul = lambda roots: __import__("functools").reduce(lambda a, x: a * ((y - x) % r) % r, roots, 1)
    mulH0, mulH1, mulH2 = mul(S0), mul(S1), mul(S2 + S2p)
    q1 = alpha * mulH0 % r * pow(mulH1, -1, r) % r
    q2 = alpha * alpha % r * mulH0 % r * pow(mulH2, -1, r) % r
    e = (r0 + r1 * q1 + r2 * q2) % r
    # A1 = F - E - J + y W2,  F = C0 + q1 C1 + q2 C2,  E = e G,  J = mulH0 W1
    A1 = g.lincomb([(1, C0), (q1, P["C1"]), (q2, P["C2"]), (-e, g.generator()), (-mulH0, P["W1"]), (y, P["W2"])])
    return dict(beta=beta, gamma=gamma, xi=xi, alpha=alpha, y=y, r0=r0, r1=r1, r2=r2, A1=A1, B1=P["W2"])


def verify_known_tau(vk, public_signals, proof, tau):
    """fflonk.verify with e(-A1, [1]_2) e(W2, [tau]_2) == 1 evaluated as A1 == tau * W2 (valid only for an SRS [tau^i] G)"""
    cx = Ctx()
    val = verifier_values(vk, public_signals, proof)
    return val["A1"] == G1(cx).lincomb([(tau, val["B1"])])


def vk_from_zkey(zkey_bytes):
    """zKey.exportVerificationKey for an FFLONK key (src/zkey_export_verificationkey.js): the fields the verifier reads"""
    from fflonk_oracle import read_fflonk_zkey
    cx = Ctx()
    zk = read_fflonk_zkey(zkey_bytes, cx)
    vk = {"nPublic": zk["nPublic"], "power": zk["power"]}
    for k in ("k1", "k2", "w3", "w4", "w8", "wr"):
        vk[k] = str(zk[k])
    vk["C0"] = [str(zk["C0"][0]), str(zk["C0"][1]), "1"]
    return vk
