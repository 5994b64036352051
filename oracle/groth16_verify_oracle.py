"""oracle/groth16_verify_oracle.py — TEST INFRASTRUCTURE ONLY: restatement of the reference's Groth16 verifier for BN254.

Follows src/groth16_verify.js:26-87: public inputs must be < r (:37-42 isWellConstructed / publicInputsAreValid), the proof points
must be on the curve, vk_x = IC[0] + sum_i pub_i * IC[i+1] (:55-61), and the pairing product
    e(-pi_a, pi_b) * e(vk_x, gamma_2) * e(pi_c, delta_2) * e(alpha_1, beta_2) == 1                      (:66-74 curve.pairingEq)
The pairing itself lives in ffjavascript / wasmcurves (absent from /root/reference as source): it is restated here as the optimal ate
pairing on BN254 in plain Python integers — G2 arithmetic on the sextic twist over Fp2, line functions embedded sparsely into
Fp12 = Fp[w]/(w^12 - 18 w^6 + 82) (u = w^6 - 9), Miller loop over 6x+2 with the two Frobenius end steps, final exponentiation by
(p^12 - 1)/r. PINNED (tests/test_oracle_golden.py): accepts the proofs the reference's own verifier accepted (tests/golden/
groth16_bn128_n1024.json, verified: true; groth16_valid_synth_*.json) and rejects tampered ones; bilinearity self-checks.
Pure Python: one verification (4 Miller loops + 1 final exponentiation) takes a few seconds.
"""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP = 29793968203157093288          # 6x + 2, x = 4965661367192848881


# ---- Fp2 = Fp[u]/(u^2 + 1): tuples (c0, c1) ------------------------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_scal(a, k): return (a[0] * k % P, a[1] * k % P)
def f2_conj(a): return (a[0], (-a[1]) % P)


def f2_inv(a):
    d = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


XI = (9, 1)                                   # w^6 = 9 + u
TWIST_B = f2_mul((3, 0), f2_inv(XI))          # E': y^2 = x^3 + 3/(9+u)
G11 = f2_pow(XI, (P - 1) // 3)                # Frobenius on the twist: x -> conj(x) * xi^((p-1)/3), y -> conj(y) * xi^((p-1)/2)
G12 = f2_pow(XI, (P - 1) // 2)
G21 = f2_pow(XI, (P * P - 1) // 3)            # p^2-Frobenius: factors lie in Fp
G22 = f2_pow(XI, (P * P - 1) // 2)


# ---- Fp12 = Fp[w]/(w^12 - 18 w^6 + 82): lists of 12 coefficients ------------------------------------------------------------
def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                if y:
                    t[i + j] += x * y
    for e in range(22, 11, -1):               # w^12 = 18 w^6 - 82
        top = t[e]
        if top:
            t[e - 6] += 18 * top
            t[e - 12] -= 82 * top
    return [x % P for x in t[:12]]


F12_ONE = [1] + [0] * 11


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_mul(a, a)
        e >>= 1
    return r


def _line(lam, xt, yt, px, py):
    """l_{T,.}(P) = yP - lam*w*xP + (lam*xT - yT)*w^3 with lam, xT, yT in Fp2 on the twist (untwist: x*w^2, y*w^3, slope*w)"""
    a = f2_scal(lam, (-px) % P)                # coefficient of w
    b = f2_sub(f2_mul(lam, xt), yt)            # coefficient of w^3
    c = [0] * 12
    c[0] = py % P
    c[1] = (a[0] - 9 * a[1]) % P; c[7] = a[1]
    c[3] = (b[0] - 9 * b[1]) % P; c[9] = b[1]
    return c


def _g2_double_step(T, px, py):
    x, y = T
    lam = f2_mul(f2_scal(f2_mul(x, x), 3), f2_inv(f2_scal(y, 2)))
    x3 = f2_sub(f2_mul(lam, lam), f2_scal(x, 2))
    y3 = f2_sub(f2_mul(lam, f2_sub(x, x3)), y)
    return (x3, y3), _line(lam, x, y, px, py)


def _g2_add_step(T, Q, px, py):
    (x1, y1), (x2, y2) = T, Q
    lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    return (x3, y3), _line(lam, x1, y1, px, py)


def miller_loop(Q, Pt):
    """Q = ((x0, x1), (y0, y1)) on the twist (G2, affine, not infinity), Pt = (x, y) in G1 (affine, not infinity)"""
    px, py = Pt
    T, f = Q, F12_ONE
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        T, l = _g2_double_step(T, px, py)
        f = f12_mul(f12_mul(f, f), l)
        if (ATE_LOOP >> i) & 1:
            T, l = _g2_add_step(T, Q, px, py)
            f = f12_mul(f, l)
    Q1 = (f2_mul(f2_conj(Q[0]), G11), f2_mul(f2_conj(Q[1]), G12))
    nQ2 = (f2_mul(Q[0], G21), f2_neg(f2_mul(Q[1], G22)))
    T, l = _g2_add_step(T, Q1, px, py)
    f = f12_mul(f, l)
    T, l = _g2_add_step(T, nQ2, px, py)
    return f12_mul(f, l)


def final_exp(f):
    return f12_pow(f, (P ** 12 - 1) // R)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for pairs of (G1 affine | None, G2 affine | None); a pair with a point at infinity contributes 1"""
    f = F12_ONE
    for g1, g2 in pairs:
        if g1 is None or g2 is None:
            continue
        f = f12_mul(f, miller_loop(g2, g1))
    return final_exp(f) == F12_ONE


# ---- G1 (affine over Fp, None = infinity) -----------------------------------------------------------------------------------
def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return (x, (lam * (a[0] - x) - a[1]) % P)


def g1_mul(a, k):
    r = None
    while k:
        if k & 1:
            r = g1_add(r, a)
        a = g1_add(a, a)
        k >>= 1
    return r


def g1_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def g1_on_curve(a):
    return a is None or (a[1] * a[1] - a[0] * a[0] * a[0] - 3) % P == 0


def g2_on_curve(q):
    if q is None:
        return True
    x, y = q
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), TWIST_B)) == (0, 0)


def _g1(obj):
    x, y, z = (int(v) for v in obj[:3])
    return None if z == 0 else (x, y)


def _g2(obj):
    (x0, x1), (y0, y1), (z0, z1) = ((int(v[0]), int(v[1])) for v in obj[:3])
    return None if (z0, z1) == (0, 0) else ((x0, x1), (y0, y1))


def groth16_verify(vk, public_signals, proof):
    """src/groth16_verify.js:26-87 on the JSON objects snarkjs itself uses (vk from zKey.exportVerificationKey, proof / publicSignals
    from groth16.prove: decimal strings, affine points with a trailing "1")."""
    if vk.get("curve", "bn128") != "bn128":
        raise ValueError("groth16_verify_oracle restates the BN254 pairing only")
    pubs = [int(s) for s in public_signals]
    if len(pubs) + 1 != len(vk["IC"]) or any(not (0 <= v < R) for v in pubs):                         # :37-42
        return False
    pi_a, pi_b, pi_c = _g1(proof["pi_a"]), _g2(proof["pi_b"]), _g1(proof["pi_c"])
    if not (g1_on_curve(pi_a) and g2_on_curve(pi_b) and g1_on_curve(pi_c)):
        return False
    ic = [_g1(x) for x in vk["IC"]]
    cpub = ic[0]                                                                                       # :55-61
    for v, pt in zip(pubs, ic[1:]):
        cpub = g1_add(cpub, g1_mul(pt, v))
    return pairing_product_is_one([(g1_neg(pi_a), pi_b), (cpub, _g2(vk["vk_gamma_2"])), (pi_c, _g2(vk["vk_delta_2"])),
                                   (_g1(vk["vk_alpha_1"]), _g2(vk["vk_beta_2"]))])                     # :66-74
