"""oracle/groth16_verify_oracle.py — TEST INFRASTRUCTURE ONLY: restatement of the reference's Groth16 verifier for BN254 and BLS12-381.

Follows src/groth16_verify.js:26-87: public inputs must be < r (:37-42 isWellConstructed / publicInputsAreValid), the proof points
must be on the curve, vk_x = IC[0] + sum_i pub_i * IC[i+1] (:55-61), and the pairing product
    e(-pi_a, pi_b) * e(vk_x, gamma_2) * e(pi_c, delta_2) * e(alpha_1, beta_2) == 1                      (:66-74 curve.pairingEq)
The pairing itself lives in ffjavascript / wasmcurves (absent from /root/reference as source): it is restated here as the optimal ate
pairing in plain Python integers — G2 arithmetic on the sextic twist over Fp2, line functions embedded sparsely into
Fp12 = Fp[w]/(w^12 - c6 w^6 - c0), Miller loop, final exponentiation by (p^12 - 1)/r:
    BN254       D-type twist y^2 = x^3 + 3/(9+u),  w^6 = 9 + u  (w^12 = 18 w^6 - 82),  loop 6x+2 with the two Frobenius end steps
    BLS12-381   M-type twist y^2 = x^3 + 4(1+u),   w^6 = 1 + u  (w^12 = 2 w^6 - 2),    loop |x| = 0xd201000000010000; x is negative,
                which inverts every pairing value alike — irrelevant for a product compared with 1, so the conjugation is left out
PINNED (tests/test_oracle_golden.py): accepts the proofs the reference's own verifier accepted on both curves (tests/golden/
groth16_bn128_n1024.json, groth16_bls12381_n1024.json, groth16_valid_synth_*.json: verified: true) and rejects tampered ones;
bilinearity self-checks. Pure Python: one verification (4 Miller loops + 1 final exponentiation) takes a few seconds.
"""


class Pairing:
    def __init__(self, name, p, r, loop, xi, twist, b, bn_end_steps):
        self.name, self.P, self.R, self.loop, self.XI, self.twist, self.bn_end_steps = name, p, r, loop, xi, twist, bn_end_steps
        P = p
        # w^6 = xi = s + u  =>  u = w^6 - s,  u^2 = -1  =>  w^12 = 2 s w^6 - (s^2 + 1)
        s = xi[0]
        assert xi[1] == 1
        self.S, self.C6, self.C0 = s, 2 * s % P, (-(s * s + 1)) % P
        self.B1 = b % P
        self.TWIST_B = self.f2_mul((b, 0), self.f2_inv(xi)) if twist == "D" else self.f2_mul((b, 0), xi)
        self.F12_ONE = [1] + [0] * 11
        if bn_end_steps:        # Frobenius on the twist: x -> conj(x) * xi^((p-1)/3), y -> conj(y) * xi^((p-1)/2)
            self.G11, self.G12 = self.f2_pow(xi, (P - 1) // 3), self.f2_pow(xi, (P - 1) // 2)
            self.G21, self.G22 = self.f2_pow(xi, (P * P - 1) // 3), self.f2_pow(xi, (P * P - 1) // 2)      # p^2-Frobenius: factors lie in Fp

    # ---- Fp2 = Fp[u]/(u^2 + 1): tuples (c0, c1) ----------------------------------------------------------------------
    def f2_add(self, a, b): return ((a[0] + b[0]) % self.P, (a[1] + b[1]) % self.P)
    def f2_sub(self, a, b): return ((a[0] - b[0]) % self.P, (a[1] - b[1]) % self.P)
    def f2_neg(self, a): return ((-a[0]) % self.P, (-a[1]) % self.P)
    def f2_mul(self, a, b): return ((a[0] * b[0] - a[1] * b[1]) % self.P, (a[0] * b[1] + a[1] * b[0]) % self.P)
    def f2_scal(self, a, k): return (a[0] * k % self.P, a[1] * k % self.P)
    def f2_conj(self, a): return (a[0], (-a[1]) % self.P)

    def f2_inv(self, a):
        d = pow((a[0] * a[0] + a[1] * a[1]) % self.P, -1, self.P)
        return (a[0] * d % self.P, (-a[1]) * d % self.P)

    def f2_pow(self, a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = self.f2_mul(r, a)
            a = self.f2_mul(a, a)
            e >>= 1
        return r

    # ---- Fp12 = Fp[w]/(w^12 - C6 w^6 - C0): lists of 12 coefficients --------------------------------------------------
    def f12_mul(self, a, b):
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    if y:
                        t[i + j] += x * y
        for e in range(22, 11, -1):
            top = t[e]
            if top:
                t[e - 6] += self.C6 * top
                t[e - 12] += self.C0 * top
        return [x % self.P for x in t[:12]]

    def f12_pow(self, a, e):
        r = self.F12_ONE
        while e:
            if e & 1:
                r = self.f12_mul(r, a)
            a = self.f12_mul(a, a)
            e >>= 1
        return r

    def _embed(self, c, k, a):
        """c += a * w^k for a in Fp2: a0 + a1 u = (a0 - S a1) + a1 w^6"""
        c[k] = (c[k] + a[0] - self.S * a[1]) % self.P
        c[k + 6] = (c[k + 6] + a[1]) % self.P

    def _line(self, lam, xt, yt, px, py):
        """The line through T with slope lam (both on the twist), evaluated at P = (px, py) in G1, up to a factor of a proper subfield
        (removed by the final exponentiation).
          D-type (untwist x w^2, y w^3, slope lam w):    yP - lam xP w + (lam xT - yT) w^3
          M-type (untwist x / w^2, y / w^3, slope lam / w), times w^3:    yP w^3 - lam xP w^2 + (lam xT - yT)"""
        a = self.f2_scal(lam, (-px) % self.P)
        b = self.f2_sub(self.f2_mul(lam, xt), yt)
        c = [0] * 12
        if self.twist == "D":
            c[0] = py % self.P
            self._embed(c, 1, a)
            self._embed(c, 3, b)
        else:
            c[3] = py % self.P
            self._embed(c, 2, a)
            self._embed(c, 0, b)
        return c

    def _g2_double_step(self, T, px, py):
        x, y = T
        lam = self.f2_mul(self.f2_scal(self.f2_mul(x, x), 3), self.f2_inv(self.f2_scal(y, 2)))
        x3 = self.f2_sub(self.f2_mul(lam, lam), self.f2_scal(x, 2))
        y3 = self.f2_sub(self.f2_mul(lam, self.f2_sub(x, x3)), y)
        return (x3, y3), self._line(lam, x, y, px, py)

    def _g2_add_step(self, T, Q, px, py):
        (x1, y1), (x2, y2) = T, Q
        lam = self.f2_mul(self.f2_sub(y2, y1), self.f2_inv(self.f2_sub(x2, x1)))
        x3 = self.f2_sub(self.f2_sub(self.f2_mul(lam, lam), x1), x2)
        y3 = self.f2_sub(self.f2_mul(lam, self.f2_sub(x1, x3)), y1)
        return (x3, y3), self._line(lam, x1, y1, px, py)

    def miller_loop(self, Q, Pt):
        """Q = ((x0, x1), (y0, y1)) on the twist (G2, affine, not infinity), Pt = (x, y) in G1 (affine, not infinity)"""
        px, py = Pt
        T, f = Q, self.F12_ONE
        for i in range(self.loop.bit_length() - 2, -1, -1):
            T, l = self._g2_double_step(T, px, py)
            f = self.f12_mul(self.f12_mul(f, f), l)
            if (self.loop >> i) & 1:
                T, l = self._g2_add_step(T, Q, px, py)
                f = self.f12_mul(f, l)
        if self.bn_end_steps:
            Q1 = (self.f2_mul(self.f2_conj(Q[0]), self.G11), self.f2_mul(self.f2_conj(Q[1]), self.G12))
            nQ2 = (self.f2_mul(Q[0], self.G21), self.f2_neg(self.f2_mul(Q[1], self.G22)))
            T, l = self._g2_add_step(T, Q1, px, py)
            f = self.f12_mul(f, l)
            T, l = self._g2_add_step(T, nQ2, px, py)
            f = self.f12_mul(f, l)
        return f

    def final_exp(self, f):
        return self.f12_pow(f, (self.P ** 12 - 1) // self.R)

    def pairing_product_is_one(self, pairs):
        """prod e(P_i, Q_i) == 1 for pairs of (G1 affine | None, G2 affine | None); a pair with a point at infinity contributes 1"""
        f = self.F12_ONE
        for g1, g2 in pairs:
            if g1 is None or g2 is None:
                continue
            f = self.f12_mul(f, self.miller_loop(g2, g1))
        return self.final_exp(f) == self.F12_ONE

    # ---- G1 (affine over Fp, None = infinity) -----------------------------------------------------------------------
    def g1_add(self, a, b):
        P = self.P
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

    def g1_mul(self, a, k):
        r = None
        while k:
            if k & 1:
                r = self.g1_add(r, a)
            a = self.g1_add(a, a)
            k >>= 1
        return r

    def g1_neg(self, a):
        return None if a is None else (a[0], (-a[1]) % self.P)

    def g1_on_curve(self, a):
        return a is None or (a[1] * a[1] - a[0] * a[0] * a[0] - self.B1) % self.P == 0

    def g2_on_curve(self, q):
        if q is None:
            return True
        x, y = q
        return self.f2_sub(self.f2_mul(y, y), self.f2_add(self.f2_mul(self.f2_mul(x, x), x), self.TWIST_B)) == (0, 0)

    def groth16_verify(self, vk, public_signals, proof):
        pubs = [int(s) for s in public_signals]
        if len(pubs) + 1 != len(vk["IC"]) or any(not (0 <= v < self.R) for v in pubs):                    # :37-42
            return False
        pi_a, pi_b, pi_c = _g1(proof["pi_a"]), _g2(proof["pi_b"]), _g1(proof["pi_c"])
        if not (self.g1_on_curve(pi_a) and self.g2_on_curve(pi_b) and self.g1_on_curve(pi_c)):
            return False
        ic = [_g1(x) for x in vk["IC"]]
        cpub = ic[0]                                                                                       # :55-61
        for v, pt in zip(pubs, ic[1:]):
            cpub = self.g1_add(cpub, self.g1_mul(pt, v))
        return self.pairing_product_is_one([(self.g1_neg(pi_a), pi_b), (cpub, _g2(vk["vk_gamma_2"])), (pi_c, _g2(vk["vk_delta_2"])),
                                            (_g1(vk["vk_alpha_1"]), _g2(vk["vk_beta_2"]))])               # :66-74


def _g1(obj):
    x, y, z = (int(v) for v in obj[:3])
    return None if z == 0 else (x, y)


def _g2(obj):
    (x0, x1), (y0, y1), (z0, z1) = ((int(v[0]), int(v[1])) for v in obj[:3])
    return None if (z0, z1) == (0, 0) else ((x0, x1), (y0, y1))


BN254 = Pairing("bn128", 21888242871839275222246405745257275088696311157297823662689037894645226208583,
                21888242871839275222246405745257275088548364400416034343698204186575808495617,
                29793968203157093288,                    # 6x + 2, x = 4965661367192848881
                (9, 1), "D", 3, True)
BLS12381 = Pairing("bls12381", 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
                   0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
                   0xd201000000010000,                   # |x|
                   (1, 1), "M", 4, False)
CURVES = {"bn128": BN254, "bls12381": BLS12381}

# BN254 under the module's historical names (tests/test_oracle_golden.py)
P, R = BN254.P, BN254.R
F12_ONE = BN254.F12_ONE
f12_mul, f12_pow, miller_loop, final_exp, g1_add, g1_mul, g1_neg = BN254.f12_mul, BN254.f12_pow, BN254.miller_loop, BN254.final_exp, BN254.g1_add, BN254.g1_mul, BN254.g1_neg


def groth16_verify(vk, public_signals, proof):
    """src/groth16_verify.js:26-87 on the JSON objects snarkjs itself uses (vk from zKey.exportVerificationKey, proof / publicSignals
    from groth16.prove: decimal strings, affine points with a trailing "1"); the curve comes from vk.curve."""
    name = vk.get("curve", "bn128")
    if name not in CURVES:
        raise ValueError("groth16_verify_oracle restates the BN254 and BLS12-381 pairings only")
    return CURVES[name].groth16_verify(vk, public_signals, proof)
