"""The unsaturated-limb field / point arithmetic of csrc/field29.cuh + msm29.cuh, executed ON THE CPU (tools/field29_hosttest.hip compiles
the same __host__ __device__ code for the host) and checked against Python big integers: Montgomery products with shared reductions, the
squaring, lazy subtractions (no limb may wrap), the zero test, packing, the R-form / R'-form conversions, G1 mixed / general additions and
the LDS-parked G2 mixed addition incl. the doubling and cancellation branches — for 9 x 29-bit limbs (BN254 Fq) and 14 x 28-bit limbs
(BLS12-381 Fq). The column sums of the product scanning are re-computed here for the worst legal operands: they must stay below 2^64."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "bin", "field29_hosttest")
SRC = os.path.join(ROOT, "tools", "field29_hosttest.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

P = {
    "bn254fq": 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    "bls12381fq_compact": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    "bls12381fq": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
}
P["bn254fr"] = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
P["bls12381fr"] = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
FORM = {"bn254fq": (8, 9, 29), "bls12381fq": (12, 14, 28), "bn254fr": (8, 9, 29), "bls12381fr": (8, 9, 29), "bls12381fq_compact": (12, 14, 28)}       # words, limbs, bits per limb
CURVES = ["bn254fq", "bls12381fq"]          # base fields: MSM accumulation
# Compact<Bls12381Fq> (field29.cuh): the same field with the products behind calls and, in the Fq2 product, the negated component formed inside
# with a 16 p offset — the point formulas run again on it, with the exact precondition checks of the host build
POINT_CURVES = CURVES + ["bls12381fq_compact"]
FR_CURVES = ["bn254fr", "bls12381fr"]       # scalar fields: NTT


def _deps():
    d = [SRC]
    for f in ("field.cuh", "field29.cuh", "msm29.cuh", "msm.cuh", "curve.cuh", "ntt29.cuh", "ntt.cuh"):
        d.append(os.path.join(ROOT, "snarkjs_amd", "csrc", f))
    return d


TOOL_BOUNDS = TOOL + "_bounds"


def _start(binary, extra_flags, errlog=None):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    if not os.path.exists(binary) or any(os.path.getmtime(d) > os.path.getmtime(binary) for d in _deps()):
        os.makedirs(os.path.dirname(binary), exist_ok=True)
        # the host pass alone, unoptimised: seconds instead of minutes; the device pass of the same headers is what snarkjs_amd/build.py compiles
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-O0", "-std=c++17", "-DZK29_CHECK"] + extra_flags + ["-I" + os.path.join(ROOT, "snarkjs_amd", "csrc"), SRC, "-o", binary])
    p = subprocess.Popen([binary], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=errlog, text=True, bufsize=1)

    def call(op, curve, words):
        p.stdin.write(op + " " + curve + " " + " ".join("%x" % w for w in words) + "\n")
        p.stdin.flush()
        line = p.stdout.readline().strip()
        assert not line.startswith("ERR"), (op, curve, line)
        return [int(t, 16) for t in line.split()]
    return p, call


@pytest.fixture(scope="module")
def tool():
    p, call = _start(TOOL, [])
    yield call
    p.stdin.close()
    p.wait(timeout=10)


@pytest.fixture(scope="module")
def tool_bounds(tmp_path_factory):
    """the same tool built with -DZK29_BOUNDS (field29.cuh): every element carries worst-case bounds, every primitive checks its precondition against them"""
    log = tmp_path_factory.mktemp("b29") / "violations.txt"
    with open(log, "w") as fh:
        p, call = _start(TOOL_BOUNDS, ["-DZK29_BOUNDS"], errlog=fh)
        call.log = str(log)
        yield call
        p.stdin.close()
        p.wait(timeout=10)


class Form:
    def __init__(self, curve):
        self.curve = curve
        self.p = P[curve]
        self.N, self.NL, self.B = FORM[curve]
        self.M = (1 << self.B) - 1
        self.Rp = 1 << (self.B * self.NL)
        self.Rinv = pow(self.Rp, -1, self.p)
        self.NP = (-pow(self.p, -1, 1 << self.B)) % (1 << self.B)

    def limbs(self, v):                    # normalised limbs of v (top limb takes the excess)
        out = []
        for i in range(self.NL):
            out.append(v & self.M if i < self.NL - 1 else v)
            v >>= self.B
        assert out[-1] < (1 << 32)
        return out

    def value(self, l):
        return sum(x << (self.B * i) for i, x in enumerate(l))

    def normalised(self, l):
        return all(x <= self.M for x in l[:-1]) and l[-1] < (1 << 32)

    def rand_norm(self, rng, bound=None):
        bound = bound or (1 << (self.B * self.NL - 3))
        return self.limbs(rng.randrange(bound))

    def lazy(self, rng, terms):            # limb-wise sum of `terms` normalised elements whose values add up below 2^(B NL - 3)
        ls = [self.limbs(rng.randrange((1 << (self.B * self.NL - 3)) // terms)) for _ in range(terms)]
        return [sum(c) for c in zip(*ls)]

    def to29(self, x):                      # canonical R'-form limbs of the field element x
        return self.limbs(x * self.Rp % self.p)

    def words(self, v):
        return [(v >> (32 * i)) & 0xffffffff for i in range(self.N)]

    # product scanning exactly as ZK_MS29_BODY: returns result limbs; asserts that no column overflows 64 bits
    def model_mont(self, pairs):
        NL, B, M = self.NL, self.B, self.M
        pl = self.limbs(self.p)
        m, r, acc = [0] * NL, [0] * NL, 0
        for k in range(NL):
            for i in range(k + 1):
                for a, b in pairs:
                    acc += a[i] * b[k - i]
            for i in range(k):
                acc += m[i] * pl[k - i]
            m[k] = ((acc & 0xffffffff) * self.NP) & M
            acc += m[k] * pl[0]
            assert acc < (1 << 64), "column overflow"
            acc >>= B
        for k in range(NL, 2 * NL):
            for i in range(k - NL + 1, NL):
                for a, b in pairs:
                    acc += a[i] * b[k - i]
            for i in range(k - NL + 1, NL):
                acc += m[i] * pl[k - i]
            assert acc < (1 << 64), "column overflow"
            r[k - NL] = (acc & 0xffffffff) if k == 2 * NL - 1 else (acc & M)
            acc >>= B
        return r


@pytest.mark.parametrize("curve", CURVES + FR_CURVES)
def test_constants(tool, curve):
    F = Form(curve)
    c = tool("consts", curve, [])
    NL = F.NL
    assert c[:3] == [F.NL, F.B, F.N]
    assert c[3] == F.NP and c[4] == pow(F.p, -1, 1 << F.B)
    assert c[5:5 + NL] == F.limbs(F.p)
    assert c[5 + NL:5 + 2 * NL] == F.limbs(F.Rp % F.p)
    assert c[5 + 2 * NL:5 + 3 * NL] == F.limbs(pow(2, 2 * F.B * NL - 32 * F.N, F.p))
    assert c[5 + 3 * NL:5 + 4 * NL] == F.limbs(pow(2, 32 * F.N, F.p))


@pytest.mark.parametrize("curve", CURVES + FR_CURVES)
def test_products(tool, curve):
    F = Form(curve)
    rng = random.Random(0x29 + F.NL)
    p, NL = F.p, F.NL

    def check(op, ops, res):
        pairs = list(zip(ops[0::2], ops[1::2]))
        assert res == F.model_mont(pairs), op                                      # limb for limb what the model of the algorithm gives
        val = sum(F.value(a) * F.value(b) for a, b in pairs)
        assert F.value(res) % p == val * F.Rinv % p
        assert F.normalised(res) and F.value(res) <= val // F.Rp + p

    for _ in range(200):
        a, b = F.rand_norm(rng), F.rand_norm(rng)
        check("mul", [a, b], tool("mul", curve, a + b))
        a, b = F.lazy(rng, 2), F.lazy(rng, 2)                                      # both operands with limbs < 2^(B+1)
        check("mul lazy2", [a, b], tool("mul", curve, a + b))
        a, b = F.lazy(rng, 4), F.rand_norm(rng, p)                                 # one operand with limbs < 2^(B+2)
        check("mul lazy4", [a, b], tool("mul", curve, a + b))
        ops = [F.rand_norm(rng, 16 * p) for _ in range(4)]
        ops[rng.randrange(4)] = F.lazy(rng, 2)
        check("mul2", ops, tool("mul2", curve, sum(ops, [])))
        ops = [F.rand_norm(rng, 16 * p) for _ in range(8)]
        check("mul4", ops, tool("mul4", curve, sum(ops, [])))
        a = F.rand_norm(rng)
        r = tool("sqr", curve, a)
        assert F.value(r) % p == F.value(a) ** 2 * F.Rinv % p and F.normalised(r) and F.value(r) <= F.value(a) ** 2 // F.Rp + p
    # worst legal operands: every limb at its bound — the 64-bit column accumulator must hold (model_mont asserts it) and the tool must agree
    top = (1 << (F.B - 3)) - 1
    mx = [F.M] * (NL - 1) + [top]
    mx2 = [2 * F.M] * (NL - 1) + [2 * top]
    mx4 = [4 * F.M] * (NL - 1) + [4 * top]
    check("mul max", [mx, mx], tool("mul", curve, mx + mx))
    check("mul max lazy2", [mx2, mx2], tool("mul", curve, mx2 + mx2))
    check("mul max lazy4", [mx4, mx], tool("mul", curve, mx4 + mx))
    check("mul2 max", [mx, mx, mx, mx2], tool("mul2", curve, mx + mx + mx + mx2))
    check("mul4 max", [mx] * 8, tool("mul4", curve, mx * 8))
    r = tool("sqr", curve, mx)
    assert F.value(r) % p == F.value(mx) ** 2 * F.Rinv % p
    # the squaring's own column bound: NL/2 cross terms of < 2^(2B+1), one square, NL reduction terms
    assert (NL // 2) * (2 * F.M) * F.M + F.M * F.M + NL * F.M * F.M < (1 << 64)


@pytest.mark.parametrize("curve", CURVES)
def test_lazy_sub_norm_zero_pack(tool, curve):
    F = Form(curve)
    rng = random.Random(0x51 + F.NL)
    p, NL, B = F.p, F.NL, F.B
    for K in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 15):
        kp = F.limbs(K * p)
        off = [kp[0] + (1 << B)] + [x + (1 << B) - 1 for x in kp[1:-1]] + [kp[-1] - 1]
        assert F.value(off) == K * p
        for _ in range(40):
            b = F.rand_norm(rng, K * p - (1 << (B * (NL - 1))))                   # any normalised subtrahend below K p - 2^(B (NL-1))
            t = F.lazy(rng, rng.choice((1, 2)))
            r = tool("sub", curve, [K] + t + b)
            assert r == [t[i] + off[i] - b[i] for i in range(NL)]                 # no limb wrapped below zero or above 2^32
            assert all(0 <= x < (1 << 32) for x in r) and F.value(r) == F.value(t) + K * p - F.value(b)
            n = tool("norm", curve, r)
            assert F.normalised(n) and F.value(n) == F.value(r)
        bmax = F.limbs(K * p - (1 << (B * (NL - 1))) - 1)                          # the largest admissible subtrahend
        r = tool("sub", curve, [K] + [0] * NL + bmax)
        assert all(0 <= x < (1 << 32) for x in r) and F.value(r) == K * p - F.value(bmax)
    for k in range(0, 20):
        assert tool("iszero", curve, F.limbs(k * p)) == [1]
        v = k * p + rng.randrange(1, p)
        assert tool("iszero", curve, F.limbs(v)) == [0]
        assert tool("iszero", curve, F.limbs(k * p + (1 << B) * rng.randrange(1, 1 << 40))) == [0]      # same low limb as k p
    for _ in range(50):
        v = rng.randrange(1 << (32 * F.N))
        l = tool("unpack", curve, F.words(v))
        assert l == F.limbs(v)
        assert tool("pack", curve, l) == F.words(v)
        x = rng.randrange(p)
        for top in (0, 1, 2):
            c = tool("canon", curve, F.limbs(x + top * p))
            assert c == F.limbs(x)
        # store: any lazy value -> canonical words in R-form (x 2^(32 N)) or R'-form; fromr: R-form words -> R'-form limbs
        a = F.lazy(rng, rng.choice((1, 2, 3)))
        xa = F.value(a) * F.Rinv % p                                               # the field element a stands for
        assert tool("store", curve, a) == F.words(xa * (1 << (32 * F.N)) % p)
        assert tool("store29", curve, a) == F.words(xa * F.Rp % p)
        r = tool("fromr", curve, F.words(x * (1 << (32 * F.N)) % p))
        assert F.value(r) % p == x * F.Rp % p and F.normalised(r) and F.value(r) < 2 * p


# ---- points: affine chord / tangent rule over Fp and Fp2 = Fp[u]/(u^2 + 1) (a = 0; the formulas never use b) -------------------
class Fp2:
    def __init__(s, p):
        s.p = p

    def add(s, a, b): return ((a[0] + b[0]) % s.p, (a[1] + b[1]) % s.p)
    def sub(s, a, b): return ((a[0] - b[0]) % s.p, (a[1] - b[1]) % s.p)
    def mul(s, a, b): return ((a[0] * b[0] - a[1] * b[1]) % s.p, (a[0] * b[1] + a[1] * b[0]) % s.p)

    def inv(s, a):
        d = pow(a[0] * a[0] + a[1] * a[1], -1, s.p)
        return (a[0] * d % s.p, -a[1] * d % s.p)
    zero, one = (0, 0), (1, 0)


class Fp1:
    def __init__(s, p):
        s.p = p

    def add(s, a, b): return (a + b) % s.p
    def sub(s, a, b): return (a - b) % s.p
    def mul(s, a, b): return a * b % s.p
    def inv(s, a): return pow(a, -1, s.p)
    zero, one = 0, 1


def aff_add(K, P1, P2):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    (x1, y1), (x2, y2) = P1, P2
    if x1 == x2:
        if K.add(y1, y2) == K.zero:
            return None
        lam = K.mul(K.mul(K.add(K.add(x1, x1), x1), x1), K.inv(K.add(y1, y1)))      # 3 x^2 / 2 y
    else:
        lam = K.mul(K.sub(y2, y1), K.inv(K.sub(x2, x1)))
    x3 = K.sub(K.sub(K.mul(lam, lam), x1), x2)
    return (x3, K.sub(K.mul(lam, K.sub(x1, x3)), y1))


def aff_neg(K, Pt):
    return None if Pt is None else (Pt[0], K.sub(K.zero, Pt[1]))


@pytest.mark.parametrize("curve", POINT_CURVES)
def test_g1_additions(tool, curve):
    F = Form(curve)
    K = Fp1(F.p)
    rng = random.Random(0x61 + F.NL)
    p, NL = F.p, F.NL

    def xyzz_value(st):
        inf, l = st[0], st[1:]
        if inf:
            return None
        X, Y, ZZ, ZZZ = (F.value(l[i * NL:(i + 1) * NL]) * F.Rinv % p for i in range(4))
        assert ZZ and (ZZ ** 3 - ZZZ ** 2) % p == 0
        return (X * pow(ZZ, -1, p) % p, Y * pow(ZZZ, -1, p) % p)

    def check_inv(st):                       # the invariants madd29 / padd29 promise for their results (units of p)
        if st[0]:
            return
        l = st[1:]
        vals = [F.value(l[i * NL:(i + 1) * NL]) for i in range(4)]
        assert all(F.normalised(l[i * NL:(i + 1) * NL]) for i in range(4))
        assert vals[0] <= 7.3 * p and vals[1] <= 3.3 * p and vals[2] <= 1.1 * p and vals[3] <= 1.1 * p

    for trial in range(6):
        pts = [(rng.randrange(p), rng.randrange(1, p)) for _ in range(12)]
        seq = list(pts)
        seq.insert(1, pts[0])                          # acc == q: the doubling branch
        seq.insert(5, None)                            # placeholder: add the negative of the running sum (-> infinity), then go on
        st = [1] + [0] * (4 * NL)
        want = None
        for q in seq:
            if q is None:
                q = aff_neg(K, want)
            neg = rng.random() < 0.5
            qq = aff_neg(K, q) if neg else q
            # the kernel negates y as 2p - y (normalised): feed that representation
            ylim = F.to29(q[1])
            if neg:
                ylim = tool("norm", curve, tool("sub", curve, [2] + [0] * NL + ylim))
            st = tool("madd", curve, st + F.to29(q[0]) + ylim)
            want = aff_add(K, want, qq)
            assert xyzz_value(st) == want
            check_inv(st)
        # general additions: fold two accumulators, an accumulator with itself (doubling), with its negative (infinity)
        st2 = [1] + [0] * (4 * NL)
        w2 = None
        for q in pts[:5]:
            st2 = tool("madd", curve, st2 + F.to29(q[0]) + F.to29(q[1]))
            w2 = aff_add(K, w2, q)
        s3 = tool("padd", curve, st + st2[1:])
        assert xyzz_value(s3) == aff_add(K, want, w2)
        check_inv(s3)
        d = tool("padd", curve, st2 + st2[1:])
        assert xyzz_value(d) == aff_add(K, w2, w2)
        check_inv(d)
        # the same point with another representative (scale ZZ by l^2, ZZZ by l^3): P + P must still take the doubling branch
        lam = rng.randrange(2, p)
        X, Y, ZZ, ZZZ = (F.value(st2[1 + i * NL:1 + (i + 1) * NL]) * F.Rinv % p for i in range(4))
        alt = [0] + F.to29(X * lam * lam % p) + F.to29(Y * lam ** 3 % p) + F.to29(ZZ * lam * lam % p) + F.to29(ZZZ * lam ** 3 % p)
        assert xyzz_value(tool("padd", curve, st2 + alt[1:])) == aff_add(K, w2, w2)
        negalt = [0] + alt[1:1 + NL] + F.to29(-Y * lam ** 3 % p) + alt[1 + 2 * NL:]
        assert tool("padd", curve, st2 + negalt[1:])[0] == 1
        assert xyzz_value(tool("padd", curve, [1] + [0] * (4 * NL) + st2[1:])) == w2
        # bucket formats: R-form words, R'-form words and the way back
        w = tool("storept", curve, st2)
        Rw = 1 << (32 * F.N)
        got = [sum(w[i * F.N + k] << (32 * k) for k in range(F.N)) for i in range(4)]
        assert got == [X * Rw % p, Y * Rw % p, ZZ * Rw % p, ZZZ * Rw % p]
        w = tool("storept29", curve, st2)
        got = [sum(w[i * F.N + k] << (32 * k) for k in range(F.N)) for i in range(4)]
        assert got == [X * F.Rp % p, Y * F.Rp % p, ZZ * F.Rp % p, ZZZ * F.Rp % p]
        assert w[4 * F.N] == 1 and xyzz_value([0] + w[4 * F.N + 1:]) == w2
        assert tool("storept29", curve, [1] + [0] * (4 * NL))[:4 * F.N + 1] == [0] * (4 * F.N + 1)


@pytest.mark.parametrize("curve", POINT_CURVES)
def test_g2_split_layout_additions(tool, curve):
    """r06: the arithmetic of k_msm_accum29_g2s (one Fq2 component per lane) — per component exactly madd29_lds's XYZZ products, offsets and carry passes, on an
    unpacked accumulator, for both limb forms (the 14-limb curve ran the packed Jacobian before): main path, doubling, cancellation, first point. What the split
    kernel does beyond madd29_lds is form K p - b for BOTH components of a right-hand operand (madd29_lds negates c1 only, c0 where a formula needs it) and
    a0 - a1 + KB p in both orders inside a square: the invariants bound both components of every value alike, so the same offsets cover them."""
    test_g2_lds_parked_additions(tool, curve, op="madd2xyzz")


@pytest.mark.parametrize("curve", POINT_CURVES)
def test_g2_lds_parked_additions(tool, curve, op="madd2seq"):
    F = Form(curve)
    K = Fp2(F.p)
    rng = random.Random(0x71 + F.NL)
    p, N = F.p, F.N
    Rw = 1 << (32 * N)
    for trial in range(6):
        pts = [((rng.randrange(p), rng.randrange(p)), (rng.randrange(p), rng.randrange(1, p))) for _ in range(10)]
        seq = [(q, rng.random() < 0.5) for q in pts]
        seq.insert(1, seq[0])                          # the same signed point twice: doubling branch
        cut = 6
        want = None
        for q, neg in seq[:cut]:
            want = aff_add(K, want, aff_neg(K, q) if neg else q)
        seq.insert(cut, (aff_neg(K, want), False))     # cancels the running sum: infinity, then the sequence goes on from there
        for upto in (1, 2, cut, cut + 1, len(seq)):
            want = None
            req = [upto]
            for q, neg in seq[:upto]:
                want = aff_add(K, want, aff_neg(K, q) if neg else q)
                req += [1 if neg else 0] + F.to29(q[0][0]) + F.to29(q[0][1]) + F.to29(q[1][0]) + F.to29(q[1][1])
            out = tool(op, curve, req)
            inf, w = out[0], out[1:]
            if want is None:
                assert inf == 1 and not any(w)
                continue
            assert inf == 0
            v = [sum(w[i * N + k] << (32 * k) for k in range(N)) for i in range(8)]
            assert all(x < p for x in v)                                            # canonical words
            Rwi = pow(Rw, -1, p)
            X, Y, ZZ, ZZZ = ((v[2 * i] * Rwi % p, v[2 * i + 1] * Rwi % p) for i in range(4))
            assert K.mul(K.mul(ZZ, ZZ), ZZ) == K.mul(ZZZ, ZZZ)
            assert (K.mul(X, K.inv(ZZ)), K.mul(Y, K.inv(ZZZ))) == want


@pytest.mark.parametrize("curve", POINT_CURVES)
def test_g2_bucket_reduction_additions(tool, curve):
    """padd29_lds (general XYZZ additions of the Fq2 row / column sums): buckets built by the accumulation path and stored as R'-form words are
    folded from the words; the sum is then added to itself through the accumulator-to-accumulator form (the tree step; doubling branch)."""
    F = Form(curve)
    K = Fp2(F.p)
    rng = random.Random(0x93 + F.NL)
    p, N = F.p, F.N
    Rwi = pow(1 << (32 * N), -1, p)

    def decode(out):
        inf, w = out[0], out[1:1 + 8 * N]
        if inf:
            assert not any(w)
            return None, out[1 + 8 * N:]
        v = [sum(w[i * N + k] << (32 * k) for k in range(N)) for i in range(8)]
        assert all(x < p for x in v)
        X, Y, ZZ, ZZZ = ((v[2 * i] * Rwi % p, v[2 * i + 1] * Rwi % p) for i in range(4))
        assert K.mul(K.mul(ZZ, ZZ), ZZ) == K.mul(ZZZ, ZZZ)
        return (K.mul(X, K.inv(ZZ)), K.mul(Y, K.inv(ZZZ))), out[1 + 8 * N:]

    for trial in range(5):
        groups = []
        for g in range(7):
            groups.append([(((rng.randrange(p), rng.randrange(p)), (rng.randrange(p), rng.randrange(1, p))), rng.random() < 0.5) for _ in range(rng.randrange(1, 5))])
        groups.insert(2, list(groups[1]))              # the same bucket twice: equal points from different representatives? same build: doubling
        groups.insert(4, [groups[0][0], (groups[0][0][0], not groups[0][0][1])])      # a bucket that cancels to infinity: skipped
        for upto in (1, 3, len(groups), len(groups) + 1):
            gs = list(groups[:upto])
            want = None                                # the points are random pairs, not points of one curve: the chord / tangent formulas are
            for grp in gs:                             # still well defined but not associative, so the expectation associates like the kernel does
                b = None
                for q, neg in grp:
                    b = aff_add(K, b, aff_neg(K, q) if neg else q)
                want = aff_add(K, want, b)
            if upto == len(groups) + 1:                # one more bucket: the negative of everything so far -> the sum is infinity
                gs.append([(aff_neg(K, want), False)])
                want = None
            req = [len(gs)]
            for grp in gs:
                req.append(len(grp))
                for q, neg in grp:
                    req += [1 if neg else 0] + F.to29(q[0][0]) + F.to29(q[0][1]) + F.to29(q[1][0]) + F.to29(q[1][1])
            out = tool("padd2", curve, req)
            got, rest = decode(out)
            assert got == want
            got2, rest = decode(rest)
            assert got2 == (aff_add(K, want, want) if want is not None else None) and not rest


@pytest.mark.parametrize("curve", POINT_CURVES)
def test_g2_row_sum_wave_flow(tool, curve):
    """k_msm_rowcol_wave29_g2's flow for one wave, emulated on the host with the kernel's own accumulator layout (stride 64): every lane
    places one R'-form bucket, six tree levels of accumulator-to-accumulator additions; empty buckets in between."""
    F = Form(curve)
    K = Fp2(F.p)
    rng = random.Random(0xA7 + F.NL)
    p, N = F.p, F.N
    Rwi = pow(1 << (32 * N), -1, p)
    words = lambda v: [(v >> (32 * k)) & 0xffffffff for k in range(N)]
    for trial in range(3):
        lanes, req = [], []
        for l in range(64):
            if trial and rng.random() < 0.3:
                lanes.append(None)
                req += [0] * (8 * N)
                continue
            q = ((rng.randrange(p), rng.randrange(p)), (rng.randrange(p), rng.randrange(1, p)))
            lam = (rng.randrange(1, p), rng.randrange(p))                    # a non-trivial representative: ZZ = lam^2, ZZZ = lam^3
            zz = K.mul(lam, lam)
            zzz = K.mul(zz, lam)
            coords = (K.mul(q[0], zz), K.mul(q[1], zzz), zz, zzz)
            lanes.append(q)
            for cdn in coords:
                for comp in cdn:
                    req += words(comp * F.Rp % p)
        # the tree associates pairwise: (l, l + 1), then (l, l + 2), ...
        cur = list(lanes)
        d = 1
        while d < 64:
            for l in range(0, 64, 2 * d):
                cur[l] = aff_add(K, cur[l], cur[l + d])
            d *= 2
        out = tool("rowsum", curve, req)
        inf, w = out[0], out[1:]
        if cur[0] is None:
            assert inf == 1
            continue
        assert inf == 0
        v = [sum(w[i * N + k] << (32 * k) for k in range(N)) for i in range(8)]
        X, Y, ZZ, ZZZ = ((v[2 * i] * Rwi % p, v[2 * i + 1] * Rwi % p) for i in range(4))
        assert (K.mul(X, K.inv(ZZ)), K.mul(Y, K.inv(ZZZ))) == cur[0]


@pytest.mark.parametrize("curve", FR_CURVES)
def test_ntt_butterfly_and_final_reduction(tool, curve):
    """ntt29.cuh: the decimation-in-time butterfly on lazy values (x + w y, x - w y + 2r) through as many stages as the largest tile has, with the
    worst admissible growth, and the final reduction of a lazy value to the canonical range (quotient estimate from the top limb)."""
    F = Form(curve)
    rng = random.Random(0x91 + F.p % 97)
    r, NL = F.p, F.NL
    # final reduction: every value below 32 r, incl. multiples of r and their neighbours
    vals = [0, 1, r - 1, r, r + 1, 31 * r, 32 * r - 1] + [k * r + d for k in range(32) for d in (-1, 0, 1) if 0 <= k * r + d < 32 * r]
    vals += [rng.randrange(32 * r) for _ in range(300)]
    for v in vals:
        assert tool("reduce", curve, F.limbs(v)) == F.limbs(v % r), hex(v)
    # butterflies: 9 stages (the largest tile), twiddles in R'-form, data values in the caller's form (any residue)
    for trial in range(20):
        x, y = F.limbs(rng.randrange(r)), F.limbs(rng.randrange(r))
        for stage in range(9):
            w = rng.randrange(r)
            out = tool("bfly", curve, x + y + F.to29(w) + [1 if stage else 0])
            x2, y2 = out[:NL], out[NL:]
            t = F.value(y) * w % r if stage else F.value(y) % r
            assert F.value(x2) % r == (F.value(x) + t) % r and F.value(y2) % r == (F.value(x) - t) % r
            assert F.normalised(x2) and F.normalised(y2)
            assert F.value(x2) < 1.4 * r + 2 * r * (stage + 1) and F.value(y2) < 1.4 * r + 2 * r * (stage + 1)      # grows by at most 2r per stage
            x, y = (x2, y2) if rng.random() < 0.5 else (y2, x2)                                                       # either output may be multiplied next
        assert tool("reduce", curve, x) == F.limbs(F.value(x) % r)
    # worst case: the largest lazy value the last stage can see (19.3 r) as BOTH operands
    big = F.limbs(int(19.3 * r))
    out = tool("bfly", curve, big + big + F.to29(r - 1) + [1])
    assert F.value(out[:NL]) % r == (F.value(big) + F.value(big) * (r - 1)) % r and F.value(out[:NL]) < 32 * r and F.value(out[NL:]) < 32 * r
    assert tool("reduce", curve, out[:NL]) == F.limbs(F.value(out[:NL]) % r)


def _no_violations(tool_bounds, curve):
    failures, col_mlog2, on = tool_bounds("bounds", curve, [])
    assert on == 1, "tool built without -DZK29_BOUNDS"
    assert failures == 0, open(tool_bounds.log).read()[-3000:]
    return col_mlog2 / 1000.0


@pytest.mark.parametrize("curve", POINT_CURVES)
def test_worst_case_bounds_of_the_point_formulas(tool_bounds, curve):
    """The point formulas again, on the tool built with -DZK29_BOUNDS: the harness gives every input the bounds its function's contract admits (accumulator
    coordinates at their invariants, table entries canonical, negated y below 2 p) — independent of the values at hand — and every product, lazy
    addition / subtraction, carry pass, zero test and final reduction checks its precondition against the bounds that follow; the results must come back
    within the invariants. Along every path these tests drive (main path, doubling, cancellation, first point, tree folds) the arithmetic is thereby
    verified for ALL admissible inputs, not only for the sampled ones."""
    test_g1_additions(tool_bounds, curve)
    test_g2_lds_parked_additions(tool_bounds, curve)
    test_g2_lds_parked_additions(tool_bounds, curve, op="madd2xyzz")            # r06: the split layout's formulas, both limb forms
    test_g2_bucket_reduction_additions(tool_bounds, curve)
    test_g2_row_sum_wave_flow(tool_bounds, curve)
    col = _no_violations(tool_bounds, curve)
    assert 60.0 < col < 64.0               # the tightest column of any product stays below 2^64 (and the tracking saw real products)


@pytest.mark.parametrize("curve", FR_CURVES)
def test_worst_case_bounds_of_the_ntt_butterfly(tool_bounds, curve):
    test_ntt_butterfly_and_final_reduction(tool_bounds, curve)
    # what a pass does to an element before its first stage: the record of the pass before (any lazy value) times the pass twiddle (a product of two table
    # entries), the caller's row factor and 1/n — and the 1.3 r the first stage's contract starts from
    F = Form(curve)
    rng = random.Random(7)
    r = F.p
    for _ in range(10):
        x, a, b, c, d = rng.randrange(19 * r), rng.randrange(r), rng.randrange(r), rng.randrange(r), rng.randrange(r)
        out = tool_bounds("nttin", curve, F.limbs(x) + F.to29(a) + F.to29(b) + F.to29(c) + F.to29(d))
        assert F.value(out) % r == x * a * b * c * d % r and F.normalised(out) and F.value(out) < 1.3 * r
    _no_violations(tool_bounds, curve)
