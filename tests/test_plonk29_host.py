"""PLONK's quotient numerator on 29-bit limbs (csrc/plonk29.cuh: what k_plonk_t29 runs per evaluation point), executed ON THE CPU by
tools/plonk29_hosttest.hip — the same __host__ __device__ body — and compared with the oracle's literal expansion of the reference's loop
(oracle/plonk_oracle.py: mul2 / mul4 follow src/mul_z.js:49-148, the terms follow src/plonk_prove.js:560-612), bit for bit, for BN254 Fr and
BLS12-381 Fr, on random and on extreme inputs (0, r - 1: the largest loaded values). A second mode instantiates the body with an interval
type and checks every precondition of the lazy arithmetic for the worst case."""
import os
import random
import struct
import subprocess
import sys
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.plonk_oracle import mul2, mul4  # noqa: E402

TOOL = os.path.join(ROOT, "tools", "bin", "plonk29_hosttest")
SRC = os.path.join(ROOT, "tools", "plonk29_hosttest.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
R = {
    "bn254fr": 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
    "bls12381fr": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
}
# constants block of plonk29.cuh
PK_BETA, PK_GAMMA, PK_K1, PK_K2, PK_ALPHA, PK_ALPHA2, PK_WN, PK_B1 = range(8)
PK_Z1, PK_Z2, PK_Z3, PK_ONE, PK_NALPHA, PK_COUNT = PK_B1 + 11, PK_B1 + 15, PK_B1 + 19, PK_B1 + 23, PK_B1 + 24, PK_B1 + 25


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    deps = [SRC] + [os.path.join(ROOT, "snarkjs_amd", "csrc", f) for f in ("field.cuh", "field29.cuh", "ntt29.cuh", "ntt.cuh", "plonk29.cuh", "mac_cols.inc")]
    if not os.path.exists(TOOL) or any(os.path.getmtime(d) > os.path.getmtime(TOOL) for d in deps):
        os.makedirs(os.path.dirname(TOOL), exist_ok=True)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "--cuda-host-only", "-O0", "-std=c++17", "-DZK29_CHECK", "-DZK29_BOUNDS", "-I" + os.path.join(ROOT, "snarkjs_amd", "csrc"), SRC, "-o", TOOL])
    return TOOL


@pytest.mark.parametrize("curve", ["bn254fr", "bls12381fr"])
def test_worst_case_bounds(tool, curve):
    out = subprocess.run([tool, "bounds", curve], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def _mont(vals, r):
    return b"".join(((v << 256) % r).to_bytes(32, "little") for v in vals)


@pytest.mark.parametrize("curve", ["bn254fr", "bls12381fr"])
@pytest.mark.parametrize("fill", ["random", "max", "zero", "mixed"])
def test_points_against_oracle(tool, curve, fill, tmp_path):
    r = R[curve]
    rng = random.Random(zlib.crc32((curve + fill).encode()))
    dom, npub = 8, 2
    n4 = 4 * dom

    def el():
        if fill == "max":
            return r - 1
        if fill == "zero":
            return 0
        if fill == "mixed":
            return rng.choice([0, 1, r - 1, r - 2, rng.randrange(r), (1 << 253) - 1])
        return rng.randrange(r)

    arr = {k: [el() for _ in range(n4)] for k in ("a", "b", "c", "z", "qm", "ql", "qr", "qo", "qc", "s1", "s2", "s3")}
    lag = [[el() for _ in range(5 * dom)] for _ in range(npub)]
    pub = [el() for _ in range(npub)]
    beta, gamma, k1, k2, alpha, wn = (el() for _ in range(6))
    b = [None] + [el() for _ in range(11)]
    w2 = rng.randrange(r)                         # the constants only need to be field elements
    Z = ([0, (-1 + w2) % r, (-2) % r, (-1 - w2) % r], [0, (-2 * w2) % r, 4, (2 * w2) % r], [0, (2 + 2 * w2) % r, (-8) % r, (2 - 2 * w2) % r])
    kv = [0] * PK_COUNT
    kv[PK_BETA], kv[PK_GAMMA], kv[PK_K1], kv[PK_K2], kv[PK_ALPHA], kv[PK_ALPHA2], kv[PK_WN] = beta, gamma, k1, k2, alpha, alpha * alpha % r, wn
    for j in range(11):
        kv[PK_B1 + j] = b[j + 1]
    for j in range(4):
        kv[PK_Z1 + j], kv[PK_Z2 + j], kv[PK_Z3 + j] = Z[0][j], Z[1][j], Z[2][j]
    kv[PK_ONE], kv[PK_NALPHA] = 1, (-alpha) % r
    w4 = rng.randrange(1, r)
    lb = 3
    lo = [pow(w4, j, r) for j in range(1 << lb)]
    hi = [pow(w4, j << lb, r) for j in range(n4 >> lb)]
    blob = struct.pack("<III", dom, npub, lb)
    for k in ("a", "b", "c", "z", "qm", "ql", "qr", "qo", "qc", "s1", "s2", "s3"):
        blob += _mont(arr[k], r)
    for row in lag:
        blob += _mont(row, r)
    blob += _mont(pub, r) + _mont(kv, r) + _mont([32 * v % r for v in kv], r) + _mont(lo, r) + _mont(hi, r)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(blob)
    res = subprocess.run([tool, "run", curve, str(fin), str(fout)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    got = fout.read_bytes()
    assert len(got) == 2 * n4 * 32
    rinv = pow(1 << 256, -1, r)
    T, Tz = [], []
    for i in range(n4):
        a, bb, c, z = arr["a"][i], arr["b"][i], arr["c"][i], arr["z"][i]
        zw = arr["z"][(n4 + 4 + i) % n4]
        w = pow(w4, i, r)
        ap, bp, cp = (b[2] + b[1] * w) % r, (b[4] + b[3] * w) % r, (b[6] + b[5] * w) % r
        zp = (b[7] * w * w + b[8] * w + b[9]) % r
        wW = w * wn % r
        zWp = (b[7] * wW * wW + b[8] * wW + b[9]) % r
        pi = 0
        for j in range(npub):
            pi = (pi - lag[j][dom + i] * pub[j]) % r
        e1, e1z = mul2(a, bb, ap, bp, i % 4, Z, r)
        e1 = (e1 * arr["qm"][i] + a * arr["ql"][i] + bb * arr["qr"][i] + c * arr["qo"][i] + pi + arr["qc"][i]) % r
        e1z = (e1z * arr["qm"][i] + ap * arr["ql"][i] + bp * arr["qr"][i] + cp * arr["qo"][i]) % r
        betaw = beta * w % r
        e2, e2z = mul4((a + betaw + gamma) % r, (bb + betaw * k1 + gamma) % r, (c + betaw * k2 + gamma) % r, z, ap, bp, cp, zp, i % 4, Z, r)
        e3, e3z = mul4((a + beta * arr["s1"][i] + gamma) % r, (bb + beta * arr["s2"][i] + gamma) % r, (c + beta * arr["s3"][i] + gamma) % r, zw, ap, bp, cp, zWp, i % 4, Z, r)
        l1 = lag[0][dom + i]
        e4 = (z - 1) * l1 % r * kv[PK_ALPHA2] % r
        e4z = zp * l1 % r * kv[PK_ALPHA2] % r
        T.append((e1 + e2 * alpha - e3 * alpha + e4) % r)
        Tz.append((e1z + e2z * alpha - e3z * alpha + e4z) % r)
    for name, exp, off in (("t", T, 0), ("tz", Tz, n4 * 32)):
        for i in range(n4):
            v = int.from_bytes(got[off + 32 * i:off + 32 * i + 32], "little")
            assert v < r, (name, i, "not canonical")
            assert v * rinv % r == exp[i], (name, i)
