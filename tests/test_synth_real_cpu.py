"""The circuit-shaped synthetic Groth16 key (snarkjs_amd/workloads/synth_zkey.py: coef_dist="real", bench.py --coef-dist real) on CPU: the shape it claims — n_coef ~ 2.5 - 3 n,
nine rows in ten with one term, a heavy tail up to one 10^5-term row from 2^17 constraints, signals distinct within a row, B density 0.4 with the B1 / B2 bases of absent
signals at infinity — and that the container it writes is one the oracle's restatement of buildABC1 (src/groth16_prove.js:147-187) accepts, row sums checked against Python integers."""
import numpy as np

import oracle_lib as O
import synth_zkey
from snarkjs_amd import binfile
from snarkjs_amd.workloads import synth_zkey as Z

R = Z.PRIMES["bn128"][2]


def test_real_coefficient_shape():
    for lg, longest in ((12, 1000), (15, 10000), (17, 100000)):
        n, m = 1 << lg, (1 << lg) - 5
        mm, cc, ss, in_b = Z.real_coefs(n, m, 2, 0x5EED)
        rows = mm.astype(np.int64) * n + cc.astype(np.int64)
        cnt = np.bincount(rows, minlength=2 * n)
        assert 2.3 * n <= mm.size <= 3.8 * n, (lg, mm.size / n)
        assert cnt.max() >= longest and (cnt == 1).mean() > 0.85
        assert np.unique(rows * m + ss.astype(np.int64)).size == mm.size, "a signal occurs twice in one row"
        assert ss.max() < m and abs(in_b.mean() - 0.4) < 0.01
        assert in_b[ss[mm == 1]].all(), "a B row uses a signal that is marked absent from B"


def test_real_key_container_and_oracle_build_abc():
    zkey, wtns = synth_zkey.make("bn128", 10, seed=9, use_device=False, coef_dist="real", witness="mixed")
    zk, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)["witness"]
    n, m = zk["domainSize"], zk["nVars"]
    rec = np.frombuffer(zk["coeffs"][4:].tobytes(), dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "u1", 32)])
    assert int(np.frombuffer(zk["coeffs"][:4].tobytes(), "<u4")[0]) == rec.size
    # the B1 / B2 bases of signals that never occur in matrix 1 are the point at infinity (all-zero bytes), the others are not
    used = np.zeros(m, bool)
    used[rec["s"][rec["m"] == 1]] = True
    b1, b2 = zk["B1"].reshape(m, 64), zk["B2"].reshape(m, 128)
    nz1, nz2 = b1.any(axis=1), b2.any(axis=1)
    assert nz1[used].all() and nz2[used].all() and np.array_equal(nz1, nz2)
    assert abs((~nz1).mean() - 0.6) < 0.02
    A, B, Cc = O.build_abc(0, zk["coeffs"], w, m, n)
    wi = [int.from_bytes(bytes(w[32 * i:32 * i + 32]), "little") for i in range(m)]
    rinv = pow(1 << 256, -1, R)
    exp = [[0] * n, [0] * n]
    for r_ in rec:
        v = int.from_bytes(bytes(r_["v"]), "little")                      # stored x R^2: the Montgomery product with a normal-form witness is (v w) x R
        exp[int(r_["m"])][int(r_["c"])] = (exp[int(r_["m"])][int(r_["c"])] + v * wi[int(r_["s"])] * rinv) % R
    for mat, got in ((0, A), (1, B)):
        for c in (0, 1, 2, n // 2, n - 2, n - 1, int(np.bincount(rec["c"][rec["m"] == mat]).argmax())):
            assert int.from_bytes(bytes(got[32 * c:32 * c + 32]), "little") == exp[mat][c], (mat, c)
