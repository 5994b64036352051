"""GPU parity of the PLONK rows: device polynomial kernels vs the Python restatement
(oracle/plonk_oracle.py) on seeded inputs, and the full device prover vs the seeded proofs of the reference."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import plonk_oracle as P
import synth


def test_product_keccak_matches_oracle_and_known_answers():
    from snarkjs_amd import plonk
    assert plonk.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    for k in (1, 3, 64, 135, 136, 137, 271, 272, 273, 1000):
        m = bytes((7 * i + k) & 255 for i in range(k))
        assert plonk.keccak256(m) == P.keccak256(m), k


@pytest.fixture(scope="module")
def env():
    from snarkjs_amd import zkmi, plonk
    zkmi.init(0)
    return zkmi, plonk, plonk._Field(0), P.Ctx()


def _dev(zkmi, cx, vals):
    return zkmi.DeviceBuffer.from_host(np.frombuffer(cx.to_mont(vals), np.uint8))


def _host(cx, buf, n):
    return cx.from_mont(buf.to_host(n * 32))


def _rand(seed, n, r):
    return [synth.to_int(b) % r for b in synth.elems(seed, n).reshape(n, 32)]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 7, 256, 2048, 2049, 5000, 70001])
def test_poly_ops_vs_oracle(env, n):
    zkmi, plonk, f, cx = env
    L, r = zkmi.lib(), cx.r
    a, b = _rand(1000 + n, n, r), _rand(2000 + n, n, r)
    k, x = _rand(3, 2, r)
    mont = lambda v: zkmi.ptr(f.mont(v))
    # axpy / scale (polynomial.js:218-284)
    for sub in (0, 1):
        for kk in (None, k):
            da, db = _dev(zkmi, cx, a), _dev(zkmi, cx, b)
            zkmi.check(L.zkmi_poly_axpy_dev(0, da.ptr, db.ptr, n, None if kk is None else mont(kk), sub))
            assert _host(cx, da, n) == P.poly_add(a, b, r, kk, -1 if sub else 1)
    da = _dev(zkmi, cx, a)
    zkmi.check(L.zkmi_poly_scale_dev(0, da.ptr, n, mont(k)))
    assert _host(cx, da, n) == [v * k % r for v in a]
    # evaluate (Horner, :174-184)
    out = np.zeros(32, np.uint8)
    da = _dev(zkmi, cx, a)
    zkmi.check(L.zkmi_poly_evaluate_dev(0, da.ptr, n, mont(x), zkmi.ptr(out)))
    assert f.unmont(out) == P.evaluate(a, x, r)
    # divByZerofier(1, beta) (:617-674): build an exactly divisible polynomial p = q * (X - beta)
    if n >= 2:
        q = a[:n - 1]
        p = [0] * n
        for i, c in enumerate(q):
            p[i] = (p[i] - x * c) % r
            p[i + 1] = (p[i + 1] + c) % r
        want = P.div_by_zerofier(p, 1, x, r)
        assert want[:n - 1] == q and want[n - 1] == 0
        dp = _dev(zkmi, cx, p)
        zkmi.check(L.zkmi_poly_div_by_zerofier_dev(0, dp.ptr, n, 1, mont(x)))
        assert _host(cx, dp, n) == want
        p[0] = (p[0] + 1) % r                       # not divisible any more
        dp = _dev(zkmi, cx, p)
        assert L.zkmi_poly_div_by_zerofier_dev(0, dp.ptr, n, 1, mont(x)) != 0
        assert b"not divisible" in L.zkmi_last_error()
    z = C.c_int(0)
    dz = _dev(zkmi, cx, [0] * n)
    zkmi.check(L.zkmi_poly_is_zero_dev(0, dz.ptr, n, C.byref(z)))
    assert z.value == 1
    zkmi.check(L.zkmi_poly_is_zero_dev(0, da.ptr, n, C.byref(z)))
    assert z.value == (0 if any(a) else 1)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 300, 4099, 70001])
def test_fused_poly_forms_vs_oracle(env, n):
    """r06 launch diet: one-launch forms of chains the reference makes call by call, against the restatement of those calls (oracle/plonk_oracle.py):
    zkmi_poly_lincomb_dev == a sequence of Polynomial.add / sub / mulScalar / addScalar, zkmi_poly_evaluate_multi_dev == Polynomial.evaluate per polynomial,
    zkmi_poly_blind_tail_dev == blindCoefficients, zkmi_poly_div_by_zerofier_enqueue + the caller's tail test == divByZerofier, zkmi_fr_batch_multi_dev == batchFromMontgomery."""
    zkmi, plonk, f, cx = env
    L, r = zkmi.lib(), cx.r
    lens = [n, max(1, n - 3), n + 6, 1, n + 2]
    polys = [_rand(500 + 10 * j + n, ln, r) for j, ln in enumerate(lens)]
    ks = [_rand(9, 5, r)[j] if j != 1 else None for j in range(5)]
    const = _rand(11, 1, r)[0]
    out_len = n + 6
    want = [0] * out_len
    for p_, k in zip(polys, ks):
        want = P.poly_add(want, p_, r, k, 1)
    devs = [_dev(zkmi, cx, p_) for p_ in polys]
    for with_const in (False, True):
        out = plonk._Poly(f, out_len, zero=False)
        plonk.lincomb(f, out, [(d.ptr, ln, k) for d, ln, k in zip(devs, lens, ks)], const if with_const else None)
        w = list(want)
        if with_const:
            w[0] = (w[0] + const) % r
        assert _host(cx, out.buf, out_len) == w
    # in place: the output is the first operand
    acc = plonk._Poly(f, out_len)
    acc.copy_from(devs[2].ptr, lens[2])
    plonk.lincomb(f, acc, [(acc.ptr, out_len, None), (devs[0].ptr, lens[0], (r - 1))])
    assert _host(cx, acc.buf, out_len) == P.poly_add(polys[2], polys[0], r, None, -1)
    # too long a term, too many terms
    with pytest.raises(zkmi.ZkmiError):
        plonk.lincomb(f, plonk._Poly(f, 2, zero=False), [(devs[2].ptr, lens[2], None)])
    with pytest.raises(zkmi.ZkmiError):
        plonk.lincomb(f, out, [(devs[0].ptr, lens[0], None)] * 17)
    # evaluations: five polynomials, two points, one of them shared by four
    x, y = _rand(13, 2, r)
    xs = [x, x, y, x, x]
    assert plonk.evaluate_many(f, [(d.ptr, ln) for d, ln in zip(devs, lens)], xs) == [P.evaluate(p_, v, r) for p_, v in zip(polys, xs)]
    # blindCoefficients in place on a buffer whose tail was never written
    for cnt in (1, 2, 3):
        if n < cnt:
            continue
        fac = _rand(17 + cnt, cnt, r)
        big = plonk._Poly(f, n + cnt, zero=False)
        zkmi.check(L.zkmi_memset_dev(big.ptr, 0xA5, (n + cnt) * 32))
        big.copy_from(devs[0].ptr, n)
        fb = np.concatenate([f.mont(v) for v in fac])
        zkmi.check(L.zkmi_poly_blind_tail_dev(0, big.ptr, n, zkmi.ptr(fb), cnt))
        assert _host(cx, big.buf, n + cnt) == P.blind(polys[0], fac, r)
    # the enqueued division leaves the quotient and a zero top coefficient iff divisible
    if n >= 2:
        q = polys[0][:n - 1]
        p_ = [0] * n
        for i, c in enumerate(q):
            p_[i] = (p_[i] - x * c) % r
            p_[i + 1] = (p_[i + 1] + c) % r
        for bump in (0, 1):
            p_[0] = (p_[0] + bump) % r
            dp = plonk._Poly(f, n, zero=False)
            dp.copy_from(_dev(zkmi, cx, p_).ptr, n)
            zkmi.check(L.zkmi_poly_div_by_zerofier_enqueue(0, dp.ptr, n, 1, zkmi.ptr(f.mont(x))))
            if bump == 0:
                assert dp.tail_is_zero(n - 1) and _host(cx, dp.buf, n) == P.div_by_zerofier(p_, 1, x, r)
            else:
                assert not dp.tail_is_zero(n - 1)
    # conversions of several arrays in one launch
    outs = [zkmi.DeviceBuffer(ln * 32) for ln in lens[:4]]
    ptrs_in = (C.c_void_p * 4)(*[d.ptr for d in devs[:4]])
    ptrs_out = (C.c_void_p * 4)(*[o.ptr for o in outs])
    ns = (C.c_size_t * 4)(*lens[:4])
    zkmi.check(L.zkmi_fr_batch_multi_dev(0, zkmi.BATCH_FROM_MONTGOMERY, ptrs_in, ptrs_out, ns, 4))
    for o, p_, ln in zip(outs, polys, lens):
        assert [int.from_bytes(bytes(o.to_host(ln * 32)[32 * i:32 * i + 32]), "little") for i in range(ln)] == p_
    zkmi.check(L.zkmi_fr_batch_multi_dev(0, zkmi.BATCH_TO_MONTGOMERY, ptrs_out, ptrs_out, ns, 4))
    for o, p_, ln in zip(outs, polys, lens):
        assert _host(cx, o, ln) == p_


@pytest.mark.gpu
@pytest.mark.parametrize("curve,lg,in_len", [(0, 4, 4), (0, 4, 5), (0, 10, 256), (0, 10, 1), (0, 14, 4096), (0, 14, 4099), (0, 18, 65536 + 3), (1, 12, 1024), (0, 8, 256), (0, 23, (1 << 21) + 1)])
def test_ntt_padded_equals_ntt_of_zero_padded_copy(env, curve, lg, in_len):
    """zkmi_ntt_padded_dev (Evaluations.fromPolynomial, evaluations.js:30-37): reading zeros behind in_len == transforming a zero-padded copy, both limb forms of the passes"""
    zkmi, plonk, _, _ = env
    L = zkmi.lib()
    N = 1 << lg
    src = zkmi.DeviceBuffer.from_host(synth.elems(4242 + lg + in_len, in_len))                     # values below 2^253: reduced in both scalar fields
    padded = zkmi.DeviceBuffer(N * 32)
    zkmi.check(L.zkmi_memset_dev(padded.ptr, 0, N * 32))
    zkmi.check(L.zkmi_memcpy_d2d(padded.ptr, src.ptr, in_len * 32))
    for inverse in (0, 1):
        want = zkmi.DeviceBuffer(N * 32)
        zkmi.check(L.zkmi_ntt_dev(curve, padded.ptr, want.ptr, lg, inverse, None, None))
        got = zkmi.DeviceBuffer(N * 32)
        zkmi.check(L.zkmi_memset_dev(got.ptr, 0x5A, N * 32))
        zkmi.check(L.zkmi_ntt_padded_dev(curve, src.ptr, in_len, got.ptr, lg, inverse))
        assert bytes(got.to_host(N * 32)) == bytes(want.to_host(N * 32)), (lg, in_len, inverse)
    assert L.zkmi_ntt_padded_dev(curve, src.ptr, 0, padded.ptr, lg, 0) != 0
    assert L.zkmi_ntt_padded_dev(curve, src.ptr, N + 1, padded.ptr, lg, 0) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 64, 1024])
def test_split_t_vs_the_reference_sequence(env, n):
    """zkmi_plonk_split_t_dev == the slices, setCoef and blinding of round 3 (plonk_prove.js:649-672)"""
    zkmi, plonk, f, cx = env
    L, r = zkmi.lib(), cx.r
    t = _rand(31 + n, 4 * n, r)
    b10, b11 = _rand(37, 2, r)
    dt = _dev(zkmi, cx, t)
    T1, T2, T3 = plonk._Poly(f, n + 1, False), plonk._Poly(f, n + 1, False), plonk._Poly(f, n + 6, False)
    zkmi.check(L.zkmi_plonk_split_t_dev(0, dt.ptr, 4 * n, n, zkmi.ptr(f.mont(b10)), zkmi.ptr(f.mont(b11)), T1.ptr, T2.ptr, T3.ptr))
    w1 = t[:n] + [b10]
    w2 = t[n:2 * n] + [b11]
    w2[0] = (w2[0] - b10) % r
    w3 = t[2 * n:3 * n + 6]
    w3[0] = (w3[0] - b11) % r
    assert _host(cx, T1.buf, n + 1) == w1 and _host(cx, T2.buf, n + 1) == w2 and _host(cx, T3.buf, n + 6) == w3


@pytest.mark.gpu
@pytest.mark.parametrize("dom", [8, 256, 4096])
def test_div_zh_vs_oracle(env, dom):
    zkmi, plonk, f, cx = env
    L, r = zkmi.lib(), cx.r
    # t = q * (X^dom - 1) with deg q < 3*dom - dom... as in computeT: length 4*dom, quotient degree < 3*dom + 6 - ... keep it exact
    q = _rand(77 + dom, 3 * dom - 4, r) + [0] * (dom + 4)
    t = [0] * (4 * dom)
    for i, c in enumerate(q[:3 * dom - 4]):
        t[i] = (t[i] - c) % r
        t[i + dom] = (t[i + dom] + c) % r
    want = P.div_zh(t, dom, 4, r)
    assert want == q
    dt = _dev(zkmi, cx, t)
    zkmi.check(L.zkmi_poly_div_zh_dev(0, dt.ptr, 4 * dom, dom, 4))
    assert _host(cx, dt, 4 * dom) == want
    t[4 * dom - 1] = 5
    dt = _dev(zkmi, cx, t)
    assert L.zkmi_poly_div_zh_dev(0, dt.ptr, 4 * dom, dom, 4) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["plonk_bn128_small", "plonk_bn128_n2048", "plonk_bls12381_small"])
def test_plonk_stages_and_golden_proof(env, golden_dir, tag):
    """Device prover == the reference's seeded proof (sha256 of the proof JSON); plonk_bls12381_small: the PLONK kernels with curve = 1
    (the reference proves PLONK on the curve of the zkey, src/plonk_prove.js:66-75; fixture: oracle/gen_golden.js plonkbls)."""
    zkmi, plonk, f, cx = env
    with open(os.path.join(golden_dir, f"{tag}.json")) as fh:
        g = json.load(fh)
    zkey = open(os.path.join(golden_dir, f"{tag}.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, f"{tag}.wtns"), "rb").read()
    blind = [bytes.fromhex(x) for x in g["blinding_mont"]]
    res = plonk.prove(zkey, wtns, blinding_mont=blind)
    assert res["publicSignals"] == g["publicSignals"]
    assert res["proof"] == g["proof"]
    assert hashlib.sha256(json.dumps(res["proof"], separators=(",", ":")).encode()).hexdigest() == g["proof_sha256"]
    with pytest.raises(ValueError):
        plonk.prove(zkey, wtns[:-32])
    # a resident key proves twice with different blinding; both verify the same public signals and differ as proofs
    key = plonk.PlonkKey(zkey)
    p1 = plonk.prove(key, wtns, blinding_mont=blind)
    p2 = plonk.prove(key, wtns)
    key.release()
    assert p1["proof"] == g["proof"] and p2["proof"] != g["proof"] and p2["publicSignals"] == g["publicSignals"]


@pytest.mark.gpu
@pytest.mark.parametrize("t29", ["1", "2", "0"])
def test_compute_z_and_t_stages_vs_oracle(env, golden_dir, t29, monkeypatch):
    """computeZ / computeT kernels against the oracle's intermediate arrays on the n = 2048 fixture; computeT in its three builds (ZKMI_PLONK_T29: 29-bit
    limbs with the products inlined / behind calls, 32-bit limbs — csrc/plonk.hip reads the variable at every call)."""
    monkeypatch.setenv("ZKMI_PLONK_T29", t29)
    zkmi, plonk, f, cx = env
    L = zkmi.lib()
    tag = "plonk_bn128_n2048"
    with open(os.path.join(golden_dir, f"{tag}.json")) as fh:
        g = json.load(fh)
    zkey = open(os.path.join(golden_dir, f"{tag}.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, f"{tag}.wtns"), "rb").read()
    st = {}
    P.plonk_prove(zkey, wtns, [bytes.fromhex(x) for x in g["blinding_mont"]], stages=st)
    key = plonk.PlonkKey(zkey)
    n = key.n
    mont = lambda v: zkmi.ptr(f.mont(v))
    dA, dB, dC = (_dev(zkmi, cx, st[k]) for k in ("A", "B", "C"))
    dZ = zkmi.DeviceBuffer(n * 32)
    zkmi.check(L.zkmi_plonk_compute_z_dev(0, dA.ptr, dB.ptr, dC.ptr, key.sec(12, n), key.sec(12, 6 * n), key.sec(12, 11 * n), n, mont(st["beta"]), mont(st["gamma"]),
                                          mont(key.k1), mont(key.k2), zkmi.ptr(f.root(key.power)), dZ.ptr))
    assert _host(cx, dZ, n) == st["Zb"]
    keep = [_dev(zkmi, cx, st[k]) for k in ("eA", "eB", "eC", "eZ")]
    ev = zkmi.PlonkEvals(keep[0].ptr, keep[1].ptr, keep[2].ptr, keep[3].ptr, key.sec(7, n), key.sec(8, n), key.sec(9, n), key.sec(10, n), key.sec(11, n),
                         key.sec(12, n), key.sec(12, 6 * n), key.sec(12, 11 * n), key.sec(13), dA.ptr)
    dT, dTz = zkmi.DeviceBuffer(4 * n * 32), zkmi.DeviceBuffer(4 * n * 32)
    blind = np.concatenate([np.frombuffer(bytes.fromhex(x), np.uint8) for x in g["blinding_mont"]])
    zkmi.check(L.zkmi_plonk_compute_t_dev(0, C.byref(ev), n, key.nPublic, zkmi.ptr(blind), mont(st["beta"]), mont(st["gamma"]), mont(st["alpha"]), mont(key.k1), mont(key.k2),
                                          zkmi.ptr(f.root(key.power)), zkmi.ptr(f.root(key.power + 2)), zkmi.ptr(f.root(2)), dT.ptr, dTz.ptr))
    assert _host(cx, dT, 4 * n) == st["T"]
    assert _host(cx, dTz, 4 * n) == st["Tz"]
    key.release()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bn128", "bls12381"])
def test_calculate_additions_device_vs_oracle(env, curve):
    """zkmi_plonk_additions_dev == the reference's sequential calculateAdditions loop (oracle/plonk_oracle.py: calculate_additions,
    src/plonk_prove.js:174-204), bit for bit, on (a) the additions section of a synthetic key with > 2^16 additions in chains of depth 3 and
    (b) a random dependency DAG of 70 001 records: operands drawn from the witness, from EARLIER internal signals (near and far, so that lanes
    wait within a wave, across waves and across blocks), one chain of depth 6 000 through consecutive records, and the edge cases of getWitness
    (:207-215): signal 0, ids at / beyond the record's own slot (the reference reads its zero-initialised buffer: 0), ids >= nVars (Fr.zero)."""
    import struct
    import synth_plonk
    zkmi, plonk, f, cx = env
    cid = 0 if curve == "bn128" else 1
    if cid:
        f, cx = plonk._Field(1), P.Ctx(48)
    L, r = zkmi.lib(), cx.r

    def run(add_buf, n_add, wit):
        n_wit = len(wit)
        d_add = zkmi.DeviceBuffer.from_host(np.frombuffer(add_buf, np.uint8))
        d_wit = zkmi.DeviceBuffer.from_host(np.frombuffer(b"".join(v.to_bytes(32, "little") for v in wit), np.uint8))
        d_int = zkmi.DeviceBuffer(32 * n_add)
        zkmi.check(L.zkmi_memset_dev(d_int.ptr, 0xA5, 32 * n_add))                 # stale contents must not leak into a result
        for _ in range(2):                                                        # twice: the ready flags of the first call must not satisfy the second
            zkmi.check(L.zkmi_plonk_additions_dev(cid, d_add.ptr, n_add, d_wit.ptr, n_wit, d_int.ptr))
        zkmi.check(L.zkmi_synchronize())
        got = [int.from_bytes(bytes(x), "little") for x in d_int.to_host(32 * n_add).reshape(n_add, 32)]
        for b in (d_add, d_wit, d_int):
            b.free()
        return got

    # (a) the section a key carries
    zkey, wtns = synth_plonk.make(curve, 17, seed=5, additions=3)
    zk = P.read_plonk_zkey(zkey, cx)
    assert zk["nAdditions"] > 1 << 16
    ws = P.read_sections(wtns)
    o = ws[2][0]
    n_wit = zk["nVars"] - zk["nAdditions"]
    wit = [int.from_bytes(wtns[o + 32 * i:o + 32 * i + 32], "little") for i in range(n_wit)]
    wit[0] = 0
    add_buf = P.sec(zk, 3)
    assert run(add_buf, zk["nAdditions"], wit) == P.calculate_additions(cx, add_buf, zk["nAdditions"], wit, zk["nVars"])

    # (b) random DAG
    rng = np.random.default_rng(0xADD + cid)
    n_wit, n_add = 5000, 70001
    n_vars = n_wit + n_add
    wit = [0] + _rand(0x77 + cid, n_wit - 1, r)
    facs = [bytes(f.mont(v)) for v in [0, 1, r - 1, 2, 12345678901234567890 % r] + _rand(0x78, 27, r)]
    recs = bytearray()
    for i in range(n_add):
        ids = []
        for k in range(2):
            t = int(rng.integers(0, 100))
            if i and 1000 <= i < 7000 and k == 0:
                ids.append(n_wit + i - 1)                                         # one long chain: record i reads record i-1
            elif t < 40 or i == 0:
                ids.append(int(rng.integers(0, n_wit)))                           # a witness signal (incl. signal 0)
            elif t < 70:
                ids.append(n_wit + int(rng.integers(max(0, i - 64), i)))          # a recent internal signal (same wave / block)
            elif t < 94:
                ids.append(n_wit + int(rng.integers(0, i)))                       # any earlier internal signal
            elif t < 97:
                ids.append(n_wit + int(rng.integers(i, n_add)))                   # its own slot or a later one: reads 0
            else:
                ids.append(n_vars + int(rng.integers(0, 1000)))                   # beyond nVars: Fr.zero
        recs += struct.pack("<II", *ids) + facs[int(rng.integers(0, len(facs)))] + facs[int(rng.integers(0, len(facs)))]
    want = P.calculate_additions(cx, bytes(recs), n_add, wit, n_vars)
    assert run(bytes(recs), n_add, wit) == want
    assert len(set(want)) > n_add // 2                                            # the DAG did not collapse to zeros


@pytest.mark.gpu
@pytest.mark.parametrize("curve,lg,depth", [("bn128", 4, 1), ("bn128", 10, 0), ("bn128", 10, 1), ("bn128", 13, 3), ("bls12381", 4, 2), ("bls12381", 10, 1), ("bls12381", 13, 1)])
def test_synthetic_plonk_key_device_vs_oracle(env, curve, lg, depth):
    """A synthetic but VALID key (tests/synth_plonk.py): the device prover must accept it (copy-constraint and divisibility
    checks) and agree with the Python restatement coefficient for coefficient (proof equality). Both curves since r04: k_plonk_t<Bls12381Fr>,
    computeZ's scans and the 4n-point paths on multi-tile BLS12-381 instances (the reference-generated BLS12-381 fixture is n = 64)."""
    import synth_plonk
    zkmi, plonk, f, cx = env
    if curve != "bn128":
        f = plonk._Field(1)
    zkey, wtns = synth_plonk.make(curve, lg, seed=lg, additions=depth)    # `depth` addition gates per multiplication gate, internal signals in chains
    blind = [bytes(f.mont(1000 + 17 * i)) for i in range(11)]
    got = plonk.prove(zkey, wtns, blinding_mont=blind)
    want_proof, want_pub = P.plonk_prove(zkey, wtns, blind)
    assert got["publicSignals"] == want_pub and got["proof"] == want_proof
    # a corrupted witness must be rejected by the copy-constraint check, like the reference does (:437-439)
    bad = bytearray(wtns)
    bad[-32] ^= 1
    with pytest.raises(Exception) as ei:
        plonk.prove(zkey, bytes(bad), blinding_mont=blind)
    assert "Copy constraints does not match" in str(ei.value) or "not divisible" in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("n,length", [(2, 4096), (3, 1000), (8, 70000), (64, 5000), (200, 1000), (4, 8)])
def test_div_by_zerofier_general_n(env, n, length):
    """divByZerofier(n, beta) for n > 1 (FFLONK's use, polynomial.js:617-674): p = q * (X^n - beta)"""
    zkmi, plonk, f, cx = env
    L, r = zkmi.lib(), cx.r
    beta = _rand(5, 1, r)[0]
    q = _rand(100 + n, length - n, r)
    p = [0] * length
    for i, c in enumerate(q):
        p[i] = (p[i] - beta * c) % r
        p[i + n] = (p[i + n] + c) % r
    want = P.div_by_zerofier(p, n, beta, r)
    assert want[:length - n] == q and not any(want[length - n:])
    dp = _dev(zkmi, cx, p)
    zkmi.check(L.zkmi_poly_div_by_zerofier_dev(0, dp.ptr, length, n, zkmi.ptr(f.mont(beta))))
    assert _host(cx, dp, length) == want
    p[1] = (p[1] + 3) % r
    dp = _dev(zkmi, cx, p)
    assert L.zkmi_poly_div_by_zerofier_dev(0, dp.ptr, length, n, zkmi.ptr(f.mont(beta))) != 0


@pytest.mark.gpu
def test_cpoly_interleave(env):
    """CPolynomial.getPolynomial (cpolynomial.js:53-73)"""
    zkmi, plonk, f, cx = env
    L, r = zkmi.lib(), cx.r
    polys = [_rand(40 + j, ln, r) if ln else None for j, ln in enumerate((100, 0, 37, 100))]
    n, out_len = 4, 512
    bufs = [_dev(zkmi, cx, p) if p else None for p in polys]
    ptrs = (C.c_void_p * n)(*[b.ptr if b else None for b in bufs])
    lens = (C.c_size_t * n)(*[len(p) if p else 0 for p in polys])
    out = zkmi.DeviceBuffer(out_len * 32)
    zkmi.check(L.zkmi_cpoly_interleave_dev(0, ptrs, lens, n, out.ptr, out_len))
    want = [0] * out_len
    for j, p in enumerate(polys):
        for i, c in enumerate(p or []):
            if i * n + j < out_len:
                want[i * n + j] = c
    assert _host(cx, out, out_len) == want


@pytest.mark.gpu
@pytest.mark.parametrize("n,deg", [(1, 0), (1000, 999), (1000, 17), (70000, 65536), (4096, 0), (300, None), (1000, 63), (1000, 64), (257, 256), (256, 255), (65, 64), (300000, 1)])
def test_poly_degree(env, n, deg):
    """Polynomial.degree (polynomial.js:165-172): index of the highest non-zero coefficient, 0 for the zero polynomial"""
    zkmi, plonk, f, cx = env
    p = [0] * n
    if deg is not None:
        for i, v in enumerate(_rand(9, deg + 1, cx.r)):
            p[i] = v
        p[deg] = p[deg] or 1
    d, dp = C.c_size_t(77), _dev(zkmi, cx, p)
    zkmi.check(zkmi.lib().zkmi_poly_degree_dev(0, dp.ptr, n, C.byref(d)))
    assert d.value == (deg or 0) == P.degree(p)


@pytest.mark.gpu
def test_fflonk_quotient_kernels_vs_oracle(env, golden_dir):
    """k_fflonk_t0/t1/t2 over the extended evaluation points == the restatement's T0, T1, T1z, T2, T2z (fflonk_prove.js:415-815)"""
    import fflonk_oracle as FF
    zkmi, plonk, f, cx = env
    L = zkmi.lib()
    tag = "fflonk_bn128_n256"
    g = json.load(open(os.path.join(golden_dir, f"{tag}.json")))
    zkey = open(os.path.join(golden_dir, f"{tag}.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, f"{tag}.wtns"), "rb").read()
    blind = [bytes.fromhex(x) for x in g["blinding_mont"]]
    st = {}
    FF.fflonk_prove(zkey, wtns, blind, stages=st)
    from snarkjs_amd import fflonk
    key = fflonk.FflonkKey(zkey)
    n = key.n
    dv = {k: _dev(zkmi, cx, st[k]) for k in ("A", "eA", "eB", "eC", "eZ")}
    ev = zkmi.PlonkEvals(dv["eA"].ptr, dv["eB"].ptr, dv["eC"].ptr, dv["eZ"].ptr, key.sec(9, n), key.sec(7, n), key.sec(8, n), key.sec(10, n), key.sec(11, n),
                         key.sec(12, n), key.sec(13, n), key.sec(14, n), key.sec(15), dv["A"].ptr)
    t0 = zkmi.DeviceBuffer(4 * n * 32)
    zkmi.check(L.zkmi_fflonk_t0_dev(0, C.byref(ev), n, key.nPublic, t0.ptr))
    assert _host(cx, t0, 4 * n) == st["T0"]
    b = [f.unmont(x) for x in blind]
    b789 = np.concatenate([f.mont(b[6]), f.mont(b[7]), f.mont(b[8])])
    t1, t1z = zkmi.DeviceBuffer(2 * n * 32), zkmi.DeviceBuffer(2 * n * 32)
    zkmi.check(L.zkmi_fflonk_t1_dev(0, dv["eZ"].ptr, key.sec(15), n, zkmi.ptr(b789), zkmi.ptr(f.root(key.power + 1)), t1.ptr, t1z.ptr))
    assert _host(cx, t1, 2 * n) == st["T1"] and _host(cx, t1z, 2 * n) == st["T1z"]
    t2, t2z = zkmi.DeviceBuffer(4 * n * 32), zkmi.DeviceBuffer(4 * n * 32)
    mp = lambda v: zkmi.ptr(f.mont(v))
    zkmi.check(L.zkmi_fflonk_t2_dev(0, C.byref(ev), n, zkmi.ptr(b789), mp(st["beta"]), mp(st["gamma"]), mp(key.k1), mp(key.k2), zkmi.ptr(f.root(key.power)),
                                    zkmi.ptr(f.root(key.power + 2)), t2.ptr, t2z.ptr))
    assert _host(cx, t2, 4 * n) == st["T2"] and _host(cx, t2z, 4 * n) == st["T2z"]
    key.release()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["fflonk_bn128_small", "fflonk_bn128_n256"])
def test_fflonk_golden_proof(env, golden_dir, tag):
    """Device FFLONK prover == the reference's seeded proof (sha256 of the proof JSON), and == the oracle on fresh blinding."""
    import fflonk_oracle as FF
    from snarkjs_amd import fflonk
    zkmi, plonk, f, cx = env
    g = json.load(open(os.path.join(golden_dir, f"{tag}.json")))
    zkey = open(os.path.join(golden_dir, f"{tag}.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, f"{tag}.wtns"), "rb").read()
    blind = [bytes.fromhex(x) for x in g["blinding_mont"]]
    res = fflonk.prove(zkey, wtns, blinding_mont=blind)
    assert res["publicSignals"] == g["publicSignals"]
    assert res["proof"] == g["proof"]
    assert hashlib.sha256(json.dumps(res["proof"], separators=(",", ":")).encode()).hexdigest() == g["proof_sha256"]
    with pytest.raises(ValueError):
        fflonk.prove(zkey, wtns[:-32])
    with pytest.raises(ValueError):
        fflonk.prove(open(os.path.join(golden_dir, "plonk_bn128_small.zkey"), "rb").read(), wtns)
    key = fflonk.FflonkKey(zkey)
    blind2 = [bytes(f.mont(v)) for v in _rand(31, 9, cx.r)]
    p2 = fflonk.prove(key, wtns, blinding_mont=blind2)
    want, _ = FF.fflonk_prove(zkey, wtns, blind2)
    assert p2["proof"] == want and p2["proof"] != g["proof"]
    p3 = fflonk.prove(key, wtns)
    key.release()
    assert p3["publicSignals"] == g["publicSignals"] and p3["proof"]["polynomials"]["C1"] != g["proof"]["polynomials"]["C1"]


@pytest.mark.gpu
@pytest.mark.parametrize("lg", [4, 10, 12])
def test_synthetic_fflonk_key_device_vs_oracle(env, lg):
    """Synthetic VALID FFLONK key (tests/synth_plonk.make_fflonk): device prover == the Python restatement, proof for proof."""
    import fflonk_oracle as FF
    import synth_plonk
    from snarkjs_amd import fflonk
    zkmi, plonk, f, cx = env
    zkey, wtns = synth_plonk.make_fflonk(lg, seed=lg)
    blind = [bytes(f.mont(2000 + 13 * i)) for i in range(9)]
    got = fflonk.prove(zkey, wtns, blinding_mont=blind)
    want_proof, want_pub = FF.fflonk_prove(zkey, wtns, blind)
    assert got["publicSignals"] == want_pub and got["proof"] == want_proof
    bad = bytearray(wtns)
    bad[-32] ^= 1
    with pytest.raises(Exception) as ei:
        fflonk.prove(zkey, bytes(bad), blinding_mont=blind)
    assert "Copy constraints does not match" in str(ei.value) or "not divisible" in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("lg", [10, 16, 18])
def test_fflonk_large_proof_verifies(env, lg):
    """2^16 / 2^18 constraints (MSMs over up to 2^22 coefficients): the device proof VERIFIES. The verifier is the restatement of
    src/fflonk_verify.js pinned to the reference's own verifier (test_fflonk_verifier_trace); the synthetic SRS has a known toy
    tau, so the final pairing is the G1 identity A1 == tau * W2 (oracle/fflonk_verify_oracle.py)."""
    import fflonk_verify_oracle as V
    import synth_plonk
    from snarkjs_amd import fflonk
    zkmi, plonk, f, cx = env
    tau = 0x1F3D5B79
    zkey, wtns = synth_plonk.make_fflonk(lg, seed=5, tau=tau)
    key = fflonk.FflonkKey(zkey)
    p1 = fflonk.prove(key, wtns)
    p2 = fflonk.prove(key, wtns)
    key.release()
    vk = V.vk_from_zkey(zkey)
    assert V.verify_known_tau(vk, p1["publicSignals"], p1["proof"], tau)
    assert V.verify_known_tau(vk, p2["publicSignals"], p2["proof"], tau)
    assert p1["proof"]["polynomials"]["C1"] != p2["proof"]["polynomials"]["C1"]           # fresh blinding every time
    # the check rejects a wrong tau, a tampered evaluation and a tampered public signal
    assert not V.verify_known_tau(vk, p1["publicSignals"], p1["proof"], tau + 1)
    bad = {"polynomials": p1["proof"]["polynomials"], "evaluations": dict(p1["proof"]["evaluations"])}
    bad["evaluations"]["a"] = str((int(bad["evaluations"]["a"]) + 1) % cx.r)
    assert not V.verify_known_tau(vk, p1["publicSignals"], bad, tau)
    assert not V.verify_known_tau(vk, [str((int(p1["publicSignals"][0]) + 1) % cx.r)], p1["proof"], tau)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,lg", [("plonk_bn128_n2048", None), ("plonk_bls12381_small", None), (None, 13), (None, 16), (None, -13)])
def test_plonk_two_proofs_in_flight_equal_serial(env, golden_dir, tag, lg):
    """plonk.prove_many (two coroutine proofs on the library's two pipeline slots, one host thread): the same proofs, bit for bit, as one
    plonk.prove after the other with the same blinding values — golden fixture first in the list; an odd number of proofs; a corrupted witness
    in the middle fails the whole call with the reference's message and leaves the library ready for the next proof."""
    import synth_plonk
    zkmi, plonk, f, cx = env
    if tag:
        with open(os.path.join(golden_dir, f"{tag}.json")) as fh:
            g = json.load(fh)
        zkey = open(os.path.join(golden_dir, f"{tag}.zkey"), "rb").read()
        wtns = open(os.path.join(golden_dir, f"{tag}.wtns"), "rb").read()
        blinds = [[bytes.fromhex(x) for x in g["blinding_mont"]]]
        fld = plonk.PlonkKey(zkey)
        f2 = fld.f
        fld.release()
    else:
        curve = "bn128" if lg > 0 else "bls12381"               # a negative size = the same on BLS12-381 (r04)
        lg = abs(lg)
        zkey, wtns = synth_plonk.make(curve, lg, seed=40 + lg)
        blinds, f2, g = [], (f if curve == "bn128" else plonk._Field(1)), None
    for k in range(len(blinds), 5):
        blinds.append([bytes(f2.mont(7000 + 131 * k + 17 * i)) for i in range(11)])
    key = plonk.PlonkKey(zkey)
    serial = [plonk.prove(key, wtns, blinding_mont=b) for b in blinds]
    many = plonk.prove_many(key, [wtns] * len(blinds), blinding_monts=blinds)
    assert [m["proof"] for m in many] == [s_["proof"] for s_ in serial]
    wres = plonk.PlonkWitness(key, wtns)                       # the witness resident on the device, shared by the proofs of both slots
    assert [m["proof"] for m in plonk.prove_many(key, [wres] * len(blinds), blinding_monts=blinds)] == [s_["proof"] for s_ in serial]
    assert plonk.prove(key, wres, blinding_mont=blinds[2])["proof"] == serial[2]["proof"]
    wres.release()
    assert all(m["publicSignals"] == serial[0]["publicSignals"] for m in many)
    if g:
        assert many[0]["proof"] == g["proof"]
    assert zkmi.lib().zkmi_pipeline_active() == 0
    bad = bytearray(wtns)
    bad[-32] ^= 1
    with pytest.raises(Exception) as ei:
        plonk.prove_many(key, [wtns, bytes(bad), wtns, wtns], blinding_monts=blinds[:4])
    assert "Copy constraints does not match" in str(ei.value) or "not divisible" in str(ei.value) or "not well calculated" in str(ei.value)
    assert zkmi.lib().zkmi_pipeline_active() == 0
    again = plonk.prove_many(key, [wtns] * 3, blinding_monts=blinds[:3])
    assert [m["proof"] for m in again] == [s_["proof"] for s_ in serial[:3]]
    assert plonk.prove(key, wtns, blinding_mont=blinds[1])["proof"] == serial[1]["proof"]
    key.release()


@pytest.mark.gpu
@pytest.mark.parametrize("curve,lg", [("bn128", 12), ("bn128", 20), ("bls12381", 16)])
def test_plonk_full_size_proof_verifies(env, curve, lg):
    """BASELINE configs[3] at its full size (2^20 constraints): the device proof VERIFIES.  The verifier is the restatement of
    src/plonk_verify.js pinned to the reference's own verifier trace (test_plonk_verifier_trace, both curves); the synthetic key has a known
    toy tau, so the final pairing is the G1 identity B1 == tau * A1 (oracle/plonk_verify_oracle.py). r04: a 2^16 proof on BLS12-381."""
    import plonk_oracle as PO
    import plonk_verify_oracle as V
    import synth_plonk
    zkmi, plonk, f, cx = env
    if curve != "bn128":
        cx = PO.Ctx(48)
    tau = 0x1F3D5B79
    zkey, wtns = synth_plonk.make(curve, lg, seed=11, tau=tau)
    res = plonk.prove(zkey, wtns)
    vk = V.vk_from_zkey(zkey)
    assert vk["curve"] == curve
    assert V.verify_known_tau(vk, res["publicSignals"], res["proof"], tau)
    # soundness of the check itself: a wrong tau, a tampered evaluation and a tampered public signal must all be rejected
    assert not V.verify_known_tau(vk, res["publicSignals"], res["proof"], tau + 1)
    bad = dict(res["proof"]); bad["eval_a"] = str((int(bad["eval_a"]) + 1) % cx.r)
    assert not V.verify_known_tau(vk, res["publicSignals"], bad, tau)
    assert not V.verify_known_tau(vk, [str((int(res["publicSignals"][0]) + 1) % cx.r)], res["proof"], tau)


@pytest.mark.gpu
@pytest.mark.parametrize("proto,lg", [("plonk", 14), ("fflonk", 12)])
def test_device_prover_equals_the_reference_wasm_prover_on_this_box(env, proto, lg):
    """The reference's OWN prover (snarkjs bundle staged in oracle/_ref: WASM + worker threads, single-threaded JS loops) on the box's host cores
    against the device-resident prover, on a synthetic valid key with addition gates, same blinding draws: proof JSON and public signals byte for
    byte. Beyond the reference-generated fixtures (n <= 2048) this is the largest size a PLONK / FFLONK proof is compared with the reference itself
    inside the suite (a reference plonk.prove takes ~1 s per 2^10 rows; tools/lab/r5_plonk_vs_ref.py runs 2^16 / 2^18 as a one-off,
    profiles/r05_plonk_vs_reference.txt)."""
    import hashlib
    import json
    import shutil
    import subprocess
    import tempfile
    import synth_plonk
    zkmi, plonk, f, cx = env
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    node, bundle = shutil.which("node"), os.path.join(root, "oracle", "_ref", "build", "snarkjs.min.js")
    if node is None or not os.path.exists(bundle):
        pytest.skip("NOT CHECKED ON THIS BOX: node or oracle/_ref is absent — `make -C oracle _ref` stages the reference's bundle in the build container and gpurun ships it "
                    "(profiles/r05_plonk_vs_reference.txt holds the one-off runs up to 2^20)")
    if proto == "plonk":
        zkey, wtns = synth_plonk.make("bn128", lg, seed=21, additions=2)
        n_draws, mod = 11, plonk
    else:
        from snarkjs_amd import fflonk as mod
        zkey, wtns = synth_plonk.make_fflonk(lg, seed=21, additions=2)
        n_draws = 9
    blind = [bytes(f.mont(77000 + 131 * i)) for i in range(n_draws)]
    got = mod.prove(zkey, wtns, blinding_mont=blind)
    with tempfile.TemporaryDirectory() as td:
        zf, wf = os.path.join(td, "k.zkey"), os.path.join(td, "k.wtns")
        open(zf, "wb").write(zkey)
        open(wf, "wb").write(wtns)
        r = subprocess.run([node, "--harmony-optional-chaining", "--harmony-nullish", "--max-old-space-size=16000", os.path.join(root, "tools", "ref_wasm_same_box.js"),
                            proto, zf, wf, ",".join(b.hex() for b in blind)], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, NTHREADS="16", SNARKJS_REF_BUNDLE=bundle, VERIFY="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["draws_used"] == n_draws and d["verified"] is True
    sha = lambda o: hashlib.sha256(json.dumps(o, separators=(",", ":")).encode()).hexdigest()
    assert sha(got["proof"]) == d["proof_json_sha256"], "device proof differs from the reference's WASM proof for the same draws"
    assert sha(got["publicSignals"]) == d["public_signals_sha256"]


@pytest.mark.gpu
def test_fflonk_prove_many_equals_prove(env, golden_dir):
    """fflonk.prove_many (r06: two coroutine proofs on the library's two pipeline slots, commitments as enqueue + collect) == fflonk.prove, proof by proof; the golden proof
    first; a bad witness in the middle fails the call and leaves the library usable."""
    import synth_plonk
    from snarkjs_amd import fflonk
    zkmi, plonk, f, cx = env
    g = json.load(open(os.path.join(golden_dir, "fflonk_bn128_n256.json")))
    zkey = open(os.path.join(golden_dir, "fflonk_bn128_n256.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, "fflonk_bn128_n256.wtns"), "rb").read()
    key = fflonk.FflonkKey(zkey)
    blinds = [[bytes.fromhex(x) for x in g["blinding_mont"]]] + [[bytes(f.mont(5000 + 97 * k + 11 * i)) for i in range(9)] for k in range(1, 5)]
    serial = [fflonk.prove(key, wtns, blinding_mont=b) for b in blinds]
    many = fflonk.prove_many(key, [wtns] * 5, blinds)
    assert many == serial and many[0]["proof"] == g["proof"]
    bad = bytearray(wtns)
    bad[-32] ^= 1
    with pytest.raises(Exception) as ei:
        fflonk.prove_many(key, [wtns, bytes(bad), wtns, wtns], blinds[:4])
    assert "Copy constraints does not match" in str(ei.value) or "not divisible" in str(ei.value) or "not well calculated" in str(ei.value)
    assert fflonk.prove_many(key, [wtns] * 3, blinds[:3]) == serial[:3] and fflonk.prove(key, wtns, blinding_mont=blinds[1]) == serial[1]
    key.release()
    zk2, wt2 = synth_plonk.make_fflonk(12, seed=12)
    b2 = [[bytes(f.mont(300 + 7 * k + i)) for i in range(9)] for k in range(3)]
    assert fflonk.prove_many(zk2, [wt2] * 3, b2) == [fflonk.prove(zk2, wt2, blinding_mont=b) for b in b2]
