"""GPU parity of the two r06 pieces of the fused Groth16 boundary:

* buildABC1 (src/groth16_prove.js:147-187) as a row-balanced sparse product: keys whose coefficient rows are shaped like a compiled circuit's
  (heavy tail up to a 10^5-term row) against the CPU oracle's literal loop, bit for bit;
* the proving key handed over as PAGES (zkmi_groth16_load_paged / _load_shard_paged: what the reference holds after readSection of a section
  beyond 2^30 bytes, src/groth16_prove.js:57-59), with gaps where a shard loader did not read.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import torch
    torch.cuda.init()
    import snarkjs_amd
    from snarkjs_amd import zkmi
    zkmi.init(0)
    return snarkjs_amd


def _abc_equal(pk, zkd, w, c):
    A, B, Cc = O.build_abc(c, zkd["coeffs"], w, zkd["nVars"], zkd["domainSize"])
    gA, gB, gC = pk.build_abc(w)
    assert np.array_equal(gA, A.reshape(-1)), "A_T differs"
    assert np.array_equal(gB, B.reshape(-1)), "B_T differs"
    assert np.array_equal(gC, Cc.reshape(-1)), "C_T differs"


@pytest.mark.parametrize("name,lg,dist", [("bn128", 10, "flat"), ("bn128", 13, "real"), ("bn128", 17, "real"), ("bls12381", 15, "real")])
def test_build_abc_against_the_oracle(zk, name, lg, dist):
    """A_T, B_T, C_T of zkmi_groth16_build_abc_dev == the oracle's restatement of the reference loop. 2^17 `real` holds ONE ROW OF 10^5 TERMS
    (3 125 segments, one wave adds their partial sums), rows of ~10^4 and ~10^3 terms and 92 % single-term rows; empty rows write the zero."""
    import synth_zkey
    from snarkjs_amd import groth16, binfile
    c = O.CURVE_ID[name]
    zkey, wtns = synth_zkey.make(name, lg, seed=0xAB0 + lg, coef_dist=dist, witness="uniform")
    zkd, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)["witness"]
    pk = groth16.ProvingKey(zkey)
    lay = pk.coef_layout()
    assert lay["n_coef"] == (zkd["coeffs"].size - 4) // 44
    if dist == "real" and lg >= 17:
        recs = np.frombuffer(zkd["coeffs"][4:].tobytes(), dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "u1", 32)])
        longest = np.bincount(recs["m"].astype(np.int64) * zkd["domainSize"] + recs["c"]).max()
        assert longest >= 100000 and lay["cut_rows"] > 50 and lay["partial_slots"] >= 3125
    # padding of the sliced layout stays marginal (segments are sorted by length before they are grouped into slices)
    assert lay["padded_terms"] <= lay["n_coef"] + 64 * 33 * 32
    _abc_equal(pk, zkd, w, c)
    # a second witness through the same resident layout (no state between calls), and a whole proof on this key == the oracle's
    w2 = np.roll(w.reshape(-1, 32), 7, axis=0).reshape(-1).copy()
    _abc_equal(pk, zkd, w2, c)
    if lg <= 13:
        r_m, s_m = O.fr_e(c, 11), O.fr_e(c, 13)
        got = pk.prove_raw(w, r_m, s_m)
        want = O.groth16_prove(c, zkd, w, r_m, s_m)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    pk.release()


def test_build_abc_degenerate_rows(zk):
    """Every row shape around the segment width (0, 1, 31, 32, 33, 64, 65 terms), repeated signals inside a row, records in arbitrary order, a
    matrix with no record at all."""
    import struct
    import synth_zkey
    from snarkjs_amd import groth16, binfile
    from snarkjs_amd.workloads import synth
    zkey, wtns = synth_zkey.make("bn128", 8, seed=77, witness="uniform")
    zkd, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)["witness"]
    n, m = zkd["domainSize"], zkd["nVars"]
    for variant in ("both", "only_b", "none"):
        rows = []
        for i, ln in enumerate((0, 1, 31, 32, 33, 64, 65, 200)):
            for mat in ((0, 1) if variant == "both" else (1,) if variant == "only_b" else ()):
                rows += [(mat, 3 * i + mat, (5 * k * (i + 1)) % m) for k in range(ln)]       # signals repeat within the long rows
        rng = np.random.default_rng(5)
        rng.shuffle(rows)
        rec = np.zeros(len(rows), dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "u1", 32)])
        if rows:
            rec["m"], rec["c"], rec["s"] = np.array(rows, dtype=np.uint32).T
            rec["v"] = synth.elems(99, len(rows)).reshape(-1, 32)
        coeffs = np.frombuffer(struct.pack("<I", len(rows)) + rec.tobytes(), np.uint8)
        z2 = _replace_section(zkey, 4, coeffs.tobytes())
        pk = groth16.ProvingKey(z2)
        _abc_equal(pk, binfile.read_groth16_zkey(z2), w, 0)
        pk.release()


def _replace_section(blob, typ, payload):
    """the same container with section `typ` replaced (iden3 binfile: magic, version, nSections, then (type u32, length u64, bytes)*)"""
    import struct
    out, off = [blob[:12]], 12
    nsec = struct.unpack_from("<I", blob, 8)[0]
    for _ in range(nsec):
        t, ln = struct.unpack_from("<IQ", blob, off)
        body = blob[off + 12:off + 12 + ln]
        off += 12 + ln
        if t == typ:
            body = payload
        out.append(struct.pack("<IQ", t, len(body)) + body)
    return b"".join(out)


@pytest.mark.parametrize("name", ["bn128", "bls12381"])
def test_paged_key_equals_flat_key(zk, golden_dir, name):
    """The reference's golden proof through zkmi_groth16_load_paged with 64 KiB pages (sections 4-9 become 2 - 8 pages each) == through the flat load."""
    from snarkjs_amd import groth16, binfile
    c = O.CURVE_ID[name]
    zkey = open(os.path.join(golden_dir, f"groth16_{name}_n1024.zkey"), "rb").read()
    wtns = open(os.path.join(golden_dir, f"groth16_{name}_n1024.wtns"), "rb").read()
    w = binfile.read_wtns(wtns)["witness"]
    r_m, s_m = O.fr_e(c, 3), O.fr_e(c, 5)
    flat = groth16.ProvingKey(zkey)
    want = [bytes(x) for x in flat.prove_raw(w, r_m, s_m)]
    flat.release()
    for page in (65536, 4096 + 44, 1 << 30):                       # odd page size: records and points straddle page borders
        pk = groth16.ProvingKey(zkey, page_bytes=page)
        assert [bytes(x) for x in pk.prove_raw(w, r_m, s_m)] == want, page
        pk.release()


def test_paged_shards_with_gaps(zk):
    """Key shards loaded from pages of which only those that intersect the shard's own byte ranges are PROVIDED (the others are gaps, as left by a
    loader that reads its slice of the file): the folded partial sums give the single-device proof; a load that needs a gap fails cleanly."""
    import ctypes as C
    import synth_zkey
    from snarkjs_amd import groth16, binfile, zkmi
    L = zkmi.lib()
    zkey, wtns = synth_zkey.make("bn128", 12, seed=31, b_zero_every=0)
    w = binfile.read_wtns(wtns)["witness"]
    r_m, s_m = O.fr_e(0, 21), O.fr_e(0, 22)
    full = groth16.ProvingKey(zkey)
    want = [bytes(x) for x in full.prove_raw(w, r_m, s_m)]
    full.release()
    world, q = 3, 32
    tot = None
    keys = []
    for rank in range(world):
        pk = groth16.ProvingKey(zkey, shard=(rank, world), page_bytes=16384, gaps=True)
        keys.append(pk)
        assert any(p is None for p in pk._pages["A"][1]), "this shard was given every page: the test does not exercise gaps"
        s = pk.sums_raw(w)
        if tot is None:
            tot = s.copy()
        else:
            for off, g in ((0, 1), (3 * q, 1), (6 * q, 2), (12 * q, 1), (15 * q, 1)):
                o = np.zeros(3 * g * q, np.uint8)
                zkmi.check(L.zkmi_point_add(0, g, zkmi.ptr(tot[off:off + 3 * g * q].copy()), zkmi.ptr(s[off:off + 3 * g * q].copy()), zkmi.ptr(o)))
                tot[off:off + 3 * g * q] = o
    got = [bytes(x) for x in keys[0].finish_raw(tot, r_m, s_m)]
    assert got == want
    # the FULL key from rank 1's pages needs bytes that lie in gaps
    d = keys[1].desc
    assert L.zkmi_groth16_load_paged(C.byref(d), 0x7001) != 0 and b"gap" in L.zkmi_last_error()
    for pk in keys:
        pk.release()


@pytest.mark.parametrize("name,lg,n_vars,dist", [("bn128", 12, 300, "flat"), ("bn128", 12, 7000, "real"), ("bn128", 14, 40000, "flat"), ("bls12381", 12, 1500, "real"),
                                                 ("bn128", 17, 60000, "flat"), ("bn128", 16, 200000, "real")])
def test_key_shapes_nvars_far_from_domain(zk, name, lg, n_vars, dist):
    """Real circuits have anything from nVars << domainSize to nVars > domainSize (the bench keys have nVars = domainSize - 5): the witness-side tables and the H table then
    differ in size, window width and bucket shape (no C / H bucket sharing, separate reductions). Whole proofs against the oracle; at the larger sizes the shard split too."""
    import synth_zkey
    from snarkjs_amd import groth16, binfile
    from snarkjs_amd import distributed as D
    c = O.CURVE_ID[name]
    zkey, wtns = synth_zkey.make(name, lg, seed=0x51 + lg, n_vars=n_vars, coef_dist=dist, b_zero_every=0 if dist == "flat" else 3)
    zkd, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)["witness"]
    assert zkd["nVars"] == n_vars and zkd["domainSize"] == 1 << lg
    r_m, s_m = O.fr_e(c, 0x77), O.fr_e(c, 0x99)
    want = O.groth16_prove(c, zkd, w, r_m, s_m)
    pk = groth16.ProvingKey(zkey)
    got = pk.prove_raw(w, r_m, s_m)
    pk.submit(zkmi_dev(w), 1)
    two = pk.collect(1, r_m, s_m)
    pk.release()
    for a, b, t in zip(got, want, two):
        assert np.array_equal(a, b) and np.array_equal(t, b)
    if lg >= 16:
        keys = []

        def make(rank):
            keys.append(groth16.ProvingKey(zkey, shard=(rank, 3), page_bytes=1 << 20, gaps=True))
            return D.DeviceShard(keys[-1], w)
        sh = D.groth16_prove_sharded_local(make, 3, r_m, s_m)
        for k in keys:
            k.release()
        assert all(np.array_equal(a, b) for a, b in zip(sh, want))


_held = []


def zkmi_dev(w):
    """the witness in device memory, kept alive for the rest of the module (a submitted proof reads it until it is collected)"""
    from snarkjs_amd import zkmi
    b = zkmi.DeviceBuffer.from_host(zkmi.u8(w))
    _held.append(b)
    return b.ptr


def test_table_msm_enqueue_collect_halves(zk):
    """zkmi_msm_table_multi_enqueue_dev + _collect (r06: the two halves of zkmi_msm_table_multi_dev, for a host that drives two proofs from one thread) == the one-call form;
    on both pipeline slots at once; a collect without an enqueue and a mismatched count fail; an enqueued call that is never collected is dropped by the next enqueue."""
    import ctypes as C
    import synth
    from snarkjs_amd import zkmi
    L = zkmi.lib()
    n, q8 = 1 << 14, 32
    d_b = zkmi.DeviceBuffer(n * 64)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(0, 1, n, 7, 11, d_b.ptr))
    h = C.c_uint64(0)
    zkmi.check(L.zkmi_msm_table_build(0, 1, d_b.ptr, n, C.byref(h)))
    sc = [zkmi.DeviceBuffer.from_host(synth.elems(0x900 + i, n)) for i in range(4)]
    ks = [n, n - 5, 0, 77]

    def arrays(idx):
        return (C.c_void_p * len(idx))(*[sc[i].ptr for i in idx]), (C.c_size_t * len(idx))(*[ks[i] for i in idx])
    want = np.zeros(4 * 96, np.uint8)
    p, k = arrays([0, 1, 2, 3])
    zkmi.check(L.zkmi_msm_table_multi_dev(h, p, k, 4, 32, zkmi.ptr(want)))
    got = np.zeros(4 * 96, np.uint8)
    zkmi.check(L.zkmi_msm_table_multi_enqueue_dev(h, p, k, 4, 32))
    zkmi.check(L.zkmi_msm_table_multi_collect(h, 4, zkmi.ptr(got)))
    aff = lambda j: [bytes(O.to_affine(0, 1, j[i * 96:(i + 1) * 96])) for i in range(len(j) // 96)]
    assert aff(got) == aff(want) and not got[2 * 96:3 * 96].any()
    # both slots hold an enqueued call at the same time
    p01, k01 = arrays([0, 1])
    p3, k3 = arrays([3])
    zkmi.check(L.zkmi_pipeline_select(0)); zkmi.check(L.zkmi_msm_table_multi_enqueue_dev(h, p01, k01, 2, 32))
    zkmi.check(L.zkmi_pipeline_select(1)); zkmi.check(L.zkmi_msm_table_multi_enqueue_dev(h, p3, k3, 1, 32))
    g1, g0 = np.zeros(96, np.uint8), np.zeros(192, np.uint8)
    zkmi.check(L.zkmi_msm_table_multi_collect(h, 1, zkmi.ptr(g1)))
    zkmi.check(L.zkmi_pipeline_select(0)); zkmi.check(L.zkmi_msm_table_multi_collect(h, 2, zkmi.ptr(g0)))
    assert aff(g0) == aff(want)[:2] and aff(g1) == aff(want)[3:]
    # misuse
    assert L.zkmi_msm_table_multi_collect(h, 2, zkmi.ptr(g0)) != 0 and b"nothing enqueued" in L.zkmi_last_error()
    zkmi.check(L.zkmi_msm_table_multi_enqueue_dev(h, p01, k01, 2, 32))
    assert L.zkmi_msm_table_multi_collect(h, 3, zkmi.ptr(got)) != 0 and b"does not match" in L.zkmi_last_error()
    zkmi.check(L.zkmi_msm_table_multi_enqueue_dev(h, p01, k01, 2, 32))          # the mismatched collect left the call in place: dropped here
    zkmi.check(L.zkmi_msm_table_multi_enqueue_dev(h, p3, k3, 1, 32))            # never collected: dropped by this one
    zkmi.check(L.zkmi_msm_table_multi_collect(h, 1, zkmi.ptr(g1)))
    assert aff(g1) == aff(want)[3:]
    zkmi.check(L.zkmi_msm_table_release(h))
    for b in sc:
        b.free()
    d_b.free()
