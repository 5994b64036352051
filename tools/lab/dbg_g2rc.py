"""r03 debugging: G2 MSM over a resident table with the Fq2 row/column sums on unsaturated limbs, small cases against the oracle"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O, synth
from snarkjs_amd import zkmi
zkmi.init()
L = zkmi.lib()
name, group = sys.argv[1], 2
c = O.CURVE_ID[name]
q8 = O.n8q(c)
for lg in (13, 15):
    n = 1 << lg
    bases = O.geom_bases(c, group, n)
    d_b = zkmi.DeviceBuffer.from_host(bases)
    h = C.c_uint64(0)
    zkmi.check(L.zkmi_msm_table_build(c, group, d_b.ptr, n, C.byref(h)))
    pats = {}
    one = np.zeros(n * 32, np.uint8); one[0] = 1
    pats["single 1"] = one
    two = np.zeros(n * 32, np.uint8); two[0] = 1; two[32] = 1
    pats["two ones"] = two
    k64 = np.zeros(n * 32, np.uint8); k64[0] = 65
    pats["single 65"] = k64
    ones = np.zeros(n * 32, np.uint8); ones[::32] = 1
    pats["all ones"] = ones
    small = np.zeros(n * 32, np.uint8); small[::32] = np.arange(n) % 251; small[1::32] = (np.arange(n) * 7) % 13
    pats["small"] = small
    pats["random"] = synth.elems(0xD6 + lg, n)
    for nm, sc in pats.items():
        want = O.to_affine(c, group, O.msm(c, group, bases, sc, n, 32))
        res = []
        for rep in range(3):
            d_s = zkmi.DeviceBuffer.from_host(sc)
            out = np.zeros(3 * group * q8, np.uint8)
            zkmi.check(L.zkmi_msm_table_dev(h, d_s.ptr, n, 32, zkmi.ptr(out)))
            res.append(bool(np.array_equal(O.to_affine(c, group, out), want)))
        print(name, "2^%d" % lg, nm, res, flush=True)
    zkmi.check(L.zkmi_msm_table_release(h))
