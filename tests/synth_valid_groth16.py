"""Synthetic VALID Groth16 proving key from a known trapdoor (replaces "the proof does not verify but is
bit-comparable" of tests/synth_zkey.py): every section is what src/zkey_new.js would write for this circuit from a ptau whose tau,
alpha, beta and whose delta contribution are the given values, so that a correct proof VERIFIES (oracle/groth16_verify_oracle.py; the
reference's own verifier for the committed small instance, tests/golden/groth16_valid_synth_*.json).

Circuit = the reference's own test circuit family (test/groth16/circuit.circom, Multiplier(N)): x_0 = a*a + b, x_i = x_{i-1}^2 + b,
public output c = x_{N-1}, public input a. Signals [1, c, a, b, x_0 .. x_{N-2}], nPublic = 2, N = n - 4 constraints on the domain
n = 2^lg:     constraint i:  (x_{i-1}) * (x_{i-1}) = x_i - b          (x_{-1} = a)

Sections (src/zkey_utils.js:229-259 / src/zkey_new.js):
  header   alpha1 = alpha G1, beta1/beta2 = beta G, gamma2 = gamma G2, delta1/delta2 = delta G             (:338-371)
  IC       IC_s = ((beta u_s + alpha v_s + w_s)(tau) / gamma) G1, s <= nPublic                            (:372-407 with the binding rows)
  coeffs   A and B matrix records (value x R^2) + the nPublic+1 binding rows A[N+s][s] = 1                 (:290-333)
  A, B1, B2  u_i(tau) G1, v_i(tau) G1, v_i(tau) G2                                                         (:409-470)
  C        ((beta u_i + alpha v_i + w_i)(tau) / delta) G1, i > nPublic
  H        the odd-indexed points of the 2n-point Lagrange basis at tau, / delta                          (:182-201): H_i = L^(2n)_{2i+1}(tau)/delta G1
u_i, v_i, w_i = the columns of the A, B, C matrices interpolated over the domain, evaluated at tau (Lagrange basis L_c(tau)).
Field vectors are handled by the CPU oracle's element-wise helpers (tests may use the oracle); the points k*G come from the device
(zkmi_gen_bases_from_scalars_dev) or, for tiny keys without a GPU, from the oracle.
"""
import struct

import numpy as np

import oracle_lib as O
from synth_zkey import PRIMES, _binfile

TRAPDOOR = {"tau": 0x1234567890ABCDEF1234567, "alpha": 0xA1FA0001, "beta": 0xBE7A0002, "gamma": 0x6A44A0003, "delta": 0xDE17A0004}


def _mont(r, v):
    return np.frombuffer(((v % r) << 256).__mod__(r).to_bytes(32, "little"), np.uint8).copy()


def _ints(buf):
    a = np.frombuffer(bytes(buf), "<u8").reshape(-1, 4).astype(object)
    return [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 for x in a]


def _rep(elem, n):
    return np.tile(np.asarray(elem, np.uint8), n)


def _points(c, group, scalars_plain, use_device):
    """k_i * G as affine Montgomery bytes; scalars: n x 32 bytes, normal form"""
    n = scalars_plain.size // 32
    q8 = O.n8q(c)
    if use_device:
        from snarkjs_amd import zkmi
        zkmi.init()
        d_s, d_o = zkmi.DeviceBuffer.from_host(scalars_plain), zkmi.DeviceBuffer(n * 2 * group * q8)
        zkmi.check(zkmi.lib().zkmi_gen_bases_from_scalars_dev(c, group, d_s.ptr, n, d_o.ptr))
        out = d_o.to_host()
        d_s.free(); d_o.free()
        return out
    out = np.zeros(n * 2 * group * q8, np.uint8)
    for i, k in enumerate(_ints(scalars_plain)):
        out[i * 2 * group * q8:(i + 1) * 2 * group * q8] = O.to_affine(c, group, O.generator_mul(c, group, k))
    return out


def layout(lg, n_public=2):
    n = 1 << lg
    N = n - 4
    m = N + 3
    sig_x = lambda i: 1 if i == N - 1 else 4 + i             # signal index of x_i
    return n, N, m, sig_x


def witness(name, lg, a=11, b=2):
    q8, q, r = PRIMES[name]
    n, N, m, sig_x = layout(lg)
    xs = [(a * a + b) % r]
    for _ in range(1, N):
        xs.append((xs[-1] * xs[-1] + b) % r)
    sig = [1, xs[N - 1], a, b] + xs[:N - 1]
    w = b"".join(v.to_bytes(32, "little") for v in sig)
    le = lambda v, k: int(v).to_bytes(k, "little")
    return _binfile(b"wtns", [(1, struct.pack("<I", 32) + le(r, 32) + struct.pack("<I", m)), (2, w)])


def make(name, lg, use_device=True, trapdoor=TRAPDOOR):
    """-> (zkey_bytes, wtns_bytes, info) with info = trapdoor-derived data a verifier needs (vk as the reference exports it)."""
    c = O.CURVE_ID[name]
    q8, q, r = PRIMES[name]
    n, N, m, sig_x = layout(lg)
    n_public = 2
    tau, alpha, beta, gamma, delta = (trapdoor[k] % r for k in ("tau", "alpha", "beta", "gamma", "delta"))
    one = O.fr_one(c)
    M = lambda v: _mont(r, v)
    scale = lambda vec, k: O.apply_key(c, vec, M(k), one)                 # vec * k
    geom = lambda cnt, first, ratio: O.apply_key(c, _rep(one, cnt), M(first), M(ratio))
    w_n = int.from_bytes(O.from_mont(c, O.fr_w(c, lg)).tobytes(), "little")
    w_2n = int.from_bytes(O.from_mont(c, O.fr_w(c, lg + 1)).tobytes(), "little")
    # Lagrange basis of the domain at tau: L_c = w^c (tau^n - 1) / (n (tau - w^c))
    wp = geom(n, 1, w_n)
    den = scale(O.vec_op(c, "sub", _rep(M(tau), n), wp), n)
    L = scale(O.vec_op(c, "mul", wp, O.batch_inverse(c, den)), (pow(tau, n, r) - 1) % r)
    Lr = L.reshape(n, 32)
    # columns of A, B, C at tau
    prev = np.array([2] + [sig_x(i) for i in range(N - 1)], np.int64)       # signal multiplied in constraint i (A and B rows)
    nxt = np.array([sig_x(i) for i in range(N)], np.int64)                  # signal x_i of the C row
    u = np.zeros((m, 32), np.uint8)
    u[prev] = Lr[:N]
    v = u.copy()
    wv = np.zeros((m, 32), np.uint8)
    wv[nxt] = Lr[:N]
    Lint = None
    sumL = O.dot(c, L[:N * 32], _rep(np.frombuffer((1).to_bytes(32, "little"), np.uint8), N))          # sum_{c<N} L_c (normal form)
    wv[3] = M(-sumL)                                                        # b enters every C row with coefficient -1
    for s in range(n_public + 1):                                           # binding rows A[N+s][s] = 1 (src/zkey_new.js:290-300)
        cur = int.from_bytes(O.from_mont(c, u[s]).tobytes(), "little")
        add = int.from_bytes(O.from_mont(c, Lr[N + s]).tobytes(), "little")
        u[s] = M(cur + add)
    u, v, wv = u.reshape(-1), v.reshape(-1), wv.reshape(-1)
    comb = O.vec_op(c, "add", O.vec_op(c, "add", scale(u, beta), scale(v, alpha)), wv)       # beta u + alpha v + w
    c_scal = scale(comb, pow(delta, -1, r))
    ic_scal = scale(comb[:(n_public + 1) * 32], pow(gamma, -1, r))
    # H_i = L^(2n)_{2i+1}(tau) / delta,  L^(2n)_j = w2n^j (tau^2n - 1) / (2n (tau - w2n^j))
    cp = geom(n, w_2n, w_n)
    den = scale(O.vec_op(c, "sub", _rep(M(tau), n), cp), 2 * n * delta)
    h_scal = scale(O.vec_op(c, "mul", cp, O.batch_inverse(c, den)), (pow(tau, 2 * n, r) - 1) % r)
    plain = lambda vec: O.from_mont(c, vec)
    A = _points(c, 1, plain(u), use_device)
    B1 = _points(c, 1, plain(v), use_device)
    B2 = _points(c, 2, plain(v), use_device)
    Cb = _points(c, 1, plain(c_scal[(n_public + 1) * 32:]), use_device)
    H = _points(c, 1, plain(h_scal), use_device)
    IC = _points(c, 1, plain(ic_scal), False)
    pt = lambda grp, k: O.to_affine(c, grp, O.generator_mul(c, grp, k))
    alpha1, beta1, beta2, gamma2, delta1, delta2 = pt(1, alpha), pt(1, beta), pt(2, beta), pt(2, gamma), pt(1, delta), pt(2, delta)
    # coefficient section: value 1 stored as R^2 (src/zkey_utils.js:174-179)
    mm = np.concatenate([np.zeros(N, "<u4"), np.ones(N, "<u4"), np.zeros(n_public + 1, "<u4")])
    cc = np.concatenate([np.arange(N, dtype="<u4"), np.arange(N, dtype="<u4"), N + np.arange(n_public + 1, dtype="<u4")])
    ss = np.concatenate([prev.astype("<u4"), prev.astype("<u4"), np.arange(n_public + 1, dtype="<u4")])
    ncoef = mm.size
    rec = np.zeros(ncoef, dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "u1", 32)])
    rec["m"], rec["c"], rec["s"] = mm, cc, ss
    rec["v"] = np.frombuffer(pow(2, 512, r).to_bytes(32, "little"), np.uint8)
    coeffs = struct.pack("<I", ncoef) + rec.tobytes()
    le = lambda val, k: int(val).to_bytes(k, "little")
    hdr = (struct.pack("<I", q8) + le(q, q8) + struct.pack("<I", 32) + le(r, 32) + struct.pack("<III", m, n_public, n)
           + alpha1.tobytes() + beta1.tobytes() + beta2.tobytes() + gamma2.tobytes() + delta1.tobytes() + delta2.tobytes())
    contributions = bytes(64) + struct.pack("<I", 0)                        # csHash + nContributions (src/zkey_utils.js:261-279)
    zkey = _binfile(b"zkey", [(1, struct.pack("<I", 1)), (2, hdr), (3, IC.tobytes()), (4, coeffs), (5, A.tobytes()), (6, B1.tobytes()),
                              (7, B2.tobytes()), (8, Cb.tobytes()), (9, H.tobytes()), (10, contributions)])
    # verification key as zKey.exportVerificationKey writes it (src/zkey_export_verificationkey.js): normal-form decimal strings
    def g1_obj(p):
        x, y = (int.from_bytes(O.fq_from_mont(c, p[i * q8:(i + 1) * q8]).tobytes(), "little") for i in range(2))
        return [str(x), str(y), "1"]

    def g2_obj(p):
        vals = [int.from_bytes(O.fq_from_mont(c, p[i * q8:(i + 1) * q8]).tobytes(), "little") for i in range(4)]
        return [[str(vals[0]), str(vals[1])], [str(vals[2]), str(vals[3])], ["1", "0"]]
    vk = {"protocol": "groth16", "curve": name, "nPublic": n_public, "vk_alpha_1": g1_obj(alpha1), "vk_beta_2": g2_obj(beta2), "vk_gamma_2": g2_obj(gamma2),
          "vk_delta_2": g2_obj(delta2), "IC": [g1_obj(IC[i * 2 * q8:(i + 1) * 2 * q8]) for i in range(n_public + 1)]}
    return zkey, witness(name, lg), {"vk": vk, "n": n, "m": m, "N": N}
