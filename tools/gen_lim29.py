#!/usr/bin/env python3
"""Generates the Lim29<C> constant blocks of snarkjs_amd/csrc/field29.cuh (unsaturated-limb Montgomery forms).

For a modulus p stored as N 32-bit words, the unsaturated form has NL limbs of B bits, Montgomery factor R' = 2^(B*NL).
usage: python tools/gen_lim29.py   (prints the C++ specialisations; paste-checked by tools/field29_hosttest)"""
MODS = {
    "Bn254Fq": (0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47, 8, 9, 29),
    "Bn254Fr": (0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001, 8, 9, 29),
    "Bls12381Fr": (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, 8, 9, 29),
    "Bls12381Fq": (0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 12, 14, 28),
}


def limbs(v, nl, b):
    out = []
    for i in range(nl):
        out.append(v & ((1 << b) - 1) if i < nl - 1 else v)
        v >>= b
    return out


def arr(v, nl, b):
    return "{" + ", ".join("0x%08xu" % x for x in limbs(v, nl, b)) + "}"


for name, (p, n, nl, b) in MODS.items():
    rp = 1 << (b * nl)
    npv = (-pow(p, -1, 1 << b)) % (1 << b)
    pinv = pow(p, -1, 1 << b)
    one = rp % p
    kin = pow(2, 2 * b * nl - 32 * n, p)
    kout = pow(2, 32 * n, p)
    print(f"template <> struct Lim29<{name}> {{")
    print(f"    static constexpr int NL = {nl}, B = {b};                  // R' = 2^{b*nl}; R'/p = {rp / p:.1f}")
    print(f"    static constexpr uint32_t NP = 0x{npv:08x}u;                    // -p^-1 mod 2^{b}")
    print(f"    static constexpr uint32_t PINV = 0x{pinv:08x}u;                  //  p^-1 mod 2^{b}")
    print(f"    ZK_HD static constexpr uint32_t p(int i) {{ constexpr uint32_t v[{nl}] = {arr(p, nl, b)}; return v[i]; }}")
    print(f"    ZK_HD static constexpr uint32_t one(int i) {{ constexpr uint32_t v[{nl}] = {arr(one, nl, b)}; return v[i]; }}     // 2^{b*nl} mod p")
    print(f"    ZK_HD static constexpr uint32_t kin(int i) {{ constexpr uint32_t v[{nl}] = {arr(kin, nl, b)}; return v[i]; }}     // 2^{2*b*nl-32*n} mod p: R-form -> R'-form")
    print(f"    ZK_HD static constexpr uint32_t kout(int i) {{ constexpr uint32_t v[{nl}] = {arr(kout, nl, b)}; return v[i]; }}    // 2^{32*n} mod p: R'-form -> R-form")
    print("};")
