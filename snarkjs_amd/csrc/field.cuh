// snarkjs_amd/csrc/field.cuh — Montgomery prime-field arithmetic for gfx950 (CDNA4), 32-bit limbs in VGPRs.
//
// Replaces the generated-WASM field kernels of wasmcurves 0.2.2 (f1m_*/frm_* : product-scanning Montgomery on
// 32-bit limbs with i64 accumulators, reference bundle build/snarkjs.min.js:1@36990) — re-designed for the CDNA4
// VALU: the workhorse is v_mad_u64_u32 (32x32+64 -> 64 with carry-out); there is no 64x64 multiplier and MFMA
// does not apply (integer carry chains, not dense contractions).
//
// Representation (identical to what snarkjs/ffjavascript keeps in memory): little-endian limbs,
// Montgomery form x·R mod p with R = 2^(32·N), fully reduced to [0,p).  N = 8 (BN254 Fr/Fq, BLS12-381 Fr) or
// 12 (BLS12-381 Fq).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef ZKMI_MUL_VARIANT
#define ZKMI_MUL_VARIANT 2      // 1 = compiler-scheduled product scanning, 2 = inline-asm MAC, 3 = CIOS
#endif

namespace zkmi {

#define ZK_DEV __device__ __forceinline__
#define ZK_HD __host__ __device__ __forceinline__

// ---- field configurations -------------------------------------------------------------------------------------
// p(i): modulus limbs; NP = -p^{-1} mod 2^32; one(i) = R mod p; r2(i) = R^2 mod p. Values are checked at library
// start-up against a host computation (zkmi_api.cpp: check_constants()).
struct Bn254Fr {
    static constexpr int N = 8;
    static constexpr uint32_t NP = 0xefffffffu;
    ZK_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    ZK_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t v[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    ZK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return v[i];
    }
};
struct Bn254Fq {
    static constexpr int N = 8;
    static constexpr uint32_t NP = 0xe4866389u;
    ZK_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return v[i];
    }
    ZK_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return v[i];
    }
    ZK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return v[i];
    }
};
struct Bls12381Fr {
    static constexpr int N = 8;
    static constexpr uint32_t NP = 0xffffffffu;
    ZK_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return v[i];
    }
    ZK_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t v[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return v[i];
    }
    ZK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return v[i];
    }
};
struct Bls12381Fq {
    static constexpr int N = 12;
    static constexpr uint32_t NP = 0xfffcfffdu;
    ZK_HD static constexpr uint32_t p(int i) {
        constexpr uint32_t v[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return v[i];
    }
    ZK_HD static constexpr uint32_t one(int i) {
        constexpr uint32_t v[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return v[i];
    }
    ZK_HD static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return v[i];
    }
};

template <class C> struct Fp {
    uint32_t l[C::N];
    using Cfg = C;
    static constexpr int N = C::N;
};

// ---- building blocks -------------------------------------------------------------------------------------------
// (acc, hi) += a*b   — 64-bit accumulator + carry counter; exactly 2 VALU instructions.
ZK_DEV void mac(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t b) {
#if ZKMI_MUL_VARIANT == 2
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "v"(b) : "vcc");
#else
    uint64_t pr = (uint64_t)a * b;
    acc += pr;
    hi += (acc < pr) ? 1u : 0u;
#endif
}
// same with the multiplier coming from an SGPR/constant (modulus limbs)
ZK_DEV void mac_c(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t k) {
#if ZKMI_MUL_VARIANT == 2
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "s"(k) : "vcc");
#else
    mac(acc, hi, a, k);
#endif
}

template <class C> ZK_DEV bool fp_is_zero(const Fp<C>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) o |= a.l[i];
    return o == 0;
}
template <class C> ZK_DEV bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}
template <class C> ZK_DEV Fp<C> fp_zero() {
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = 0;
    return r;
}
template <class C> ZK_DEV Fp<C> fp_one() {
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = C::one(i);
    return r;
}
template <class C> ZK_DEV Fp<C> fp_r2() {
    Fp<C> r;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = C::r2(i);
    return r;
}

// Modular add / sub / neg as plain carry chains: __builtin_addc / __builtin_subc lower to v_add_co_u32 + v_addc_co_u32 (one VALU
// instruction per limb). The earlier uint64_t formulation compiled to 64-bit adds, shifts and moves — 56 issue slots per operation
// against ~25 here — which mattered most where additions are dense: an Fq2 mixed addition holds ~70 of them (30 % of its time).
// r = t - p if t >= p (t given with an extra top carry bit `top`), else t
template <class C> ZK_DEV void fp_cond_sub_p(Fp<C>& r, const uint32_t* t, uint32_t top) {
    uint32_t d[C::N];
    unsigned bw = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) d[i] = __builtin_subc(t[i], C::p(i), bw, &bw);
    const bool use_d = (top != 0) || (bw == 0);
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = use_d ? d[i] : t[i];
}
template <class C> ZK_DEV Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
    uint32_t t[C::N];
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) t[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    Fp<C> r;
    fp_cond_sub_p<C>(r, t, (uint32_t)c);
    return r;
}
// a + b without the final correction: for reduced inputs the sum is < 2p < R/2. Only valid as an operand of fp_mul (the Montgomery
// product of operands < 2p is still < 2p before its own correction because 4p < R for every modulus here) — used for the operand
// sums of the Karatsuba Fq2 product.
template <class C> ZK_DEV Fp<C> fp_add_noreduce(const Fp<C>& a, const Fp<C>& b) {
    static_assert(C::p(C::N - 1) < 0x40000000u, "fp_add_noreduce needs 4p < R");
    Fp<C> r;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    return r;
}
template <class C> ZK_DEV Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
    uint32_t t[C::N];
    unsigned bw = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) t[i] = __builtin_subc(a.l[i], b.l[i], bw, &bw);
    const uint32_t mask = (uint32_t)0 - (uint32_t)bw;
    Fp<C> r;
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = __builtin_addc(t[i], C::p(i) & mask, c, &c);
    return r;
}
template <class C> ZK_DEV Fp<C> fp_neg(const Fp<C>& a) {
    Fp<C> r;
    unsigned bw = 0;
    const bool z = fp_is_zero(a);
#pragma unroll
    for (int i = 0; i < C::N; i++) { const uint32_t x = __builtin_subc(C::p(i), a.l[i], bw, &bw); r.l[i] = z ? 0u : x; }
    return r;
}
template <class C> ZK_DEV Fp<C> fp_dbl(const Fp<C>& a) { return fp_add(a, a); }

// Montgomery product a·b·R^{-1} mod p, fully reduced.
template <class C> ZK_DEV Fp<C> fp_mul(const Fp<C>& a, const Fp<C>& b) {
    constexpr int N = C::N;
    Fp<C> r;
#if ZKMI_MUL_VARIANT == 3
    // CIOS (coarsely integrated operand scanning): per row, t += a_i·b then t = (t + m·p)/2^32.
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t x = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t x = (uint64_t)t[N] + c;
        t[N] = (uint32_t)x;
        t[N + 1] = (uint32_t)(x >> 32);
        uint32_t m = t[0] * C::NP;
        c = ((uint64_t)m * C::p(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            uint64_t y = (uint64_t)m * C::p(j) + t[j] + c;
            t[j - 1] = (uint32_t)y;
            c = y >> 32;
        }
        x = (uint64_t)t[N] + c;
        t[N - 1] = (uint32_t)x;
        t[N] = t[N + 1] + (uint32_t)(x >> 32);
    }
    fp_cond_sub_p<C>(r, t, t[N]);
#else
    // FIPS (finely integrated product scanning): column k accumulates a_i·b_{k-i} and m_i·p_{k-i} into a
    // 64-bit accumulator + carry counter; 2N^2+N multiplies, 2 VALU instructions per multiply.
    uint32_t m[N], t[N];
    uint64_t acc = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mac(acc, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mac_c(acc, hi, m[i], C::p(k - i));
        m[k] = (uint32_t)acc * C::NP;
        mac_c(acc, hi, m[k], C::p(0));
        acc = (acc >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac(acc, hi, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac_c(acc, hi, m[i], C::p(k - i));
        t[k - N] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    fp_cond_sub_p<C>(r, t, (uint32_t)acc);
#endif
    return r;
}
template <class C> ZK_DEV Fp<C> fp_sqr(const Fp<C>& a) { return fp_mul(a, a); }
template <class C> ZK_DEV Fp<C> fp_to_mont(const Fp<C>& a) { return fp_mul(a, fp_r2<C>()); }
template <class C> ZK_DEV Fp<C> fp_from_mont(const Fp<C>& a) {
    Fp<C> o = fp_zero<C>();
    o.l[0] = 1;
    return fp_mul(a, o);
}
// a^e for a small public exponent (square-and-multiply, MSB first)
template <class C> ZK_DEV Fp<C> fp_pow_u32(const Fp<C>& a, uint32_t e) {
    Fp<C> r = fp_one<C>();
    for (int i = 31; i >= 0; i--) {
        r = fp_sqr(r);
        if ((e >> i) & 1) r = fp_mul(r, a);
    }
    return r;
}
// Fermat inverse a^(p-2); 0 -> 0. (Used only off the hot path; batched inversion is used where it matters.)
template <class C> __device__ __noinline__ Fp<C> fp_inv(const Fp<C>& a) {
    Fp<C> r = fp_one<C>();
    Fp<C> base = a;
    // exponent p-2, LSB first
    uint32_t bw = 2;
    for (int i = 0; i < C::N; i++) {
        uint32_t pi = C::p(i);
        uint32_t e = pi - bw;
        bw = (pi < bw) ? 1u : 0u;
        for (int b = 0; b < 32; b++) {
            if ((e >> b) & 1) r = fp_mul(r, base);
            base = fp_sqr(base);
        }
    }
    return r;
}

// ---- memory access: elements are 4·N contiguous bytes, 16-byte aligned -----------------------------------------
template <class C> ZK_DEV Fp<C> fp_load(const void* p) {
    Fp<C> r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) {
        uint4 v = q[i];
        r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
    }
    return r;
}
template <class C> ZK_DEV void fp_store(void* p, const Fp<C>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < C::N / 4; i++) q[i] = make_uint4(a.l[4 * i], a.l[4 * i + 1], a.l[4 * i + 2], a.l[4 * i + 3]);
}

}  // namespace zkmi
