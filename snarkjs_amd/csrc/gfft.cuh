// snarkjs_amd/csrc/gfft.cuh — FFT / inverse FFT over GROUP elements and G.batchApplyKey for gfx950 (the ceremony-side
// bulk operations).
//
// Replaces ffjavascript's engine_fft for G1 / G2 (build/snarkjs.min.js:1@215859 with the g1m_/g2m_fftMix/_fftJoin/_fftFinal kernels of
// wasmcurves: the same radix-2 butterflies as Fr.fft, with "multiply by a twiddle" = G.timesFr) and engine_applykey for G1 / G2
// (@211529, g1m_/g2m_batchApplyKey): callers src/powersoftau_preparephase2.js:87 (G.lagrangeEvaluations -> G.ifft),
// src/mpc_applykey.js:44-70, src/powersoftau_verify.js:358-482.
//   X_k = sum_j w^(jk) P_j,  w = Fr.w[log n], natural order in and out, the inverse scaled by 1/n.
// A group butterfly costs one 254-bit scalar multiplication (~3500 field multiplications) for 2 x 64..192 bytes of data: purely
// ALU-bound, so the layout is the plain one — points live as XYZZ in one work array, one lane per butterfly and one launch per
// stage (log n launches), decimation in time over a bit-reversed load. The twiddle w_{2h}^j comes from the NTT module's split
// power tables (one field multiplication), is brought to normal form and consumed as a signed binary (NAF) digit stream:
// ~254 doublings + ~85 additions per butterfly.
#pragma once
#include "curve.cuh"

namespace zkmi {

struct GfftTw {
    const uint32_t* T_lo;
    const uint32_t* T_hi;
    uint32_t log_lb, log_n;
};

// k*Q for a plain 256-bit scalar k (8 words, normal form), LSB-first non-adjacent form: digit in {-1, 0, 1}
template <class F> __device__ __noinline__ XYZZ<F> pt_mul_naf(const XYZZ<F>& q_in, const uint32_t (&k_in)[8]) {
    XYZZ<F> acc, q = q_in;
    pt_set_inf(acc);
    if (pt_is_inf(q)) return acc;
    uint32_t k[9];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = k_in[i];
    k[8] = 0;
    for (int bit = 0; bit < 258; bit++) {
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nz |= k[i];
        if (!nz) break;
        if (k[0] & 1u) {
            const bool neg = (k[0] & 3u) == 3u;                  // k mod 4 == 3: digit -1, k += 1
            XYZZ<F> t = q;
            if (neg) {
                t.Y = f_neg(t.Y);
                uint32_t c = 1;
#pragma unroll
                for (int i = 0; i < 9; i++) { uint32_t s = k[i] + c; c = (s < c) ? 1u : 0u; k[i] = s; }
            } else k[0] &= ~1u;
            acc = pt_add(acc, t);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = (k[i] >> 1) | (k[i + 1] << 31);
        k[8] >>= 1;
        q = pt_dbl(q);
    }
    return acc;
}
template <class F> ZK_DEV XYZZ<F> pt_from_affine(const Affine<F>& a) {
    XYZZ<F> p;
    if (pt_is_inf(a)) { pt_set_inf(p); return p; }
    p.X = a.x; p.Y = a.y; f_set_one(p.ZZ); f_set_one(p.ZZZ);
    return p;
}
template <class F> ZK_DEV Affine<F> pt_to_affine(const XYZZ<F>& q) {
    Affine<F> r;
    if (pt_is_inf(q)) { f_set_zero(r.x); f_set_zero(r.y); return r; }
    F i3 = f_inv(q.ZZZ), i2 = f_sqr(f_mul(q.ZZ, i3));            // 1/ZZ = (ZZ/ZZZ)^2
    r.x = f_mul(q.X, i2); r.y = f_mul(q.Y, i3);
    return r;
}
template <class FrC> ZK_DEV void fr_to_scalar(uint32_t (&k)[8], const Fp<FrC>& m) {
    const Fp<FrC> nrm = fp_from_mont(m);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = nrm.l[i];
}

// work[bitrev(i)] = P_i (affine in, XYZZ out)
template <class F> __global__ void __launch_bounds__(256)
k_gfft_load(const uint32_t* __restrict__ in, uint32_t* __restrict__ work, uint32_t n, uint32_t log_n) {
    constexpr int FW = FieldWords<F>::value;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> a; pt_load(a, in + (size_t)i * 2 * FW);
    const uint32_t j = log_n ? (__brev(i) >> (32 - log_n)) : 0u;
    pt_store(work + (size_t)j * 4 * FW, pt_from_affine(a));
}
// one DIT stage: (a, b) -> (a + w b, a - w b), w = root^(j * n / (2h)) with root = Fr.w[log n] (or its inverse)
template <class F, class FrC> __global__ void __launch_bounds__(256)
k_gfft_stage(uint32_t* __restrict__ work, uint32_t n, uint32_t st, GfftTw tw) {
    constexpr int PW = 4 * FieldWords<F>::value;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n / 2) return;
    const uint32_t h = 1u << (st - 1), j = b & (h - 1), lo = ((b >> (st - 1)) << st) + j, hi = lo + h;
    XYZZ<F> A, B;
    pt_load(A, work + (size_t)lo * PW); pt_load(B, work + (size_t)hi * PW);
    if (j) {
        const uint64_t e = (uint64_t)j << (tw.log_n - st);
        const Fp<FrC> lo_t = fp_load<FrC>(tw.T_lo + (size_t)(e & ((1ull << tw.log_lb) - 1)) * 8), hi_t = fp_load<FrC>(tw.T_hi + (size_t)(e >> tw.log_lb) * 8);
        uint32_t k[8];
        fr_to_scalar<FrC>(k, fp_mul(lo_t, hi_t));
        B = pt_mul_naf<F>(B, k);
    }
    XYZZ<F> nB = B;
    nB.Y = f_neg(nB.Y);
    pt_store(work + (size_t)lo * PW, pt_add(A, B));
    pt_store(work + (size_t)hi * PW, pt_add(A, nB));
}
// XYZZ -> affine, with the 1/n factor of the inverse transform (scale = n^-1 in Montgomery form, or nullptr)
template <class F, class FrC> __global__ void __launch_bounds__(256)
k_gfft_store(const uint32_t* __restrict__ work, uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ scale) {
    constexpr int FW = FieldWords<F>::value;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ<F> p; pt_load(p, work + (size_t)i * 4 * FW);
    if (scale) {
        uint32_t k[8];
        fr_to_scalar<FrC>(k, fp_load<FrC>(scale));
        p = pt_mul_naf<F>(p, k);
    }
    const Affine<F> a = pt_to_affine(p);
    f_store(out + (size_t)i * 2 * FW, a.x); f_store(out + (size_t)i * 2 * FW + FW, a.y);
}
// G.batchApplyKey: out_i = (first * inc^i) * P_i, affine in and out; `consts` = first | inc | inc^256 (Montgomery): a lane starts from
// first * inc^i computed as first * (inc^256)^(i / 256) * inc^(i % 256) by square-and-multiply on the small exponents
template <class F, class FrC> __global__ void __launch_bounds__(256)
k_g_apply_key(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ consts) {
    constexpr int FW = FieldWords<F>::value;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fp<FrC> first = fp_load<FrC>(consts), inc = fp_load<FrC>(consts + 8), inc256 = fp_load<FrC>(consts + 16);
    const Fp<FrC> f = fp_mul(first, fp_mul(fp_pow_u32(inc256, i >> 8), fp_pow_u32(inc, i & 255u)));
    uint32_t k[8];
    fr_to_scalar<FrC>(k, f);
    Affine<F> a; pt_load(a, in + (size_t)i * 2 * FW);
    const Affine<F> r = pt_to_affine(pt_mul_naf<F>(pt_from_affine(a), k));
    f_store(out + (size_t)i * 2 * FW, r.x); f_store(out + (size_t)i * 2 * FW + FW, r.y);
}

}  // namespace zkmi
