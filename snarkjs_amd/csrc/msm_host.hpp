// snarkjs_amd/csrc/msm_host.hpp — host driver of the device Pippenger (launch sequence + final window fold).
#pragma once
#include <string.h>
#include <algorithm>
#include "host_field.hpp"
#include "msm.cuh"
#include "zkmi_common.hpp"

namespace zkmi {

static inline int ilog2_sz(size_t n) { int l = 0; while (((size_t)1 << (l + 1)) <= n) l++; return l; }

// Window width for n terms. Signed digits -> 2^(c-1) buckets per window; the table balances n·W mixed additions in
// k_msm_accum against the latency-bound bucket reduction (depth O(c)); re-tuned on MI355X.
static inline int msm_pick_c(size_t n) {
    int lg = ilog2_sz(n ? n : 1);
    static const int T[] = {2, 2, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 9, 10, 11, 11, 12, 13, 13, 14, 15, 15, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    return T[lg > 31 ? 31 : lg];
}

// Host-side adapters: F = device field type; HC = host curve over the matching host field.
template <class F> struct HostOf;
template <> struct HostOf<Fp<Bn254Fq>> { typedef host::HField<4> FT; static FT make() { return host::HField<4>::from_cfg<Bn254Fq>(); } };
template <> struct HostOf<Fp2<Bn254Fq>> { typedef host::HField2<4> FT; static FT make() { return FT{host::HField<4>::from_cfg<Bn254Fq>()}; } };
template <> struct HostOf<Fp<Bls12381Fq>> { typedef host::HField<6> FT; static FT make() { return host::HField<6>::from_cfg<Bls12381Fq>(); } };
template <> struct HostOf<Fp2<Bls12381Fq>> { typedef host::HField2<6> FT; static FT make() { return FT{host::HField<6>::from_cfg<Bls12381Fq>()}; } };

// Fold W window sums (device XYZZ, little-endian words) into one Jacobian point: sum_w 2^(c·w)·P_w.
template <class F> void msm_fold_windows(const uint32_t* win, int W, int c, uint8_t* out_jac) {
    typedef typename HostOf<F>::FT FT;
    typedef typename FT::E E;
    host::HCurve<FT> cv{HostOf<F>::make()};
    const FT& Fh = cv.F;
    constexpr int FW = FieldWords<F>::value;
    typename host::HCurve<FT>::P acc = cv.zero();
    for (int w = W - 1; w >= 0; w--) {
        if (!cv.is_zero(acc)) for (int k = 0; k < c; k++) acc = cv.dbl(acc);
        E X, Y, ZZ, ZZZ;
        const uint32_t* p = win + (size_t)w * 4 * FW;
        memcpy(&X, p, 4 * FW); memcpy(&Y, p + FW, 4 * FW); memcpy(&ZZ, p + 2 * FW, 4 * FW); memcpy(&ZZZ, p + 3 * FW, 4 * FW);
        if (ZZ.is_zero()) continue;
        // XYZZ -> Jacobian with Z = ZZ·ZZZ:  X' = X·ZZ·ZZZ^2, Y' = Y·ZZ^3·ZZZ^2
        typename host::HCurve<FT>::P q;
        E z2 = Fh.sqr(ZZZ), zzX = Fh.mul(ZZ, z2);
        q.X = Fh.mul(X, zzX);
        q.Y = Fh.mul(Y, Fh.mul(Fh.sqr(ZZ), zzX));
        q.Z = Fh.mul(ZZ, ZZZ);
        acc = cv.add(acc, q);
    }
    if (cv.is_zero(acc)) { memset(out_jac, 0, 3 * 4 * FW); return; }
    memcpy(out_jac, &acc.X, 4 * FW); memcpy(out_jac + 4 * FW, &acc.Y, 4 * FW); memcpy(out_jac + 8 * FW, &acc.Z, 4 * FW);
}

template <class F> int to_affine_host(const uint8_t* jac, uint8_t* aff) {
    typedef typename HostOf<F>::FT FT;
    host::HCurve<FT> cv{HostOf<F>::make()};
    constexpr int FW = FieldWords<F>::value;
    typename host::HCurve<FT>::P p;
    memcpy(&p.X, jac, 4 * FW); memcpy(&p.Y, jac + 4 * FW, 4 * FW); memcpy(&p.Z, jac + 8 * FW, 4 * FW);
    typename FT::E x, y;
    cv.to_affine(p, x, y);
    memcpy(aff, &x, 4 * FW); memcpy(aff + 4 * FW, &y, 4 * FW);
    return ZKMI_OK;
}

template <class F, int NW> int msm_launch_digits(const uint8_t* d_scalars, const MsmShape& sh, uint32_t* counts, uint32_t* starts, uint32_t* cursor,
                                                uint32_t* sorted, hipStream_t st) {
    const unsigned blocks = (unsigned)((sh.n + 255) / 256);
    hipLaunchKernelGGL((k_msm_count<NW>), dim3(blocks), dim3(256), 0, st, d_scalars, sh, counts);
    hipLaunchKernelGGL(k_msm_scan, dim3(sh.W), dim3(1024), 0, st, counts, starts, sh.nb);
    hipLaunchKernelGGL((k_msm_scatter<NW>), dim3(blocks), dim3(256), 0, st, d_scalars, sh, starts, cursor, sorted);
    return ZKMI_OK;
}

// Full device MSM: bases (affine, device), scalars (plain integers, device) -> Jacobian point on the host.
template <class F> int msm_run(const void* d_bases, const void* d_scalars, size_t n, size_t sb, uint8_t* out_jac) {
    constexpr int FW = FieldWords<F>::value, PW = 4 * FW;
    Ctx& cx = ctx();
    if (n == 0) { memset(out_jac, 0, 3 * 4 * FW); return ZKMI_OK; }
    if (n >= (1ull << 31)) return fail(ZKMI_ERR_UNSUPPORTED, "msm: n >= 2^31");
    if (sb == 0 || sb > 64) return fail(ZKMI_ERR_UNSUPPORTED, "msm: scalar size must be 1..64 bytes");
    MsmShape sh;
    sh.n = (uint32_t)n; sh.sb = (int)sb;
    sh.c = cx.msm_c_override ? cx.msm_c_override : msm_pick_c(n);
    sh.W = (int)((8 * sb + 1 + sh.c - 1) / sh.c);
    sh.nb = 1u << (sh.c - 1);
    const size_t total = (size_t)sh.W * sh.nb;
    hipStream_t st = cx.stream;

    uint32_t *counts, *sorted, *buckets, *order, *hist, *redA0, *redR0, *redA1, *redR1;
    ZK_TRY(ws_get("msm.counts", 3 * total * 4, (void**)&counts));        // counts | starts | cursor
    uint32_t *starts = counts + total, *cursor = starts + total;
    ZK_TRY(ws_get("msm.sorted", (size_t)sh.W * n * 4, (void**)&sorted));
    ZK_TRY(ws_get("msm.buckets", total * PW * 4, (void**)&buckets));
    ZK_TRY(ws_get("msm.order", total * 4, (void**)&order));
    const uint32_t BINS = 1024;
    ZK_TRY(ws_get("msm.hist", BINS * 4, (void**)&hist));
    const uint32_t G = std::min<uint32_t>(8u, sh.nb);
    const uint32_t m1 = sh.nb / G;
    ZK_TRY(ws_get("msm.redA0", (size_t)sh.W * m1 * PW * 4, (void**)&redA0));
    ZK_TRY(ws_get("msm.redR0", (size_t)sh.W * m1 * PW * 4, (void**)&redR0));
    constexpr int M = (PW * 4 * 2 * 256 <= 128 * 1024) ? 256 : 128;    // LDS: 2 arrays of M points
    const uint32_t m2 = (m1 + M - 1) / M;
    ZK_TRY(ws_get("msm.redA1", (size_t)sh.W * std::max(m2, 1u) * PW * 4, (void**)&redA1));
    ZK_TRY(ws_get("msm.redR1", (size_t)sh.W * std::max(m2, 1u) * PW * 4, (void**)&redR1));

    ZK_HIP(hipEventRecord(cx.ev0, st));
    ZK_HIP(hipMemsetAsync(counts, 0, 3 * total * 4, st));
    ZK_HIP(hipMemsetAsync(hist, 0, BINS * 4, st));
    const uint8_t* sc = (const uint8_t*)d_scalars;
    if (sb <= 4) msm_launch_digits<F, 1>(sc, sh, counts, starts, cursor, sorted, st);
    else if (sb <= 32) msm_launch_digits<F, 8>(sc, sh, counts, starts, cursor, sorted, st);
    else msm_launch_digits<F, 16>(sc, sh, counts, starts, cursor, sorted, st);
    // bucket order by size (descending)
    const unsigned tb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_msm_size_hist, dim3(tb), dim3(256), 0, st, counts, (uint32_t)total, BINS - 1, hist);
    hipLaunchKernelGGL(k_msm_size_scan, dim3(1), dim3(64), 0, st, hist, BINS);
    hipLaunchKernelGGL(k_msm_size_scatter, dim3(tb), dim3(256), 0, st, counts, (uint32_t)total, BINS - 1, hist, order);
    // accumulate
    hipLaunchKernelGGL((k_msm_accum<F>), dim3(tb), dim3(256), 0, st, (const uint32_t*)d_bases, sh, counts, starts, sorted, order, buckets);
    // reduce: level 1 (sequential groups of G), then block levels until one point per window
    const uint32_t tg = sh.W * m1;
    hipLaunchKernelGGL((k_msm_reduce_seq<F>), dim3((tg + 255) / 256), dim3(256), 0, st, buckets, sh.nb, G, m1, tg, redA0, redR0);
    uint32_t m = m1;
    int log_scale = ilog2_sz(G);
    uint32_t *inA = redA0, *inR = redR0, *outA = redA1, *outR = redR1;
    const size_t lds_bytes = (size_t)2 * M * PW * 4;
    static bool attr_set = false;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute((const void*)k_msm_reduce_block<F, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    for (;;) {
        const uint32_t blocks = (m + M - 1) / M;
        const int fin = blocks == 1;
        hipLaunchKernelGGL((k_msm_reduce_block<F, M>), dim3(sh.W * blocks), dim3(M), lds_bytes, st, inA, inR, m, blocks, log_scale, fin, outA, outR);
        std::swap(inA, outA); std::swap(inR, outR);
        if (fin) break;
        m = blocks;
        log_scale += ilog2_sz(M);
    }
    ZK_HIP(hipEventRecord(cx.ev1, st));
    std::vector<uint32_t> win((size_t)sh.W * PW);
    ZK_HIP(hipMemcpyAsync(win.data(), inA, win.size() * 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    float ms = 0;
    hipEventElapsedTime(&ms, cx.ev0, cx.ev1);
    cx.last_ms = ms;
    msm_fold_windows<F>(win.data(), sh.W, sh.c, out_jac);
    return ZKMI_OK;
}

template <class F, class FrC> int gen_bases_run(const uint8_t* gen_affine_host, size_t n, uint64_t f, uint64_t g, void* d_out) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    uint32_t* d_gen;
    ZK_TRY(ws_get("gen.generator", 2 * FW * 4, (void**)&d_gen));
    ZK_HIP(hipMemcpyAsync(d_gen, gen_affine_host, 2 * FW * 4, hipMemcpyHostToDevice, cx.stream));
    hipLaunchKernelGGL((k_gen_geometric_bases<F, FrC>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, d_gen, (uint32_t)n, f, g, (uint32_t*)d_out);
    ZK_HIP(hipStreamSynchronize(cx.stream));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

}  // namespace zkmi
