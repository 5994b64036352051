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

// ---- stage 1-3 (field independent, msm_sort.hip): signed-digit recoding + counting sort of all windows -----------
// The plan points into the library's named scratch buffers; it stays valid until the next msm_sort on the stream.
struct MsmPlan {
    MsmShape sh;
    size_t total = 0;                 // W * nb buckets
    uint32_t cap = 0;                 // lane-group granularity (msm.cuh: k_msm_assign)
    size_t lane_bound = 0, multi_bound = 0;
    uint32_t *counts = nullptr, *starts = nullptr, *sorted = nullptr;
    uint32_t *lane_g = nullptr, *lane_sub = nullptr, *meta = nullptr, *giants = nullptr;
};
constexpr int MSM_TB = 128, MSM_LOG_TB = 7;      // tree block: 128 lanes x 384 B (BLS12-381 G2 XYZZ) = 48 KiB of LDS
int msm_sort(const void* d_scalars, size_t n, size_t sb, MsmPlan& plan);

// One MSM in flight: per-window sums land in a pinned host slot; msm_fold turns them into the Jacobian result.
struct MsmJob {
    int W = 0, c = 0;
    uint32_t* h_win = nullptr;
};
constexpr int MSM_JOB_SLOTS = 8;
constexpr size_t MSM_JOB_SLOT_BYTES = 128 * 1024;
int msm_job_slot(int slot, MsmJob& job);

// ---- stage 4-5: bucket accumulation + reduction for one base table over an existing plan ------------------------------
// skip: the scalar with index i pairs with base (i - skip); indices < skip are ignored. This lets several MSMs share
// one digit sort (Groth16: A, B1, B2 over the witness and C over witness[nPublic+1:], src/groth16_prove.js:85-97).
template <class F> int msm_accumulate(const void* d_bases, const MsmPlan& pl, uint32_t skip, MsmJob& job) {
    constexpr int FW = FieldWords<F>::value, PW = 4 * FW;
    Ctx& cx = ctx();
    const MsmShape& sh = pl.sh;
    const size_t total = pl.total;
    hipStream_t st = cx.stream;
    uint32_t *buckets, *redA0, *redR0, *redA1, *redR1, *lane_partials, *block_partials;
    ZK_TRY(ws_get("msm.buckets", total * PW * 4, (void**)&buckets));
    ZK_TRY(ws_get("msm.lane_partials", std::max<size_t>(pl.multi_bound, 1) * PW * 4, (void**)&lane_partials));
    const size_t tree_blocks = pl.multi_bound / MSM_TB + 1;
    ZK_TRY(ws_get("msm.block_partials", tree_blocks * PW * 4, (void**)&block_partials));
    const uint32_t G = std::min<uint32_t>(8u, sh.nb);
    const uint32_t m1 = sh.nb / G;
    ZK_TRY(ws_get("msm.redA0", (size_t)sh.W * m1 * PW * 4, (void**)&redA0));
    ZK_TRY(ws_get("msm.redR0", (size_t)sh.W * m1 * PW * 4, (void**)&redR0));
    constexpr int M = (PW * 4 * 2 * 256 <= 128 * 1024) ? 256 : 128;    // LDS: 2 arrays of M points
    const uint32_t m2 = (m1 + M - 1) / M;
    ZK_TRY(ws_get("msm.redA1", (size_t)sh.W * std::max(m2, 1u) * PW * 4, (void**)&redA1));
    ZK_TRY(ws_get("msm.redR1", (size_t)sh.W * std::max(m2, 1u) * PW * 4, (void**)&redR1));
    if ((size_t)sh.W * PW * 4 > MSM_JOB_SLOT_BYTES) return fail(ZKMI_ERR_UNSUPPORTED, "msm: too many windows");
    static bool tree_attr = false;
    const size_t tree_lds = (size_t)MSM_TB * PW * 4;
    if (!tree_attr) {
        ZK_HIP(hipFuncSetAttribute((const void*)k_msm_tree<F, MSM_TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tree_lds));
        ZK_HIP(hipFuncSetAttribute((const void*)k_msm_giant<F, MSM_TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tree_lds));
        tree_attr = true;
    }
    hipLaunchKernelGGL((k_msm_accum<F>), dim3((unsigned)((pl.lane_bound + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_bases, sh, skip, pl.cap, pl.counts,
                       pl.starts, pl.sorted, pl.lane_g, pl.lane_sub, pl.meta, buckets, lane_partials);
    hipLaunchKernelGGL((k_msm_tree<F, MSM_TB>), dim3((unsigned)tree_blocks), dim3(MSM_TB), tree_lds, st, lane_partials, pl.lane_g, pl.counts, pl.cap, pl.meta, buckets,
                       block_partials);
    hipLaunchKernelGGL((k_msm_giant<F, MSM_TB>), dim3((unsigned)tree_blocks), dim3(MSM_TB), tree_lds, st, pl.giants, pl.meta, block_partials, buckets);
    // reduce: level 1 (sequential groups of G), then block levels until one point per window
    const uint32_t tg = sh.W * m1;
    hipLaunchKernelGGL((k_msm_reduce_seq<F>), dim3((tg + 255) / 256), dim3(256), 0, st, buckets, pl.counts, sh.nb, G, m1, tg, redA0, redR0);
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
    job.W = sh.W; job.c = sh.c;
    ZK_HIP(hipMemcpyAsync(job.h_win, inA, (size_t)sh.W * PW * 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
// after the stream has been synchronised
template <class F> void msm_fold(const MsmJob& job, uint8_t* out_jac) { msm_fold_windows<F>(job.h_win, job.W, job.c, out_jac); }

// Full device MSM: bases (affine, device), scalars (plain integers, device) -> Jacobian point on the host.
template <class F> int msm_run(const void* d_bases, const void* d_scalars, size_t n, size_t sb, uint8_t* out_jac) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    if (n == 0) { memset(out_jac, 0, 3 * 4 * FW); return ZKMI_OK; }
    hipStream_t st = cx.stream;
    MsmPlan pl;
    MsmJob job;
    ZK_TRY(msm_job_slot(0, job));
    ZK_HIP(hipEventRecord(cx.ev0, st));
    ZK_TRY(msm_sort(d_scalars, n, sb, pl));
    ZK_TRY(msm_accumulate<F>(d_bases, pl, 0, job));
    ZK_HIP(hipEventRecord(cx.ev1, st));
    ZK_HIP(hipStreamSynchronize(st));
    float ms = 0;
    if (hipEventElapsedTime(&ms, cx.ev0, cx.ev1) == hipSuccess) cx.last_ms = ms;
    msm_fold<F>(job, out_jac);
    return ZKMI_OK;
}

// non-template entry points (msm_bn254.hip / msm_bls12381.hip) for callers that must not instantiate the kernels again
int msm_accumulate_dispatch(int curve, int group, const void* d_bases, const MsmPlan& pl, uint32_t skip, MsmJob& job);
int msm_fold_dispatch(int curve, int group, const MsmJob& job, uint8_t* out_jac);

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
