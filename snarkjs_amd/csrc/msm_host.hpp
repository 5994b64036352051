// snarkjs_amd/csrc/msm_host.hpp — host driver of the device Pippenger (launch sequence + final window fold).
#pragma once
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "host_field.hpp"
#include "msm.cuh"
#include "msm29.cuh"
#include <type_traits>
#include "zkmi_common.hpp"

namespace zkmi {

static inline int ilog2_sz(size_t n) { int l = 0; while (((size_t)1 << (l + 1)) <= n) l++; return l; }

// Window width for n terms. Signed digits -> 2^(c-1) buckets per window; the table balances n·W mixed additions in
// k_msm_accum against the latency-bound bucket reduction (depth O(c)); re-tuned on MI355X.
static inline int msm_pick_c(size_t n) {
    int lg = ilog2_sz(n ? n : 1);
    if (lg < 62 && n > (((size_t)3) << lg) / 2) lg++;      // round to the nearest power of two (nVars is usually just below one)
    // (17..21 use 15: see msm_precomp_c about the fill of the top digit window)
    static const int T[] = {2, 2, 2, 3, 3, 4, 4, 5, 6, 7, 8, 9, 9, 10, 11, 11, 12, 15, 15, 15, 15, 15, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    return T[lg > 31 ? 31 : lg];
}

// Host-side adapters: F = device field type; HC = host curve over the matching host field.
template <class F> struct HostOf;
template <> struct HostOf<Fp<Bn254Fq>> { typedef host::HField<4> FT; static FT make() { return host::HField<4>::from_cfg<Bn254Fq>(); } };
template <> struct HostOf<Fp2<Bn254Fq>> { typedef host::HField2<4> FT; static FT make() { return FT{host::HField<4>::from_cfg<Bn254Fq>()}; } };
template <> struct HostOf<Fp<Bls12381Fq>> { typedef host::HField<6> FT; static FT make() { return host::HField<6>::from_cfg<Bls12381Fq>(); } };
template <> struct HostOf<Fp2<Bls12381Fq>> { typedef host::HField2<6> FT; static FT make() { return FT{host::HField<6>::from_cfg<Bls12381Fq>()}; } };

// XYZZ (device words) -> host Jacobian with Z = ZZ*ZZZ:  X' = X*ZZ*ZZZ^2, Y' = Y*ZZ^3*ZZZ^2
template <class F> typename host::HCurve<typename HostOf<F>::FT>::P xyzz_to_jac(const host::HCurve<typename HostOf<F>::FT>& cv, const uint32_t* p) {
    typedef typename HostOf<F>::FT FT;
    typedef typename FT::E E;
    constexpr int FW = FieldWords<F>::value;
    const FT& Fh = cv.F;
    E X, Y, ZZ, ZZZ;
    memcpy(&X, p, 4 * FW); memcpy(&Y, p + FW, 4 * FW); memcpy(&ZZ, p + 2 * FW, 4 * FW); memcpy(&ZZZ, p + 3 * FW, 4 * FW);
    if (ZZ.is_zero()) return cv.zero();
    typename host::HCurve<FT>::P q;
    E z2 = Fh.sqr(ZZZ), zzX = Fh.mul(ZZ, z2);
    q.X = Fh.mul(X, zzX);
    q.Y = Fh.mul(Y, Fh.mul(Fh.sqr(ZZ), zzX));
    q.Z = Fh.mul(ZZ, ZZZ);
    return q;
}
// Fold the per-window device results into one Jacobian point. Per window w the device leaves (see msm.cuh, bucket
// reduction): win[(2w)*PW] = WR = sum_r r*Row_r, win[(2w+1)*PW] = WC = sum_c c*Col_c and tot[(2w)*PW] = T = sum_r Row_r;
// S_w = 2^cbits * WR + WC + T, result = sum_w 2^(c*w) * S_w (the reference also recombines windows on the host, @213360).
template <class F> void msm_fold_windows(const uint32_t* win, const uint32_t* tot, int W, int c, int cbits, uint8_t* out_jac) {
    typedef typename HostOf<F>::FT FT;
    host::HCurve<FT> cv{HostOf<F>::make()};
    constexpr int FW = FieldWords<F>::value, PW = 4 * FW;
    typename host::HCurve<FT>::P acc = cv.zero();
    for (int w = W - 1; w >= 0; w--) {
        if (!cv.is_zero(acc)) for (int k = 0; k < c; k++) acc = cv.dbl(acc);
        auto wr = xyzz_to_jac<F>(cv, win + (size_t)(2 * w) * PW);
        for (int k = 0; k < cbits && !cv.is_zero(wr); k++) wr = cv.dbl(wr);
        auto s = cv.add(cv.add(wr, xyzz_to_jac<F>(cv, win + (size_t)(2 * w + 1) * PW)), xyzz_to_jac<F>(cv, tot + (size_t)(2 * w) * PW));
        acc = cv.add(acc, s);
    }
    if (cv.is_zero(acc)) { memset(out_jac, 0, 3 * 4 * FW); return; }
    memcpy(out_jac, &acc.X, 4 * FW); memcpy(out_jac + 4 * FW, &acc.Y, 4 * FW); memcpy(out_jac + 8 * FW, &acc.Z, 4 * FW);
}

// out = a + b for two Jacobian points in the C-ABI format (host, O(1)): combines per-GPU partial MSM results.
template <class F> int point_add_host(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    typedef typename HostOf<F>::FT FT;
    host::HCurve<FT> cv{HostOf<F>::make()};
    constexpr int FW = FieldWords<F>::value;
    typename host::HCurve<FT>::P p, q;
    memcpy(&p.X, a, 4 * FW); memcpy(&p.Y, a + 4 * FW, 4 * FW); memcpy(&p.Z, a + 8 * FW, 4 * FW);
    memcpy(&q.X, b, 4 * FW); memcpy(&q.Y, b + 4 * FW, 4 * FW); memcpy(&q.Z, b + 8 * FW, 4 * FW);
    auto r = cv.add(p, q);
    if (cv.is_zero(r)) { memset(out, 0, 12 * FW); return ZKMI_OK; }
    memcpy(out, &r.X, 4 * FW); memcpy(out + 4 * FW, &r.Y, 4 * FW); memcpy(out + 8 * FW, &r.Z, 4 * FW);
    return ZKMI_OK;
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
    int slot = 0;                     // which set of plan buffers (0/1): two plans stay alive when their MSMs are reduced in one batch
    size_t total = 0;                 // W * nb buckets
    uint32_t cap = 0;                 // lane-group granularity (msm.cuh: k_msm_assign)
    size_t lane_bound = 0, multi_bound = 0;
    uint32_t *counts = nullptr, *starts = nullptr, *sorted = nullptr;
    uint32_t *lane_g = nullptr, *lane_sub = nullptr, *meta = nullptr, *giants = nullptr;
};
constexpr int MSM_TB = 128, MSM_LOG_TB = 7;      // tree block: 128 lanes x 384 B (BLS12-381 G2 XYZZ) = 48 KiB of LDS
// precomp_c != 0: the bases of every MSM run over this plan are pre-computed window tables built with that c
// d_dropmask: optional bitmap over the scalars; set bits are left out of the lists (bases at infinity)
int msm_sort(const void* d_scalars, size_t n, size_t sb, MsmPlan& plan, int plan_slot = 0, int precomp_c = 0, size_t table_stride = 0, const uint32_t* d_dropmask = nullptr);

// One MSM in flight: per-window sums land in a pinned host slot; msm_fold turns them into the Jacobian result.
struct MsmJob {
    int W = 0, c = 0, cbits = 0, slot = 0;
    bool merged = false;                            // accumulated into another job's buckets (msm_accumulate `into`)
    bool r29 = false;                               // the buckets hold R'-form words (msm29.cuh): reduced by k_msm_rowcol_wave29
    int bitsums = 0;                                // h_win holds per (array, k) plain sums (k_msm_bitsums) instead of weighted sums
    uint32_t nb = 0;
    uint32_t* h_win = nullptr;                      // pinned host: 2W weighted sums then 2W totals (XYZZ)
    const uint32_t *buckets = nullptr, *counts = nullptr;   // device: this job's complete buckets (between accumulate and reduce)
    hipEvent_t acc0 = nullptr, acc1 = nullptr;      // bracket the k_msm_accum launch (bench.py roofline: live kernel time)
};
constexpr int MSM_JOB_SLOTS = 8;
constexpr size_t MSM_JOB_SLOT_BYTES = 512 * 1024;
int msm_job_slot(int slot, MsmJob& job);

// ---- stage 4: bucket accumulation for one base table over an existing plan ---------------------------------------------
// skip: the scalar with index i pairs with base (i - skip); indices < skip are ignored. This lets several MSMs share
// one digit sort (Groth16: A, B1, B2 over the witness and C over witness[nPublic+1:], src/groth16_prove.js:85-97).
// Leaves the complete buckets of this MSM in the bucket buffer of job.slot; msm_reduce finishes the job.
// into (optional): an already accumulated job of the same shape; this MSM's points are added into ITS buckets (the two results are
// only needed as a sum) and `job` is marked merged: msm_reduce skips it and msm_fold returns the point at infinity for it.
template <class F> int msm_accumulate(const void* d_bases, const MsmPlan& pl, uint32_t skip, MsmJob& job, const uint32_t* d_infmask = nullptr, MsmJob* into = nullptr) {
    constexpr int FW = FieldWords<F>::value, PW = 4 * FW;
    constexpr bool WIDE = FW > 12;
    Ctx& cx = ctx();
    const MsmShape& sh = pl.sh;
    const size_t total = pl.total;
    hipStream_t st = cx.stream;
    const std::string sfx = "." + std::to_string(job.slot);
    uint32_t *buckets, *lane_partials, *block_partials;
    const uint32_t* prev_counts = nullptr;
    if (into) {
        if (WIDE) return fail(ZKMI_ERR_UNSUPPORTED, "msm_accumulate: merge mode is implemented for G1 only");
        if (into->W != sh.W || into->c != sh.c || into->nb != sh.nb || !into->buckets) return fail(ZKMI_ERR_INVALID, "msm_accumulate: merge target of a different shape");
        buckets = const_cast<uint32_t*>(into->buckets);
        prev_counts = into->counts;
    } else ZK_TRY(ws_get("msm.buckets" + sfx, total * PW * 4, (void**)&buckets));
    ZK_TRY(ws_get("msm.lane_partials", std::max<size_t>(pl.multi_bound, 1) * PW * 4, (void**)&lane_partials));
    const size_t tree_blocks = pl.multi_bound / MSM_TB + 1;
    ZK_TRY(ws_get("msm.block_partials", tree_blocks * PW * 4, (void**)&block_partials));
    static bool tree_attr = false;
    const size_t tree_lds = (size_t)MSM_TB * PW * 4;
    if (!tree_attr) {
        ZK_HIP(hipFuncSetAttribute((const void*)k_msm_tree<F, MSM_TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tree_lds));
        ZK_HIP(hipFuncSetAttribute((const void*)k_msm_giant<F, MSM_TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tree_lds));
        tree_attr = true;
    }
    constexpr unsigned AT = WIDE ? MsmAccumBlock<F>::value : 256;      // threads per accumulation block
    const size_t acc_lds = WIDE ? (size_t)PW * AT * 4 : 0;              // WIDE: XYZZ accumulators live in LDS
    static bool acc_attr = false;
    if (WIDE && !acc_attr) {
        ZK_HIP(hipFuncSetAttribute((const void*)k_msm_accum<F, WIDE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds));
        acc_attr = true;
    }
    if (job.acc0) ZK_HIP(hipEventRecord(job.acc0, st));
    bool launched = false, r29_buckets = false;
    typedef typename F::Cfg C;
    // base arrays registered in the R'-form of field29.cuh (window tables of resident bases; the library's own upload of a plain zkmi_msm
    // call): the hot loop on unsaturated limbs (9 x 29 bits BN254, 14 x 28 bits BLS12-381)
    auto r29 = cx.r29_tables.find(d_bases);
    const bool table29 = r29 != cx.r29_tables.end();
    if constexpr (!WIDE) {
        if (table29) {
            // ZKMI_R29_REDUCE=0: buckets leave the kernel in the reference's R-form and the generic row / column sums reduce them
            static const bool r29_reduce = !(getenv("ZKMI_R29_REDUCE") && atoi(getenv("ZKMI_R29_REDUCE")) == 0) && !(getenv("ZKMI_ROWCOL_WAVE") && atoi(getenv("ZKMI_ROWCOL_WAVE")) == 0);
            launched = true;
            r29_buckets = into ? into->r29 : (r29_reduce && (sh.c - 1) / 2 >= 6);       // the wave row/column sums need >= 64 buckets per row and column
            const uint32_t* mask = d_infmask ? d_infmask : r29->second;
            // Threads per block. A workgroup is placed only when EVERY one of its waves finds registers: a 256-thread block needs a free slot
            // on all four SIMDs of a CU. While the Fq2 bucket reduction runs beside it on the auxiliary stream (256-register waves on two of the
            // four SIMDs), a second 14-limb accumulation block (224 registers per wave) no longer fits and the CU drops from eight to four
            // accumulation waves (r03 trace: B1 4.5 ms against 2.1 ms for the same work alone); 128-thread blocks still fill the other two SIMDs.
            static const unsigned acc_t_env = getenv("ZKMI_ACC29_BLOCK") ? (unsigned)atoi(getenv("ZKMI_ACC29_BLOCK")) : 0u;
            const unsigned AT29 = (acc_t_env == 64 || acc_t_env == 128 || acc_t_env == 256) ? acc_t_env : (Lim29<C>::NL > 9 ? 128u : 256u);
            const dim3 grid((unsigned)((pl.lane_bound + AT29 - 1) / AT29));
            // compact_code(): the instantiation with CALLED products where the inlined loop exceeds the instruction cache (14-limb curve)
            // and this box fetches instructions slowly beyond it (field29.cuh: Compact)
            const bool cc = Lim29<C>::NL > 9 && (compact_code() & 1);
#define ZK_LAUNCH_ACC29(CC, MG) hipLaunchKernelGGL((k_msm_accum29<CC, MG>), grid, dim3(AT29), 0, st, (const uint32_t*)d_bases, mask, sh, skip, pl.cap, pl.counts, pl.starts, \
                                                   pl.sorted, pl.lane_g, pl.lane_sub, pl.meta, buckets, lane_partials, prev_counts, (int)r29_buckets)
            bool done = false;
            if constexpr (Lim29<C>::NL > 9) if (cc) { if (into) ZK_LAUNCH_ACC29(Compact<C>, true); else ZK_LAUNCH_ACC29(Compact<C>, false); done = true; }
            if (!done) { if (into) ZK_LAUNCH_ACC29(C, true); else ZK_LAUNCH_ACC29(C, false); }
#undef ZK_LAUNCH_ACC29
        }
    } else {
        if (table29 && into) return fail(ZKMI_ERR_UNSUPPORTED, "msm_accumulate: no merge mode over an R'-form G2 table");
        if (table29) {
            launched = true;
            constexpr unsigned T29 = Accum29G2<C>::T;
            constexpr size_t lds29 = Accum29G2<C>::lds_bytes;              // 4 coordinates x 2 components per lane
            static bool a29 = false;
            if (!a29) {
                ZK_HIP(hipFuncSetAttribute((const void*)k_msm_accum29_g2<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds29));
                if constexpr (Lim29<C>::NL > 9) ZK_HIP(hipFuncSetAttribute((const void*)k_msm_accum29_g2<Compact<C>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds29));
                a29 = true;
            }
            // BN254's 72 KB loop loses more to the calls (3.5 -> 7.8 ms on a healthy box) than a slow-fetch box costs it (+8 %): 14-limb curve only
            const bool cc = Lim29<C>::NL > 9 && (compact_code() & 2) != 0;
            // The Fq2 buckets stay in R'-form and k_msm_rowcol_wave29_g2 forms the row / column sums on the same limbs (r03 A/B, same box:
            // BLS12-381 51.1 / 50.2 against 50.3 / 50.0 proofs/s, BN254 105.9 against 105.5); ZKMI_R29_REDUCE_G2=0: R-form buckets and the
            // generic 32-bit kernel
            static const bool g2r = !(getenv("ZKMI_R29_REDUCE_G2") && atoi(getenv("ZKMI_R29_REDUCE_G2")) == 0);
            static const bool wave_ok = !(getenv("ZKMI_ROWCOL_WAVE") && atoi(getenv("ZKMI_ROWCOL_WAVE")) == 0);
            // on a slow-fetch box (compact_code() bit 3) the Fq2 row / column sums go back to the generic 32-bit kernel, whose 55 - 84 KB of code
            // was not affected there, instead of the 320 - 750 KB of k_msm_rowcol_wave29_g2 (7.1 instead of 2.1 ms on such a box)
            r29_buckets = wave_ok && g2r && !(compact_code() & 8) && (sh.c - 1) / 2 >= 6;
            bool done = false;
            // r06: one Fq2 component per lane, accumulators in registers (msm29.cuh: k_msm_accum29_g2s) — BN254: 168 VGPRs, 3 waves per SIMD, bit-identical buckets;
            // BLS12-381: XYZZ in 248 VGPRs without a spill instead of the packed Jacobian in LDS with 111 spilled registers (another representative of the same
            // bucket), and a hot loop of 61 KB instead of 118 KB — it fits the instruction cache, so it also takes the place of the Compact instantiation on a
            // slow-fetch box. ZKMI_G2_SPLIT=0: the LDS-parked layouts (and their Compact twin where compact_code() asks for it); ZKMI_G2_SPLIT_BLS=0: 14-limb only
            {
                static const bool split_on = !(getenv("ZKMI_G2_SPLIT") && atoi(getenv("ZKMI_G2_SPLIT")) == 0) &&
                                             !(Lim29<C>::NL > 9 && getenv("ZKMI_G2_SPLIT_BLS") && atoi(getenv("ZKMI_G2_SPLIT_BLS")) == 0);
                if (split_on) {
                    hipLaunchKernelGGL((k_msm_accum29_g2s<C>), dim3((unsigned)((pl.lane_bound + G2S_SLOTS - 1) / G2S_SLOTS)), dim3(256), 0, st, (const uint32_t*)d_bases,
                                       d_infmask ? d_infmask : r29->second, sh, skip, pl.cap, pl.counts, pl.starts, pl.sorted, pl.lane_g, pl.lane_sub, pl.meta, buckets,
                                       lane_partials, (int)r29_buckets);
                    done = true;
                }
            }
            if constexpr (Lim29<C>::NL > 9) if (cc && !done) {
                hipLaunchKernelGGL((k_msm_accum29_g2<Compact<C>>), dim3((unsigned)((pl.lane_bound + T29 - 1) / T29)), dim3(T29), lds29, st, (const uint32_t*)d_bases,
                                   d_infmask ? d_infmask : r29->second, sh, skip, pl.cap, pl.counts, pl.starts, pl.sorted, pl.lane_g, pl.lane_sub, pl.meta, buckets,
                                   lane_partials, (int)r29_buckets);
                done = true;
            }
            if (!done) hipLaunchKernelGGL((k_msm_accum29_g2<C>), dim3((unsigned)((pl.lane_bound + T29 - 1) / T29)), dim3(T29), lds29, st, (const uint32_t*)d_bases,
                                    d_infmask ? d_infmask : r29->second, sh, skip, pl.cap, pl.counts, pl.starts, pl.sorted, pl.lane_g, pl.lane_sub, pl.meta, buckets, lane_partials,
                                    (int)r29_buckets);
        }
    }
    if (into && !launched && into->r29) return fail(ZKMI_ERR_INVALID, "msm_accumulate: merge target holds R'-form buckets");
    if constexpr (!WIDE) if (into && !launched) {
        launched = true;
        hipLaunchKernelGGL((k_msm_accum<F, false, true>), dim3((unsigned)((pl.lane_bound + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_bases, d_infmask, sh, skip, pl.cap, pl.counts,
                           pl.starts, pl.sorted, pl.lane_g, pl.lane_sub, pl.meta, buckets, lane_partials, prev_counts);
    }
    if (!launched)
        hipLaunchKernelGGL((k_msm_accum<F, WIDE, false>), dim3((unsigned)((pl.lane_bound + AT - 1) / AT)), dim3(AT), acc_lds, st, (const uint32_t*)d_bases, d_infmask, sh, skip, pl.cap, pl.counts,
                           pl.starts, pl.sorted, pl.lane_g, pl.lane_sub, pl.meta, buckets, lane_partials, prev_counts);
    if (job.acc1) ZK_HIP(hipEventRecord(job.acc1, st));
    if (cx.msm_stats && cx.d_addcount && job.slot < MSM_JOB_SLOTS) {           // outside the bracketed launch
        const int ci = cx.pipe * MSM_JOB_SLOTS + job.slot;
        const uint32_t* mask = d_infmask ? d_infmask : (table29 ? r29->second : nullptr);
        ZK_HIP(hipMemsetAsync(cx.d_addcount + ci, 0, 8, st));
        hipLaunchKernelGGL(k_msm_count_adds, dim3(1024), dim3(256), 0, st, pl.sorted, pl.counts, pl.starts, (uint32_t)total, skip, mask, cx.d_addcount + ci);
        ZK_HIP(hipMemcpyAsync(cx.h_addcount + ci, cx.d_addcount + ci, 8, hipMemcpyDeviceToHost, st));
    }
    hipLaunchKernelGGL((k_msm_tree<F, MSM_TB>), dim3((unsigned)std::min<size_t>(tree_blocks, 512)), dim3(MSM_TB), tree_lds, st, lane_partials, pl.lane_g, pl.counts, pl.cap, pl.meta, buckets,
                       block_partials, (int)r29_buckets);
    hipLaunchKernelGGL((k_msm_giant<F, MSM_TB>), dim3((unsigned)std::min<size_t>(tree_blocks, 256)), dim3(MSM_TB), tree_lds, st, pl.giants, pl.meta, block_partials, buckets, (int)r29_buckets);
    job.W = sh.W; job.c = sh.c; job.nb = sh.nb; job.buckets = buckets; job.counts = pl.counts;
    job.merged = into != nullptr;
    job.r29 = r29_buckets;
    if (into) {
        uint32_t* cmb;
        ZK_TRY(ws_get("msm.cmbcounts." + std::to_string(into->slot), total * 4, (void**)&cmb));
        hipLaunchKernelGGL(k_msm_counts_add, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, prev_counts, pl.counts, (uint32_t)total, cmb);
        into->counts = cmb;
    }
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

// ---- stage 5: bucket reduction of up to MSM_MAX_BATCH accumulated jobs of the same shape, in one set of launches ------
// aux: run on the library's auxiliary stream with its own scratch buffers (a latency-bound reduction can then overlap the
// throughput-bound accumulations of other MSMs); the caller orders the streams with events.
template <class F> int msm_reduce(MsmJob* const* jobs, int njobs, bool aux = false) {
    constexpr int FW = FieldWords<F>::value, PW = 4 * FW;
    Ctx& cx = ctx();
    hipStream_t st = aux ? cx.aux_stream : cx.stream;
    const std::string ax = aux ? ".aux" : "";
    if (njobs < 1 || njobs > MSM_MAX_BATCH) return fail(ZKMI_ERR_INVALID, "msm_reduce: bad batch size");
    const int W = jobs[0]->W, c = jobs[0]->c;
    const uint32_t nb = jobs[0]->nb;
    MsmReduceBatch rb;
    rb.njobs = njobs;
    for (int i = 0; i < njobs; i++) {
        if (jobs[i]->W != W || jobs[i]->c != c) return fail(ZKMI_ERR_INVALID, "msm_reduce: jobs of different shape");
        rb.buckets[i] = jobs[i]->buckets; rb.counts[i] = jobs[i]->counts;
    }
    const uint32_t rbits = (uint32_t)(c - 1) / 2, cbits = (uint32_t)(c - 1) - rbits, C = 1u << cbits;
    const size_t VW = (size_t)njobs * W * 2;                           // arrays of C points: (job, window, row|col)
    constexpr int M = (PW * 4 * 2 * 256 <= 128 * 1024) ? 256 : 128;    // k_msm_wsum: 2 LDS arrays of M points
    const uint32_t m2 = (C + M - 1) / M;
    uint32_t *rc, *a0, *r0, *a1, *r1;
    ZK_TRY(ws_get("msm.rowcol" + ax, VW * C * PW * 4, (void**)&rc));
    ZK_TRY(ws_get("msm.redA0" + ax, std::max<size_t>(VW * m2, VW * (cbits + 1)) * PW * 4, (void**)&a0));
    ZK_TRY(ws_get("msm.redR0" + ax, VW * m2 * PW * 4, (void**)&r0));
    ZK_TRY(ws_get("msm.redA1" + ax, VW * PW * 4, (void**)&a1));
    ZK_TRY(ws_get("msm.redR1" + ax, VW * PW * 4, (void**)&r1));
    static bool attr_set = false;
    const size_t lds_ws = (size_t)2 * M * PW * 4;
    if (!attr_set) {
        ZK_HIP(hipFuncSetAttribute((const void*)k_msm_wsum<F, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws));
        attr_set = true;
    }
    const size_t n_out = VW * C;
    // Row/Col sums in stages: L lanes per sum (~16 additions each), then folds of 8 partials at a time
    // (Fq2 additions are ~50 us of latency each and the total work is small: spread wider and fold 4 at a time)
    constexpr bool WIDE_R = FW > 12;
    const uint32_t seq = WIDE_R ? 4 : 16, maxL = WIDE_R ? 256 : 64, foldK = WIDE_R ? 4 : 8;
    uint32_t L = 1;
    while (L < maxL && (C / L) > seq) L <<= 1;
    // one wave per sum with the fold inside the launch (k_msm_rowcol_wave) whenever every sum has >= 64 buckets;
    // ZKMI_ROWCOL_WAVE=0 keeps the staged k_msm_rowcol + k_msm_fold sequence
    static const bool wave_env = !(getenv("ZKMI_ROWCOL_WAVE") && atoi(getenv("ZKMI_ROWCOL_WAVE")) == 0);
    const bool wave_rc = wave_env && rbits >= 6 && cbits >= 6;
    bool all_r29 = true;                                     // R'-form buckets and the row / column sums on the same limbs
    for (int i = 0; i < njobs; i++) all_r29 = all_r29 && jobs[i]->r29;
    for (int i = 0; i < njobs; i++) if (jobs[i]->r29 != jobs[0]->r29) return fail(ZKMI_ERR_INVALID, "msm_reduce: jobs with different bucket formats");
    if (jobs[0]->r29 && !(all_r29 && wave_rc)) return fail(ZKMI_ERR_UNSUPPORTED, "msm_reduce: R'-form buckets need the wave row/column sums");
    if (all_r29 && wave_rc) {
        if constexpr (FW <= 12) {
            typedef typename F::Cfg C;
            constexpr size_t lds29 = (size_t)256 * 4 * Lim29<C>::NL * 4;
            static bool rc29_attr = false;
            if (!rc29_attr) {
                ZK_HIP(hipFuncSetAttribute((const void*)k_msm_rowcol_wave29<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds29));
                if constexpr (Lim29<C>::NL > 9) ZK_HIP(hipFuncSetAttribute((const void*)k_msm_rowcol_wave29<Compact<C>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds29));
                rc29_attr = true;
            }
            size_t rc_blocks = (n_out + 3) / 4;
            static const int aux_cap29 = getenv("ZKMI_AUX_RC_SUMS") ? atoi(getenv("ZKMI_AUX_RC_SUMS")) : 512;
            if (aux && aux_cap29 > 0) rc_blocks = std::min<size_t>(rc_blocks, (size_t)aux_cap29 / 4);
            // slow-fetch box, 14-limb curve (254 KB inlined): 5.9 -> 3.0 ms there; BN254's 114 KB kernel gains nothing from the calls (measured)
            bool done = false;
            if constexpr (Lim29<C>::NL > 9) if (compact_code() & 4) {
                hipLaunchKernelGGL((k_msm_rowcol_wave29<Compact<C>>), dim3((unsigned)rc_blocks), dim3(256), lds29, st, rb, (uint32_t)W, nb, rbits, cbits, rc);
                done = true;
            }
            if (!done) hipLaunchKernelGGL((k_msm_rowcol_wave29<C>), dim3((unsigned)rc_blocks), dim3(256), lds29, st, rb, (uint32_t)W, nb, rbits, cbits, rc);
        } else {
            typedef typename F::Cfg C;
            constexpr int T = Reduce29G2<C>::T;
            constexpr size_t lds29 = Reduce29G2<C>::lds_bytes;
            static bool rc29_attr = false;
            if (!rc29_attr) { ZK_HIP(hipFuncSetAttribute((const void*)k_msm_rowcol_wave29_g2<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds29)); rc29_attr = true; }
            size_t rc_blocks = (n_out + T / 64 - 1) / (T / 64);
            static const int aux_cap29 = getenv("ZKMI_AUX_RC_SUMS") ? atoi(getenv("ZKMI_AUX_RC_SUMS")) : 512;
            if (aux && aux_cap29 > 0) rc_blocks = std::min<size_t>(rc_blocks, (size_t)aux_cap29 / (T / 64));
            hipLaunchKernelGGL((k_msm_rowcol_wave29_g2<C>), dim3((unsigned)rc_blocks), dim3(T), lds29, st, rb, (uint32_t)W, nb, rbits, cbits, rc);
        }
    } else if (wave_rc) {
        constexpr int T = MsmRcBlock<F>::value;
        const size_t lds_rc = (size_t)T * PW * 4;
        static bool rc_attr = false;
        if (!rc_attr) { ZK_HIP(hipFuncSetAttribute((const void*)k_msm_rowcol_wave<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rc)); rc_attr = true; }
        size_t rc_blocks = (n_out + T / 64 - 1) / (T / 64);
        // see k_msm_rowcol_wave: on the auxiliary stream at most aux_cap sums (waves) are in flight, a quarter of the chip's CUs at two
        // 256-lane blocks per CU; the rest of the CUs stay with the main stream
        static const int aux_cap = getenv("ZKMI_AUX_RC_SUMS") ? atoi(getenv("ZKMI_AUX_RC_SUMS")) : 512;
        if (aux && aux_cap > 0) rc_blocks = std::min<size_t>(rc_blocks, (size_t)aux_cap / (T / 64));
        hipLaunchKernelGGL((k_msm_rowcol_wave<F>), dim3((unsigned)rc_blocks), dim3(T), lds_rc, st, rb, (uint32_t)W, nb, rbits, cbits, rc);
    } else {
    uint32_t *p0, *p1;
    ZK_TRY(ws_get("msm.rcpart0" + ax, n_out * L * PW * 4, (void**)&p0));
    ZK_TRY(ws_get("msm.rcpart1" + ax, std::max<size_t>(n_out * L / 4, 1) * PW * 4, (void**)&p1));
    {
        uint32_t* dst = L == 1 ? rc : p0;
        hipLaunchKernelGGL((k_msm_rowcol<F>), dim3((unsigned)((n_out * L + 255) / 256)), dim3(256), 0, st, rb, (uint32_t)W, nb, rbits, cbits, L, dst);
        uint32_t parts = L;
        const uint32_t* src = dst;
        while (parts > 1) {
            const uint32_t K = std::min<uint32_t>(parts, foldK);
            parts /= K;
            uint32_t* d2 = parts == 1 ? rc : (src == p0 ? p1 : p0);
            hipLaunchKernelGGL((k_msm_fold<F>), dim3((unsigned)((n_out * parts + 255) / 256)), dim3(256), 0, st, src, d2, n_out * parts, K);
            src = d2;
        }
    }
    }
    const bool bitsums = VW * (cbits + 1) <= 256;            // few arrays (pre-computed tables): plain sums only, host does the weighting
    const uint32_t* resA = nullptr;
    const uint32_t* resR = nullptr;
    size_t perA = 0;                                         // words per job in resA
    if (bitsums) {
        constexpr int MB = (PW * 4 * 256 <= 64 * 1024) ? 256 : 128;
        static bool battr = false;
        if (!battr) { ZK_HIP(hipFuncSetAttribute((const void*)k_msm_bitsums<F, MB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MB * PW * 4))); battr = true; }
        if constexpr (WIDE_R) {
            constexpr int TB = MsmAccumBlock<F>::value;
            static bool blattr = false;
            if (!blattr) { ZK_HIP(hipFuncSetAttribute((const void*)k_msm_bitsums_lds<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(TB * PW * 4))); blattr = true; }
            if (wave_env) hipLaunchKernelGGL((k_msm_bitsums_lds<F>), dim3((unsigned)(VW * (cbits + 1))), dim3(TB), (size_t)TB * PW * 4, st, rc, C, cbits, a0);
            else hipLaunchKernelGGL((k_msm_bitsums<F, MB>), dim3((unsigned)(VW * (cbits + 1))), dim3(MB), (size_t)MB * PW * 4, st, rc, C, cbits, a0);
        } else
        hipLaunchKernelGGL((k_msm_bitsums<F, MB>), dim3((unsigned)(VW * (cbits + 1))), dim3(MB), (size_t)MB * PW * 4, st, rc, C, cbits, a0);
        resA = a0;
        perA = (size_t)2 * W * (cbits + 1) * PW;
    } else {
        uint32_t m = C;
        int log_scale = 0;
        const uint32_t *inA = nullptr, *inR = rc;
        uint32_t *outA = a0, *outR = r0;
        for (;;) {
            const uint32_t blocks = (m + M - 1) / M;
            hipLaunchKernelGGL((k_msm_wsum<F, M>), dim3((unsigned)(VW * blocks)), dim3(M), lds_ws, st, inA, inR, m, blocks, log_scale, outA, outR);
            inA = outA; inR = outR;
            if (blocks == 1) break;
            outA = (outA == a0) ? a1 : a0; outR = (outR == r0) ? r1 : r0;
            m = blocks;
            log_scale += ilog2_sz(M);
        }
        resA = inA; resR = inR;
        perA = (size_t)2 * W * PW;
    }
    for (int i = 0; i < njobs; i++) {
        MsmJob& job = *jobs[i];
        job.cbits = (int)cbits;
        job.bitsums = bitsums ? 1 : 0;
        if (perA * 4 * 2 > MSM_JOB_SLOT_BYTES) return fail(ZKMI_ERR_UNSUPPORTED, "msm: too many windows");
        ZK_HIP(hipMemcpyAsync(job.h_win, resA + (size_t)i * perA, perA * 4, hipMemcpyDeviceToHost, st));
        if (!bitsums) ZK_HIP(hipMemcpyAsync(job.h_win + perA, resR + (size_t)i * perA, perA * 4, hipMemcpyDeviceToHost, st));
    }
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
// after the stream has been synchronised
template <class F> void msm_fold(const MsmJob& job, uint8_t* out_jac) {
    constexpr int PW = 4 * FieldWords<F>::value;
    if (job.merged) { memset(out_jac, 0, 3 * PW); return; }              // its points are inside the merge target's result
    if (!job.bitsums) { msm_fold_windows<F>(job.h_win, job.h_win + (size_t)2 * job.W * PW, job.W, job.c, job.cbits, out_jac); return; }
    // plain bit sums -> weighted sums by Horner on the host, laid out as msm_fold_windows expects
    typedef typename HostOf<F>::FT FT;
    host::HCurve<FT> cv{HostOf<F>::make()};
    const int nb1 = job.cbits + 1;
    std::vector<uint32_t> win((size_t)2 * job.W * PW), tot((size_t)2 * job.W * PW);
    auto put = [&](uint32_t* dst, const typename host::HCurve<FT>::P& p) {      // Jacobian -> XYZZ words (ZZ = Z^2, ZZZ = Z^3)
        constexpr int FW = FieldWords<F>::value;
        if (cv.is_zero(p)) { memset(dst, 0, PW * 4); return; }
        auto z2 = cv.F.sqr(p.Z), z3 = cv.F.mul(z2, p.Z);
        memcpy(dst, &p.X, 4 * FW); memcpy(dst + FW, &p.Y, 4 * FW); memcpy(dst + 2 * FW, &z2, 4 * FW); memcpy(dst + 3 * FW, &z3, 4 * FW);
    };
    for (int a = 0; a < 2 * job.W; a++) {
        const uint32_t* src = job.h_win + (size_t)a * nb1 * PW;
        auto acc = cv.zero();
        for (int k = job.cbits - 1; k >= 0; k--) { if (!cv.is_zero(acc)) acc = cv.dbl(acc); acc = cv.add(acc, xyzz_to_jac<F>(cv, src + (size_t)k * PW)); }
        put(win.data() + (size_t)a * PW, acc);
        memcpy(tot.data() + (size_t)a * PW, src + (size_t)job.cbits * PW, PW * 4);
    }
    msm_fold_windows<F>(win.data(), tot.data(), job.W, job.c, job.cbits, out_jac);
}

// Full device MSM: bases (affine, device), scalars (plain integers, device) -> Jacobian point on the host.
template <class F> int msm_run(const void* d_bases, const void* d_scalars, size_t n, size_t sb, uint8_t* out_jac) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    if (n == 0) { memset(out_jac, 0, 3 * 4 * FW); return ZKMI_OK; }
    hipStream_t st = cx.stream;
    MsmPlan pl;
    MsmJob job;
    ZK_TRY(msm_job_slot(0, job));
    if (!cx.ev0_held) ZK_HIP(hipEventRecord(cx.ev0, st));
    ZK_TRY(msm_sort(d_scalars, n, sb, pl));
    ZK_TRY(msm_accumulate<F>(d_bases, pl, 0, job));
    MsmJob* jp = &job;
    ZK_TRY(msm_reduce<F>(&jp, 1));
    ZK_HIP(hipEventRecord(cx.ev1, st));
    ZK_HIP(hipStreamSynchronize(st));
    float ms = 0;
    if (hipEventElapsedTime(&ms, cx.ev0, cx.ev1) == hipSuccess) cx.last_ms = ms;
    msm_fold<F>(job, out_jac);
    return ZKMI_OK;
}

// Window width and digit count of a pre-computed table for n resident bases and sb-byte scalars
static inline int msm_precomp_c(size_t n) {
    int lg = ilog2_sz(n ? n : 1);
    if (lg < 62 && n > (((size_t)3) << lg) / 2) lg++;
    int c = std::max(8, std::min(21, lg));
    // 254/255-bit scalars: keep the top digit window well filled (a 2..7-bit top window funnels millions of scalars into a
    // handful of buckets and serialises the sort's atomics): for large n snap to c in {15, 16, 17, 20}
    if (lg >= 14) { static const int snap[] = {15, 15, 16, 17, 17, 20, 20, 20}; c = snap[std::min(std::max(c, 14), 21) - 14]; }
    return c;
}
static inline int msm_digits(size_t sb, int c) { return (int)((8 * sb + 1 + c - 1) / c); }
// d_table: Wd * n affine points (device). Built once per resident base set.
template <class F> int msm_precompute(const void* d_bases, size_t n, int c, int Wd, void* d_table) {
    Ctx& cx = ctx();
    hipLaunchKernelGGL((k_msm_precompute<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, (const uint32_t*)d_bases, (uint32_t)n, c, Wd, (uint32_t*)d_table);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
int msm_precompute_dispatch(int curve, int group, const void* d_bases, size_t n, int c, int Wd, void* d_table);
template <class F> int msm_infmask(const void* d_points, size_t n, uint32_t* d_mask) {
    Ctx& cx = ctx();
    ZK_HIP(hipMemsetAsync(d_mask, 0, ((n + 31) / 32) * 4, cx.stream));
    if (n) hipLaunchKernelGGL((k_msm_infmask<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, (const uint32_t*)d_points, n, d_mask);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
// MSM of the first k scalars against a pre-computed table (stride points per row, window width c)
template <class F> int msm_run_table(const void* d_table, size_t stride, int c, const void* d_scalars, size_t k, size_t sb, uint8_t* out_jac);
int msm_table_dispatch(int curve, int group, const void* d_table, size_t stride, int c, const void* d_scalars, size_t k, size_t sb, uint8_t* out_jac);
// Several MSMs against the same table (independent scalar vectors): one sort + accumulation each, ONE batched bucket reduction
int msm_table_multi_dispatch(int curve, int group, const void* d_table, size_t stride, int c, const void* const* d_scalars, const size_t* ks, int count, size_t sb,
                             uint8_t* out_jacs);

// non-template entry points (msm_bn254.hip / msm_bls12381.hip) for callers that must not instantiate the kernels again
int msm_accumulate_dispatch(int curve, int group, const void* d_bases, const MsmPlan& pl, uint32_t skip, MsmJob& job, const uint32_t* d_infmask = nullptr, MsmJob* into = nullptr);
// d_mask: ceil(n/32) words, zeroed by the callee; bit i = point i is the point at infinity
int msm_infmask_dispatch(int curve, int group, const void* d_points, size_t n, uint32_t* d_mask);
int msm_reduce_dispatch(int curve, int group, MsmJob* const* jobs, int njobs, bool aux = false);

int msm_fold_dispatch(int curve, int group, const MsmJob& job, uint8_t* out_jac);

template <class F> int msm_run_table(const void* d_table, size_t stride, int c, const void* d_scalars, size_t k, size_t sb, uint8_t* out_jac) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    if (k == 0) { memset(out_jac, 0, 3 * 4 * FW); return ZKMI_OK; }
    hipStream_t st = cx.stream;
    MsmPlan pl;
    MsmJob job;
    ZK_TRY(msm_job_slot(0, job));
    ZK_HIP(hipEventRecord(cx.ev0, st));
    ZK_TRY(msm_sort(d_scalars, k, sb, pl, 0, c, stride));
    ZK_TRY(msm_accumulate<F>(d_table, pl, 0, job));
    MsmJob* jp = &job;
    ZK_TRY(msm_reduce<F>(&jp, 1));
    ZK_HIP(hipEventRecord(cx.ev1, st));
    ZK_HIP(hipStreamSynchronize(st));
    float ms = 0;
    if (hipEventElapsedTime(&ms, cx.ev0, cx.ev1) == hipSuccess) cx.last_ms = ms;
    msm_fold<F>(job, out_jac);
    return ZKMI_OK;
}

// The MSMs of one zkmi_msm_table_multi call that have been enqueued and not yet collected, per pipeline slot (r06: enqueue / collect split, so that a
// host that drives two proofs from one thread — plonk.py prove_many, js proveMany — can put the accumulations of proof B on the device underneath the
// latency-bound bucket reduction of proof A instead of waiting inside A's call)
struct MsmMultiPending {
    bool live = false;
    int count = 0, fw = 0;
    size_t ks[MSM_MAX_BATCH] = {};
    MsmJob job[MSM_MAX_BATCH];
};
MsmMultiPending& msm_multi_pending(int pipe);              // msm_sort.hip

template <class F> int msm_run_table_multi_enqueue(const void* d_table, size_t stride, int c, const void* const* d_scalars, const size_t* ks, int count, size_t sb) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    if (count < 1 || count > MSM_MAX_BATCH) return fail(ZKMI_ERR_INVALID, "msm_table_multi: 1..4 MSMs per call");
    MsmMultiPending& P = msm_multi_pending(cx.pipe);
    hipStream_t st = cx.stream;
    if (P.live) {                                                      // a call that was enqueued and never collected (its proof was abandoned after an error): drop it
        ZK_HIP(hipStreamSynchronize(st));
        P.live = false;
    }
    MsmPlan pl[MSM_MAX_BATCH];
    MsmJob* job = P.job;
    MsmJob* jp[MSM_MAX_BATCH];
    int live = 0;
    ZK_HIP(hipEventRecord(cx.ev0, st));
    // The digit sorts are memory-bound, the accumulations ALU-bound: with more than one MSM in the call the sorts go to the
    // auxiliary stream, so that sort i+1 runs underneath accumulation i (ZKMI_MULTI_OVERLAP=0 keeps one stream).
    static const bool ov_env = !(getenv("ZKMI_MULTI_OVERLAP") && atoi(getenv("ZKMI_MULTI_OVERLAP")) == 0);
    int n_live = 0;
    for (int i = 0; i < count; i++) n_live += ks[i] != 0;
    const bool ov = ov_env && n_live > 1;
    hipStream_t aux = nullptr;
    if (ov) {
        ZK_TRY(ensure_aux_stream());
        aux = cx.aux_stream;
        if (!cx.sort_ev[0]) for (int i = 0; i < 5; i++) ZK_HIP(hipEventCreateWithFlags(&cx.sort_ev[i], hipEventDisableTiming));
        ZK_HIP(hipEventRecord(cx.sort_ev[0], st));                   // the scalars are ready on the main stream at this point
        ZK_HIP(hipStreamWaitEvent(aux, cx.sort_ev[0], 0));
    }
    // r06: sort i and accumulation i are enqueued ALTERNATELY (sort 0, accumulation 0, sort 1, ...). Enqueueing every sort first left the main stream empty for
    // as long as the host needed to launch them (16 launches each: 0.35 ms between the end of sort 0 and the start of accumulation 0 in the kernel trace of a PLONK round).
    for (int i = 0; i < count; i++) {
        job[i] = MsmJob();
        if (ks[i] == 0) continue;
        ZK_TRY(msm_job_slot(i, job[i]));
        if (ov) {
            cx.stream = aux;
            int rc = msm_sort(d_scalars[i], ks[i], sb, pl[i], i, c, stride);
            if (!rc && hipEventRecord(cx.sort_ev[1 + i], aux) != hipSuccess) rc = fail(ZKMI_ERR_HIP, "hipEventRecord");
            cx.stream = st;
            ZK_TRY(rc);
            ZK_HIP(hipStreamWaitEvent(st, cx.sort_ev[1 + i], 0));
        } else ZK_TRY(msm_sort(d_scalars[i], ks[i], sb, pl[i], i, c, stride));
        ZK_TRY(msm_accumulate<F>(d_table, pl[i], 0, job[i]));
        jp[live++] = &job[i];
    }
    if (live) ZK_TRY(msm_reduce<F>(jp, live));
    ZK_HIP(hipEventRecord(cx.ev1, st));
    P.live = true; P.count = count; P.fw = FW;
    for (int i = 0; i < count; i++) P.ks[i] = ks[i];
    return ZKMI_OK;
}
template <class F> int msm_run_table_multi_collect(int count, uint8_t* out_jacs) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    MsmMultiPending& P = msm_multi_pending(cx.pipe);
    if (!P.live) return fail(ZKMI_ERR_INVALID, "msm_table_multi: nothing enqueued in this pipeline slot");
    if (P.fw != FW || P.count != count) { return fail(ZKMI_ERR_INVALID, "msm_table_multi: collect does not match what was enqueued (table or count)"); }
    P.live = false;
    ZK_HIP(hipStreamSynchronize(cx.stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, cx.ev0, cx.ev1) == hipSuccess) cx.last_ms = ms;
    for (int i = 0; i < P.count; i++) {
        if (P.ks[i] == 0) memset(out_jacs + (size_t)i * 12 * FW, 0, 12 * FW);
        else msm_fold<F>(P.job[i], out_jacs + (size_t)i * 12 * FW);
    }
    return ZKMI_OK;
}
template <class F> int msm_run_table_multi(const void* d_table, size_t stride, int c, const void* const* d_scalars, const size_t* ks, int count, size_t sb, uint8_t* out_jacs) {
    int rc = msm_run_table_multi_enqueue<F>(d_table, stride, c, d_scalars, ks, count, sb);
    if (rc) { msm_multi_pending(ctx().pipe).live = false; return rc; }
    return msm_run_table_multi_collect<F>(count, out_jacs);
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

template <class F> int gen_scalar_bases_run(const uint8_t* gen_affine_host, const void* d_scalars, size_t n, void* d_out) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    uint32_t* d_gen;
    ZK_TRY(ws_get("gen.generator", 2 * FW * 4, (void**)&d_gen));
    ZK_HIP(hipMemcpyAsync(d_gen, gen_affine_host, 2 * FW * 4, hipMemcpyHostToDevice, cx.stream));
    hipLaunchKernelGGL((k_gen_scalar_bases<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, d_gen, (const uint32_t*)d_scalars, (uint32_t)n, (uint32_t*)d_out);
    ZK_HIP(hipStreamSynchronize(cx.stream));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

}  // namespace zkmi
