// snarkjs_amd/csrc/zkmi_common.hpp — library context shared by the translation units of libzkmi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/zkmi.h"
#include "../../include/zkmi_diag.h"

namespace zkmi {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define ZK_HIP(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return ::zkmi::fail(ZKMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
#define ZK_TRY(expr)            \
    do {                        \
        int _rc = (expr);       \
        if (_rc) return _rc;    \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct NttPlan {
    unsigned log_n = 0;
    int n_pass = 0;
    uint32_t l[4] = {0, 0, 0, 0};
    uint32_t log_lb = 0;
    uint32_t *T_lo = nullptr, *T_hi = nullptr, *T_hi_last = nullptr, *n_inv = nullptr;
    uint32_t* LT[4] = {nullptr, nullptr, nullptr, nullptr};
    // the same tables in the R'-form of field29.cuh (t 2^261 mod r, packed canonical words) for the passes of ntt29.cuh; LT29 in bit-reversed order
    uint32_t *T_lo29 = nullptr, *T_hi29 = nullptr, *T_hi_last29 = nullptr, *n_inv29 = nullptr;
    uint32_t* LT29[4] = {nullptr, nullptr, nullptr, nullptr};
};

// Per-pipeline-slot resources. Two Groth16 proofs can be in flight (zkmi_groth16_submit_dev / _collect): the latency-bound tail of
// proof k (bucket reductions, result copies, host folds) then runs underneath the throughput-bound front of proof k+1. Each slot
// owns its streams, events, pinned result slots and — through ws_get, which prefixes buffer names with the slot — its scratch
// buffers. select_pipe() swaps the slot's resources into the Ctx fields the kernel drivers use.
struct PipeRes {
    bool init = false;
    hipStream_t own_stream = nullptr, stream = nullptr, aux_stream = nullptr;
    hipEvent_t aux_ev[2] = {}, sort_ev[5] = {}, job_ev[16] = {};
    uint8_t* pinned = nullptr;
};

struct Ctx {
    bool ready = false;
    int pipe = 0;                                           // active pipeline slot (0 | 1)
    PipeRes saved[2];
    int device = -1;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_ms = 0.0;
    bool ev0_held = false;                                  // the caller of msm_run has recorded ev0 already (zkmi_msm_dev: its conversion pass is inside the reported time)
    int msm_c_override = 0;
    std::map<std::string, DevBuf> ws;                       // named scratch buffers (grow-only)
    std::map<std::tuple<int, unsigned, int>, NttPlan> plans;  // (curve, log_n, inverse)
    std::map<std::string, DevBuf> ntt_prescale;             // cached row-factor tables of fused NTT pre-scales
    std::map<void*, std::pair<size_t, int>> user_allocs;    // zkmi_dev_alloc blocks: (bytes, pipeline slot that allocated it)
    std::map<size_t, std::vector<void*>> pool[2];           // freed zkmi_dev_alloc blocks by size, per pipeline slot (reuse is stream-ordered within a slot)
    size_t pool_bytes = 0, pool_limit = (size_t)96 << 30;
    std::map<uint64_t, void*> groth16;                      // zkmi_groth16 resident keys (groth16.hip)
    // window tables kept in the R'-form of field29.cuh (base pointer -> infinity bitmap): msm_accumulate runs them through k_msm_accum29
    std::map<const void*, const uint32_t*> r29_tables;
    hipStream_t aux_stream = nullptr;                       // second stream for latency-bound reductions (groth16.hip)
    hipEvent_t aux_ev[2] = {};
    hipEvent_t sort_ev[5] = {};                             // groth16.hip: digit sorts on the auxiliary stream (start, B, witness, join, H)
    hipEvent_t job_ev[16] = {};                             // per MSM job slot: events around k_msm_accum
    uint8_t* pinned = nullptr;                              // pinned host slots for MSM window sums
    bool msm_stats = false;                                 // zkmi_msm_stats: count the mixed additions of every accumulation launch
    unsigned long long *d_addcount = nullptr, *h_addcount = nullptr;      // per (pipeline slot, job slot): device counters, pinned host copies
};
Ctx& ctx();
int require_ctx();
int select_pipe(int p);
// What changes on a box that fetches instructions slowly beyond the instruction cache — bit 0: G1 accumulation of the 14-limb curve runs its
// Compact instantiation (field29.cuh), 1: the same for its G2 accumulation, 2: Compact G1 row/column sums of that curve, 3: the Fq2 row/column
// sums go back to the generic 32-bit kernel, 4: PLONK's quotient numerator by the 32-bit kernels with called products (8 - 12 KB per part) instead of the inlined 29-bit ones (41 - 52 KB). ZKMI_COMPACT_CODE=<mask> fixes it; otherwise the box is probed once
// (zkmi_calibrate_code_fetch: a 210 KB loop against a 17 KB loop of the same products) and bits 1, 2, 3 are set when the big loop runs below 0.85 (r05: bits 0 and 4 no longer —
// those loops fit the instruction cache since r04 and measured faster inlined on a slow-fetch box, profiles/r05_slow_fetch_box_ab.txt).
int compact_code();
int dev_alloc_big(void** p, size_t bytes);                // hipMalloc; ZKMI_CONTIG=1: physically contiguous VRAM first (zkmi_api.hip: measured slower)
int ensure_aux_stream();                                  // creates Ctx::aux_stream (+ aux_ev) on first use                                     // make pipeline slot p (0 | 1) the active one
// scratch buffer `name` with at least `bytes` capacity (contents undefined)
int ws_get(const std::string& name, size_t bytes, void** out);

// pages <-> device
int upload_pages(const zkmi_pages& pg, size_t total_bytes, void* d_dst);
int download_pages(const void* d_src, size_t total_bytes, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages);

// per-module entry points (implemented in msm_*.hip / ntt.hip)
int msm_dev_dispatch(int curve, int group, const void* d_bases, const void* d_scalars, size_t n, size_t sb, uint8_t* out_jac);
int gen_bases_dispatch(int curve, int group, size_t n, uint64_t f, uint64_t g, void* d_out);
int ntt_dev_dispatch(int curve, const void* d_in, void* d_out, unsigned log_n, int inverse, const uint8_t* first, const uint8_t* inc);
// `batch` transforms of one size per launch: member k at d_in + k*in_stride / d_out + k*out_stride ELEMENTS
int ntt_dev_batch_dispatch(int curve, const void* d_in, size_t in_stride, void* d_out, size_t out_stride, unsigned batch, unsigned log_n, int inverse, const uint8_t* first,
                           const uint8_t* inc);
int ntt_dev_padded_dispatch(int curve, const void* d_in, size_t in_len, void* d_out, unsigned log_n, int inverse);
// Fr.batchTo/FromMontgomery on up to four arrays in one launch (plonk.hip)
int fr_convert_multi_dispatch(int curve, int op, const void* const* d_in, void* const* d_out, const size_t* ns, int count);
int apply_key_dev_dispatch(int curve, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc);
int fr_batch_dev_dispatch(int curve, int op, const void* d_in, void* d_out, size_t n);
int join_abc_dev_dispatch(int curve, const void* a, const void* b, const void* c, void* out, size_t n);
int to_affine_dispatch(int curve, int group, const uint8_t* jac, uint8_t* aff);
// inc of the Groth16 coset step: Fr.shift when power == Fr.s, else Fr.w[power+1] (src/groth16_prove.js:64), Montgomery bytes
int fr_coset_inc(int curve, unsigned power, uint8_t* out32);
int fr_root(int curve, unsigned i, uint8_t* out32);      // Fr.w[i], Montgomery bytes
// Convert a freshly built window table (n_points affine points, R-form) to the R'-form of field29.cuh in place and register it together
// with its infinity bitmap; returns without doing anything for curves / groups that have no 29-bit path (or with ZKMI_R29=0).
int msm_table_to_r29(int curve, int group, void* d_table, size_t n_points, const uint32_t* d_infmask);
void msm_table_forget_r29(const void* d_table);
int ntt_power_tables(int curve, unsigned L, int inverse, const uint32_t** T_lo, const uint32_t** T_hi, uint32_t* log_lb, const uint32_t** n_inv);

inline int n8q_of(int curve) { return curve == ZKMI_CURVE_BN128 ? 32 : 48; }

}  // namespace zkmi
