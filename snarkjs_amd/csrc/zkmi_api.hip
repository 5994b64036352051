// snarkjs_amd/csrc/zkmi_api.hip — C-ABI entry points of libzkmi.so (include/zkmi.h): context, memory, page
// marshalling and dispatch to the per-curve kernel drivers.  No CPU fallback: every compute entry point requires a
// HIP device and fails with ZKMI_ERR_NO_DEVICE otherwise.
#include <string.h>
#include <thread>
#include <algorithm>
#include "msm_host.hpp"
#include "zkmi_common.hpp"

namespace zkmi {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }

static Ctx g_ctx;
Ctx& ctx() { return g_ctx; }

// Every entry point passes through here: a thread that has not called into the library before (e.g. a libuv pool thread of the
// N-API addon's *Async functions) is bound to the context's device first — HIP's current device is per thread.
int require_ctx() {
    static thread_local bool bound = false;
    if (!g_ctx.ready) ZK_TRY(zkmi_init(0));
    if (!bound) { ZK_HIP(hipSetDevice(g_ctx.device)); bound = true; }
    return ZKMI_OK;
}

int select_pipe(int p) {
    Ctx& cx = g_ctx;
    if (p < 0 || p > 1) return fail(ZKMI_ERR_INVALID, "pipeline slot must be 0 or 1");
    if (p == cx.pipe) return ZKMI_OK;
    PipeRes& cur = cx.saved[cx.pipe];
    cur.init = true; cur.own_stream = cx.own_stream; cur.stream = cx.stream; cur.aux_stream = cx.aux_stream; cur.pinned = cx.pinned;
    memcpy(cur.aux_ev, cx.aux_ev, sizeof cur.aux_ev); memcpy(cur.sort_ev, cx.sort_ev, sizeof cur.sort_ev); memcpy(cur.job_ev, cx.job_ev, sizeof cur.job_ev);
    PipeRes& nx = cx.saved[p];
    if (!nx.init) {
        ZK_HIP(hipStreamCreateWithFlags(&nx.own_stream, hipStreamNonBlocking));
        nx.stream = nx.own_stream;
        nx.init = true;
    }
    cx.own_stream = nx.own_stream; cx.stream = nx.stream; cx.aux_stream = nx.aux_stream; cx.pinned = nx.pinned;
    memcpy(cx.aux_ev, nx.aux_ev, sizeof nx.aux_ev); memcpy(cx.sort_ev, nx.sort_ev, sizeof nx.sort_ev); memcpy(cx.job_ev, nx.job_ev, sizeof nx.job_ev);
    cx.pipe = p;
    return ZKMI_OK;
}

// Allocation of the large device buffers (window tables, bucket arrays, NTT work arrays, the host's pooled buffers). ZKMI_CONTIG=1 asks for
// PHYSICALLY CONTIGUOUS VRAM first (hipExtMallocWithFlags + hipDeviceMallocContiguous, falling back to hipMalloc) — an experiment against the
// box-to-box spread of the gather- and stream-heavy kernels (r02: the same binary measured 3.3 ms and 5.8 ms for the G2 accumulation on two
// boxes). Measured on a box in the fast state (tools/gpu_ab9.sh): contiguous ranges are SLOWER — buildABC 0.37 vs 0.14 ms, the NTT chain 1.29 vs
// 1.11 ms, the G2 accumulation 3.5 vs 3.3 ms, 90.8 vs 99.8 proofs/s — presumably because a contiguous range interleaves over fewer HBM
// channels than the driver's default placement. Default: off.
extern "C" int zkmi_calibrate_code_fetch(double* small_loop_gmul_per_s, double* big_loop_gmul_per_s);
static int g_compact_code = -1;
int compact_code() {
    if (g_compact_code >= 0) return g_compact_code;
    if (getenv("ZKMI_COMPACT_CODE")) return g_compact_code = atoi(getenv("ZKMI_COMPACT_CODE")) & 31;
    double small = 0, big = 0;
    if (zkmi_calibrate_code_fetch(&small, &big) != ZKMI_OK || small <= 0) return g_compact_code = 0;
    // r05 (profiles/r05_slow_fetch_box_ab.txt, a box with ratio 0.80): since r04 shrank the inlined loops, the 14-limb G1 accumulation (38 KB loop) and
    // PLONK's 29-bit quotient kernels (41-52 KB per part) fit the instruction cache and run FASTER inlined on such a box than with called products
    // (BLS12-381 38.8-39.1 against 35.5-35.7 proofs/s, PLONK 38.2-38.3 against 36.7-37.1); the 14-limb G2 accumulation (118 KB) and the row / column
    // sums still want the called products there (all inlined: 28.5 proofs/s). So a slow-fetch box gets bits 1, 2, 3 — not 0 and 4.
    return g_compact_code = (big / small < 0.85) ? 14 : 0;
}
extern "C" int zkmi_compact_code(void) { return g_ctx.ready ? compact_code() : -1; }
int dev_alloc_big(void** p, size_t bytes) {
    static const bool contig = getenv("ZKMI_CONTIG") && atoi(getenv("ZKMI_CONTIG")) == 1;
    static const size_t min_bytes = getenv("ZKMI_CONTIG_MIN") ? (size_t)atoll(getenv("ZKMI_CONTIG_MIN")) : ((size_t)2 << 20);
    if (contig && bytes >= min_bytes) {
        if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess) return ZKMI_OK;
        (void)hipGetLastError();                                  // not available / no contiguous range: plain allocation
    }
    ZK_HIP(hipMalloc(p, bytes));
    return ZKMI_OK;
}

static bool g_ws_oom = false;            // the last ws_get failed in its ALLOCATION with out-of-memory (and in nothing else)
int ws_get(const std::string& name, size_t bytes, void** out) {
    DevBuf& b = g_ctx.ws[g_ctx.pipe ? "P1:" + name : name];
    g_ws_oom = false;
    if (b.cap < bytes) {
        if (b.p) { ZK_HIP(hipStreamSynchronize(g_ctx.stream)); ZK_HIP(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
        size_t cap = bytes + bytes / 8 + 256;
        const hipError_t e = hipMalloc(&b.p, cap);
        if (e != hipSuccess) {
            b.p = nullptr;
            g_ws_oom = e == hipErrorOutOfMemory;
            return fail(ZKMI_ERR_HIP, std::string("hipMalloc(") + std::to_string(cap) + " bytes, scratch \"" + name + "\"): " + hipGetErrorString(e));
        }
        b.cap = cap;
    }
    *out = b.p;
    return ZKMI_OK;
}

// drop the scratch buffer `name` of the active pipeline slot (a large one-off: the copy of a standalone MSM's bases)
static void ws_drop(const std::string& name) {
    auto it = g_ctx.ws.find(g_ctx.pipe ? "P1:" + name : name);
    if (it == g_ctx.ws.end() || !it->second.p) return;
    (void)hipStreamSynchronize(g_ctx.stream);
    (void)hipFree(it->second.p);
    g_ctx.ws.erase(it);
}

static size_t pages_total(const zkmi_pages& pg) { size_t t = 0; for (int i = 0; i < pg.n_pages; i++) t += pg.len[i]; return t; }

int upload_pages(const zkmi_pages& pg, size_t total_bytes, void* d_dst) {
    if (pages_total(pg) < total_bytes) return fail(ZKMI_ERR_INVALID, "input buffer shorter than n elements");
    size_t off = 0;
    for (int i = 0; i < pg.n_pages && off < total_bytes; i++) {
        size_t k = pg.len[i] < total_bytes - off ? pg.len[i] : total_bytes - off;
        if (k) ZK_HIP(hipMemcpyAsync((uint8_t*)d_dst + off, pg.ptr[i], k, hipMemcpyHostToDevice, g_ctx.stream));
        off += k;
    }
    return ZKMI_OK;
}
int download_pages(const void* d_src, size_t total_bytes, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages) {
    size_t cap = 0;
    for (int i = 0; i < n_out_pages; i++) cap += out_len[i];
    if (cap < total_bytes) return fail(ZKMI_ERR_INVALID, "output buffer too small");
    size_t off = 0;
    for (int i = 0; i < n_out_pages && off < total_bytes; i++) {
        size_t k = out_len[i] < total_bytes - off ? out_len[i] : total_bytes - off;
        if (k) ZK_HIP(hipMemcpyAsync(out_ptr[i], (const uint8_t*)d_src + off, k, hipMemcpyDeviceToHost, g_ctx.stream));
        off += k;
    }
    ZK_HIP(hipStreamSynchronize(g_ctx.stream));
    return ZKMI_OK;
}

int msm_bn254(int group, const void*, const void*, size_t, size_t, uint8_t*);
int msm_bls12381(int group, const void*, const void*, size_t, size_t, uint8_t*);
int msm_accumulate_bn254(int group, const void*, const MsmPlan&, uint32_t, MsmJob&, const uint32_t*, MsmJob*);
int msm_table_to_r29_bn254(int group, void* d_table, size_t n_points);
int msm_table_to_r29_bls12381(int group, void* d_table, size_t n_points);
int msm_accumulate_bls12381(int group, const void*, const MsmPlan&, uint32_t, MsmJob&, const uint32_t*, MsmJob*);
int msm_infmask_bn254(int group, const void*, size_t, uint32_t*);
int msm_infmask_bls12381(int group, const void*, size_t, uint32_t*);
int msm_reduce_bn254(int group, MsmJob* const*, int, bool);
int msm_precompute_bn254(int group, const void*, size_t, int, int, void*);
int msm_precompute_bls12381(int group, const void*, size_t, int, int, void*);
int msm_table_bn254(int group, const void*, size_t, int, const void*, size_t, size_t, uint8_t*);
int msm_table_bls12381(int group, const void*, size_t, int, const void*, size_t, size_t, uint8_t*);
int msm_table_multi_bn254(int group, const void*, size_t, int, const void* const*, const size_t*, int, size_t, uint8_t*);
int msm_table_multi_bls12381(int group, const void*, size_t, int, const void* const*, const size_t*, int, size_t, uint8_t*);
int msm_table_multi_enqueue_bn254(int group, const void*, size_t, int, const void* const*, const size_t*, int, size_t);
int msm_table_multi_enqueue_bls12381(int group, const void*, size_t, int, const void* const*, const size_t*, int, size_t);
int msm_table_multi_collect_bn254(int group, int, uint8_t*);
int msm_table_multi_collect_bls12381(int group, int, uint8_t*);
int msm_reduce_bls12381(int group, MsmJob* const*, int, bool);
int msm_fold_bn254(int group, const MsmJob&, uint8_t*);
int msm_fold_bls12381(int group, const MsmJob&, uint8_t*);
int gen_bases_bn254(int group, size_t, uint64_t, uint64_t, void*);
int gen_scalar_bases_bn254(int group, const void*, size_t, void*);
int gen_scalar_bases_bls12381(int group, const void*, size_t, void*);
int gen_bases_bls12381(int group, size_t, uint64_t, uint64_t, void*);
int to_affine_bn254(int group, const uint8_t*, uint8_t*);
int point_add_bn254(int group, const uint8_t*, const uint8_t*, uint8_t*);
int point_add_bls12381(int group, const uint8_t*, const uint8_t*, uint8_t*);
int to_affine_bls12381(int group, const uint8_t*, uint8_t*);

static int check_cg(int curve, int group) {
    if (curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) return fail(ZKMI_ERR_INVALID, "unknown curve");
    if (group != 1 && group != 2) return fail(ZKMI_ERR_INVALID, "Invalid group");
    return ZKMI_OK;
}
int msm_dev_dispatch(int curve, int group, const void* d_bases, const void* d_scalars, size_t n, size_t sb, uint8_t* out) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? msm_bn254(group, d_bases, d_scalars, n, sb, out) : msm_bls12381(group, d_bases, d_scalars, n, sb, out);
}
int msm_accumulate_dispatch(int curve, int group, const void* d_bases, const MsmPlan& pl, uint32_t skip, MsmJob& job, const uint32_t* d_infmask, MsmJob* into) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? msm_accumulate_bn254(group, d_bases, pl, skip, job, d_infmask, into) : msm_accumulate_bls12381(group, d_bases, pl, skip, job, d_infmask, into);
}
int msm_infmask_dispatch(int curve, int group, const void* d_points, size_t n, uint32_t* d_mask) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? msm_infmask_bn254(group, d_points, n, d_mask) : msm_infmask_bls12381(group, d_points, n, d_mask);
}
int msm_precompute_dispatch(int curve, int group, const void* d_bases, size_t n, int c, int Wd, void* d_table) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? msm_precompute_bn254(group, d_bases, n, c, Wd, d_table) : msm_precompute_bls12381(group, d_bases, n, c, Wd, d_table);
}
int msm_table_to_r29(int curve, int group, void* d_table, size_t n_points, const uint32_t* d_infmask) {
    // ZKMI_R29=0: every curve on saturated 32-bit limbs; ZKMI_R29_BLS=0: BLS12-381 only; ZKMI_R29_G2=0: G2 tables only (A/B switches)
    static const bool on = !(getenv("ZKMI_R29") && atoi(getenv("ZKMI_R29")) == 0);
    static const bool bls_on = !(getenv("ZKMI_R29_BLS") && atoi(getenv("ZKMI_R29_BLS")) == 0);
    static const int g2_on = getenv("ZKMI_R29_G2") ? atoi(getenv("ZKMI_R29_G2")) : 1;
    if (!on || (group != 1 && !(group == 2 && g2_on)) || !d_infmask) return ZKMI_OK;
    if (curve == ZKMI_CURVE_BN128) ZK_TRY(msm_table_to_r29_bn254(group, d_table, n_points));
    else if (curve == ZKMI_CURVE_BLS12381 && bls_on) ZK_TRY(msm_table_to_r29_bls12381(group, d_table, n_points));
    else return ZKMI_OK;
    g_ctx.r29_tables[d_table] = d_infmask;
    return ZKMI_OK;
}
void msm_table_forget_r29(const void* d_table) { g_ctx.r29_tables.erase(d_table); }
int msm_table_dispatch(int curve, int group, const void* d_table, size_t stride, int c, const void* d_scalars, size_t k, size_t sb, uint8_t* out) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? msm_table_bn254(group, d_table, stride, c, d_scalars, k, sb, out) : msm_table_bls12381(group, d_table, stride, c, d_scalars, k, sb, out);
}
int ensure_aux_stream() {
    if (g_ctx.aux_stream) return ZKMI_OK;
    // highest priority: the auxiliary stream carries latency-bound work (few waves, long dependency chains) and memory-bound digit
    // sorts that must not queue behind the main stream's throughput-bound kernels for wave slots
    int prio_least = 0, prio_greatest = 0;
    ZK_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    ZK_HIP(hipStreamCreateWithPriority(&g_ctx.aux_stream, hipStreamNonBlocking, getenv("ZKMI_AUX_PRIO") ? atoi(getenv("ZKMI_AUX_PRIO")) : prio_greatest));
    ZK_HIP(hipEventCreateWithFlags(&g_ctx.aux_ev[0], hipEventDisableTiming));
    ZK_HIP(hipEventCreateWithFlags(&g_ctx.aux_ev[1], hipEventDisableTiming));
    return ZKMI_OK;
}
int msm_reduce_dispatch(int curve, int group, MsmJob* const* jobs, int njobs, bool aux) {
    ZK_TRY(check_cg(curve, group));
    if (aux) ZK_TRY(ensure_aux_stream());
    if (njobs == 0) return ZKMI_OK;                               // only makes sure the auxiliary stream exists
    return curve == ZKMI_CURVE_BN128 ? msm_reduce_bn254(group, jobs, njobs, aux) : msm_reduce_bls12381(group, jobs, njobs, aux);
}
int msm_fold_dispatch(int curve, int group, const MsmJob& job, uint8_t* out_jac) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? msm_fold_bn254(group, job, out_jac) : msm_fold_bls12381(group, job, out_jac);
}
int gen_bases_dispatch(int curve, int group, size_t n, uint64_t f, uint64_t g, void* d_out) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? gen_bases_bn254(group, n, f, g, d_out) : gen_bases_bls12381(group, n, f, g, d_out);
}
int to_affine_dispatch(int curve, int group, const uint8_t* jac, uint8_t* aff) {
    ZK_TRY(check_cg(curve, group));
    return curve == ZKMI_CURVE_BN128 ? to_affine_bn254(group, jac, aff) : to_affine_bls12381(group, jac, aff);
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

const char* zkmi_version(void) { return "snarkjs-amd 0.1 (gfx950)"; }
const char* zkmi_last_error(void) { return g_err.c_str(); }

int zkmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int zkmi_init(int device) {
    if (g_ctx.ready && (device < 0 || device == g_ctx.device)) return ZKMI_OK;
    if (g_ctx.ready) return fail(ZKMI_ERR_INVALID, "zkmi_init: already bound to another device (one process per GPU)");
    int n = zkmi_device_count();
    if (n <= 0) return fail(ZKMI_ERR_NO_DEVICE, "no HIP device visible: the MI355X backend has no CPU fallback");
    if (device < 0) device = 0;
    if (device >= n) return fail(ZKMI_ERR_INVALID, "zkmi_init: device index out of range");
    ZK_HIP(hipSetDevice(device));
    ZK_HIP(hipStreamCreateWithFlags(&g_ctx.own_stream, hipStreamNonBlocking));
    g_ctx.stream = g_ctx.own_stream;
    ZK_HIP(hipEventCreate(&g_ctx.ev0));
    ZK_HIP(hipEventCreate(&g_ctx.ev1));
    g_ctx.device = device;
    g_ctx.ready = true;
    return ZKMI_OK;
}
int zkmi_set_stream(void* s) {
    ZK_TRY(require_ctx());
    g_ctx.stream = s ? (hipStream_t)s : g_ctx.own_stream;
    return ZKMI_OK;
}
int zkmi_synchronize(void) {
    ZK_TRY(require_ctx());
    ZK_HIP(hipStreamSynchronize(g_ctx.stream));
    return ZKMI_OK;
}
double zkmi_last_kernel_ms(void) {
    if (!g_ctx.ready) return 0.0;
    if (hipEventSynchronize(g_ctx.ev1) != hipSuccess) return 0.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, g_ctx.ev0, g_ctx.ev1) != hipSuccess) return g_ctx.last_ms;
    return ms;
}
double zkmi_msm_accum_ms(int slot) {
    if (!g_ctx.ready || slot < 0 || slot >= 8 || !g_ctx.job_ev[2 * slot]) return -1.0;
    float ms = 0;
    if (hipEventSynchronize(g_ctx.job_ev[2 * slot + 1]) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, g_ctx.job_ev[2 * slot], g_ctx.job_ev[2 * slot + 1]) != hipSuccess) return -1.0;
    return ms;
}
int zkmi_msm_stats(int enable) {
    ZK_TRY(require_ctx());
    if (enable && !g_ctx.d_addcount) {
        ZK_HIP(hipMalloc((void**)&g_ctx.d_addcount, 2 * MSM_JOB_SLOTS * 8));
        ZK_HIP(hipHostMalloc((void**)&g_ctx.h_addcount, 2 * MSM_JOB_SLOTS * 8, hipHostMallocDefault));
        memset(g_ctx.h_addcount, 0, 2 * MSM_JOB_SLOTS * 8);
    }
    g_ctx.msm_stats = enable != 0;
    return ZKMI_OK;
}
double zkmi_msm_accum_additions(int slot) {
    if (!g_ctx.ready || !g_ctx.h_addcount || slot < 0 || slot >= MSM_JOB_SLOTS) return -1.0;
    return (double)g_ctx.h_addcount[g_ctx.pipe * MSM_JOB_SLOTS + slot];
}
// Device buffers handed to the host are pooled by size: a prover allocates and drops the same few sizes every proof, and
// hipMalloc / hipFree are synchronous and slow (and the first touch of fresh VRAM costs milliseconds of page-table set-up).
struct FreeFence { hipEvent_t ev[2] = {nullptr, nullptr}; bool armed[2] = {false, false}; };
static std::map<void*, FreeFence> g_free_fence;              // pooled zkmi_dev_alloc blocks: what the other slot had queued when they were freed
int zkmi_dev_alloc(size_t bytes, void** d_ptr) {
    ZK_TRY(require_ctx());
    if (!bytes) bytes = 1;
    auto& pool = g_ctx.pool[g_ctx.pipe];
    auto it = pool.find(bytes);
    *d_ptr = nullptr;
    if (it != pool.end()) {
        // Oldest block first, and only a block whose fences have completed: a block freed while the OTHER pipeline slot still had work queued
        // that may read it (zkmi_dev_free records that slot's streams) is not handed out before that work is done. Nothing ever WAITS for a
        // fence — a proof that frees a buffer and asks for the same size again must not stall behind the other proof in flight; when every
        // pooled block of the size is still fenced a fresh one is allocated and the pool grows to its steady state.
        auto& v = it->second;
        for (size_t i = 0; i < v.size() && !*d_ptr; i++) {
            bool ready = true;
            auto fe = g_free_fence.find(v[i]);
            if (fe != g_free_fence.end())
                for (int k = 0; k < 2; k++) if (fe->second.armed[k]) {
                    if (hipEventQuery(fe->second.ev[k]) == hipSuccess) fe->second.armed[k] = false;
                    else { ready = false; (void)hipGetLastError(); }       // hipErrorNotReady is an answer, not a failure: keep it out of the error state
                }
            if (ready) { *d_ptr = v[i]; v.erase(v.begin() + (long)i); g_ctx.pool_bytes -= bytes; }
        }
    }
    if (!*d_ptr) ZK_TRY(dev_alloc_big(d_ptr, bytes));
    g_ctx.user_allocs[*d_ptr] = std::make_pair(bytes, g_ctx.pipe);
    return ZKMI_OK;
}
int zkmi_dev_free(void* d_ptr) {
    ZK_TRY(require_ctx());
    if (!d_ptr) return ZKMI_OK;
    auto it = g_ctx.user_allocs.find(d_ptr);
    if (it == g_ctx.user_allocs.end()) return fail(ZKMI_ERR_INVALID, "zkmi_dev_free: not a zkmi_dev_alloc pointer");
    const size_t bytes = it->second.first;
    const int slot = it->second.second;
    g_ctx.user_allocs.erase(it);
    // Stream-ordered reuse: a freed block may still be read by kernels queued on its slot's stream, so it goes back to the pool of the slot
    // that allocated it (whichever slot is active when the host drops it) and is only ever handed out to work queued behind those kernels.
    if (g_ctx.pool_bytes + bytes <= g_ctx.pool_limit) {
        // Buffers shared by both slots (a witness or key-side array allocated in slot 0 and read by kernels queued in slot 1) may still be in
        // use on the OTHER slot's streams when the host drops them. The block keeps a fence per stream of the other slot (events recorded
        // here); whoever takes it out of the pool waits for them on its own stream (zkmi_dev_alloc). Nothing waits at free time: an early
        // version made the allocating slot's stream wait here and thereby serialised the two proofs in flight (PLONK 38.4 -> 36.0 proofs/s).
        const int other = 1 - slot;
        static const bool fence_on = !(getenv("ZKMI_POOL_FENCE") && atoi(getenv("ZKMI_POOL_FENCE")) == 0);      // A/B switch of the fences' host cost
        const bool other_live = fence_on && (other == g_ctx.pipe ? true : g_ctx.saved[other].init);
        if (other_live) {
            FreeFence& ff = g_free_fence[d_ptr];
            hipStream_t theirs[2] = {other == g_ctx.pipe ? g_ctx.stream : g_ctx.saved[other].stream, other == g_ctx.pipe ? g_ctx.aux_stream : g_ctx.saved[other].aux_stream};
            for (int k = 0; k < 2; k++) {
                if (!theirs[k]) continue;
                if (!ff.ev[k]) ZK_HIP(hipEventCreateWithFlags(&ff.ev[k], hipEventDisableTiming));
                ZK_HIP(hipEventRecord(ff.ev[k], theirs[k]));
                ff.armed[k] = true;
            }
        }
        g_ctx.pool[slot][bytes].push_back(d_ptr);
        g_ctx.pool_bytes += bytes;
    } else {
        if (g_ctx.saved[slot].init && slot != g_ctx.pipe) (void)hipStreamSynchronize(g_ctx.saved[slot].stream);
        auto fe = g_free_fence.find(d_ptr);
        if (fe != g_free_fence.end()) { for (auto e : fe->second.ev) if (e) (void)hipEventDestroy(e); g_free_fence.erase(fe); }
        ZK_HIP(hipFree(d_ptr));
    }
    return ZKMI_OK;
}
// Host-orchestrated provers (PLONK, FFLONK) with two proofs in flight from ONE host thread: every library call works on the active pipeline
// slot — its stream and events, its scratch buffers (ws_get prefixes the names), its pool of zkmi_dev_alloc blocks, its ring of constants.
int zkmi_pipeline_select(int slot) {
    ZK_TRY(require_ctx());
    return select_pipe(slot);
}
int zkmi_pipeline_active(void) { return g_ctx.ready ? g_ctx.pipe : 0; }
int zkmi_memcpy_h2d(void* d, const void* h, size_t bytes) {
    ZK_TRY(require_ctx());
    ZK_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, g_ctx.stream));
    ZK_HIP(hipStreamSynchronize(g_ctx.stream));
    return ZKMI_OK;
}
int zkmi_memcpy_d2h(void* h, const void* d, size_t bytes) {
    ZK_TRY(require_ctx());
    ZK_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, g_ctx.stream));
    ZK_HIP(hipStreamSynchronize(g_ctx.stream));
    return ZKMI_OK;
}

int zkmi_host_register(void* host_ptr, size_t bytes) {
    ZK_TRY(require_ctx());
    if (!host_ptr || !bytes) return fail(ZKMI_ERR_INVALID, "host_register: null or empty range");
    ZK_HIP(hipHostRegister(host_ptr, bytes, hipHostRegisterPortable));
    return ZKMI_OK;
}
int zkmi_host_unregister(void* host_ptr) {
    ZK_TRY(require_ctx());
    if (!host_ptr) return ZKMI_OK;
    ZK_HIP(hipHostUnregister(host_ptr));
    return ZKMI_OK;
}
int zkmi_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes) {
    ZK_TRY(require_ctx());
    if (bytes) ZK_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, g_ctx.stream));
    return ZKMI_OK;
}
int zkmi_memset_dev(void* d_dst, int value, size_t bytes) {
    ZK_TRY(require_ctx());
    if (bytes) ZK_HIP(hipMemsetAsync(d_dst, value, bytes, g_ctx.stream));
    return ZKMI_OK;
}
int zkmi_gen_bases_from_scalars_dev(int curve, int group, const void* d_scalars, size_t n, void* d_out) {
    ZK_TRY(require_ctx());
    ZK_TRY(check_cg(curve, group));
    if (!d_scalars || !d_out) return fail(ZKMI_ERR_INVALID, "gen_bases_from_scalars: null argument");
    if (!n) return ZKMI_OK;
    return curve == ZKMI_CURVE_BN128 ? gen_scalar_bases_bn254(group, d_scalars, n, d_out) : gen_scalar_bases_bls12381(group, d_scalars, n, d_out);
}
int zkmi_msm_set_window_bits(int c) {
    if (c < 0 || c > 20) return fail(ZKMI_ERR_INVALID, "window bits must be 0 (auto) or 1..20");
    g_ctx.msm_c_override = c;
    return ZKMI_OK;
}
// ---- resident base tables --------------------------------------------------------------------------------------------------
struct MsmTable { void* p = nullptr; size_t n = 0; int c = 0, Wd = 0, curve = 0, group = 0; uint32_t* mask = nullptr; };
static std::map<uint64_t, MsmTable> g_tables;
static uint64_t g_next_table = 1;
static int table_build(int curve, int group, const void* d_bases, size_t n, MsmTable& t) {
    const size_t pb = (size_t)2 * group * n8q_of(curve);
    t.curve = curve; t.group = group; t.n = n;
    // ZKMI_TABLE_C=<c>: window width of the tables built here (A/B experiments: PLONK's SRS table at c = 17 against the built-in 20)
    static const int c_env = getenv("ZKMI_TABLE_C") ? atoi(getenv("ZKMI_TABLE_C")) : 0;
    t.c = c_env > 0 ? c_env : msm_precomp_c(n);
    t.Wd = msm_digits(32, t.c);
    if ((size_t)t.Wd * n >= (1ull << 31)) return fail(ZKMI_ERR_UNSUPPORTED, "msm table: too many points");
    ZK_TRY(dev_alloc_big(&t.p, (size_t)t.Wd * n * pb));
    ZK_TRY(msm_precompute_dispatch(curve, group, d_bases, n, t.c, t.Wd, t.p));
    ZK_HIP(hipStreamSynchronize(g_ctx.stream));
    return ZKMI_OK;
}
static void table_free(MsmTable& t) {
    if (t.p) { msm_table_forget_r29(t.p); (void)hipFree(t.p); }
    if (t.mask) (void)hipFree(t.mask);
    t = MsmTable();
}
// library-private tables (handles, the content-addressed cache behind zkmi_msm): kept in the R'-form of field29.cuh where that path exists
// (needs the infinity bitmap); on any failure nothing stays registered or allocated
static int table_build_r29(int curve, int group, const void* d_bases, size_t n, MsmTable& t) {
    int rc = table_build(curve, group, d_bases, n, t);
    if (!rc && hipMalloc((void**)&t.mask, (((size_t)t.Wd * n + 31) / 32) * 4 + 16) != hipSuccess) rc = fail(ZKMI_ERR_HIP, "hipMalloc: infinity bitmap of a window table");
    if (!rc) rc = msm_infmask_dispatch(curve, group, t.p, (size_t)t.Wd * n, t.mask);
    if (!rc) rc = msm_table_to_r29(curve, group, t.p, (size_t)t.Wd * n, t.mask);
    if (!rc && hipStreamSynchronize(g_ctx.stream) != hipSuccess) rc = fail(ZKMI_ERR_HIP, "hipStreamSynchronize: window table");
    if (rc) table_free(t);
    return rc;
}
int zkmi_msm_table_build(int curve, int group, const void* d_bases, size_t n, uint64_t* handle) {
    ZK_TRY(require_ctx());
    ZK_TRY(check_cg(curve, group));
    if (!handle || !d_bases || !n) return fail(ZKMI_ERR_INVALID, "msm_table_build: bad argument");
    MsmTable t;
    ZK_TRY(table_build_r29(curve, group, d_bases, n, t));
    *handle = g_next_table++;
    g_tables[*handle] = t;
    return ZKMI_OK;
}
int zkmi_msm_table_dev(uint64_t handle, const void* d_scalars, size_t k, size_t scalar_bytes, uint8_t* out) {
    ZK_TRY(require_ctx());
    auto it = g_tables.find(handle);
    if (it == g_tables.end()) return fail(ZKMI_ERR_INVALID, "msm_table_dev: unknown table");
    const MsmTable& t = it->second;
    if (!out) return fail(ZKMI_ERR_INVALID, "null output");
    if (k > t.n) return fail(ZKMI_ERR_INVALID, "msm_table_dev: more scalars than resident bases");
    if (scalar_bytes == 0 || scalar_bytes > 32) return fail(ZKMI_ERR_UNSUPPORTED, "msm_table_dev: tables are built for scalars of at most 32 bytes");
    return msm_table_dispatch(t.curve, t.group, t.p, t.n, t.c, d_scalars, k, scalar_bytes, out);
}
int zkmi_msm_table_multi_dev(uint64_t handle, const void* const* d_scalars, const size_t* ks, int count, size_t scalar_bytes, uint8_t* out_jacobians) {
    ZK_TRY(require_ctx());
    auto it = g_tables.find(handle);
    if (it == g_tables.end()) return fail(ZKMI_ERR_INVALID, "msm_table_multi_dev: unknown table");
    const MsmTable& t = it->second;
    if (!d_scalars || !ks || !out_jacobians) return fail(ZKMI_ERR_INVALID, "null argument");
    if (scalar_bytes == 0 || scalar_bytes > 32) return fail(ZKMI_ERR_UNSUPPORTED, "msm_table_multi_dev: tables are built for scalars of at most 32 bytes");
    for (int i = 0; i < count; i++) if (ks[i] > t.n) return fail(ZKMI_ERR_INVALID, "msm_table_multi_dev: more scalars than resident bases");
    return t.curve == ZKMI_CURVE_BN128 ? msm_table_multi_bn254(t.group, t.p, t.n, t.c, d_scalars, ks, count, scalar_bytes, out_jacobians)
                                       : msm_table_multi_bls12381(t.group, t.p, t.n, t.c, d_scalars, ks, count, scalar_bytes, out_jacobians);
}
int zkmi_msm_table_multi_enqueue_dev(uint64_t handle, const void* const* d_scalars, const size_t* ks, int count, size_t scalar_bytes) {
    ZK_TRY(require_ctx());
    auto it = g_tables.find(handle);
    if (it == g_tables.end()) return fail(ZKMI_ERR_INVALID, "msm_table_multi_enqueue_dev: unknown table");
    const MsmTable& t = it->second;
    if (!d_scalars || !ks) return fail(ZKMI_ERR_INVALID, "null argument");
    if (scalar_bytes == 0 || scalar_bytes > 32) return fail(ZKMI_ERR_UNSUPPORTED, "msm_table_multi_enqueue_dev: tables are built for scalars of at most 32 bytes");
    for (int i = 0; i < count; i++) if (ks[i] > t.n) return fail(ZKMI_ERR_INVALID, "msm_table_multi_enqueue_dev: more scalars than resident bases");
    return t.curve == ZKMI_CURVE_BN128 ? msm_table_multi_enqueue_bn254(t.group, t.p, t.n, t.c, d_scalars, ks, count, scalar_bytes)
                                       : msm_table_multi_enqueue_bls12381(t.group, t.p, t.n, t.c, d_scalars, ks, count, scalar_bytes);
}
int zkmi_msm_table_multi_enqueue_mont_dev(uint64_t handle, const void* const* d_polys, const size_t* ks, int count) {
    ZK_TRY(require_ctx());
    auto it = g_tables.find(handle);
    if (it == g_tables.end()) return fail(ZKMI_ERR_INVALID, "msm_table_multi_enqueue_mont_dev: unknown table");
    const MsmTable& t = it->second;
    if (!d_polys || !ks) return fail(ZKMI_ERR_INVALID, "null argument");
    if (count < 1 || count > 4) return fail(ZKMI_ERR_INVALID, "msm_table_multi: 1..4 MSMs per call");
    void* sc[4] = {};
    for (int i = 0; i < count; i++) {
        if (ks[i] > t.n) return fail(ZKMI_ERR_INVALID, "msm_table_multi_enqueue_mont_dev: more scalars than resident bases");
        if (ks[i]) ZK_TRY(ws_get("msm.commit_sc." + std::to_string(i), ks[i] * 32, &sc[i]));
    }
    ZK_TRY(fr_convert_multi_dispatch(t.curve, ZKMI_BATCH_FROM_MONTGOMERY, d_polys, sc, ks, count));
    return t.curve == ZKMI_CURVE_BN128 ? msm_table_multi_enqueue_bn254(t.group, t.p, t.n, t.c, sc, ks, count, 32) : msm_table_multi_enqueue_bls12381(t.group, t.p, t.n, t.c, sc, ks, count, 32);
}
int zkmi_msm_table_multi_collect(uint64_t handle, int count, uint8_t* out_jacobians) {
    ZK_TRY(require_ctx());
    auto it = g_tables.find(handle);
    if (it == g_tables.end()) return fail(ZKMI_ERR_INVALID, "msm_table_multi_collect: unknown table");
    if (!out_jacobians) return fail(ZKMI_ERR_INVALID, "null argument");
    const MsmTable& t = it->second;
    return t.curve == ZKMI_CURVE_BN128 ? msm_table_multi_collect_bn254(t.group, count, out_jacobians) : msm_table_multi_collect_bls12381(t.group, count, out_jacobians);
}
int zkmi_msm_table_info(uint64_t handle, int* curve, int* group, size_t* n) {
    auto it = g_tables.find(handle);
    if (it == g_tables.end()) return fail(ZKMI_ERR_INVALID, "msm_table_info: unknown table");
    if (curve) *curve = it->second.curve;
    if (group) *group = it->second.group;
    if (n) *n = it->second.n;
    return ZKMI_OK;
}
int zkmi_msm_table_release(uint64_t handle) {
    auto it = g_tables.find(handle);
    if (it == g_tables.end()) return ZKMI_OK;
    if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream);
    table_free(it->second);
    g_tables.erase(it);
    return ZKMI_OK;
}
static unsigned long long g_msm_dev_fallbacks = 0;       // zkmi_msm_dev calls that ran on the saturated-limb path for want of scratch memory
unsigned long long zkmi_msm_dev_fallbacks(void) { return g_msm_dev_fallbacks; }
int zkmi_msm_dev(int curve, int group, const void* d_bases, const void* d_scalars, size_t n, size_t scalar_bytes, uint8_t* out) {
    ZK_TRY(require_ctx());
    ZK_TRY(check_cg(curve, group));
    if (!out) return fail(ZKMI_ERR_INVALID, "null output");
    // The caller's bases are the reference's R-form and stay untouched: the accumulation runs on the library's own copy, moved to R'-form on
    // the device (r04: one 2 n pb-byte pass + 5 / 8 doublings per coordinate, ~ 40 us at 2^20 — 1.5 % of the MSM it makes 1.4 x faster), so
    // that the standalone MSM uses the same unsaturated-limb kernels as a resident key. ZKMI_MSM_DEV_R29=0: the r01 saturated-limb kernel.
    static const bool conv = !(getenv("ZKMI_MSM_DEV_R29") && atoi(getenv("ZKMI_MSM_DEV_R29")) == 0);
    if (!conv || n == 0 || !d_bases) return msm_dev_dispatch(curve, group, d_bases, d_scalars, n, scalar_bytes, out);
    const size_t pb = (size_t)2 * group * n8q_of(curve);
    void* d_b = nullptr;
    uint32_t* d_mask = nullptr;
    // The copy is scratch the caller never asked for (n pb bytes: 4 - 6 GB at 2^26 G1 points). When it cannot be had, the MSM still runs — on the
    // caller's own bases with the saturated-limb kernel, as before r04 — instead of failing with out-of-memory; and a copy beyond
    // ZKMI_MSM_DEV_KEEP_BYTES (default 2 GiB) is released after the call rather than kept for the life of the process.
    static const size_t keep_limit = [] { const char* e = getenv("ZKMI_MSM_DEV_KEEP_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)2 << 30); }();
    if (ws_get("api.dev_bases29", n * pb, &d_b) != ZKMI_OK || ws_get("api.dev_basemask", ((n + 31) / 32) * 4 + 16, (void**)&d_mask) != ZKMI_OK) {
        // ONLY an allocation that ran out of memory takes the slower path on the caller's own bases; anything else (a sticky fault of an earlier
        // kernel surfacing in the stream sync or the free inside ws_get) is this call's error and is reported as such
        if (!g_ws_oom) return ZKMI_ERR_HIP;
        (void)hipGetLastError();                                   // the failed allocation is not this call's result
        ws_drop("api.dev_bases29");
        g_msm_dev_fallbacks++;
        return msm_dev_dispatch(curve, group, d_bases, d_scalars, n, scalar_bytes, out);
    }
    ZK_HIP(hipEventRecord(g_ctx.ev0, g_ctx.stream));              // zkmi_last_kernel_ms covers the copy and the conversion too
    ZK_HIP(hipMemcpyAsync(d_b, d_bases, n * pb, hipMemcpyDeviceToDevice, g_ctx.stream));
    int rc = msm_infmask_dispatch(curve, group, d_b, n, d_mask);
    if (!rc) rc = msm_table_to_r29(curve, group, d_b, n, d_mask);  // registers d_b when the 29-bit path exists for (curve, group); else a no-op
    if (!rc) {
        struct Ev0Held { Ev0Held() { g_ctx.ev0_held = true; } ~Ev0Held() { g_ctx.ev0_held = false; } } held;      // msm_run: "the start event is recorded already", for exactly this dispatch
        rc = msm_dev_dispatch(curve, group, d_b, d_scalars, n, scalar_bytes, out);
    }
    msm_table_forget_r29(d_b);
    if (n * pb > keep_limit) ws_drop("api.dev_bases29");
    return rc;
}
// ---- content-addressed cache of resident base tables behind zkmi_msm ---------------------------------------------------------
// zkey sections and SRS slices are static, so the same bytes come back on every proof; their pre-computed window tables stay on
// the device. The cache is keyed by the CONTENT of the base buffer — a 128-bit hash of every 64 KiB chunk, computed by the
// library on each call (host threads, a few GB/s per thread) — never by a caller-supplied fingerprint: two buffers that differ in one
// interior point cannot share a table. Policy:
//   * 1st sight of a buffer: plain MSM, only its chunk hashes are remembered; 2nd sight: the table is built (a one-proof CLI run
//     never pays k_msm_precompute); an MSM over a PREFIX of a resident buffer (PLONK's PTau.slice(0, n+2..n+6)) re-uses its table,
//     and building a table drops resident tables that are prefixes of it;
//   * tables that would not fit (Wd*n >= 2^31 entries, or more bytes than the budget) are never built: plain bases every time;
//   * least-recently-used tables are freed when the resident bytes exceed the budget (ZKMI_BASE_CACHE_BYTES, default 64 GiB).
constexpr size_t BC_CHUNK = 64 * 1024;
struct BcHash { uint64_t a = 0, b = 0; bool operator==(const BcHash& o) const { return a == o.a && b == o.b; } bool operator!=(const BcHash& o) const { return !(*this == o); } };
static BcHash bc_hash_bytes(const uint8_t* p, size_t len) {
    uint64_t h[4] = {0x9e3779b97f4a7c15ull ^ len, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
    size_t i = 0;
    for (; i + 32 <= len; i += 32) {
        uint64_t w[4];
        memcpy(w, p + i, 32);
        for (int k = 0; k < 4; k++) { h[k] = (h[k] ^ w[k]) * 0xff51afd7ed558ccdull; h[k] ^= h[k] >> 29; }
    }
    for (; i < len; i++) { h[i & 3] = (h[i & 3] ^ p[i]) * 0x100000001b3ull; }
    BcHash r;
    r.a = (h[0] ^ (h[1] << 1 | h[1] >> 63)) * 0xc4ceb9fe1a85ec53ull + h[2];
    r.b = (h[2] ^ (h[3] << 7 | h[3] >> 57)) * 0xff51afd7ed558ccdull + h[0] + (h[1] >> 3);
    r.a ^= r.a >> 31; r.b ^= r.b >> 33;
    return r;
}
// hash of chunk c of the first `total` bytes of a paged buffer (a chunk may straddle pages: gathered through a bounce buffer)
static BcHash bc_chunk_hash_one(const zkmi_pages& pg, const std::vector<size_t>& start, size_t total, size_t c, std::vector<uint8_t>& bounce) {
    const size_t off = c * BC_CHUNK, len = std::min(BC_CHUNK, total - off);
    int page = 0;
    while (page + 1 < pg.n_pages && start[page + 1] <= off) page++;
    if (off + len <= start[page + 1]) return bc_hash_bytes(pg.ptr[page] + (off - start[page]), len);
    bounce.resize(len);
    size_t done = 0;
    for (int q = page; q < pg.n_pages && done < len; q++) {
        const size_t o = off + done - start[q], k = std::min(len - done, pg.len[q] - o);
        memcpy(bounce.data() + done, pg.ptr[q] + o, k);
        done += k;
    }
    return bc_hash_bytes(bounce.data(), len);
}
// chunk hashes of the first `total` bytes of a paged buffer
static void bc_chunk_hashes(const zkmi_pages& pg, size_t total, std::vector<BcHash>& out) {
    const size_t nch = (total + BC_CHUNK - 1) / BC_CHUNK;
    out.assign(nch, BcHash());
    std::vector<size_t> start((size_t)pg.n_pages + 1, 0);
    for (int i = 0; i < pg.n_pages; i++) start[i + 1] = start[i] + pg.len[i];
    auto work = [&](size_t c0, size_t c1) {
        std::vector<uint8_t> bounce;
        for (size_t c = c0; c < c1; c++) out[c] = bc_chunk_hash_one(pg, start, total, c, bounce);
    };
    unsigned nt = std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    if (nch < 64) nt = 1;
    if (nt == 1) { work(0, nch); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, nch * t / nt, nch * (t + 1) / nt);
    for (auto& x : th) x.join();
}
struct BcEntry {
    int curve = 0, group = 0;
    size_t n = 0;                       // points hashed (and resident, when table.p)
    std::vector<BcHash> chunks;         // per 64 KiB of base bytes; the last one may cover a partial chunk
    MsmTable table;                     // table.p == nullptr: seen, not resident
    bool no_table = false;              // would exceed the limits: never build
    uint64_t last_use = 0, uses = 0;
    // (first page pointer, byte length) of caller buffers whose EVERY byte has been checked against this resident table, and how often each
    // came back since: the sampled re-check below applies to these only
    struct Seen { const uint8_t* p; size_t total; uint64_t hits; };
    std::vector<Seen> verified;
    size_t bytes() const { return table.p ? (size_t)table.Wd * table.n * 2 * table.group * n8q_of(table.curve) : 0; }
};
static std::vector<BcEntry> g_bc;
static uint64_t g_bc_clock = 0;
static size_t bc_budget() {
    static const size_t v = [] { const char* e = getenv("ZKMI_BASE_CACHE_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)64 << 30); }();
    return v;
}
// In-place x -> x * 2^SH mod p (SH = r29_shift: 5 for BN254, 8 for BLS12-381) on `len` bytes of whole base-field elements
static void bc_elems_to_r29(int curve, uint8_t* bytes, size_t len) {
    const size_t es = (size_t)n8q_of(curve);
    if (curve == ZKMI_CURVE_BN128) {
        const auto F = host::HField<4>::from_cfg<Bn254Fq>();
        for (size_t o = 0; o + es <= len; o += es) { host::HFp<4> v; memcpy(v.v, bytes + o, es); for (int k = 0; k < r29_shift<Bn254Fq>(); k++) v = F.dbl(v); memcpy(bytes + o, v.v, es); }
    } else {
        const auto F = host::HField<6>::from_cfg<Bls12381Fq>();
        for (size_t o = 0; o + es <= len; o += es) { host::HFp<6> v; memcpy(v.v, bytes + o, es); for (int k = 0; k < r29_shift<Bls12381Fq>(); k++) v = F.dbl(v); memcpy(bytes + o, v.v, es); }
    }
}
static size_t bc_resident_bytes() { size_t t = 0; for (auto& e : g_bc) t += e.bytes(); return t; }
static void bc_free(BcEntry& e) {
    if (e.table.p) { if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream); table_free(e.table); }
    e.verified.clear();
}
// does the resident entry e hold, as a prefix, exactly the `total` bytes whose chunk hashes are `q`? Whole chunks are compared by
// hash; a trailing partial chunk of the query is compared byte for byte against row 0 of the table (the plain bases).
static int bc_prefix_match(const BcEntry& e, const zkmi_pages& pg, size_t total, const std::vector<BcHash>& q, bool* match) {
    *match = false;
    const size_t pb = (size_t)2 * e.group * n8q_of(e.curve), ebytes = e.n * pb;
    if (total > ebytes) return ZKMI_OK;
    const size_t full = total / BC_CHUNK, tail = total - full * BC_CHUNK;
    for (size_t c = 0; c < full; c++) if (q[c] != e.chunks[c]) return ZKMI_OK;
    if (tail) {
        if (total == ebytes) { if (q[full] != e.chunks[full]) return ZKMI_OK; }
        else {
            if (!e.table.p) return ZKMI_OK;                  // nothing to compare the partial chunk with
            // compared from the first byte of the field element that holds the chunk boundary (BLS12-381's 48-byte elements straddle 64 KiB
            // boundaries): the range is whole elements on both ends
            const size_t es = (size_t)n8q_of(e.curve);
            const size_t off = full * BC_CHUNK - (full * BC_CHUNK) % es, len = total - off;
            std::vector<uint8_t> dev(len), host(len);
            ZK_HIP(hipMemcpy(dev.data(), (const uint8_t*)e.table.p + off, len, hipMemcpyDeviceToHost));
            size_t done = 0, base = 0;
            for (int i = 0; i < pg.n_pages && done < len; i++) {
                if (off + done < base + pg.len[i]) {
                    const size_t o = off + done - base, k = std::min(len - done, pg.len[i] - o);
                    memcpy(host.data() + done, pg.ptr[i] + o, k);
                    done += k;
                }
                base += pg.len[i];
            }
            // row 0 of an R'-form table holds x 2^SH for every coordinate x of the plain bases: bring the caller's bytes to the same form
            if (g_ctx.r29_tables.count(e.table.p)) bc_elems_to_r29(e.curve, host.data(), len);
            if (memcmp(dev.data(), host.data(), len)) return ZKMI_OK;
        }
    }
    *match = true;
    return ZKMI_OK;
}
int zkmi_msm(int curve, int group, zkmi_pages bases, zkmi_pages scalars, size_t n, size_t scalar_bytes, uint64_t key, uint8_t* out) {
    ZK_TRY(require_ctx());
    ZK_TRY(check_cg(curve, group));
    if (!out) return fail(ZKMI_ERR_INVALID, "null output");
    const size_t pb = (size_t)2 * group * n8q_of(curve);
    // base_cache_key is a set of permission bits (include/zkmi.h). Until r04 any non-zero value meant "may cache": a caller still passing an arbitrary
    // identity (1002, 1011 ...) would silently get no caching, or the sampled re-check it never asked for — refuse it instead
    if (key & ~(ZKMI_BASES_CACHE | ZKMI_BASES_IMMUTABLE))
        return fail(ZKMI_ERR_INVALID, "msm: base_cache_key has unknown bits set (it is a set of permission bits: ZKMI_BASES_CACHE = 1, ZKMI_BASES_IMMUTABLE = 2; it carries no identity)");
    if ((key & ZKMI_BASES_IMMUTABLE) && !(key & ZKMI_BASES_CACHE)) return fail(ZKMI_ERR_INVALID, "msm: ZKMI_BASES_IMMUTABLE without ZKMI_BASES_CACHE");
    if (n == 0) { memset(out, 0, 3 * group * n8q_of(curve)); return ZKMI_OK; }
    if (pages_total(scalars) != n * scalar_bytes) return fail(ZKMI_ERR_INVALID, "Scalar size does not match");
    if (pages_total(bases) < n * pb) return fail(ZKMI_ERR_INVALID, "input buffer shorter than n elements");
    void *d_b = nullptr, *d_s = nullptr;
    ZK_TRY(ws_get("api.scalars", n * scalar_bytes, &d_s));
    ZK_TRY(upload_pages(scalars, n * scalar_bytes, d_s));
    // A resident key comes back with every proof in the SAME host buffer. Hashing all of it on every call (64 MB per 2^20 G1 MSM) costs more than
    // the device part of the call (3.28 ms through N-API against 1.92 ms on the device). A caller that PROMISES not to edit such a buffer
    // (ZKMI_BASES_IMMUTABLE, include/zkmi.h) gets a sampled re-check instead: a buffer (first page pointer, length) whose every byte was checked
    // against a resident table on an earlier call is re-checked by its first and last whole chunk and 30 chunks at pseudo-random positions that
    // change from call to call (2 MB), and in full again on every 32nd sight; anything else (another pointer, another length, a failed sample)
    // takes the full content hash. Without the promise (the default since r05) EVERY call takes the full hash: the result always follows the
    // bytes passed, as the reference's does. ZKMI_BASE_HASH_FULL=1: the full hash whatever was promised.
    const bool allow_cache = (key & ZKMI_BASES_CACHE) != 0, promised = allow_cache && (key & ZKMI_BASES_IMMUTABLE) != 0;
    static const bool hash_full = getenv("ZKMI_BASE_HASH_FULL") && atoi(getenv("ZKMI_BASE_HASH_FULL")) == 1;
    if (promised && scalar_bytes <= 32 && !hash_full && bases.n_pages >= 1) {
        const size_t total = n * pb;
        for (auto& e : g_bc) {
            if (!e.table.p || e.curve != curve || e.group != group || total > e.n * pb) continue;
            BcEntry::Seen* sn = nullptr;
            for (auto& v : e.verified) if (v.p == bases.ptr[0] && v.total == total) sn = &v;
            if (!sn) continue;
            if ((++sn->hits & 31u) == 0) break;                          // periodic full check: fall through to the complete hash
            const size_t full = total / BC_CHUNK;                        // whole chunks of the query (their hashes are position-independent of the tail)
            bool same = true;
            if (full) {
                std::vector<size_t> start((size_t)bases.n_pages + 1, 0);
                for (int i = 0; i < bases.n_pages; i++) start[i + 1] = start[i] + bases.len[i];
                std::vector<uint8_t> bounce;
                uint64_t x = 0x9e3779b97f4a7c15ull * (sn->hits + 1) + (uint64_t)(uintptr_t)bases.ptr[0];
                for (int k = 0; k < 32 && same; k++) {
                    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
                    const size_t c = k == 0 ? 0 : (k == 1 ? full - 1 : (size_t)((x * 0x2545f4914f6cdd1dull) % full));
                    same = bc_chunk_hash_one(bases, start, total, c, bounce) == e.chunks[c];
                }
            }
            if (!same) { e.verified.clear(); break; }                    // the buffer changed: full path decides
            e.last_use = ++g_bc_clock; e.uses++;
            return msm_table_dispatch(curve, group, e.table.p, e.table.n, e.table.c, d_s, n, scalar_bytes, out);
        }
    }
    if (allow_cache && scalar_bytes <= 32) {
        std::vector<BcHash> q;
        bc_chunk_hashes(bases, n * pb, q);
        BcEntry* hit = nullptr;                      // resident table that holds these bases as a prefix
        BcEntry* seen = nullptr;                     // exact buffer seen before, no table yet
        for (auto& e : g_bc) {
            if (e.curve != curve || e.group != group || e.chunks.empty()) continue;
            if (e.chunks[0] != q[0] && e.n * pb >= BC_CHUNK && n * pb >= BC_CHUNK) continue;          // different first 64 KiB
            if (e.table.p) { bool m = false; ZK_TRY(bc_prefix_match(e, bases, n * pb, q, &m)); if (m) { hit = &e; break; } }
            else if (e.n == n && e.chunks == q) seen = &e;
        }
        if (!hit && seen && !seen->no_table) {
            // 2nd sight: build the table, unless it cannot exist
            MsmTable t;
            const int c = msm_precomp_c(n), Wd = msm_digits(32, c);
            const size_t tbytes = (size_t)Wd * n * pb;
            if ((size_t)Wd * n >= (1ull << 31) || tbytes > bc_budget()) seen->no_table = true;
            else {
                // make room: least recently used first
                while (bc_resident_bytes() + tbytes > bc_budget()) {
                    BcEntry* lru = nullptr;
                    for (auto& e : g_bc) if (e.table.p && (!lru || e.last_use < lru->last_use)) lru = &e;
                    if (!lru) break;
                    bc_free(*lru);
                }
                void* raw = nullptr;
                ZK_HIP(hipMalloc(&raw, n * pb));
                int rc = upload_pages(bases, n * pb, raw);
                if (!rc) rc = table_build_r29(curve, group, raw, n, t);
                (void)hipFree(raw);
                if (rc == ZKMI_OK) {
                    seen->table = t;
                    hit = seen;
                    // resident tables that are prefixes of the new one are redundant now
                    for (auto& e : g_bc)
                        if (&e != seen && e.table.p && e.curve == curve && e.group == group && e.n < n) {
                            bool pre = e.chunks.size() <= q.size();
                            const size_t full = e.n * pb / BC_CHUNK;
                            for (size_t k = 0; pre && k < full; k++) pre = e.chunks[k] == q[k];
                            if (pre && (e.n * pb) % BC_CHUNK == 0) bc_free(e);
                        }
                }
                // a failed build (out of device memory) falls through to the plain path
            }
        }
        if (!hit && !seen) {
            BcEntry e;
            e.curve = curve; e.group = group; e.n = n; e.chunks = q;
            if (g_bc.size() >= 256) {                // bound the bookkeeping: forget the oldest entry that holds no table
                size_t victim = g_bc.size();
                for (size_t i = 0; i < g_bc.size(); i++) if (!g_bc[i].table.p && (victim == g_bc.size() || g_bc[i].last_use < g_bc[victim].last_use)) victim = i;
                if (victim < g_bc.size()) g_bc.erase(g_bc.begin() + victim);
            }
            g_bc.push_back(std::move(e));
            seen = &g_bc.back();
        }
        BcEntry* used = hit ? hit : seen;
        used->last_use = ++g_bc_clock; used->uses++;
        if (hit) {
            // every byte of this buffer has just been compared with the table (hashes of whole chunks, bytes of a partial tail)
            bool known = false;
            for (auto& v : hit->verified) if (v.p == bases.ptr[0] && v.total == n * pb) known = true;
            if (!known) { if (hit->verified.size() >= 16) hit->verified.erase(hit->verified.begin()); hit->verified.push_back({bases.ptr[0], n * pb, 0}); }
            return msm_table_dispatch(curve, group, hit->table.p, hit->table.n, hit->table.c, d_s, n, scalar_bytes, out);
        }
    }
    ZK_TRY(ws_get("api.bases", n * pb, &d_b));
    ZK_TRY(upload_pages(bases, n * pb, d_b));
    // the upload is the library's own copy: moved to R'-form in place (one pass of 5 / 8 doublings per coordinate), so that plain-base MSMs
    // run the unsaturated-limb accumulation too; the registration lives for this call only
    uint32_t* d_mask = nullptr;
    ZK_TRY(ws_get("api.basemask", ((n + 31) / 32) * 4 + 16, (void**)&d_mask));
    ZK_TRY(msm_infmask_dispatch(curve, group, d_b, n, d_mask));
    ZK_TRY(msm_table_to_r29(curve, group, d_b, n, d_mask));
    const int rc = msm_dev_dispatch(curve, group, d_b, d_s, n, scalar_bytes, out);
    msm_table_forget_r29(d_b);
    return rc;
}
// key != 0 / 0: both drop every cached base table (the cache is content-addressed; per-key release has no meaning any more)
int zkmi_release_bases(uint64_t key) {
    (void)key;
    for (auto& e : g_bc) bc_free(e);
    g_bc.clear();
    return ZKMI_OK;
}
// introspection for tests: resident tables, their bytes, remembered (table-less) buffers
int zkmi_base_cache_stats(uint64_t* n_tables, uint64_t* table_bytes, uint64_t* n_seen) {
    uint64_t t = 0, s = 0;
    for (auto& e : g_bc) { if (e.table.p) t++; else s++; }
    if (n_tables) *n_tables = t;
    if (table_bytes) *table_bytes = bc_resident_bytes();
    if (n_seen) *n_seen = s;
    return ZKMI_OK;
}

int zkmi_ntt_dev(int curve, const void* d_in, void* d_out, unsigned log_n, int inverse, const uint8_t* first, const uint8_t* inc) {
    ZK_TRY(require_ctx());
    return ntt_dev_dispatch(curve, d_in, d_out, log_n, inverse, first, inc);
}
int zkmi_ntt_padded_dev(int curve, const void* d_in, size_t in_len, void* d_out, unsigned log_n, int inverse) {
    ZK_TRY(require_ctx());
    if (!d_in || !d_out) return fail(ZKMI_ERR_INVALID, "null argument");
    return ntt_dev_padded_dispatch(curve, d_in, in_len, d_out, log_n, inverse);
}
int zkmi_ntt(int curve, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out, unsigned log_n, int inverse,
             const uint8_t* first, const uint8_t* inc) {
    ZK_TRY(require_ctx());
    if (log_n > 40) return fail(ZKMI_ERR_INVALID, "fft: size out of range");
    const size_t bytes = ((size_t)1 << log_n) * 32;
    if (pages_total(in) != bytes) return fail(ZKMI_ERR_INVALID, "fft must be multiple of 2");
    void *d_i, *d_o;
    ZK_TRY(ws_get("api.ntt_in", bytes, &d_i));
    ZK_TRY(ws_get("api.ntt_out", bytes, &d_o));
    ZK_TRY(upload_pages(in, bytes, d_i));
    ZK_TRY(ntt_dev_dispatch(curve, d_i, d_o, log_n, inverse, first, inc));
    return download_pages(d_o, bytes, out_ptr, out_len, n_out);
}
int zkmi_fr_batch_apply_key_dev(int curve, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc) {
    ZK_TRY(require_ctx());
    return apply_key_dev_dispatch(curve, d_in, d_out, n, first, inc);
}
int zkmi_fr_batch_apply_key(int curve, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out, size_t n,
                            const uint8_t* first, const uint8_t* inc) {
    ZK_TRY(require_ctx());
    void *d_i, *d_o;
    ZK_TRY(ws_get("api.b_in", n * 32, &d_i)); ZK_TRY(ws_get("api.b_out", n * 32, &d_o));
    ZK_TRY(upload_pages(in, n * 32, d_i));
    ZK_TRY(apply_key_dev_dispatch(curve, d_i, d_o, n, first, inc));
    return download_pages(d_o, n * 32, out_ptr, out_len, n_out);
}
int zkmi_fr_batch_dev(int curve, int op, const void* d_in, void* d_out, size_t n) {
    ZK_TRY(require_ctx());
    return fr_batch_dev_dispatch(curve, op, d_in, d_out, n);
}
int zkmi_fr_batch(int curve, int op, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out, size_t n) {
    ZK_TRY(require_ctx());
    void *d_i, *d_o;
    ZK_TRY(ws_get("api.b_in", n * 32, &d_i)); ZK_TRY(ws_get("api.b_out", n * 32, &d_o));
    ZK_TRY(upload_pages(in, n * 32, d_i));
    ZK_TRY(fr_batch_dev_dispatch(curve, op, d_i, d_o, n));
    return download_pages(d_o, n * 32, out_ptr, out_len, n_out);
}
int zkmi_groth16_join_abc_dev(int curve, const void* a, const void* b, const void* c, void* out, size_t n) {
    ZK_TRY(require_ctx());
    return join_abc_dev_dispatch(curve, a, b, c, out, n);
}
int zkmi_groth16_join_abc(int curve, zkmi_pages a, zkmi_pages b, zkmi_pages c, uint8_t* const* out_ptr, const size_t* out_len, int n_out, size_t n) {
    ZK_TRY(require_ctx());
    void *d_a, *d_b, *d_c, *d_o;
    ZK_TRY(ws_get("api.j_a", n * 32, &d_a)); ZK_TRY(ws_get("api.j_b", n * 32, &d_b)); ZK_TRY(ws_get("api.j_c", n * 32, &d_c)); ZK_TRY(ws_get("api.b_out", n * 32, &d_o));
    ZK_TRY(upload_pages(a, n * 32, d_a)); ZK_TRY(upload_pages(b, n * 32, d_b)); ZK_TRY(upload_pages(c, n * 32, d_c));
    ZK_TRY(join_abc_dev_dispatch(curve, d_a, d_b, d_c, d_o, n));
    return download_pages(d_o, n * 32, out_ptr, out_len, n_out);
}
int zkmi_gen_geometric_bases_dev(int curve, int group, size_t n, uint64_t f, uint64_t g, void* d_out) {
    ZK_TRY(require_ctx());
    if (n >= (1ull << 32)) return fail(ZKMI_ERR_UNSUPPORTED, "n too large");
    return gen_bases_dispatch(curve, group, n, f, g, d_out);
}
int zkmi_to_affine(int curve, int group, const uint8_t* jac, uint8_t* aff) { return to_affine_dispatch(curve, group, jac, aff); }
int zkmi_fr_root(int curve, unsigned i, uint8_t* out32) {
    if (curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) return fail(ZKMI_ERR_INVALID, "unknown curve");
    if (!out32) return fail(ZKMI_ERR_INVALID, "null argument");
    return fr_root(curve, i, out32);
}
int zkmi_point_add(int curve, int group, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    ZK_TRY(check_cg(curve, group));
    if (!a || !b || !out) return fail(ZKMI_ERR_INVALID, "null argument");
    return curve == ZKMI_CURVE_BN128 ? point_add_bn254(group, a, b, out) : point_add_bls12381(group, a, b, out);
}

}  // extern "C"
