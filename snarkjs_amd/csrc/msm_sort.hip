// snarkjs_amd/csrc/msm_sort.hip — field-independent front half of the device Pippenger: signed-digit recoding of the
// scalars and a counting sort of (point index, sign) into per-(window, bucket) lists; plus bucket ordering by size.
// One sort can feed several accumulations (msm_accumulate) when MSMs share their scalars.
#include <string.h>
#include <mutex>
#include "msm_host.hpp"

namespace zkmi {

template <int NW> static int msm_launch_digits(const uint8_t* d_scalars, const MsmShape& sh, uint32_t* counts, uint32_t* starts, uint32_t* cursor,
                                               uint32_t* sorted, uint32_t* part, const uint32_t* dropmask, hipStream_t st) {
    const unsigned blocks = (unsigned)((sh.n + 255) / 256);
    hipLaunchKernelGGL((k_msm_count<NW>), dim3(blocks), dim3(256), 0, st, d_scalars, sh, dropmask, counts);
    const uint32_t total = (uint32_t)sh.W * sh.nb, nparts = (total + MSM_SCAN_CHUNK - 1) / MSM_SCAN_CHUNK;
    hipLaunchKernelGGL(k_msm_scan_sums, dim3(nparts), dim3(256), 0, st, counts, total, part);
    hipLaunchKernelGGL(k_msm_scan_top, dim3(1), dim3(1024), 0, st, part, nparts);
    hipLaunchKernelGGL(k_msm_scan_final, dim3(nparts), dim3(256), 0, st, counts, total, part, starts);
    hipLaunchKernelGGL((k_msm_scatter<NW>), dim3(blocks), dim3(256), 0, st, d_scalars, sh, dropmask, starts, cursor, sorted);
    return ZKMI_OK;
}

// Two-level LDS radix partition (msm.cuh: k_rsort_*): same outputs (counts, starts, sorted) without global atomics.
// lb = bits of the low key (2^lb buckets per partition); fused_cap != 0: partitions of at most that many pairs are finished by k_rsort_part.
// pairs staged in LDS by k_rsort_part, by low-key width: 2^10-bucket partitions 36 000 (4 B each + 2^lb + 1024 words = 152 KB: one block per CU),
// 2^9-bucket partitions 15 000 (66 KB: two blocks per CU, and room beside an LDS-parked G2 accumulation block)
static uint32_t rsort_part_cap(uint32_t lb) { return lb == 10 ? 36000u : 15000u; }
static bool rsort_fused_on() {
    static const bool on = !(getenv("ZKMI_RSORT_FUSED") && atoi(getenv("ZKMI_RSORT_FUSED")) == 0);
    return on;
}
// The fused level 2 stages a partition in up to 152 KB of LDS (gfx950: 160 KB per CU). Checked ONCE per process against the bound device — the library
// binds one device per process — under std::call_once (zkmi_msm runs on libuv pool threads too): the opt-in attribute is set there, and a device
// that cannot give that much LDS takes the chunked level 2 (the r04 path) instead of failing at launch.
static bool rsort_fused_fits() {
    static std::once_flag once;
    static bool fits = false;
    std::call_once(once, [] {
        const int need = (int)(((size_t)RSORT_BINS + 1024 + rsort_part_cap(10)) * 4);
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) { (void)hipGetLastError(); return; }
        // gfx950 (MI355X) has 160 KB of LDS per CU; elsewhere the device's own figure decides (64 KB on earlier CDNA: the chunked path)
        const bool gfx950 = strncmp(pr.gcnArchName, "gfx950", 6) == 0;
        if (!gfx950 && (size_t)pr.maxSharedMemoryPerMultiProcessor < (size_t)need) return;
        if (hipFuncSetAttribute((const void*)k_rsort_part, hipFuncAttributeMaxDynamicSharedMemorySize, need) != hipSuccess) { (void)hipGetLastError(); return; }
        fits = true;
    });
    return fits;
}
// The staged level-1 scatter (msm.cuh: k_rsort_scatter1_staged) needs Wd x RSORT_TILE pairs of LDS beside its cursors: taken where that fits the device
// (checked once per instantiation, like the fused level 2 above), the direct scatter otherwise.
template <int NW> static bool rsort_staged_fits(size_t lds) {
    static std::once_flag once;
    static size_t limit = 0;
    std::call_once(once, [] {
        if (getenv("ZKMI_RSORT_STAGED") && atoi(getenv("ZKMI_RSORT_STAGED")) == 0) return;
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) { (void)hipGetLastError(); return; }
        const size_t want = strncmp(pr.gcnArchName, "gfx950", 6) == 0 ? (size_t)156 * 1024 : (size_t)pr.maxSharedMemoryPerMultiProcessor;
        if (hipFuncSetAttribute((const void*)k_rsort_scatter1_staged<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess) { (void)hipGetLastError(); return; }
        limit = want;
    });
    return lds <= limit;
}
// low-key bits for this shape: 10 (ZKMI_RSORT_LB=9: 9) when the fused level 2 can take a typical partition (entries / partitions <= 0.9 cap), else 11
static uint32_t rsort_low_bits(const MsmShape& sh) {
    static const uint32_t want = getenv("ZKMI_RSORT_LB") ? (uint32_t)atoi(getenv("ZKMI_RSORT_LB")) : 10u;
    const size_t total = (size_t)sh.W * sh.nb, entries = (size_t)sh.Wd * sh.n;
    if (rsort_fused_on() && (want == 9 || want == 10) && (uint32_t)sh.c > want && (total >> want) <= RSORT_MAX_PARTS && (total >> want) >= 1 &&
        entries / (total >> want) <= (size_t)rsort_part_cap(want) * 9 / 10)
        return want;
    return RSORT_LOW_BITS;
}
template <int NW> static int msm_launch_digits_radix(const uint8_t* d_scalars, const MsmShape& sh, uint32_t* counts, uint32_t* starts, uint32_t* sorted,
                                                     const uint32_t* dropmask, hipStream_t st, uint32_t* sched_zero, uint32_t sched_zero_words) {
    const uint32_t lb = rsort_fused_fits() ? rsort_low_bits(sh) : (uint32_t)RSORT_LOW_BITS;
    const uint32_t fused_cap = (lb < (uint32_t)RSORT_LOW_BITS) ? rsort_part_cap(lb) : 0u;
    const uint32_t total = (uint32_t)sh.W * sh.nb, P = total >> lb;
    const uint32_t nblk = (uint32_t)((sh.n + RSORT_TILE - 1) / RSORT_TILE);
    const size_t nbh = (size_t)P * nblk + 1, emax = (size_t)sh.Wd * sh.n, nch_max = emax / RSORT_CHUNK + P + 1;
    uint32_t *bh, *part, *ck, *h2;
    uint2* tmp;
    ZK_TRY(ws_get("msm.rs_bh", 2 * nbh * 4, (void**)&bh));
    uint32_t* bhoff = bh + nbh;
    const uint32_t nsp = (uint32_t)((nbh + MSM_SCAN_CHUNK - 1) / MSM_SCAN_CHUNK);
    ZK_TRY(ws_get("msm.rs_part", ((size_t)nsp + 2) * 4, (void**)&part));
    ZK_TRY(ws_get("msm.rs_tmp", emax * 8, (void**)&tmp));
    ZK_TRY(ws_get("msm.rs_chunks", (3 * nch_max + P + 2 + 4) * 4, (void**)&ck));
    uint32_t *pchunk0 = ck + 3 * nch_max, *meta = pchunk0 + P + 2;
    ZK_TRY(ws_get("msm.rs_h2", nch_max * RSORT_BINS * 4, (void**)&h2));
    hipLaunchKernelGGL((k_rsort_hist1<NW>), dim3(nblk), dim3(256), 0, st, d_scalars, sh, dropmask, P, lb, bh);
    hipLaunchKernelGGL(k_msm_scan_sums, dim3(nsp), dim3(256), 0, st, bh, (uint32_t)nbh, part);
    hipLaunchKernelGGL(k_msm_scan_top, dim3(1), dim3(1024), 0, st, part, nsp);
    hipLaunchKernelGGL(k_msm_scan_final, dim3(nsp), dim3(256), 0, st, bh, (uint32_t)nbh, part, bhoff);
    // level-1 scatter: ranked in LDS and written as runs where a block's pairs fit (ZKMI_RSORT_STAGED=0: the direct scatter everywhere)
    const uint32_t stage_cap = (uint32_t)sh.Wd * RSORT_TILE;
    const size_t stage_lds = ((size_t)2 * P + 1024 + (size_t)2 * stage_cap) * 4;
    // ... and where a block's share of a partition is long enough to be worth a run: >= 8 pairs on uniform scalars (26 at 2^20 terms with 512 partitions; with
    // thousands of partitions and 2 - 3 pairs per block and partition there is nothing to coalesce, and the staged kernel reads two entries of the scanned matrix per
    // partition where the direct one reads one)
    if (rsort_staged_fits<NW>(stage_lds) && (size_t)stage_cap >= (size_t)8 * P)
        hipLaunchKernelGGL((k_rsort_scatter1_staged<NW>), dim3(nblk), dim3(1024), stage_lds, st, d_scalars, sh, dropmask, P, lb, bhoff, stage_cap, tmp);
    else
        hipLaunchKernelGGL((k_rsort_scatter1<NW>), dim3(nblk), dim3(256), 0, st, d_scalars, sh, dropmask, P, lb, bhoff, tmp);
    hipLaunchKernelGGL(k_rsort_chunks, dim3(1), dim3(1024), 0, st, bhoff, P, nblk, fused_cap, pchunk0, ck, meta, sched_zero, sched_zero_words);
    if (fused_cap) {
        const size_t lds = ((size_t)(1u << lb) + 1024 + fused_cap) * 4;
        hipLaunchKernelGGL(k_rsort_part, dim3(P), dim3(1024), lds, st, tmp, bhoff, nblk, lb, fused_cap, counts, starts, sorted);
    }
    // the chunk kernels walk the table (meta[0] chunks, usually none: every block that only reads meta[0] and leaves still costs ~10 ns of dispatch —
    // 17 us per launch at 1 024 blocks, measured standalone); real witnesses leave a few dozen chunks (the bucket of digit 1), 256 blocks take them in one pass
    // — where the fused level 2 is on. Without it (2^11-bucket partitions: MSMs beyond ~2^20 terms) EVERY partition is cut into chunks (26 000 of them at 2^24): one block each
    const unsigned chunk_blocks = (unsigned)(fused_cap ? std::min<size_t>(nch_max, 256) : nch_max);
    hipLaunchKernelGGL(k_rsort_hist2, dim3(chunk_blocks), dim3(256), 0, st, tmp, ck, meta, h2);
    hipLaunchKernelGGL(k_rsort_scan2, dim3(P), dim3(1024), 0, st, bhoff, nblk, lb, fused_cap ? 1u : 0u, pchunk0, h2, counts, starts);
    hipLaunchKernelGGL(k_rsort_scatter2, dim3(chunk_blocks), dim3(256), 0, st, tmp, ck, meta, h2, starts, lb, sorted);
    return ZKMI_OK;
}
static bool msm_use_radix(const MsmShape& sh) {
    static const bool on = !(getenv("ZKMI_RSORT") && atoi(getenv("ZKMI_RSORT")) == 0);
    const size_t total = (size_t)sh.W * sh.nb;
    return on && sh.c > RSORT_LOW_BITS && (total >> RSORT_LOW_BITS) <= RSORT_MAX_PARTS && (size_t)sh.Wd * sh.n >= (1u << 17);
}

int msm_sort(const void* d_scalars, size_t n, size_t sb, MsmPlan& pl, int plan_slot, int precomp_c, size_t table_stride, const uint32_t* d_dropmask) {
    Ctx& cx = ctx();
    pl.slot = plan_slot;
    const std::string sfx = ".p" + std::to_string(plan_slot);
    if (n == 0 || n >= (1ull << 31)) return fail(ZKMI_ERR_UNSUPPORTED, "msm: n must be in [1, 2^31)");
    if (sb == 0 || sb > 64) return fail(ZKMI_ERR_UNSUPPORTED, "msm: scalar size must be 1..64 bytes");
    MsmShape& sh = pl.sh;
    sh.n = (uint32_t)n; sh.sb = (int)sb;
    sh.precomp = precomp_c ? 1 : 0;
    sh.c = precomp_c ? precomp_c : (cx.msm_c_override ? cx.msm_c_override : msm_pick_c(n));
    sh.Wd = msm_digits(sb, sh.c);
    sh.W = sh.precomp ? 1 : sh.Wd;
    sh.stride = (uint32_t)(table_stride ? table_stride : n);
    sh.nb = 1u << (sh.c - 1);
    if (sh.precomp && (size_t)sh.Wd * sh.stride >= (1ull << 31)) return fail(ZKMI_ERR_UNSUPPORTED, "msm: pre-computed table too large for 31-bit indices");
    const size_t total = (size_t)sh.W * sh.nb;
    pl.total = total;
    hipStream_t st = cx.stream;
    uint32_t *counts, *hist;
    ZK_TRY(ws_get("msm.counts" + sfx, 3 * total * 4, (void**)&counts));        // counts | starts | cursor
    pl.counts = counts; pl.starts = counts + total;
    uint32_t* cursor = pl.starts + total;
    ZK_TRY(ws_get("msm.sorted" + sfx, (size_t)sh.Wd * n * 4, (void**)&pl.sorted));
    // lane-group schedule (k_msm_classify/_class_scan/_assign)
    // cap: points per lane. Large MSMs are ALU-bound: one lane per typical bucket (cap = pow2ceil(2 * average size)) keeps
    // the combine tree idle; small MSMs are latency-bound: shrink cap until ~250k lanes exist.
    const size_t avg = ((size_t)(sh.precomp ? sh.Wd : 1) * n + sh.nb - 1) / sh.nb;
    uint32_t cap_big = 16, cap_fill = 8;
    while (cap_big < 2 * avg && cap_big < MSM_MAX_CAP) cap_big <<= 1;
    while ((size_t)2 * cap_fill <= (size_t)sh.Wd * n / 250000 && cap_fill < MSM_MAX_CAP) cap_fill <<= 1;
    static const int cap_env = getenv("ZKMI_CAP") ? atoi(getenv("ZKMI_CAP")) : 0;
    const uint32_t cap = cap_env ? (uint32_t)cap_env : std::min(cap_big, cap_fill);
    pl.cap = cap;
    pl.multi_bound = (size_t)sh.Wd * n / cap + 1;                          // lanes of multi-lane groups: sum 2^floor(log2(cnt/cap))
    pl.lane_bound = total + pl.multi_bound;
    ZK_TRY(ws_get("msm.lanes" + sfx, 2 * pl.lane_bound * 4, (void**)&pl.lane_g));
    pl.lane_sub = pl.lane_g + pl.lane_bound;
    const size_t giant_bound = pl.multi_bound / MSM_TB + 1;
    ZK_TRY(ws_get("msm.hist" + sfx, (3 * MSM_NKEYS + 8 + 3 * giant_bound) * 4, (void**)&hist));     // hist | off | cursor | meta | giants
    uint32_t *koff = hist + MSM_NKEYS, *kcur = koff + MSM_NKEYS;
    pl.meta = kcur + MSM_NKEYS; pl.giants = pl.meta + 8;
    const uint32_t sched_words = 3 * MSM_NKEYS + 8;                          // zeroed by the radix sort's chunk kernel on its way (one launch less), by a fill otherwise
    const uint8_t* sc = (const uint8_t*)d_scalars;
    uint32_t* part;
    ZK_TRY(ws_get("msm.scanpart" + sfx, (total / MSM_SCAN_CHUNK + 2) * 4, (void**)&part));
    if (msm_use_radix(sh)) {
        if (sb <= 4) { ZK_TRY(msm_launch_digits_radix<1>(sc, sh, pl.counts, pl.starts, pl.sorted, d_dropmask, st, hist, sched_words)); }
        else if (sb <= 32) { ZK_TRY(msm_launch_digits_radix<8>(sc, sh, pl.counts, pl.starts, pl.sorted, d_dropmask, st, hist, sched_words)); }
        else { ZK_TRY(msm_launch_digits_radix<16>(sc, sh, pl.counts, pl.starts, pl.sorted, d_dropmask, st, hist, sched_words)); }
    } else {
    ZK_HIP(hipMemsetAsync(hist, 0, sched_words * 4, st));
    ZK_HIP(hipMemsetAsync(counts, 0, 3 * total * 4, st));
    if (sb <= 4) msm_launch_digits<1>(sc, sh, pl.counts, pl.starts, cursor, pl.sorted, part, d_dropmask, st);
    else if (sb <= 32) msm_launch_digits<8>(sc, sh, pl.counts, pl.starts, cursor, pl.sorted, part, d_dropmask, st);
    else msm_launch_digits<16>(sc, sh, pl.counts, pl.starts, cursor, pl.sorted, part, d_dropmask, st);
    }
    const unsigned tb = std::min<unsigned>((unsigned)((total + 255) / 256), MSM_SCHED_BLOCKS);      // contiguous runs of buckets per block (msm.cuh: msm_sched_run)
    hipLaunchKernelGGL(k_msm_classify, dim3(tb), dim3(256), 0, st, pl.counts, (uint32_t)total, cap, hist);
    hipLaunchKernelGGL(k_msm_assign, dim3(tb), dim3(256), 0, st, pl.counts, (uint32_t)total, cap, (uint32_t)MSM_LOG_TB, hist, kcur, pl.lane_g, pl.lane_sub, pl.giants, pl.meta);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

MsmMultiPending& msm_multi_pending(int pipe) {
    static MsmMultiPending pend[2];
    return pend[pipe & 1];
}
int msm_job_slot(int slot, MsmJob& job) {
    Ctx& cx = ctx();
    if (slot < 0 || slot >= MSM_JOB_SLOTS) return fail(ZKMI_ERR_INVALID, "msm: bad job slot");
    if (!cx.pinned) ZK_HIP(hipHostMalloc((void**)&cx.pinned, MSM_JOB_SLOTS * MSM_JOB_SLOT_BYTES, hipHostMallocDefault));
    job.slot = slot;
    job.h_win = (uint32_t*)(cx.pinned + (size_t)slot * MSM_JOB_SLOT_BYTES);
    if (!cx.job_ev[2 * slot]) { ZK_HIP(hipEventCreate(&cx.job_ev[2 * slot])); ZK_HIP(hipEventCreate(&cx.job_ev[2 * slot + 1])); }
    job.acc0 = cx.job_ev[2 * slot]; job.acc1 = cx.job_ev[2 * slot + 1];
    return ZKMI_OK;
}

}  // namespace zkmi
