// snarkjs_amd/csrc/calib.hip — two 20-millisecond probes of the BOX, reported next to a benchmark line so that a low number can be read:
// the MI355X boxes a run lands on differ (r02: the same commit measured 99.5 - 101.1 proofs/s on three boxes, 80.6 on one whose multiply rate
// was half, 74.2 on one with normal arithmetic and slow random access).
//   mul29: a chain of Montgomery products on 29-bit limbs at 8 waves per SIMD (tools/fieldbench29's k_chain29): G products/s
//   gather128: random 128-byte reads (8 x 16 bytes per lane, as the G2 accumulation issues them) over a 2 GiB table: GB/s
#include "field29.cuh"
#include "zkmi_common.hpp"

namespace zkmi {

template <class C> __global__ void __launch_bounds__(256) k_calib_mul29(uint32_t* out, int iters) {
    Fp29<C> a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & mask29<C>(); b.l[i] = (blockIdx.x * 2246822519u + i * 3266489917u) & mask29<C>(); }
    a.l[8] &= 0xffffu; b.l[8] &= 0xffffu;
    for (int it = 0; it < iters; it++) { a = mul29(a, b); b = mul29(b, a); }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) acc ^= a.l[i] ^ b.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// the same chain with a loop body of 2 REP products laid out as straight-line code (REP = 4: ~17 KB, inside the 64 KB instruction cache two
// CUs share; REP = 48: ~210 KB, the size of the BLS12-381 accumulation loops), at the accumulation kernels' two waves per SIMD: the ratio of the
// two rates says what instruction fetch beyond the cache costs on THIS box (r03: on some boxes the kernels with loops beyond the cache ran
// 1.7 - 3 x slower while both other probes read healthy)
template <class C, int K> struct CalibRep {                 // K x 2 products as straight-line code (a `#pragma unroll` over 48 bodies is only partly honoured)
    static __device__ __forceinline__ void run(Fp29<C>& a, Fp29<C>& b) { a = mul29(a, b); b = mul29(b, a); CalibRep<C, K - 1>::run(a, b); }
};
template <class C> struct CalibRep<C, 0> { static __device__ __forceinline__ void run(Fp29<C>&, Fp29<C>&) {} };
template <class C, int REP> __global__ void __launch_bounds__(256, 2) k_calib_code(uint32_t* out, int iters) {
    Fp29<C> a, b;
#pragma unroll
    for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & mask29<C>(); b.l[i] = (blockIdx.x * 2246822519u + i * 3266489917u) & mask29<C>(); }
    a.l[8] &= 0xffffu; b.l[8] &= 0xffffu;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        CalibRep<C, REP>::run(a, b);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) acc ^= a.l[i] ^ b.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_calib_gather128(const uint4* __restrict__ tab, size_t n_rows, uint32_t* __restrict__ out, int reps) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint64_t h = (uint64_t)t * 0x9E3779B97F4A7C15ull;
    uint32_t acc = 0;
    for (int r = 0; r < reps; r++) {
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const uint4* p = tab + (h % n_rows) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        h += acc;                                       // the next address depends on the data: no reordering across repetitions
    }
    out[t] = acc;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" int zkmi_calibrate_box(double* mul29_gmul_per_s, double* gather128_gb_per_s) {
    ZK_TRY(require_ctx());
    Ctx& cx = ctx();
    hipStream_t st = cx.stream;
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0)); ZK_HIP(hipEventCreate(&e1));
    float ms = 0;
    const int blocks = 256 * 8, iters = 400;                          // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    uint32_t* d_out;
    ZK_TRY(ws_get("calib.out", (size_t)4 << 20, (void**)&d_out));
    double best = 0;
    for (int rep = 0; rep < 3; rep++) {
        ZK_HIP(hipEventRecord(e0, st));
        hipLaunchKernelGGL((k_calib_mul29<Bn254Fq>), dim3(blocks), dim3(256), 0, st, d_out, iters);
        ZK_HIP(hipEventRecord(e1, st));
        ZK_HIP(hipEventSynchronize(e1));
        ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
        best = std::max(best, (double)blocks * 256 * iters * 2 / (ms * 1e-3) / 1e9);
    }
    if (mul29_gmul_per_s) *mul29_gmul_per_s = best;
    void* tab = nullptr;
    const size_t bytes = (size_t)2 << 30;
    if (hipMalloc(&tab, bytes) != hipSuccess) { (void)hipGetLastError(); if (gather128_gb_per_s) *gather128_gb_per_s = 0; }
    else {
        ZK_HIP(hipMemsetAsync(tab, 0x5a, bytes, st));
        const int gblocks = 1 << 12, reps = 16;                       // 2^20 lanes x 16 dependent gathers of 128 bytes
        best = 0;
        for (int rep = 0; rep < 3; rep++) {
            ZK_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_calib_gather128, dim3(gblocks), dim3(256), 0, st, (const uint4*)tab, bytes / 128, d_out, reps);
            ZK_HIP(hipEventRecord(e1, st));
            ZK_HIP(hipEventSynchronize(e1));
            ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
            best = std::max(best, (double)gblocks * 256 * reps * 128 / (ms * 1e-3) / 1e9);
        }
        if (gather128_gb_per_s) *gather128_gb_per_s = best;
        ZK_HIP(hipStreamSynchronize(st));
        (void)hipFree(tab);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

extern "C" int zkmi_calibrate_code_fetch(double* small_loop_gmul_per_s, double* big_loop_gmul_per_s) {
    ZK_TRY(require_ctx());
    Ctx& cx = ctx();
    hipStream_t st = cx.stream;
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0)); ZK_HIP(hipEventCreate(&e1));
    uint32_t* d_out;
    ZK_TRY(ws_get("calib.out", (size_t)4 << 20, (void**)&d_out));
    const int blocks = 256 * 2;                                        // two 4-wave workgroups per CU: 2 waves per SIMD
    float ms = 0;
    double best[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++)
        for (int which = 0; which < 2; which++) {
            ZK_HIP(hipEventRecord(e0, st));
            if (which == 0) hipLaunchKernelGGL((k_calib_code<Bn254Fq, 4>), dim3(blocks), dim3(256), 0, st, d_out, 480);
            else hipLaunchKernelGGL((k_calib_code<Bn254Fq, 48>), dim3(blocks), dim3(256), 0, st, d_out, 40);
            ZK_HIP(hipEventRecord(e1, st));
            ZK_HIP(hipEventSynchronize(e1));
            ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
            best[which] = std::max(best[which], (double)blocks * 256 * 3840 / (ms * 1e-3) / 1e9);      // 480 x 8 = 40 x 96 = 3840 products per lane
        }
    if (small_loop_gmul_per_s) *small_loop_gmul_per_s = best[0];
    if (big_loop_gmul_per_s) *big_loop_gmul_per_s = best[1];
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
