// tools/maddbench29_g2.hip — VERDICT r05 #6: "measure — do not estimate — the one-Fq2-component-per-lane accumulator".
// Mixed additions per second of the BN254 G2 accumulation, arithmetic only (operands in registers, no table, no lists):
//   k_g2_lds    the shipped madd29_lds (msm29.cuh): one point per lane, XYZZ accumulator parked in LDS (288 B per lane), 256-lane blocks
//   k_g2_split  the same formulas with ONE Fq2 COMPONENT PER LANE (msm29.cuh: madd29_split): the two components of a point sit eight lanes apart in a
//               row of sixteen, the accumulator is half as wide and stays in REGISTERS, every Fq2 product fetches the partner's operands with DPP moves
// Both kernels produce the same canonical words (checked here), so the comparison is of two layouts of one computation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I snarkjs_amd/csrc -I include -o tools/bin/maddbench29_g2 tools/maddbench29_g2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "msm29.cuh"
using namespace zkmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef Bn254Fq Cq;
constexpr int NLQ = Lim29<Cq>::NL;

// ---- the shipped layout -------------------------------------------------------------------------------------------------------------------
template <int MINW> __global__ void __launch_bounds__(256, MINW) k_g2_lds(const uint32_t* pts, uint32_t* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_b[];
    typedef LdsAcc29<Cq, 256, false> Acc;
    const Acc A{lds_b + threadIdx.x};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool inf = true;
    // as in k_msm_accum29_g2: the point arrives packed (here from a 16 KB table that stays in the caches) and is unpacked by the iteration that uses it
    for (int it = 0; it < iters; it++) {
        const uint32_t* p = pts + ((i + (it & 1)) & 127) * 32;
        const F2x<Cq> x{from_r256<Cq>(p), from_r256<Cq>(p + 8)}, y{from_r256<Cq>(p + 16), from_r256<Cq>(p + 24)};
        madd29_lds<Cq>(A, inf, x, y);
    }
    store_xyzz29_lds<Cq, Acc, false>(out + (size_t)i * 64, A, inf);
}

// ---- one component per lane: msm29.cuh's madd29_split (the library's own routine since the measurement below) -----------------------------------
template <int MINW> __global__ void __launch_bounds__(256, MINW) k_g2_split(const uint32_t* pts, uint32_t* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    const int comp = (threadIdx.x >> 3) & 1;                          // lanes 8-15 of every row of 16 hold c1
    const int i = ((lane >> 4) << 3) | (lane & 7);                    // the point this lane works on: 8 per row
    const uint32_t dbl = comp ? 0xffffffffu : 0u;
    AccS29<Cq> a;
    bool inf = true;
    for (int it = 0; it < iters; it++) {
        const uint32_t* p = pts + ((i + (it & 1)) & 127) * 32 + comp * 8;
        const Fp29<Cq> x = from_r256<Cq>(p), y = from_r256<Cq>(p + 16);
        madd29_split<Cq>(a, inf, x, y, dbl);
    }
    uint32_t* o = out + (size_t)i * 64 + comp * 8;
    store_r256<Cq, false>(o, a.X); store_r256<Cq, false>(o + 16, a.Y); store_r256<Cq, false>(o + 32, a.ZZ); store_r256<Cq, false>(o + 48, a.ZZZ);
#endif
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount, iters = 128;
    // 128 "points": four arbitrary canonical Fq elements each (the formulas are algebraic; nothing here needs a curve point, only P != 0)
    static uint32_t h[128 * 32];
    {
        const uint32_t pw[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};   // BN254 q
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (int e = 0; e < 128 * 4; e++) {
            uint32_t* w = h + e * 8;
            for (int k = 0; k < 8; k++) { s = s * 6364136223846793005ull + 1442695040888963407ull; w[k] = (uint32_t)(s >> 32); }
            w[7] &= 0x1fffffffu;                                       // < 2^253 < q: canonical
            (void)pw;
        }
    }
    uint32_t *pts, *oa, *ob;
    const size_t max_lanes = (size_t)CUS * 4 * 256;
    CK(hipMalloc(&pts, sizeof h)); CK(hipMalloc(&oa, max_lanes * 64 * 4)); CK(hipMalloc(&ob, max_lanes * 64 * 4));
    CK(hipMemcpy(pts, h, sizeof h, hipMemcpyHostToDevice));
    const size_t lds = (size_t)256 * 8 * NLQ * 4;                     // 73 728 B: two blocks per CU
    CK(hipFuncSetAttribute((const void*)k_g2_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // same canonical words from both layouts
    for (int n_add : {1, 2, 3, 37}) {
        hipLaunchKernelGGL((k_g2_lds<2>), dim3(2), dim3(256), lds, 0, pts, oa, n_add);
        hipLaunchKernelGGL((k_g2_split<2>), dim3(4), dim3(256), 0, 0, pts, ob, n_add);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> a(512 * 64), b(512 * 64);
        CK(hipMemcpy(a.data(), oa, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ob, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, per[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t k = 0; k < a.size(); k++) if (a[k] != b[k]) { bad++; per[(k & 63) >> 3]++; }
        printf("split layout == LDS layout on 512 accumulators after %d additions: %s (%zu differing words; by component X0 X1 Y0 Y1 ZZ0 ZZ1 ZZZ0 ZZZ1: %zu %zu %zu %zu %zu %zu %zu %zu); X[0] = %08x %08x | %08x %08x\n",
               n_add, bad ? "NO" : "yes", bad, per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7], a[0], a[1], b[0], b[1]);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); float best = 1e30f; for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best * 1e-3; };
    {
        const int blocks = CUS * 2;                                   // the shipped kernel's residency: two 256-lane blocks per CU
        const double n = (double)blocks * 256 * iters;
        const double t = timeit([&] { hipLaunchKernelGGL((k_g2_lds<2>), dim3(blocks), dim3(256), lds, 0, pts, oa, iters); });
        printf("LDS-parked accumulator, one point per lane, 2 waves per SIMD:           %.3f G additions/s\n", n / t * 1e-9);
    }
    for (int bpc : {2, 3, 4}) {                                       // blocks per CU = waves per SIMD offered; the register count decides what runs
        const int blocks = CUS * bpc;
        const double n = (double)blocks * 128 * iters;                // two lanes per point
        const double t2 = timeit([&] { hipLaunchKernelGGL((k_g2_split<2>), dim3(blocks), dim3(256), 0, 0, pts, ob, iters); });
        const double t3 = timeit([&] { hipLaunchKernelGGL((k_g2_split<3>), dim3(blocks), dim3(256), 0, 0, pts, ob, iters); });
        const double t4 = timeit([&] { hipLaunchKernelGGL((k_g2_split<4>), dim3(blocks), dim3(256), 0, 0, pts, ob, iters); });
        printf("one component per lane, %d blocks per CU offered: bounds(256,2) %.3f  bounds(256,3) %.3f  bounds(256,4) %.3f G additions/s\n", bpc, n / t2 * 1e-9, n / t3 * 1e-9, n / t4 * 1e-9);
    }
    return 0;
}
