// tools/maddbench29.hip — mixed additions per second: pt_madd on saturated 32-bit limbs (curve.cuh) vs madd29 (msm29.cuh), no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -I snarkjs_amd/csrc -I include -o tools/bin/maddbench29 tools/maddbench29.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "msm29.cuh"
using namespace zkmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MINW> __global__ void __launch_bounds__(256, MINW) k_madd32(const uint32_t* pts, uint32_t* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Affine<Fp<Bn254Fq>> q0, q1;
    pt_load(q0, pts + (i & 255) * 16); pt_load(q1, pts + ((i + 1) & 255) * 16);
    XYZZ<Fp<Bn254Fq>> acc; pt_set_inf(acc);
    for (int it = 0; it < iters; it++) { pt_madd(acc, (it & 1) ? q1 : q0); }
    pt_store(out + (size_t)i * 32, acc);
}
template <int MINW> __global__ void __launch_bounds__(256, MINW) k_madd29(const uint32_t* pts, uint32_t* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Aff29<Bn254Fq> q0, q1;
    const uint32_t* p0 = pts + (i & 255) * 16; const uint32_t* p1 = pts + ((i + 1) & 255) * 16;
    q0.x = from_r256<Bn254Fq>(p0); q0.y = from_r256<Bn254Fq>(p0 + 8); q1.x = from_r256<Bn254Fq>(p1); q1.y = from_r256<Bn254Fq>(p1 + 8);
    XYZZ29<Bn254Fq> acc; bool inf = true;
    for (int it = 0; it < iters; it++) { madd29(acc, inf, (it & 1) ? q1 : q0); }
    store_xyzz29(out + (size_t)i * 32, acc, inf);
}
// affine points of y^2 = x^3 + 3 are not needed for timing, but the two kernels must agree on real points: host supplies k*G multiples
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount, iters = 256;
    // 256 points: small multiples of the generator computed on the device with the 32-bit path
    uint32_t *pts, *o32, *o29;
    CK(hipMalloc(&pts, 256 * 64)); CK(hipMalloc(&o32, (size_t)CUS * 4 * 256 * 128)); CK(hipMalloc(&o29, (size_t)CUS * 4 * 256 * 128));
    {
        // even slots: G = (1, 2), odd slots: 3G (Montgomery form, R = 2^256): lane i alternates between its slot and the next one
        const uint32_t g1x[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        const uint32_t g1y[8] = {0x8b1e1b3au, 0xa6ba871bu, 0xeb8e167bu, 0x14f1d651u, 0xf0f28c58u, 0xccdd46deu, 0x340fbe5eu, 0x1c14ef83u};
        const uint32_t g3x[8] = {0x248b11a0u, 0x9d831b4eu, 0x77f54b7eu, 0x91f18c06u, 0xbfda15d0u, 0x0ee5ea95u, 0x28c70539u, 0x10f0baf6u};
        const uint32_t g3y[8] = {0x983653eau, 0xbe1523eau, 0x7c629a1au, 0xe86e4817u, 0x6d2a214au, 0x51cc9f8eu, 0xca757913u, 0x014925f0u};
        static uint32_t h[256 * 16];
        for (int i = 0; i < 256; i++) for (int k = 0; k < 8; k++) { h[i * 16 + k] = (i & 1) ? g3x[k] : g1x[k]; h[i * 16 + 8 + k] = (i & 1) ? g3y[k] : g1y[k]; }
        CK(hipMemcpy(pts, h, sizeof h, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); float best = 1e30f; for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best * 1e-3; };
    for (int wps : {2, 3, 4}) {
        const int blocks = CUS * wps;
        const double n = (double)blocks * 256 * iters;
        double t32 = timeit([&] { hipLaunchKernelGGL((k_madd32<1>), dim3(blocks), dim3(256), 0, 0, pts, o32, iters); });
        double t29a = timeit([&] { hipLaunchKernelGGL((k_madd29<1>), dim3(blocks), dim3(256), 0, 0, pts, o29, iters); });
        double t29b = timeit([&] { hipLaunchKernelGGL((k_madd29<3>), dim3(blocks), dim3(256), 0, 0, pts, o29, iters); });
        double t29c = timeit([&] { hipLaunchKernelGGL((k_madd29<4>), dim3(blocks), dim3(256), 0, 0, pts, o29, iters); });
        printf("blocks/CU=%d  madd32 %.2f G/s   madd29(bounds 1) %.2f G/s   madd29(bounds 3) %.2f G/s   madd29(bounds 4) %.2f G/s\n", wps, n / t32 * 1e-9, n / t29a * 1e-9, n / t29b * 1e-9, n / t29c * 1e-9);
    }
    // agreement of the two paths on the final accumulators (both in the reference's R-form)
    std::vector<uint32_t> a(256 * 32), b(256 * 32);
    hipLaunchKernelGGL((k_madd32<1>), dim3(1), dim3(256), 0, 0, pts, o32, 37); hipLaunchKernelGGL((k_madd29<1>), dim3(1), dim3(256), 0, 0, pts, o29, 37);
    CK(hipMemcpy(a.data(), o32, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o29, b.size() * 4, hipMemcpyDeviceToHost));
    printf("same XYZZ representative: %s\n", memcmp(a.data(), b.data(), a.size() * 4) ? "NO (projective representatives may differ only if the formulas differ)" : "yes");
    return 0;
}
