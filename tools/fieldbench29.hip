// tools/fieldbench29.hip — ceilings of the unsaturated-limb Montgomery product of csrc/field29.cuh on this chip: 9 x 29-bit limbs (BN254 Fq / Fr,
// BLS12-381 Fr) and 14 x 28-bit limbs (BLS12-381 Fq), the LIBRARY's own mul29 / sqr29 (not a copy), as chains of dependent products at 1, 2, 4
// and 8 waves per SIMD, one and two independent chains per lane. bench.py's int_alu.peak quotes these numbers (profiles/rNN_fieldbench29.txt).
// Correctness of the device compilation: store_r256(mul29(from_r256(a), from_r256(b))) must equal the host's Montgomery product of the same
// R-form words (the CPU-side check of the same code is tests/test_field29_host.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I snarkjs_amd/csrc -o tools/bin/fieldbench29 tools/fieldbench29.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "field29.cuh"
#include "host_field.hpp"
using namespace zkmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class C> __global__ void k_ops29(const uint32_t* a, const uint32_t* b, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fp29<C> x = from_r256<C>(a + (size_t)i * C::N), y = from_r256<C>(b + (size_t)i * C::N);
    Fp29<C> r = (i & 1) ? mul29(x, y) : ((i & 2) ? mul29_2(x, y, zero29<C>(), y) : mul29(y, x));
    if ((i & 7) == 7) r = sqr29(x);                          // lanes 7 mod 8: a^2 (the host checks against a * a)
    store_r256<C, false>(out + (size_t)i * C::N, r);
}
template <class C, int ILP, bool SQR> __global__ void __launch_bounds__(256) k_chain29(const uint32_t* a, uint32_t* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fp29<C> x[ILP], y = load29_packed<C>(a + (size_t)(i & 1023) * C::N);
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = load29_packed<C>(a + (size_t)((i + 7 * k + 1) & 1023) * C::N);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = SQR ? sqr29(x[k]) : mul29(x[k], y);
    }
    Fp29<C> s = x[0];
#pragma unroll
    for (int k = 1; k < ILP; k++) s = add29(s, x[k]);
    store_r256<C, true>(out + (size_t)i * C::N, s);
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 7; rng_state ^= rng_state >> 9; return rng_state * 0x2545F4914F6CDD1Dull; }

template <class C, int L> int run(const char* name) {
    auto F = host::HField<L>::template from_cfg<C>();
    const int n = 4096, W = C::N;
    std::vector<uint32_t> a((size_t)n * W), b((size_t)n * W);
    for (int i = 0; i < n; i++) {
        host::HFp<L> x, y;
        for (int k = 0; k < L; k++) { x.v[k] = rnd(); y.v[k] = rnd(); }
        x.v[L - 1] &= F.p[L - 1] >> 1; y.v[L - 1] &= F.p[L - 1] >> 1;               // canonical (< p)
        if (i == 0) x = F.zero();
        if (i == 1) { for (int k = 0; k < L; k++) x.v[k] = F.p[k]; x.v[0] -= 1; y = x; }
        memcpy(&a[(size_t)i * W], x.v, 4 * W); memcpy(&b[(size_t)i * W], y.v, 4 * W);
    }
    uint32_t *da, *db, *dm;
    size_t bytes = (size_t)n * 4 * W;
    CK(hipMalloc(&da, bytes)); CK(hipMalloc(&db, bytes)); CK(hipMalloc(&dm, bytes));
    CK(hipMemcpy(da, a.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_ops29<C>), dim3(n / 256), dim3(256), 0, 0, da, db, dm, n);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> m((size_t)n * W);
    CK(hipMemcpy(m.data(), dm, bytes, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        host::HFp<L> x, y, r; memcpy(x.v, &a[(size_t)i * W], 4 * W); memcpy(y.v, &b[(size_t)i * W], 4 * W); memcpy(r.v, &m[(size_t)i * W], 4 * W);
        auto want = ((i & 7) == 7) ? F.mul(x, x) : F.mul(x, y);
        if (!(r == want)) { if (bad < 3) printf("  %s mismatch at %d\n", name, i); bad++; }
    }
    printf("[%d x %d-bit limbs] %s: device mul29 / mul29_2 / sqr29 vs host Montgomery product: %s (%d mismatches)\n", Lim29<C>::NL, Lim29<C>::B, name, bad ? "FAIL" : "ok", bad);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount, iters = 256;
    uint32_t* out; CK(hipMalloc(&out, (size_t)CUS * 8 * 256 * 4 * W));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); float best = 1e30f; for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best * 1e-3; };
    for (int wps : {1, 2, 4, 8}) {
        int blocks = CUS * wps;
        double t1 = timeit([&] { hipLaunchKernelGGL((k_chain29<C, 1, false>), dim3(blocks), dim3(256), 0, 0, da, out, iters); });
        double t2 = timeit([&] { hipLaunchKernelGGL((k_chain29<C, 2, false>), dim3(blocks), dim3(256), 0, 0, da, out, iters); });
        double t3 = timeit([&] { hipLaunchKernelGGL((k_chain29<C, 1, true>), dim3(blocks), dim3(256), 0, 0, da, out, iters); });
        double muls1 = (double)blocks * 256 * iters;
        printf("  wps=%d  mul ILP1 %.2f Gmul/s  ILP2 %.2f Gmul/s  sqr %.2f G/s\n", wps, muls1 / t1 * 1e-9, 2 * muls1 / t2 * 1e-9, muls1 / t3 * 1e-9);
    }
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dm)); CK(hipFree(out));
    return bad;
}
int main() {
    int bad = 0;
    bad += run<Bn254Fq, 4>("bn254_fq");
    bad += run<Bls12381Fq, 6>("bls12381_fq");
    bad += run<Bn254Fr, 4>("bn254_fr");
    bad += run<Bls12381Fr, 4>("bls12381_fr");
    return bad ? 1 : 0;
}
