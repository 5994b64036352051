// tools/fieldbench29.hip — gate experiment for unsaturated 29-bit limbs (DESIGN.md 7 "next" #1): a 254-bit Montgomery product on 9 limbs of
// 29 bits with R' = 2^261. Column sums of 18 products of < 2^58 fit a 64-bit accumulator, so every MAC is ONE v_mad_u64_u32 (no carry
// counter): 171 MACs + 9 m-computations + 18 shifts + 9 masks, against 136 x 2 + 18 for the saturated 8 x 32-bit product scanning.
// Checks r * 2^5 == mont256(a, b) (mod p) on the host and prints Gmul/s in the harness of tools/fieldbench.hip.
//   hipcc --offload-arch=gfx950 -O3 -I snarkjs_amd/csrc -o tools/bin/fieldbench29 tools/fieldbench29.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "field.cuh"
#include "host_field.hpp"
using namespace zkmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr uint32_t M29 = (1u << 29) - 1;
template <class C> struct L29 {
    // limb k = bits [29k, 29k+29) of p
    __host__ __device__ static constexpr uint32_t p(int k) {
        const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
        uint64_t lo = w < C::N ? C::p(w) : 0u, hi = (w + 1 < C::N) ? C::p(w + 1) : 0u;
        return (uint32_t)(((lo | (hi << 32)) >> sh) & M29);
    }
    // -p^-1 mod 2^29
    __host__ __device__ static constexpr uint32_t np() {
        uint32_t p0 = C::p(0), inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2u - p0 * inv;
        return (0u - inv) & M29;
    }
};
__device__ __forceinline__ void mad1(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
__device__ __forceinline__ void mad1c(uint64_t& acc, uint32_t a, uint32_t k) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc"); }

template <class C> __device__ __forceinline__ void mul29(uint32_t (&r)[9], const uint32_t (&a)[9], const uint32_t (&b)[9]) {
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mad1(acc, a[i], b[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mad1c(acc, m[i], L29<C>::p(k - i));
        m[k] = ((uint32_t)acc * L29<C>::np()) & M29;
        mad1c(acc, m[k], L29<C>::p(0));
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad1(acc, a[i], b[k - i]);
#pragma unroll
        for (int i = k - 8; i < 9; i++) mad1c(acc, m[i], L29<C>::p(k - i));
        r[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
}
template <class C> __device__ __forceinline__ void unpack29(uint32_t (&l)[9], const uint32_t* w) {     // 8 x 32-bit words -> 9 limbs
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
        uint64_t v = w[wi];
        if (wi + 1 < 8) v |= (uint64_t)w[wi + 1] << 32;
        l[k] = (uint32_t)(v >> sh) & M29;
    }
}
__device__ __forceinline__ void pack29(uint32_t* w, const uint32_t (&l)[9]) {                          // 9 limbs (< 2^29 each, value < 2^256) -> 8 words
    uint64_t acc = 0; int bits = 0, wi = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc |= (uint64_t)l[k] << bits; bits += 29;
        if (bits >= 32) { w[wi++] = (uint32_t)acc; acc >>= 32; bits -= 32; }
    }
    if (wi < 8) w[wi] = (uint32_t)acc;
}
template <class C> __global__ void k_ops29(const uint32_t* a, const uint32_t* b, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[9], y[9], r[9];
    unpack29<C>(x, a + i * 8); unpack29<C>(y, b + i * 8);
    mul29<C>(r, x, y);
    pack29(out + i * 8, r);
}
template <class C, int ILP> __global__ void k_chain29(const uint32_t* a, uint32_t* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[ILP][9], y[9];
    unpack29<C>(y, a + (i & 1023) * 8);
#pragma unroll
    for (int k = 0; k < ILP; k++) unpack29<C>(x[k], a + ((i + 7 * k + 1) & 1023) * 8);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) { uint32_t t[9]; mul29<C>(t, x[k], y);
#pragma unroll
            for (int q = 0; q < 9; q++) x[k][q] = t[q]; }
    }
    uint32_t s[9];
#pragma unroll
    for (int q = 0; q < 9; q++) { s[q] = 0; for (int k = 0; k < ILP; k++) s[q] ^= x[k][q]; }
    pack29(out + (size_t)i * 8, s);
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 7; rng_state ^= rng_state >> 9; return rng_state * 0x2545F4914F6CDD1Dull; }

template <class C> int run(const char* name) {
    auto F = host::HField<4>::template from_cfg<C>();
    const int n = 4096;
    std::vector<uint32_t> a(n * 8), b(n * 8);
    for (int i = 0; i < n; i++) {
        host::HFp<4> x, y;
        for (int k = 0; k < 4; k++) { x.v[k] = rnd(); y.v[k] = rnd(); }
        x.v[3] &= F.p[3] >> 1; y.v[3] &= F.p[3] >> 1;
        if (i == 0) x = F.zero();
        if (i == 1) { for (int k = 0; k < 4; k++) x.v[k] = F.p[k]; x.v[0] -= 1; y = x; }
        memcpy(&a[i * 8], x.v, 32); memcpy(&b[i * 8], y.v, 32);
    }
    uint32_t *da, *db, *dm;
    size_t bytes = (size_t)n * 32;
    CK(hipMalloc(&da, bytes)); CK(hipMalloc(&db, bytes)); CK(hipMalloc(&dm, bytes));
    CK(hipMemcpy(da, a.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_ops29<C>), dim3(n / 256), dim3(256), 0, 0, da, db, dm, n);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> m(n * 8);
    CK(hipMemcpy(m.data(), dm, bytes, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        host::HFp<4> x, y, r; memcpy(x.v, &a[i * 8], 32); memcpy(y.v, &b[i * 8], 32); memcpy(r.v, &m[i * 8], 32);
        // r may be in [0, 2p): bring into [0, p), then r * 2^5 must equal mont256(x, y) = x*y*2^-256
        if (host::HField<4>::cmp(r.v, F.p) >= 0) { uint64_t bw = 0; for (int k = 0; k < 4; k++) { unsigned __int128 d = (unsigned __int128)r.v[k] - F.p[k] - bw; r.v[k] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; } }
        for (int k = 0; k < 5; k++) r = F.dbl(r);
        auto want = F.mul(x, y);
        if (!(r == want)) { if (bad < 3) printf("  %s mul29 mismatch at %d\n", name, i); bad++; }
    }
    printf("[29-bit limbs] %s: correctness %s (%d mismatches)\n", name, bad ? "FAIL" : "ok", bad);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount, iters = 512;
    uint32_t* out; CK(hipMalloc(&out, (size_t)CUS * 8 * 256 * 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); float best = 1e30f; for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best * 1e-3; };
    for (int wps : {1, 2, 4, 8}) {
        int blocks = CUS * wps;
        double t1 = timeit([&] { hipLaunchKernelGGL((k_chain29<C, 1>), dim3(blocks), dim3(256), 0, 0, da, out, iters); });
        double t2 = timeit([&] { hipLaunchKernelGGL((k_chain29<C, 2>), dim3(blocks), dim3(256), 0, 0, da, out, iters); });
        double muls1 = (double)blocks * 256 * iters, muls2 = muls1 * 2;
        printf("  wps=%d  mul ILP1 %.2f Gmul/s (%.0f cyc/wave-mul)  ILP2 %.2f Gmul/s (%.0f cyc)\n", wps, muls1 / t1 * 1e-9,
               t1 * prop.clockRate * 1e3 / (iters * (double)wps), muls2 / t2 * 1e-9, t2 * prop.clockRate * 1e3 / (iters * 2.0 * wps));
    }
    return bad;
}
int main() {
    int bad = 0;
    bad += run<Bn254Fq>("bn254_fq");
    bad += run<Bn254Fr>("bn254_fr");
    return bad ? 1 : 0;
}
