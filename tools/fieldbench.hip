// tools/fieldbench.hip — correctness (vs host_field.hpp) and throughput of the device Montgomery multiplier.
// Build per variant: hipcc --offload-arch=gfx950 -O3 -DZKMI_MUL_VARIANT=k -I snarkjs_amd/csrc -o fieldbench_vk tools/fieldbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "field.cuh"
#include "host_field.hpp"
using namespace zkmi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class C> __global__ void k_ops(const uint32_t* a, const uint32_t* b, uint32_t* mul, uint32_t* add, uint32_t* sub, uint32_t* inv, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<C> x = fp_load<C>(a + i * C::N), y = fp_load<C>(b + i * C::N);
    fp_store<C>(mul + i * C::N, fp_mul(x, y));
    fp_store<C>(add + i * C::N, fp_add(x, y));
    fp_store<C>(sub + i * C::N, fp_sub(x, y));
    if (i < 64) fp_store<C>(inv + i * C::N, fp_inv(x));
}
template <class C, int ILP> __global__ void k_chain(const uint32_t* a, uint32_t* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fp<C> x[ILP], y = fp_load<C>(a + (i & 1023) * C::N);
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = fp_load<C>(a + ((i + 7 * k + 1) & 1023) * C::N);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = fp_mul(x[k], y);
    }
    Fp<C> s = x[0];
#pragma unroll
    for (int k = 1; k < ILP; k++) s = fp_add(s, x[k]);
    fp_store<C>(out + (size_t)i * C::N, s);
}
template <class C> __global__ void k_chain_addsub(const uint32_t* a, uint32_t* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fp<C> x = fp_load<C>(a + (i & 1023) * C::N), y = fp_load<C>(a + ((i + 1) & 1023) * C::N);
    for (int it = 0; it < iters; it++) { x = fp_add(x, y); y = fp_sub(y, x); }
    fp_store<C>(out + (size_t)i * C::N, fp_add(x, y));
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 7; rng_state ^= rng_state >> 9; return rng_state * 0x2545F4914F6CDD1Dull; }

template <class C, int L> int run(const char* name) {
    auto F = host::HField<L>::template from_cfg<C>();
    const int n = 4096, N = C::N;
    std::vector<uint32_t> a(n * N), b(n * N);
    for (int i = 0; i < n; i++) {
        typename host::HField<L>::E x, y;
        for (int k = 0; k < L; k++) { x.v[k] = rnd(); y.v[k] = rnd(); }
        x.v[L - 1] &= F.p[L - 1] >> 1; y.v[L - 1] &= F.p[L - 1] >> 1;      // < p
        if (i == 0) x = F.zero();
        if (i == 1) { for (int k = 0; k < L; k++) x.v[k] = F.p[k]; x.v[0] -= 1; y = x; }   // p-1
        if (i == 2) y = F.One();
        memcpy(&a[i * N], x.v, 4 * N); memcpy(&b[i * N], y.v, 4 * N);
    }
    uint32_t *da, *db, *dm, *dadd, *dsub, *dinv;
    size_t bytes = (size_t)n * N * 4;
    CK(hipMalloc(&da, bytes)); CK(hipMalloc(&db, bytes)); CK(hipMalloc(&dm, bytes)); CK(hipMalloc(&dadd, bytes)); CK(hipMalloc(&dsub, bytes)); CK(hipMalloc(&dinv, bytes));
    CK(hipMemcpy(da, a.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_ops<C>), dim3(n / 256), dim3(256), 0, 0, da, db, dm, dadd, dsub, dinv, n);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> m(n * N), ad(n * N), sb(n * N), iv(n * N);
    CK(hipMemcpy(m.data(), dm, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(ad.data(), dadd, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sb.data(), dsub, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(iv.data(), dinv, bytes, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; i++) {
        typename host::HField<L>::E x, y; memcpy(x.v, &a[i * N], 4 * N); memcpy(y.v, &b[i * N], 4 * N);
        auto wm = F.mul(x, y), wa = F.add(x, y), ws = F.sub(x, y);
        if (memcmp(wm.v, &m[i * N], 4 * N)) { if (bad < 3) printf("  %s mul mismatch at %d\n", name, i); bad++; }
        if (memcmp(wa.v, &ad[i * N], 4 * N)) { if (bad < 3) printf("  %s add mismatch at %d\n", name, i); bad++; }
        if (memcmp(ws.v, &sb[i * N], 4 * N)) { if (bad < 3) printf("  %s sub mismatch at %d\n", name, i); bad++; }
        if (i < 64) { auto wi = F.inv(x); if (memcmp(wi.v, &iv[i * N], 4 * N)) { if (bad < 3) printf("  %s inv mismatch at %d\n", name, i); bad++; } }
    }
    printf("[variant %d] %s: correctness %s (%d mismatches)\n", ZKMI_MUL_VARIANT, name, bad ? "FAIL" : "ok", bad);
    // throughput
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUS = prop.multiProcessorCount, iters = 512;
    uint32_t* out; CK(hipMalloc(&out, (size_t)CUS * 8 * 256 * N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); float best = 1e30f; for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } return best * 1e-3; };
    for (int wps : {1, 2, 4, 8}) {
        int blocks = CUS * wps;
        double t1 = timeit([&] { hipLaunchKernelGGL((k_chain<C, 1>), dim3(blocks), dim3(256), 0, 0, da, out, iters); });
        double t2 = timeit([&] { hipLaunchKernelGGL((k_chain<C, 2>), dim3(blocks), dim3(256), 0, 0, da, out, iters); });
        double ta = timeit([&] { hipLaunchKernelGGL((k_chain_addsub<C>), dim3(blocks), dim3(256), 0, 0, da, out, iters * 8); });
        double muls1 = (double)blocks * 256 * iters, muls2 = muls1 * 2, as = (double)blocks * 256 * iters * 8 * 2;
        printf("  wps=%d  mul ILP1 %.2f Gmul/s (%.0f cyc/wave-mul)  ILP2 %.2f Gmul/s (%.0f cyc)  add/sub %.1f Gop/s\n", wps, muls1 / t1 * 1e-9,
               t1 * prop.clockRate * 1e3 / (iters * (double)wps), muls2 / t2 * 1e-9, t2 * prop.clockRate * 1e3 / (iters * 2.0 * wps), as / ta * 1e-9);
    }
    CK(hipFree(out));
    return bad;
}
int main() {
    int bad = 0;
    bad += run<Bn254Fr, 4>("bn254_fr");
    bad += run<Bn254Fq, 4>("bn254_fq");
    bad += run<Bls12381Fr, 4>("bls12381_fr");
    bad += run<Bls12381Fq, 6>("bls12381_fq");
    return bad ? 1 : 0;
}
