// tools/maddbench.hip — throughput of the MSM hot loop body (XYZZ mixed addition) per field type (development aid).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I snarkjs_amd/csrc [-DZKMI_FP2_NOINLINE=1] tools/maddbench.hip -o tools/maddbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "curve.cuh"
using namespace zkmi;

#ifndef BENCH_MINBLK
#define BENCH_MINBLK 1
#endif
#ifndef BENCH_PREFETCH
#define BENCH_PREFETCH 1
#endif
template <class F> __global__ void __launch_bounds__(256, BENCH_MINBLK) k_bench(const uint32_t* __restrict__ table, uint32_t T, int iters, uint32_t* __restrict__ out) {
    constexpr int FW = FieldWords<F>::value;
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<F> acc;
    pt_set_inf(acc);
    uint32_t idx = (lane * 2654435761u) % T;
    Affine<F> qn;
    pt_load(qn, table + (size_t)idx * 2 * FW);
    for (int i = 0; i < iters; i++) {
#if BENCH_PREFETCH
        Affine<F> q = qn;
        idx = (idx * 1664525u + 1013904223u) % T;
        pt_load(qn, table + (size_t)idx * 2 * FW);
#else
        Affine<F> q;
        idx = (idx * 1664525u + 1013904223u) % T;
        pt_load(q, table + (size_t)idx * 2 * FW);
#endif
        pt_madd(acc, q);
    }
    pt_store(out + (size_t)lane * 4 * FW, acc);
}

template <class F> void run(const char* name, int muls_per_madd) {
    constexpr int FW = FieldWords<F>::value;
    const uint32_t T = 1 << 16;
    std::vector<uint32_t> h((size_t)T * 2 * FW);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 3; }     // arbitrary residues: timing only
    uint32_t *d_t, *d_o;
    const int blocks = 256 * 16, iters = 64;
    hipMalloc(&d_t, h.size() * 4);
    hipMalloc(&d_o, (size_t)blocks * 256 * 4 * FW * 4);
    hipMemcpy(d_t, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_bench<F>), dim3(blocks), dim3(256), 0, 0, d_t, T, iters, d_o);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)k_bench<F>);
    double madds = (double)blocks * 256 * iters;
    printf("%-16s %8.3f ms  %7.2f Gmadd/s  %7.1f Gmul-equiv/s  vgpr=%d scratch=%zu bytes code=? \n", name, best, madds / best / 1e6, madds * muls_per_madd / best / 1e6,
           fa.numRegs, (size_t)fa.localSizeBytes);
    hipFree(d_t); hipFree(d_o);
}

int main() {
#ifndef BENCH_ONLY_G2
    run<Fp<Bn254Fq>>("bn254 G1", 10);
    run<Fp<Bls12381Fq>>("bls12-381 G1", 10);
#endif
    run<Fp2<Bn254Fq>>("bn254 G2", 28);
    run<Fp2<Bls12381Fq>>("bls12-381 G2", 28);
    return 0;
}
