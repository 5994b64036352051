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

template <class C> ZK_DEV void f_from_words(Fp2<C>& v, const uint32_t* w) {
#pragma unroll
    for (int i = 0; i < C::N; i++) { v.c0.l[i] = w[i]; v.c1.l[i] = w[C::N + i]; } }
template <class C> ZK_DEV void f_to_words(uint32_t* w, const Fp2<C>& v) {
#pragma unroll
    for (int i = 0; i < C::N; i++) { w[i] = v.c0.l[i]; w[C::N + i] = v.c1.l[i]; } }
// variant: accumulator parked in LDS (transposed: dword i of lane t at i*256 + t), coordinates pulled into registers only
// while they are needed — keeps the Fq2 working set under 256 VGPRs without scratch spills.
template <class F> struct LdsAcc {
    static constexpr int FW = FieldWords<F>::value;
    uint32_t* base;     // &lds[threadIdx.x]
    ZK_DEV void get(int coord, F& v) const { uint32_t w[FW]; 
#pragma unroll
        for (int i = 0; i < FW; i++) w[i] = base[(coord * FW + i) * 256];
        f_from_words(v, w); }
    ZK_DEV void put(int coord, const F& v) const { uint32_t w[FW]; f_to_words(w, v);
#pragma unroll
        for (int i = 0; i < FW; i++) base[(coord * FW + i) * 256] = w[i]; }
};
// acc += q with acc in LDS; caller guarantees q != inf; flag `inf` kept in a register
template <class F> ZK_DEV void pt_madd_lds(const LdsAcc<F>& A, bool& inf, const Affine<F>& q) {
    if (inf) { A.put(0, q.x); A.put(1, q.y); F one; f_set_one(one); A.put(2, one); A.put(3, one); inf = false; return; }
    F t, P, R;
    A.get(2, t); P = f_mul(q.x, t);            // U2
    A.get(0, t); P = f_sub(P, t);              // P = U2 - X
    A.get(3, t); R = f_mul(q.y, t);            // S2
    A.get(1, t); R = f_sub(R, t);              // R = S2 - Y
    if (f_is_zero(P)) {
        if (f_is_zero(R)) { XYZZ<F> d = pt_dbl_affine(q); A.put(0, d.X); A.put(1, d.Y); A.put(2, d.ZZ); A.put(3, d.ZZZ); }
        else inf = true;
        return;
    }
    F PP = f_sqr(P);
    A.get(2, t); A.put(2, f_mul(t, PP));       // ZZ' = ZZ*PP
    A.get(0, t); F Q = f_mul(t, PP);           // Q = X*PP
    F PPP = f_mul(P, PP);
    A.get(3, t); A.put(3, f_mul(t, PPP));      // ZZZ' = ZZZ*PPP
    F X3 = f_sub(f_sub(f_sqr(R), PPP), f_dbl(Q));
    A.put(0, X3);
    A.get(1, t);
    A.put(1, f_sub(f_mul(R, f_sub(Q, X3)), f_mul(t, PPP)));
}
template <class F> __global__ void __launch_bounds__(256, 2) k_bench_lds(const uint32_t* __restrict__ table, uint32_t T, int iters, uint32_t* __restrict__ out) {
    constexpr int FW = FieldWords<F>::value;
    extern __shared__ uint32_t lds[];
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    LdsAcc<F> A{lds + threadIdx.x};
    bool inf = true;
    uint32_t idx = (lane * 2654435761u) % T;
    for (int i = 0; i < iters; i++) {
        Affine<F> q;
        idx = (idx * 1664525u + 1013904223u) % T;
        pt_load(q, table + (size_t)idx * 2 * FW);
        pt_madd_lds(A, inf, q);
    }
    XYZZ<F> acc;
    if (inf) pt_set_inf(acc); else { A.get(0, acc.X); A.get(1, acc.Y); A.get(2, acc.ZZ); A.get(3, acc.ZZZ); }
    pt_store(out + (size_t)lane * 4 * FW, acc);
}
template <class F> void run_lds(const char* name, int muls_per_madd) {
    constexpr int FW = FieldWords<F>::value;
    const uint32_t T = 1 << 16;
    std::vector<uint32_t> h((size_t)T * 2 * FW);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 3; }
    uint32_t *d_t, *d_o;
    const int blocks = 256 * 16, iters = 64;
    hipMalloc(&d_t, h.size() * 4);
    hipMalloc(&d_o, (size_t)blocks * 256 * 4 * FW * 4);
    hipMemcpy(d_t, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const size_t ldsb = (size_t)4 * FW * 256 * 4;
    hipFuncSetAttribute((const void*)k_bench_lds<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_bench_lds<F>), dim3(blocks), dim3(256), ldsb, 0, d_t, T, iters, d_o);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)k_bench_lds<F>);
    double madds = (double)blocks * 256 * iters;
    printf("%-16s LDS-acc %8.3f ms  %7.2f Gmadd/s  %7.1f Gmul-equiv/s  vgpr=%d scratch=%zu bytes (err=%s)\n", name, best, madds / best / 1e6, madds * muls_per_madd / best / 1e6,
           fa.numRegs, (size_t)fa.localSizeBytes, hipGetErrorString(hipGetLastError()));
    hipFree(d_t); hipFree(d_o);
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
    run_lds<Fp2<Bn254Fq>>("bn254 G2", 28);
    run<Fp2<Bls12381Fq>>("bls12-381 G2", 28);
    return 0;
}
