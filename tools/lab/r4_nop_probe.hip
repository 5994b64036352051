#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void mad(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc"); }
// V1: one dependent chain
extern "C" __global__ void k_v1(const uint32_t* a, const uint32_t* b, uint64_t* o) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; i++) { x[i] = a[threadIdx.x * 9 + i]; y[i] = b[threadIdx.x * 9 + i]; }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) mad(acc, x[i], y[8 - i]);
    o[threadIdx.x] = acc;
}
// V2: two chains, interleaved in program order
extern "C" __global__ void k_v2(const uint32_t* a, const uint32_t* b, uint64_t* o) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; i++) { x[i] = a[threadIdx.x * 9 + i]; y[i] = b[threadIdx.x * 9 + i]; }
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { mad(acc0, x[i], y[8 - i]); mad(acc1, x[i], y[i]); }
    o[threadIdx.x] = acc0 ^ acc1;
}
// V3: one chain, builtin multiply-add instead of inline asm (compiler's own v_mad_u64_u32)
extern "C" __global__ void k_v3(const uint32_t* a, const uint32_t* b, uint64_t* o) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; i++) { x[i] = a[threadIdx.x * 9 + i]; y[i] = b[threadIdx.x * 9 + i]; }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) acc += (uint64_t)x[i] * y[8 - i];
    o[threadIdx.x] = acc;
}
// V4: one chain, carry-out to an SGPR pair other than vcc
__device__ __forceinline__ void mad_s(uint64_t& acc, uint32_t a, uint32_t b) { uint64_t c; asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "v"(b)); }
extern "C" __global__ void k_v4(const uint32_t* a, const uint32_t* b, uint64_t* o) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; i++) { x[i] = a[threadIdx.x * 9 + i]; y[i] = b[threadIdx.x * 9 + i]; }
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) mad_s(acc, x[i], y[8 - i]);
    o[threadIdx.x] = acc;
}
// truly serial: each MAC's multiplicand is the low word of the previous accumulator
extern "C" __global__ void k_serial(const uint32_t* b, uint64_t* o) {
    uint32_t y[9];
    for (int i = 0; i < 9; i++) y[i] = b[threadIdx.x * 9 + i];
    uint64_t acc = y[0];
#pragma unroll
    for (int i = 0; i < 9; i++) acc = acc + (uint64_t)(uint32_t)acc * y[i];
    o[threadIdx.x] = acc;
}
// product-scanning column sums in plain C: 3 columns of a 9x9 product
extern "C" __global__ void k_cols(const uint32_t* a, const uint32_t* b, uint64_t* o) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; i++) { x[i] = a[threadIdx.x * 9 + i]; y[i] = b[threadIdx.x * 9 + i]; }
    uint64_t acc = 0; uint32_t r[3];
#pragma unroll
    for (int k = 6; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)x[i] * y[k - i];
        r[k - 6] = (uint32_t)acc & 0x1fffffffu; acc >>= 29;
    }
    o[threadIdx.x] = acc + r[0] + r[1] + r[2];
}
// constant (SGPR / literal) multiplicand
extern "C" __global__ void k_const(const uint32_t* a, uint64_t* o) {
    uint32_t x[4];
    for (int i = 0; i < 4; i++) x[i] = a[threadIdx.x * 4 + i];
    uint64_t acc = 0;
    acc += (uint64_t)x[0] * 0x187cfd47u; acc += (uint64_t)x[1] * 0x010460b6u; acc += (uint64_t)x[2] * 0x1c72a34fu; acc += (uint64_t)x[3] * 0x02d522d0u;
    o[threadIdx.x] = acc;
}
// two chains interleaved, each with its OWN hard-coded carry-out SGPR pair (no shared register between consecutive statements)
__device__ __forceinline__ void madA(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, s[90:91], %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "s90", "s91"); }
__device__ __forceinline__ void madB(uint64_t& acc, uint32_t a, uint32_t b) { asm("v_mad_u64_u32 %0, s[92:93], %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "s92", "s93"); }
extern "C" __global__ void k_two(const uint32_t* a, const uint32_t* b, uint64_t* o) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; i++) { x[i] = a[threadIdx.x * 9 + i]; y[i] = b[threadIdx.x * 9 + i]; }
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { madA(acc0, x[i], y[8 - i]); madB(acc1, x[i], y[i]); }
    o[threadIdx.x] = acc0 ^ acc1;
}
// one asm statement holding TWO multiply-adds of two chains
__device__ __forceinline__ void mad2(uint64_t& a0, uint64_t& a1, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1" : "+v"(a0), "+v"(a1) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
extern "C" __global__ void k_pair(const uint32_t* a, const uint32_t* b, uint64_t* o) {
    uint32_t x[9], y[9];
    for (int i = 0; i < 9; i++) { x[i] = a[threadIdx.x * 9 + i]; y[i] = b[threadIdx.x * 9 + i]; }
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) mad2(acc0, acc1, x[i], y[8 - i], x[i], y[i]);
    o[threadIdx.x] = acc0 ^ acc1;
}
