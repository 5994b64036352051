// tools/ubench_int.hip — instruction-rate microbenchmarks that calibrate the field-arithmetic design for gfx950.
// Not part of the product path. Build: hipcc --offload-arch=gfx950 -O3 -o ubench_int tools/ubench_int.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;

// K independent chains per lane, each ITERS long
template <int K> __global__ void k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[K];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = k + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K> __global__ void k_mad64_addc(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[K]; uint32_t hi[K];
    uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) { acc[k] = k + threadIdx.x; hi[k] = 0; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[k]), "+v"(hi[k]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k] + hi[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K> __global__ void k_mullo(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[K]; uint32_t y = b + blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = a + k + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[k]) : "v"(y));
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K> __global__ void k_mulhi(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[K]; uint32_t y = b + blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = a + k + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[k]) : "v"(y));
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K> __global__ void k_addco(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t lo[K], hi[K]; uint32_t y = b + blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) { lo[k] = a + k + threadIdx.x; hi[k] = k; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo[k]), "+v"(hi[k]) : "v"(y) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += lo[k] + hi[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K> __global__ void k_add64(uint64_t* out, uint32_t a, uint32_t b) {   // v_lshl_add_u64
    uint64_t acc[K]; uint64_t y = b + blockIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = a + k + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(y));
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K> __global__ void k_mad24(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t acc[K]; uint32_t y = b + blockIdx.x, z = a ^ 0x1234;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = a + k + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(y), "v"(z));
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K> __global__ void k_fma64(uint64_t* out, uint32_t a, uint32_t b) {
    double acc[K]; double y = 1.0 + 1e-9 * b, z = 1e-3 * a;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = a + k + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(y), "v"(z));
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
template <int K> __global__ void k_fma32(uint64_t* out, uint32_t a, uint32_t b) {
    float acc[K]; float y = 1.0f + 1e-7f * b, z = 1e-3f * a;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = a + k + threadIdx.x;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(y), "v"(z));
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < K; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

template <typename F> double time_kernel(F launch, int reps = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int CUS = prop.multiProcessorCount;
    uint64_t* out; CK(hipMalloc(&out, (size_t)CUS * 8 * 1024 * 8));
    // waves per SIMD sweep: blocks of 256 threads (4 waves = 1 per SIMD); blocks per CU = wps
    for (int wps : {1, 2, 4}) {
        int blocks = CUS * wps;
#define RUN(NAME, KERN, K, OPS_PER)                                                                              \
        {                                                                                                       \
            double t = time_kernel([&] { hipLaunchKernelGGL((KERN<K>), dim3(blocks), dim3(256), 0, 0, out, 3u, 5u); }); \
            double ops = (double)blocks * 256 * ITERS * K * OPS_PER;                                             \
            double per_simd_cycle = ops / 64.0 / (CUS * 4.0) / (t * prop.clockRate * 1e3);                        \
            printf("%-22s K=%d wps=%d  %.3f ms  %.2f Tops/s  wave-inst/clk/SIMD %.3f (cyc/inst %.2f)\n", NAME, K, wps, t * 1e3, ops / t * 1e-12, per_simd_cycle, 1.0 / per_simd_cycle); \
        }
        RUN("v_mad_u64_u32", k_mad64, 1, 1) RUN("v_mad_u64_u32", k_mad64, 4, 1) RUN("v_mad_u64_u32", k_mad64, 8, 1)
        RUN("mad64+addc (2 inst)", k_mad64_addc, 4, 2) RUN("mad64+addc (2 inst)", k_mad64_addc, 8, 2)
        RUN("v_mul_lo_u32", k_mullo, 1, 1) RUN("v_mul_lo_u32", k_mullo, 8, 1)
        RUN("v_mul_hi_u32", k_mulhi, 8, 1)
        RUN("add_co+addc (2 inst)", k_addco, 1, 2) RUN("add_co+addc (2 inst)", k_addco, 8, 2)
        RUN("v_lshl_add_u64", k_add64, 1, 1) RUN("v_lshl_add_u64", k_add64, 8, 1)
        RUN("v_mad_u32_u24", k_mad24, 8, 1)
        RUN("v_fma_f64", k_fma64, 1, 1) RUN("v_fma_f64", k_fma64, 8, 1)
        RUN("v_fma_f32", k_fma32, 8, 1)
    }
    return 0;
}
