// tools/gatherbench.hip — calibration of rocprofv3's FETCH_SIZE for the MSM's access pattern: random 64-byte (G1 affine point) and
// 128-byte (G2) gathers, 16-byte loads per lane as pt_load issues them, from a table far larger than L2 + Infinity Cache.
// The guide calibrates FETCH_SIZE only for wide coalesced streaming reads (x2); VERDICT r01 weak #7 asks for the gather case.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gatherbench tools/gatherbench.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o g -- tools/bin/gatherbench
// Known bytes per launch: lanes x PB (printed). Kernels: k_gather<4> = 64 B per lane, k_gather<8> = 128 B, k_stream = coalesced.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int V> __global__ void k_gather(const uint4* __restrict__ tab, size_t n_pts, uint32_t* __restrict__ out, uint32_t lanes) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    uint64_t h = (uint64_t)t * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint4* p = tab + (h % n_pts) * V;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < V; i++) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    out[t] = acc;
}
// r06: a 128-byte entry read by TWO lanes eight apart in a row of sixteen, each its own 2 x 32 bytes (k_msm_accum29_g2s: one Fq2 component per lane)
__global__ void k_gather_pair(const uint4* __restrict__ tab, size_t n_pts, uint32_t* __restrict__ out, uint32_t lanes) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    const uint32_t comp = (t >> 3) & 1u, slot = ((t >> 4) << 3) | (t & 7u);
    uint64_t h = (uint64_t)slot * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint4* p = tab + (h % n_pts) * 8 + comp * 2;
    const uint4 a = p[0], b = p[1], c = p[4], d = p[5];
    out[t] = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
}
// the 14-limb curve's 192-byte G2 entry the same way: each of the two lanes reads 2 x 48 bytes (x.c_s at 48 s, y.c_s at 96 + 48 s)
__global__ void k_gather_pair192(const uint4* __restrict__ tab, size_t n_pts, uint32_t* __restrict__ out, uint32_t lanes) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    const uint32_t comp = (t >> 3) & 1u, slot = ((t >> 4) << 3) | (t & 7u);
    uint64_t h = (uint64_t)slot * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint4* p = tab + (h % n_pts) * 12 + comp * 3;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) { const uint4 a = p[i], b = p[6 + i]; acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w; }
    out[t] = acc;
}
// the same 64-byte gathers, but lane t only picks inside a window of `win_pts` points that slides with the lane index: what the MSM
// would do if every bucket list were sorted by table row (all lanes in flight gather from the same ~64 MB row at the same time)
__global__ void k_gather_window(const uint4* __restrict__ tab, size_t n_pts, size_t win_pts, uint32_t* __restrict__ out, uint32_t lanes) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    uint64_t h = (uint64_t)t * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const size_t base = (size_t)((double)t / lanes * (double)(n_pts - win_pts));
    const uint4* p = tab + (base + h % win_pts) * 4;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    out[t] = acc;
}
__global__ void k_stream(const uint4* __restrict__ tab, uint32_t* __restrict__ out, size_t n_vec) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_vec) return;
    uint4 v = tab[t];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = 1;
}
int main() {
    const size_t bytes = (size_t)4 << 30;                 // 4 GiB table
    uint4* tab; uint32_t* out;
    const uint32_t lanes = 1u << 22;
    hipMalloc(&tab, bytes); hipMalloc(&out, lanes * 4);
    hipMemset(tab, 0x5a, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); k_gather<4><<<lanes / 256, 256>>>(tab, bytes / 64, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("k_gather<4>  %u lanes x 64 B  = %.1f MB  %.3f ms  %.1f GB/s\n", lanes, lanes * 64.0 / 1e6, ms, lanes * 64.0 / ms / 1e6);
        hipEventRecord(e0); k_gather<8><<<lanes / 256, 256>>>(tab, bytes / 128, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("k_gather<8>  %u lanes x 128 B = %.1f MB  %.3f ms  %.1f GB/s\n", lanes, lanes * 128.0 / 1e6, ms, lanes * 128.0 / ms / 1e6);
        hipEventRecord(e0); k_gather_pair<<<lanes / 256, 256>>>(tab, bytes / 128, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("k_gather_pair %u lanes, %u entries x 128 B = %.1f MB  %.3f ms  %.1f GB/s\n", lanes, lanes / 2, lanes * 64.0 / 1e6, ms, lanes * 64.0 / ms / 1e6);
        hipEventRecord(e0); k_gather_pair192<<<lanes / 256, 256>>>(tab, bytes / 192, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("k_gather_pair192 %u lanes, %u entries x 192 B = %.1f MB  %.3f ms  %.1f GB/s\n", lanes, lanes / 2, lanes * 96.0 / 1e6, ms, lanes * 96.0 / ms / 1e6);
        // r06: the 14-limb curve's table entries (BLS12-381: 96 B per G1 point, 192 B per G2 point; entries are 96 / 192-byte aligned only)
        hipEventRecord(e0); k_gather<6><<<lanes / 256, 256>>>(tab, bytes / 96, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("k_gather<6>  %u lanes x 96 B  = %.1f MB  %.3f ms  %.1f GB/s\n", lanes, lanes * 96.0 / 1e6, ms, lanes * 96.0 / ms / 1e6);
        hipEventRecord(e0); k_gather<12><<<lanes / 256, 256>>>(tab, bytes / 192, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("k_gather<12> %u lanes x 192 B = %.1f MB  %.3f ms  %.1f GB/s\n", lanes, lanes * 192.0 / 1e6, ms, lanes * 192.0 / ms / 1e6);
        for (size_t win_mb : {16, 64, 256}) {
            const size_t n13 = ((size_t)832 << 20) / 64;      // a 2^20-point G1 window table: 13 rows x 64 MB
            hipEventRecord(e0); k_gather_window<<<lanes / 256, 256>>>(tab, n13, (win_mb << 20) / 64, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); printf("k_gather_window %zu MB window over 832 MB: %.3f ms  %.1f GB/s\n", win_mb, ms, lanes * 64.0 / ms / 1e6);
        }
        {
            const size_t n13 = ((size_t)832 << 20) / 64;
            hipEventRecord(e0); k_gather<4><<<lanes / 256, 256>>>(tab, n13, out, lanes); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); printf("k_gather<4> random over 832 MB: %.3f ms  %.1f GB/s\n", ms, lanes * 64.0 / ms / 1e6);
        }
        const size_t nv = ((size_t)1 << 30) / 16;
        hipEventRecord(e0); k_stream<<<(unsigned)(nv / 256), 256>>>(tab, out, nv); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("k_stream     1 GiB coalesced   = %.1f MB  %.3f ms  %.1f GB/s\n", 1073.7, ms, 1073.7 / ms);
    }
    return 0;
}
