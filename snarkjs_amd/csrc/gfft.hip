// snarkjs_amd/csrc/gfft.hip — host driver + C-ABI of the group-element FFT and G.batchApplyKey (gfft.cuh).
#include <string.h>
#include "gfft.cuh"
#include "host_field.hpp"
#include "zkmi_common.hpp"

namespace zkmi {

template <class F, class FrC> static int gfft_run(int curve, const void* d_in, void* d_out, unsigned L, int inverse) {
    constexpr int FW = FieldWords<F>::value;
    Ctx& cx = ctx();
    hipStream_t st = cx.stream;
    const size_t n = (size_t)1 << L;
    if (L > 28) return fail(ZKMI_ERR_UNSUPPORTED, "group fft: at most 2^28 points");
    uint32_t* work;
    ZK_TRY(ws_get("gfft.work", n * 4 * FW * 4, (void**)&work));
    GfftTw tw{nullptr, nullptr, 0, L};
    const uint32_t* n_inv = nullptr;
    if (L) ZK_TRY(ntt_power_tables(curve, L, inverse, &tw.T_lo, &tw.T_hi, &tw.log_lb, &n_inv));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    ZK_HIP(hipEventRecord(cx.ev0, st));
    hipLaunchKernelGGL((k_gfft_load<F>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)d_in, work, (uint32_t)n, L);
    for (unsigned s = 1; s <= L; s++)
        hipLaunchKernelGGL((k_gfft_stage<F, FrC>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, work, (uint32_t)n, s, tw);
    hipLaunchKernelGGL((k_gfft_store<F, FrC>), dim3(blocks), dim3(256), 0, st, work, (uint32_t*)d_out, (uint32_t)n, (inverse && L) ? n_inv : nullptr);
    ZK_HIP(hipEventRecord(cx.ev1, st));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
template <class F, class FrC> static int g_apply_key_run(const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc) {
    Ctx& cx = ctx();
    if (!n) return ZKMI_OK;
    if (n >= (1ull << 32)) return fail(ZKMI_ERR_UNSUPPORTED, "group batchApplyKey: at most 2^32 - 1 points");
    const host::HField<4> Fr = host::HField<4>::from_cfg<FrC>();
    host::HFp<4> g;
    memcpy(g.v, inc, 32);
    const host::HFp<4> step = Fr.pow_u64(g, 256);
    uint32_t* d_k;
    ZK_TRY(ws_get("gfft.applykey", 96, (void**)&d_k));
    uint8_t h[96];
    memcpy(h, first, 32); memcpy(h + 32, inc, 32); memcpy(h + 64, step.v, 32);
    ZK_HIP(hipMemcpyAsync(d_k, h, 96, hipMemcpyHostToDevice, cx.stream));
    ZK_HIP(hipStreamSynchronize(cx.stream));                  // `h` is a stack buffer
    ZK_HIP(hipEventRecord(cx.ev0, cx.stream));
    hipLaunchKernelGGL((k_g_apply_key<F, FrC>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, (const uint32_t*)d_in, (uint32_t*)d_out, (uint32_t)n, d_k);
    ZK_HIP(hipEventRecord(cx.ev1, cx.stream));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
static int gfft_dispatch(int curve, int group, const void* d_in, void* d_out, unsigned L, int inverse) {
    if (curve == ZKMI_CURVE_BN128) return group == 1 ? gfft_run<Fp<Bn254Fq>, Bn254Fr>(curve, d_in, d_out, L, inverse) : gfft_run<Fp2<Bn254Fq>, Bn254Fr>(curve, d_in, d_out, L, inverse);
    return group == 1 ? gfft_run<Fp<Bls12381Fq>, Bls12381Fr>(curve, d_in, d_out, L, inverse) : gfft_run<Fp2<Bls12381Fq>, Bls12381Fr>(curve, d_in, d_out, L, inverse);
}
static int g_apply_key_dispatch(int curve, int group, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc) {
    if (curve == ZKMI_CURVE_BN128) return group == 1 ? g_apply_key_run<Fp<Bn254Fq>, Bn254Fr>(d_in, d_out, n, first, inc) : g_apply_key_run<Fp2<Bn254Fq>, Bn254Fr>(d_in, d_out, n, first, inc);
    return group == 1 ? g_apply_key_run<Fp<Bls12381Fq>, Bls12381Fr>(d_in, d_out, n, first, inc) : g_apply_key_run<Fp2<Bls12381Fq>, Bls12381Fr>(d_in, d_out, n, first, inc);
}
static size_t pages_bytes(const zkmi_pages& pg) { size_t t = 0; for (int i = 0; i < pg.n_pages; i++) t += pg.len[i]; return t; }

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zkmi_group_fft_dev(int curve, int group, const void* d_in, void* d_out, unsigned log_n, int inverse) {
    ZK_TRY(require_ctx());
    if ((curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || (group != 1 && group != 2)) return fail(ZKMI_ERR_INVALID, "group fft: unknown curve or group");
    if (!d_in || !d_out) return fail(ZKMI_ERR_INVALID, "group fft: null buffer");
    return gfft_dispatch(curve, group, d_in, d_out, log_n, inverse);
}
int zkmi_group_fft(int curve, int group, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, unsigned log_n, int inverse) {
    ZK_TRY(require_ctx());
    if ((curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || (group != 1 && group != 2)) return fail(ZKMI_ERR_INVALID, "group fft: unknown curve or group");
    if (log_n > 28) return fail(ZKMI_ERR_UNSUPPORTED, "group fft: at most 2^28 points");
    const size_t bytes = ((size_t)1 << log_n) * 2 * group * n8q_of(curve);
    if (pages_bytes(in) != bytes) return fail(ZKMI_ERR_INVALID, "fft must be multiple of 2");
    void *d_i, *d_o;
    ZK_TRY(ws_get("api.gfft_in", bytes, &d_i));
    ZK_TRY(ws_get("api.gfft_out", bytes, &d_o));
    ZK_TRY(upload_pages(in, bytes, d_i));
    ZK_TRY(gfft_dispatch(curve, group, d_i, d_o, log_n, inverse));
    return download_pages(d_o, bytes, out_ptr, out_len, n_out_pages);
}
int zkmi_group_batch_apply_key_dev(int curve, int group, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc) {
    ZK_TRY(require_ctx());
    if ((curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || (group != 1 && group != 2)) return fail(ZKMI_ERR_INVALID, "group batchApplyKey: unknown curve or group");
    if (!first || !inc || (n && (!d_in || !d_out))) return fail(ZKMI_ERR_INVALID, "group batchApplyKey: null argument");
    return g_apply_key_dispatch(curve, group, d_in, d_out, n, first, inc);
}
int zkmi_group_batch_apply_key(int curve, int group, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, size_t n, const uint8_t* first,
                               const uint8_t* inc) {
    ZK_TRY(require_ctx());
    if ((curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || (group != 1 && group != 2)) return fail(ZKMI_ERR_INVALID, "group batchApplyKey: unknown curve or group");
    if (!first || !inc) return fail(ZKMI_ERR_INVALID, "group batchApplyKey: null argument");
    const size_t bytes = n * 2 * group * n8q_of(curve);
    if (pages_bytes(in) < bytes) return fail(ZKMI_ERR_INVALID, "input buffer shorter than n elements");
    if (!n) return ZKMI_OK;
    void *d_i, *d_o;
    ZK_TRY(ws_get("api.gfft_in", bytes, &d_i));
    ZK_TRY(ws_get("api.gfft_out", bytes, &d_o));
    ZK_TRY(upload_pages(in, bytes, d_i));
    ZK_TRY(g_apply_key_dispatch(curve, group, d_i, d_o, n, first, inc));
    return download_pages(d_o, bytes, out_ptr, out_len, n_out_pages);
}

}  // extern "C"
