// snarkjs_amd/csrc/gconv.hip — host driver + C-ABI of the point-format conversions (gconv.cuh).
#include <string.h>
#include "gconv.cuh"
#include "host_field.hpp"
#include "zkmi_common.hpp"

namespace zkmi {

// curve constant b of y^2 = x^3 + b in Montgomery form: 3 (BN254 G1), 3/(9+u) (BN254 G2), 4 (BLS12-381 G1), 4(1+u) (BLS12-381 G2)
template <class C> static void curve_b_words(int curve, int group, uint32_t* out) {
    constexpr int L = C::N / 2;
    const host::HField<L> F = host::HField<L>::template from_cfg<C>();
    const host::HFp<L> k = F.from_u64(curve == ZKMI_CURVE_BN128 ? 3 : 4);
    memset(out, 0, (size_t)group * C::N * 4);
    if (group == 1) { memcpy(out, k.v, C::N * 4); return; }
    host::HField2<L> F2;
    F2.F = F;
    host::HFp2<L> b;
    if (curve == ZKMI_CURVE_BN128) b = F2.mul(host::HFp2<L>{k, F.zero()}, F2.inv(host::HFp2<L>{F.from_u64(9), F.One()}));
    else b = host::HFp2<L>{k, k};
    memcpy(out, b.c0.v, C::N * 4);
    memcpy(out + C::N, b.c1.v, C::N * 4);
}

template <class C> static int gconv_run(int curve, int group, int kind, const void* d_in, void* d_out, size_t n) {
    Ctx& cx = ctx();
    hipStream_t st = cx.stream;
    if (!n) return ZKMI_OK;
    if (n >= (1ull << 31)) return fail(ZKMI_ERR_UNSUPPORTED, "point conversion: at most 2^31 - 1 points");
    const unsigned blocks = (unsigned)((n + 255) / 256);
    ZK_HIP(hipEventRecord(cx.ev0, st));
    if (kind == ZKMI_CONV_LEM_TO_U || kind == ZKMI_CONV_U_TO_LEM) {
        const uint64_t ne = (uint64_t)n * 2 * group;
        hipLaunchKernelGGL((k_gconv_elems<C>), dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_in, (uint32_t*)d_out, ne, group, kind == ZKMI_CONV_LEM_TO_U ? 1 : 0);
    } else if (kind == ZKMI_CONV_LEM_TO_C) {
        if (group == 1) hipLaunchKernelGGL((k_gconv_compress<Fp<C>>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)d_in, (uint32_t*)d_out, (uint32_t)n);
        else hipLaunchKernelGGL((k_gconv_compress<Fp2<C>>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)d_in, (uint32_t*)d_out, (uint32_t)n);
    } else {
        uint32_t h[2 * C::N + 4] = {0};
        curve_b_words<C>(curve, group, h + 4);                     // word 0: the "not on the curve" flag, b from byte 16 (16-byte aligned loads)
        uint32_t* d_k;
        ZK_TRY(ws_get("gconv.consts", sizeof h, (void**)&d_k));
        ZK_HIP(hipMemcpyAsync(d_k, h, sizeof h, hipMemcpyHostToDevice, st));
        ZK_HIP(hipStreamSynchronize(st));                           // `h` is a stack buffer
        if (group == 1) hipLaunchKernelGGL((k_gconv_decompress<Fp<C>>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)d_in, (uint32_t*)d_out, (uint32_t)n, d_k + 4, d_k);
        else hipLaunchKernelGGL((k_gconv_decompress<Fp2<C>>), dim3(blocks), dim3(256), 0, st, (const uint32_t*)d_in, (uint32_t*)d_out, (uint32_t)n, d_k + 4, d_k);
        uint32_t bad = 0;
        ZK_HIP(hipMemcpyAsync(&bad, d_k, 4, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipEventRecord(cx.ev1, st));
        ZK_HIP(hipStreamSynchronize(st));
        if (bad) return fail(ZKMI_ERR_INVALID, "batchCtoLEM: a compressed point is not on the curve");
        return ZKMI_OK;
    }
    ZK_HIP(hipEventRecord(cx.ev1, st));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
static int gconv_check(int curve, int group, int kind) {
    if ((curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || (group != 1 && group != 2)) return fail(ZKMI_ERR_INVALID, "point conversion: unknown curve or group");
    if (kind < ZKMI_CONV_LEM_TO_U || kind > ZKMI_CONV_C_TO_LEM) return fail(ZKMI_ERR_INVALID, "point conversion: unknown kind");
    return ZKMI_OK;
}
static int gconv_dispatch(int curve, int group, int kind, const void* d_in, void* d_out, size_t n) {
    return curve == ZKMI_CURVE_BN128 ? gconv_run<Bn254Fq>(curve, group, kind, d_in, d_out, n) : gconv_run<Bls12381Fq>(curve, group, kind, d_in, d_out, n);
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zkmi_group_convert_dev(int curve, int group, int kind, const void* d_in, void* d_out, size_t n) {
    ZK_TRY(require_ctx());
    ZK_TRY(gconv_check(curve, group, kind));
    if (n && (!d_in || !d_out)) return fail(ZKMI_ERR_INVALID, "point conversion: null buffer");
    return gconv_dispatch(curve, group, kind, d_in, d_out, n);
}
int zkmi_group_convert(int curve, int group, int kind, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, size_t n) {
    ZK_TRY(require_ctx());
    ZK_TRY(gconv_check(curve, group, kind));
    const size_t full = 2 * (size_t)group * n8q_of(curve);
    const size_t in_bytes = n * (kind == ZKMI_CONV_C_TO_LEM ? full / 2 : full), out_bytes = n * (kind == ZKMI_CONV_LEM_TO_C ? full / 2 : full);
    size_t have = 0;
    for (int i = 0; i < in.n_pages; i++) have += in.len[i];
    if (have != in_bytes) return fail(ZKMI_ERR_INVALID, "point conversion: input length does not match n points");
    if (!n) return ZKMI_OK;
    void *d_i, *d_o;
    ZK_TRY(ws_get("api.gfft_in", in_bytes, &d_i));
    ZK_TRY(ws_get("api.gfft_out", out_bytes, &d_o));
    ZK_TRY(upload_pages(in, in_bytes, d_i));
    ZK_TRY(gconv_dispatch(curve, group, kind, d_i, d_o, n));
    return download_pages(d_o, out_bytes, out_ptr, out_len, n_out_pages);
}

}  // extern "C"
