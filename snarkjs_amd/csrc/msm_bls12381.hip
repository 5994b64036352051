// snarkjs_amd/csrc/msm_bls12381.hip — BLS12-381 instantiations of the MSM pipeline (12-word Fq; 14 x 28-bit limbs in the accumulation kernels; G1 and G2).
#include "msm_host.hpp"

namespace zkmi {

static const uint64_t BLS_G1[2][6] = {
    {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL, 0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL},
    {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL, 0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL}};
static const uint64_t BLS_G2[4][6] = {
    {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL},
    {0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL},
    {0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL},
    {0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL}};

static void bls_generator(int group, uint8_t* out) {
    auto F = host::HField<6>::from_cfg<Bls12381Fq>();
    const int k = 2 * group;
    for (int i = 0; i < k; i++) {
        host::HFp<6> e;
        memcpy(e.v, group == 1 ? BLS_G1[i] : BLS_G2[i], 48);
        e = F.to_mont(e);
        memcpy(out + 48 * i, e.v, 48);
    }
}

int msm_bls12381(int group, const void* d_bases, const void* d_scalars, size_t n, size_t sb, uint8_t* out_jac) {
    if (group == 1) return msm_run<Fp<Bls12381Fq>>(d_bases, d_scalars, n, sb, out_jac);
    return msm_run<Fp2<Bls12381Fq>>(d_bases, d_scalars, n, sb, out_jac);
}
int msm_table_to_r29_bls12381(int group, void* d_table, size_t n_points) {
    Ctx& cx = ctx();
    const size_t elems = n_points * 2 * (size_t)group;         // base-field elements: 2 per G1 point, 4 per G2 point
    hipLaunchKernelGGL((k_table_to_r29<Bls12381Fq>), dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, cx.stream, (uint32_t*)d_table, elems);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
int msm_accumulate_bls12381(int group, const void* d_bases, const MsmPlan& pl, uint32_t skip, MsmJob& job, const uint32_t* d_infmask, MsmJob* into) {
    if (group == 1) return msm_accumulate<Fp<Bls12381Fq>>(d_bases, pl, skip, job, d_infmask, into);
    return msm_accumulate<Fp2<Bls12381Fq>>(d_bases, pl, skip, job, d_infmask, into);
}
int msm_infmask_bls12381(int group, const void* d_points, size_t n, uint32_t* d_mask) {
    if (group == 1) return msm_infmask<Fp<Bls12381Fq>>(d_points, n, d_mask);
    return msm_infmask<Fp2<Bls12381Fq>>(d_points, n, d_mask);
}
int msm_precompute_bls12381(int group, const void* d_bases, size_t n, int c, int Wd, void* d_table) {
    if (group == 1) return msm_precompute<Fp<Bls12381Fq>>(d_bases, n, c, Wd, d_table);
    return msm_precompute<Fp2<Bls12381Fq>>(d_bases, n, c, Wd, d_table);
}
int msm_table_bls12381(int group, const void* d_table, size_t stride, int c, const void* d_scalars, size_t k, size_t sb, uint8_t* out) {
    if (group == 1) return msm_run_table<Fp<Bls12381Fq>>(d_table, stride, c, d_scalars, k, sb, out);
    return msm_run_table<Fp2<Bls12381Fq>>(d_table, stride, c, d_scalars, k, sb, out);
}
int msm_table_multi_bls12381(int group, const void* d_table, size_t stride, int c, const void* const* d_scalars, const size_t* ks, int count, size_t sb, uint8_t* outs) {
    if (group == 1) return msm_run_table_multi<Fp<Bls12381Fq>>(d_table, stride, c, d_scalars, ks, count, sb, outs);
    return msm_run_table_multi<Fp2<Bls12381Fq>>(d_table, stride, c, d_scalars, ks, count, sb, outs);
}
int msm_table_multi_enqueue_bls12381(int group, const void* d_table, size_t stride, int c, const void* const* d_scalars, const size_t* ks, int count, size_t sb) {
    if (group == 1) return msm_run_table_multi_enqueue<Fp<Bls12381Fq>>(d_table, stride, c, d_scalars, ks, count, sb);
    return msm_run_table_multi_enqueue<Fp2<Bls12381Fq>>(d_table, stride, c, d_scalars, ks, count, sb);
}
int msm_table_multi_collect_bls12381(int group, int count, uint8_t* outs) {
    if (group == 1) return msm_run_table_multi_collect<Fp<Bls12381Fq>>(count, outs);
    return msm_run_table_multi_collect<Fp2<Bls12381Fq>>(count, outs);
}
int msm_reduce_bls12381(int group, MsmJob* const* jobs, int njobs, bool aux) {
    if (group == 1) return msm_reduce<Fp<Bls12381Fq>>(jobs, njobs, aux);
    return msm_reduce<Fp2<Bls12381Fq>>(jobs, njobs, aux);
}
int msm_fold_bls12381(int group, const MsmJob& job, uint8_t* out_jac) {
    if (group == 1) msm_fold<Fp<Bls12381Fq>>(job, out_jac); else msm_fold<Fp2<Bls12381Fq>>(job, out_jac);
    return ZKMI_OK;
}
int gen_bases_bls12381(int group, size_t n, uint64_t f, uint64_t g, void* d_out) {
    uint8_t gen[192];
    bls_generator(group, gen);
    if (group == 1) return gen_bases_run<Fp<Bls12381Fq>, Bls12381Fr>(gen, n, f, g, d_out);
    return gen_bases_run<Fp2<Bls12381Fq>, Bls12381Fr>(gen, n, f, g, d_out);
}
int gen_scalar_bases_bls12381(int group, const void* d_scalars, size_t n, void* d_out) {
    uint8_t gen[192];
    bls_generator(group, gen);
    if (group == 1) return gen_scalar_bases_run<Fp<Bls12381Fq>>(gen, d_scalars, n, d_out);
    return gen_scalar_bases_run<Fp2<Bls12381Fq>>(gen, d_scalars, n, d_out);
}
int point_add_bls12381(int group, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (group == 1) return point_add_host<Fp<Bls12381Fq>>(a, b, out);
    return point_add_host<Fp2<Bls12381Fq>>(a, b, out);
}
int to_affine_bls12381(int group, const uint8_t* jac, uint8_t* aff) {
    if (group == 1) return to_affine_host<Fp<Bls12381Fq>>(jac, aff);
    return to_affine_host<Fp2<Bls12381Fq>>(jac, aff);
}

}  // namespace zkmi
