// snarkjs_amd/csrc/msm_bn254.hip — BN254 (bn128) instantiations of the MSM pipeline: G1 over Fq, G2 over Fq2.
#include "msm_host.hpp"

namespace zkmi {

// generators, normal form (x, y) / ((x.c0,x.c1),(y.c0,y.c1)); converted to Montgomery form on first use
static const uint64_t BN254_G1[2][4] = {{1, 0, 0, 0}, {2, 0, 0, 0}};
static const uint64_t BN254_G2[4][4] = {
    {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL},
    {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL},
    {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL},
    {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL}};

static void bn254_generator(int group, uint8_t* out) {
    auto F = host::HField<4>::from_cfg<Bn254Fq>();
    const int k = 2 * group;
    for (int i = 0; i < k; i++) {
        host::HFp<4> e;
        memcpy(e.v, group == 1 ? BN254_G1[i] : BN254_G2[i], 32);
        e = F.to_mont(e);
        memcpy(out + 32 * i, e.v, 32);
    }
}

int msm_bn254(int group, const void* d_bases, const void* d_scalars, size_t n, size_t sb, uint8_t* out_jac) {
    if (group == 1) return msm_run<Fp<Bn254Fq>>(d_bases, d_scalars, n, sb, out_jac);
    return msm_run<Fp2<Bn254Fq>>(d_bases, d_scalars, n, sb, out_jac);
}
int msm_table_to_r29_bn254(int group, void* d_table, size_t n_points) {
    Ctx& cx = ctx();
    const size_t elems = n_points * 2 * (size_t)group;         // base-field elements: 2 per G1 point, 4 per G2 point
    hipLaunchKernelGGL((k_table_to_r29<Bn254Fq>), dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, cx.stream, (uint32_t*)d_table, elems);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
int msm_accumulate_bn254(int group, const void* d_bases, const MsmPlan& pl, uint32_t skip, MsmJob& job, const uint32_t* d_infmask, MsmJob* into) {
    if (group == 1) return msm_accumulate<Fp<Bn254Fq>>(d_bases, pl, skip, job, d_infmask, into);
    return msm_accumulate<Fp2<Bn254Fq>>(d_bases, pl, skip, job, d_infmask, into);
}
int msm_infmask_bn254(int group, const void* d_points, size_t n, uint32_t* d_mask) {
    if (group == 1) return msm_infmask<Fp<Bn254Fq>>(d_points, n, d_mask);
    return msm_infmask<Fp2<Bn254Fq>>(d_points, n, d_mask);
}
int msm_precompute_bn254(int group, const void* d_bases, size_t n, int c, int Wd, void* d_table) {
    if (group == 1) return msm_precompute<Fp<Bn254Fq>>(d_bases, n, c, Wd, d_table);
    return msm_precompute<Fp2<Bn254Fq>>(d_bases, n, c, Wd, d_table);
}
int msm_table_bn254(int group, const void* d_table, size_t stride, int c, const void* d_scalars, size_t k, size_t sb, uint8_t* out) {
    if (group == 1) return msm_run_table<Fp<Bn254Fq>>(d_table, stride, c, d_scalars, k, sb, out);
    return msm_run_table<Fp2<Bn254Fq>>(d_table, stride, c, d_scalars, k, sb, out);
}
int msm_table_multi_bn254(int group, const void* d_table, size_t stride, int c, const void* const* d_scalars, const size_t* ks, int count, size_t sb, uint8_t* outs) {
    if (group == 1) return msm_run_table_multi<Fp<Bn254Fq>>(d_table, stride, c, d_scalars, ks, count, sb, outs);
    return msm_run_table_multi<Fp2<Bn254Fq>>(d_table, stride, c, d_scalars, ks, count, sb, outs);
}
int msm_table_multi_enqueue_bn254(int group, const void* d_table, size_t stride, int c, const void* const* d_scalars, const size_t* ks, int count, size_t sb) {
    if (group == 1) return msm_run_table_multi_enqueue<Fp<Bn254Fq>>(d_table, stride, c, d_scalars, ks, count, sb);
    return msm_run_table_multi_enqueue<Fp2<Bn254Fq>>(d_table, stride, c, d_scalars, ks, count, sb);
}
int msm_table_multi_collect_bn254(int group, int count, uint8_t* outs) {
    if (group == 1) return msm_run_table_multi_collect<Fp<Bn254Fq>>(count, outs);
    return msm_run_table_multi_collect<Fp2<Bn254Fq>>(count, outs);
}
int msm_reduce_bn254(int group, MsmJob* const* jobs, int njobs, bool aux) {
    if (group == 1) return msm_reduce<Fp<Bn254Fq>>(jobs, njobs, aux);
    return msm_reduce<Fp2<Bn254Fq>>(jobs, njobs, aux);
}
int msm_fold_bn254(int group, const MsmJob& job, uint8_t* out_jac) {
    if (group == 1) msm_fold<Fp<Bn254Fq>>(job, out_jac); else msm_fold<Fp2<Bn254Fq>>(job, out_jac);
    return ZKMI_OK;
}
int gen_bases_bn254(int group, size_t n, uint64_t f, uint64_t g, void* d_out) {
    uint8_t gen[128];
    bn254_generator(group, gen);
    if (group == 1) return gen_bases_run<Fp<Bn254Fq>, Bn254Fr>(gen, n, f, g, d_out);
    return gen_bases_run<Fp2<Bn254Fq>, Bn254Fr>(gen, n, f, g, d_out);
}
int gen_scalar_bases_bn254(int group, const void* d_scalars, size_t n, void* d_out) {
    uint8_t gen[128];
    bn254_generator(group, gen);
    if (group == 1) return gen_scalar_bases_run<Fp<Bn254Fq>>(gen, d_scalars, n, d_out);
    return gen_scalar_bases_run<Fp2<Bn254Fq>>(gen, d_scalars, n, d_out);
}
int point_add_bn254(int group, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (group == 1) return point_add_host<Fp<Bn254Fq>>(a, b, out);
    return point_add_host<Fp2<Bn254Fq>>(a, b, out);
}
int to_affine_bn254(int group, const uint8_t* jac, uint8_t* aff) {
    if (group == 1) return to_affine_host<Fp<Bn254Fq>>(jac, aff);
    return to_affine_host<Fp2<Bn254Fq>>(jac, aff);
}

}  // namespace zkmi
