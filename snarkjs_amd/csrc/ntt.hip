// snarkjs_amd/csrc/ntt.hip — host driver for the Fr NTT and the element-wise Fr batch kernels (both curves).
#include <string.h>
#include <algorithm>
#include "host_field.hpp"
#include "ntt.cuh"
#include "ntt29.cuh"
#include "zkmi_common.hpp"

namespace zkmi {

typedef host::HField<4> HFr;
typedef host::HFp<4> HE;

struct FrRoots {
    HFr F;
    int s = 0;
    HE w[33], wi[33], shift;
};
// Fr.w[] as ffjavascript defines it (build/snarkjs.min.js:1@185893): nqr = smallest quadratic
// non-residue >= 2, s = 2-adicity of r-1, w[s] = nqr^((r-1)/2^s), w[i] = w[i+1]^2.
template <class C> static const FrRoots& fr_roots() {
    static FrRoots R;
    static bool init = false;
    if (init) return R;
    R.F = HFr::from_cfg<C>();
    const HFr& F = R.F;
    uint64_t rm1[4], e[4], half[4];
    memcpy(rm1, F.p, 32); rm1[0] -= 1;
    memcpy(e, rm1, 32);
    int s = 0;
    while (!(e[0] & 1)) { for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0); s++; }
    for (int i = 0; i < 4; i++) half[i] = (rm1[i] >> 1) | (i < 3 ? rm1[i + 1] << 63 : 0);
    HE negone = F.neg(F.One()), nqr;
    for (uint64_t x = 2;; x++) { nqr = F.from_u64(x); if (F.pow(nqr, half, 4) == negone) break; }
    R.s = s;
    R.shift = F.sqr(nqr);
    R.w[s] = F.pow(nqr, e, 4);
    for (int i = s - 1; i >= 0; i--) R.w[i] = F.sqr(R.w[i + 1]);
    for (int i = 0; i <= s; i++) R.wi[i] = F.inv(R.w[i]);
    init = true;
    return R;
}

static int upload_table(const std::vector<HE>& t, uint32_t** d) {
    ZK_HIP(hipMalloc((void**)d, std::max<size_t>(t.size(), 1) * 32));
    if (!t.empty()) ZK_HIP(hipMemcpy(*d, t.data(), t.size() * 32, hipMemcpyHostToDevice));
    return ZKMI_OK;
}

template <class C> static int get_plan(int curve, unsigned L, int inverse, NttPlan** out) {
    Ctx& cx = ctx();
    auto key = std::make_tuple(curve, L, inverse);
    auto it = cx.plans.find(key);
    if (it != cx.plans.end()) { *out = &it->second; return ZKMI_OK; }
    const FrRoots& R = fr_roots<C>();
    const HFr& F = R.F;
    NttPlan P;
    P.log_n = L;
    if (L <= (unsigned)NTT_TILE_LOG) { P.n_pass = 1; P.l[0] = L; }
    else {
        static const unsigned maxbits = getenv("ZKMI_NTT_DIGIT") ? (unsigned)atoi(getenv("ZKMI_NTT_DIGIT")) : 8u;
        P.n_pass = (int)((L + maxbits - 1) / maxbits);
        unsigned rem = L;
        for (int i = 0; i < P.n_pass; i++) { unsigned li = (rem + (P.n_pass - i) - 1) / (P.n_pass - i); P.l[i] = li; rem -= li; }
    }
    const HE root = inverse ? R.wi[L] : R.w[L];
    P.log_lb = (L + 1) / 2;
    const size_t nlo = (size_t)1 << P.log_lb, nhi = (size_t)1 << (L - P.log_lb);
    std::vector<HE> tlo(nlo), thi(nhi), thil(nhi);
    tlo[0] = F.One();
    for (size_t i = 1; i < nlo; i++) tlo[i] = F.mul(tlo[i - 1], root);
    HE step = F.mul(tlo[nlo - 1], root);           // root^(2^lb)
    thi[0] = F.One();
    for (size_t i = 1; i < nhi; i++) thi[i] = F.mul(thi[i - 1], step);
    HE ninv = F.inv(F.from_u64((uint64_t)1 << L));
    for (size_t i = 0; i < nhi; i++) thil[i] = inverse ? F.mul(thi[i], ninv) : thi[i];
    ZK_TRY(upload_table(tlo, &P.T_lo)); ZK_TRY(upload_table(thi, &P.T_hi)); ZK_TRY(upload_table(thil, &P.T_hi_last));
    std::vector<HE> one(1, ninv);
    ZK_TRY(upload_table(one, &P.n_inv));
    // R'-form copies for ntt29.cuh: t 2^256 (the host's Montgomery form) times 2^5
    auto to29 = [&](std::vector<HE> v) { for (auto& x : v) for (int k = 0; k < 5; k++) x = F.dbl(x); return v; };
    ZK_TRY(upload_table(to29(tlo), &P.T_lo29)); ZK_TRY(upload_table(to29(thi), &P.T_hi29)); ZK_TRY(upload_table(to29(thil), &P.T_hi_last29));
    ZK_TRY(upload_table(to29(one), &P.n_inv29));
    for (int i = 0; i < P.n_pass; i++) {
        const unsigned li = P.l[i];
        const HE* wt = inverse ? R.wi : R.w;
        std::vector<HE> lt((size_t)1 << (li ? li - 1 : 0));
        lt[0] = F.One();
        for (size_t k = 1; k < lt.size(); k++) lt[k] = F.mul(lt[k - 1], wt[li]);      // w_{Ni} = Fr.w[li]
        ZK_TRY(upload_table(lt, &P.LT[i]));
        // ntt29.cuh: U[m] = w_{Ni}^(bit reversal of m over li - 1 bits)
        std::vector<HE> u(lt.size());
        for (size_t mI = 0; mI < lt.size(); mI++) { size_t rv = 0; for (unsigned b = 0; b + 1 < li; b++) if ((mI >> b) & 1) rv |= (size_t)1 << (li - 2 - b); u[mI] = lt[rv]; }
        ZK_TRY(upload_table(to29(u), &P.LT29[i]));
    }
    cx.plans[key] = P;
    *out = &cx.plans[key];
    return ZKMI_OK;
}

// batch > 1: `batch` transforms of the same size in ONE launch per pass (gridDim.y): member k reads d_in + k*in_stride elements and writes
// d_out + k*out_stride elements (Groth16's A, B, C chains: three times the blocks per launch, a third of the launches and launch tails)
template <class C> static int ntt_run(int curve, const void* d_in, void* d_out, unsigned L, int inverse, const uint8_t* first, const uint8_t* inc, unsigned batch = 1, size_t in_stride = 0,
                                      size_t out_stride = 0, size_t in_len = 0) {
    Ctx& cx = ctx();
    const FrRoots& R = fr_roots<C>();
    if ((int)L > R.s) return fail(ZKMI_ERR_UNSUPPORTED, "fft: log2(n) exceeds the 2-adicity of Fr (the reference's n = 2^(s+1) coset case is not supported)");
    hipStream_t st = cx.stream;
    const size_t n = (size_t)1 << L;
    if ((first == nullptr) != (inc == nullptr)) return fail(ZKMI_ERR_INVALID, "fft: prescale needs both first and inc");
    if (L == 0 && batch > 1) return fail(ZKMI_ERR_UNSUPPORTED, "fft: batched launches need n > 1");
    if (L == 0) {
        if (first) {
            HE x, f; std::vector<uint8_t> tmp(32);
            ZK_HIP(hipMemcpyAsync(tmp.data(), d_in, 32, hipMemcpyDeviceToHost, st)); ZK_HIP(hipStreamSynchronize(st));
            memcpy(x.v, tmp.data(), 32); memcpy(f.v, first, 32); x = R.F.mul(x, f);
            ZK_HIP(hipMemcpyAsync(d_out, x.v, 32, hipMemcpyHostToDevice, st)); ZK_HIP(hipStreamSynchronize(st));
        } else if (d_in != d_out) ZK_HIP(hipMemcpyAsync(d_out, d_in, 32, hipMemcpyDeviceToDevice, st));
        return ZKMI_OK;
    }
    NttPlan* P;
    ZK_TRY((get_plan<C>(curve, L, inverse, &P)));
    const int p = P->n_pass;
    // pre-scale row tables: rowinc_i[j] = (i == 0 ? first : 1) * inc^(j*S_i)
    // ZKMI_NTT29=1: the passes on 9 x 29-bit limbs (ntt29.cuh) instead of the saturated 32-bit ones (ntt.cuh). Built and measured in r03
    // (profiles/NOTES.md): bit-identical, and NOT faster — 2^20 0.147 vs 0.149 ms, in-proof chain 1.07-1.25 vs 1.07-1.18 ms on the same box,
    // 2^24 chain 22.2 vs 20.9 ms: a pass is bound by instruction issue INCLUDING its LDS traffic, and nine 4-byte limb planes cost 45 LDS
    // instructions per butterfly where two 16-byte planes cost 10, which eats what the cheaper product (207 vs ~290 instructions) saves.
    // The 32-bit passes stay the default; the 29-bit ones are kept behind this switch with their own parity test.
    // r04 (three vector planes per tile, column-statement products): the 29-bit passes are faster up to 2^22 (2^16 0.0275 vs 0.033 ms, 2^20 0.151 vs 0.165,
    // 2^22 0.571 vs 0.618 forward) and slower from 2^24 (their 48-byte records between passes are 1.5 x the traffic: 2.66 vs 2.59 ms), profiles/r04_ntt29_ab.txt.
    // ZKMI_NTT29=1 / 0 forces one form; unset: the 29-bit passes up to ZKMI_NTT29_MAX_LOG (default 22).
    static const int env29 = getenv("ZKMI_NTT29") ? atoi(getenv("ZKMI_NTT29")) : -1;
    static const unsigned max29 = getenv("ZKMI_NTT29_MAX_LOG") ? (unsigned)atoi(getenv("ZKMI_NTT29_MAX_LOG")) : 22u;
    const bool use29 = env29 >= 0 ? env29 == 1 : L <= max29;
    uint32_t* d_rowinc = nullptr;
    size_t rowoff[4] = {0, 0, 0, 0};
    if (first) {
        size_t tot = 0;
        for (int i = 0; i < p; i++) { rowoff[i] = tot; tot += (size_t)1 << P->l[i]; }
        std::string ck((const char*)first, 32);
        ck.append((const char*)inc, 32);
        ck.push_back((char)curve); ck.push_back((char)L); ck.push_back(use29 ? '9' : '2');
        DevBuf& cb = cx.ntt_prescale[ck];
        if (!cb.p) {                                  // tables are O(sum 2^l_i) elements: built once per (size, first, inc)
            if (cx.ntt_prescale.size() > 64) {         // bound the cache: drop everything but the new entry
                ZK_HIP(hipStreamSynchronize(st));
                for (auto it = cx.ntt_prescale.begin(); it != cx.ntt_prescale.end();) {
                    if (it->first == ck) { ++it; continue; }
                    if (it->second.p) (void)hipFree(it->second.p);
                    it = cx.ntt_prescale.erase(it);
                }
            }
            const HFr& F = R.F;
            HE f, g; memcpy(f.v, first, 32); memcpy(g.v, inc, 32);
            std::vector<HE> tab(tot);
            unsigned logS = L;
            for (int i = 0; i < p; i++) {
                logS -= P->l[i];
                HE b = g;
                for (unsigned k = 0; k < logS; k++) b = F.sqr(b);          // inc^(S_i)
                HE cur = (i == 0) ? f : F.One();
                for (size_t j = 0; j < ((size_t)1 << P->l[i]); j++) { tab[rowoff[i] + j] = cur; cur = F.mul(cur, b); }
            }
            if (use29) for (auto& x : tab) for (int k = 0; k < 5; k++) x = F.dbl(x);       // R'-form factors for ntt29.cuh
            DevBuf& nb = cx.ntt_prescale[ck];
            ZK_HIP(hipMalloc(&nb.p, tot * 32));
            nb.cap = tot * 32;
            ZK_HIP(hipMemcpyAsync(nb.p, tab.data(), tot * 32, hipMemcpyHostToDevice, st));
            ZK_HIP(hipStreamSynchronize(st));     // `tab` is a stack-owned staging buffer
            d_rowinc = (uint32_t*)nb.p;
        } else d_rowinc = (uint32_t*)cb.p;
    }
    // work buffer for the in-place middle passes (the caller's input is never modified); ntt29: 48-byte lazy records
    uint32_t* work = nullptr;
    const size_t rec = use29 ? NTT29_REC : 8;                       // words per work-array element
    if (p > 1) ZK_TRY(ws_get(use29 ? "ntt.work29" : "ntt.work", n * rec * 4 * batch, (void**)&work));
    static bool attr = false;
    if (!attr) {
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt_pass_strided<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt_pass_last<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt29_pass_strided<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt29_pass_strided<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt29_pass_last<C, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ZK_HIP(hipFuncSetAttribute((const void*)k_ntt29_pass_last<C, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr = true;
    }
    ZK_HIP(hipEventRecord(cx.ev0, st));
    NttPassArgs a;
    a.log_n = L; a.n_pass = (uint32_t)p; a.log_lb = P->log_lb; a.T_lo = use29 ? P->T_lo29 : P->T_lo;
    a.in_len = in_len >= n ? 0 : in_len;
    for (int i = 0; i < 4; i++) a.l[i] = P->l[i];
    unsigned logS = L;
    for (int i = 0; i < p - 1; i++) {
        logS -= P->l[i];
        a.pass = (uint32_t)i; a.T_hi = use29 ? P->T_hi29 : P->T_hi; a.LT = use29 ? P->LT29[i] : P->LT[i]; a.scale = nullptr;
        a.rowinc = d_rowinc ? d_rowinc + rowoff[i] * 8 : nullptr;
        a.log_ch = std::min<unsigned>(NTT_TILE_LOG - P->l[i], logS);
        const size_t N = (size_t)1 << P->l[i], E = N << a.log_ch;
        const size_t tile29 = ZKMI_NTT_VARIANT >= 1 ? ((N + 1) << a.log_ch) : E;                      // ntt29.cuh: the variants keep the strided tiles transposed
        const size_t lds = use29 ? (size_t)(Tile29::words((uint32_t)tile29) + Tile29::words((uint32_t)(N / 2)) + Tile29::words((uint32_t)N)) * 4 : (2 * E + N + 2 * N) * 16;
        const unsigned tiles = (unsigned)(n >> (P->l[i] + a.log_ch));
        const uint32_t* src = (i == 0) ? (const uint32_t*)d_in : work;
        a.in_bs = (i == 0) ? (uint64_t)in_stride * 8 : (uint64_t)n * rec; a.out_bs = (uint64_t)n * rec;
        if (!use29) hipLaunchKernelGGL((k_ntt_pass_strided<C>), dim3(tiles, batch), dim3(NTT_THREADS), lds, st, src, work, a);
        else if (i == 0) hipLaunchKernelGGL((k_ntt29_pass_strided<C, false>), dim3(tiles, batch), dim3(NTT29_THREADS), lds, st, src, work, a);
        else hipLaunchKernelGGL((k_ntt29_pass_strided<C, true>), dim3(tiles, batch), dim3(NTT29_THREADS), lds, st, src, work, a);
    }
    {
        const int i = p - 1;
        a.pass = (uint32_t)i; a.T_hi = use29 ? P->T_hi_last29 : P->T_hi_last; a.LT = use29 ? P->LT29[i] : P->LT[i];
        a.rowinc = d_rowinc ? d_rowinc + rowoff[i] * 8 : nullptr;
        a.scale = (p == 1 && inverse) ? (use29 ? P->n_inv29 : P->n_inv) : nullptr;
        a.log_ch = (p == 1) ? 0 : std::min<unsigned>(NTT_TILE_LOG - P->l[i], P->l[0]);
        const size_t N = (size_t)1 << P->l[i];
        const size_t lds = use29 ? (size_t)(Tile29::words((uint32_t)((N + 1) << a.log_ch)) + Tile29::words((uint32_t)std::max<size_t>(N / 2, 1))) * 4 : (2 * ((N + 1) << a.log_ch) + N) * 16;
        const unsigned tiles = (unsigned)(n >> (P->l[i] + a.log_ch));
        const uint32_t* src = (p == 1) ? (const uint32_t*)d_in : work;
        uint32_t* dst = (uint32_t*)d_out;
        if (p == 1 && d_in == d_out) { /* single tile: loads complete before stores */ }
        a.in_bs = (p == 1) ? (uint64_t)in_stride * 8 : (uint64_t)n * rec; a.out_bs = (uint64_t)out_stride * 8;
        if (!use29) hipLaunchKernelGGL((k_ntt_pass_last<C>), dim3(tiles, batch), dim3(NTT_THREADS), lds, st, src, dst, a);
        else if (p == 1) hipLaunchKernelGGL((k_ntt29_pass_last<C, false>), dim3(tiles, batch), dim3(NTT29_THREADS), lds, st, src, dst, a);
        else hipLaunchKernelGGL((k_ntt29_pass_last<C, true>), dim3(tiles, batch), dim3(NTT29_THREADS), lds, st, src, dst, a);
    }
    ZK_HIP(hipEventRecord(cx.ev1, st));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

// split power tables of root = Fr.w[L] (or its inverse): root^e = T_lo[e & (2^lb - 1)] * T_hi[e >> lb]; n_inv = (2^L)^-1 (Montgomery). Shared
// with the group FFT (gfft.hip), whose twiddles are the same roots.
int ntt_power_tables(int curve, unsigned L, int inverse, const uint32_t** T_lo, const uint32_t** T_hi, uint32_t* log_lb, const uint32_t** n_inv) {
    NttPlan* P = nullptr;
    if (curve == ZKMI_CURVE_BN128) { if ((int)L > fr_roots<Bn254Fr>().s) return fail(ZKMI_ERR_UNSUPPORTED, "fft: log2(n) exceeds the 2-adicity of Fr"); ZK_TRY((get_plan<Bn254Fr>(curve, L, inverse, &P))); }
    else if (curve == ZKMI_CURVE_BLS12381) { if ((int)L > fr_roots<Bls12381Fr>().s) return fail(ZKMI_ERR_UNSUPPORTED, "fft: log2(n) exceeds the 2-adicity of Fr"); ZK_TRY((get_plan<Bls12381Fr>(curve, L, inverse, &P))); }
    else return fail(ZKMI_ERR_INVALID, "unknown curve");
    *T_lo = P->T_lo; *T_hi = P->T_hi; *log_lb = P->log_lb; *n_inv = P->n_inv;
    return ZKMI_OK;
}

int ntt_dev_dispatch(int curve, const void* d_in, void* d_out, unsigned log_n, int inverse, const uint8_t* first, const uint8_t* inc) {
    if (curve == ZKMI_CURVE_BN128) return ntt_run<Bn254Fr>(curve, d_in, d_out, log_n, inverse, first, inc);
    if (curve == ZKMI_CURVE_BLS12381) return ntt_run<Bls12381Fr>(curve, d_in, d_out, log_n, inverse, first, inc);
    return fail(ZKMI_ERR_INVALID, "unknown curve");
}
// the input holds in_len < 2^log_n elements, the rest is read as zero (in place allowed when the buffer itself is 2^log_n long: d_in == d_out)
int ntt_dev_padded_dispatch(int curve, const void* d_in, size_t in_len, void* d_out, unsigned log_n, int inverse) {
    if (in_len == 0 || (log_n < 63 && in_len > ((size_t)1 << log_n))) return fail(ZKMI_ERR_INVALID, "fft: the input length must be in [1, 2^log_n]");
    if (log_n == 0) return ntt_dev_dispatch(curve, d_in, d_out, log_n, inverse, nullptr, nullptr);
    if (curve == ZKMI_CURVE_BN128) return ntt_run<Bn254Fr>(curve, d_in, d_out, log_n, inverse, nullptr, nullptr, 1, 0, 0, in_len);
    if (curve == ZKMI_CURVE_BLS12381) return ntt_run<Bls12381Fr>(curve, d_in, d_out, log_n, inverse, nullptr, nullptr, 1, 0, 0, in_len);
    return fail(ZKMI_ERR_INVALID, "unknown curve");
}
int ntt_dev_batch_dispatch(int curve, const void* d_in, size_t in_stride, void* d_out, size_t out_stride, unsigned batch, unsigned log_n, int inverse, const uint8_t* first,
                           const uint8_t* inc) {
    if (batch < 1 || batch > 16) return fail(ZKMI_ERR_INVALID, "fft: batch must be 1..16");
    if (curve == ZKMI_CURVE_BN128) return ntt_run<Bn254Fr>(curve, d_in, d_out, log_n, inverse, first, inc, batch, in_stride, out_stride);
    if (curve == ZKMI_CURVE_BLS12381) return ntt_run<Bls12381Fr>(curve, d_in, d_out, log_n, inverse, first, inc, batch, in_stride, out_stride);
    return fail(ZKMI_ERR_INVALID, "unknown curve");
}

template <class C> static int apply_key_run(const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc) {
    Ctx& cx = ctx();
    if (!n) return ZKMI_OK;
    const HFr F = HFr::from_cfg<C>();
    constexpr int PER = 16, T = 256;
    HE g; memcpy(g.v, inc, 32);
    HE step = F.pow_u64(g, T);
    uint32_t* d_k;
    ZK_TRY(ws_get("fr.applykey", 96, (void**)&d_k));
    uint8_t h[96]; memcpy(h, first, 32); memcpy(h + 32, inc, 32); memcpy(h + 64, step.v, 32);
    ZK_HIP(hipMemcpyAsync(d_k, h, 96, hipMemcpyHostToDevice, cx.stream));
    ZK_HIP(hipStreamSynchronize(cx.stream));
    ZK_HIP(hipEventRecord(cx.ev0, cx.stream));
    hipLaunchKernelGGL((k_apply_key<C, PER>), dim3((unsigned)((n + T * PER - 1) / (T * PER))), dim3(T), 0, cx.stream, (const uint32_t*)d_in, (uint32_t*)d_out, n, d_k, d_k + 8, d_k + 16);
    ZK_HIP(hipEventRecord(cx.ev1, cx.stream));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
int fr_root(int curve, unsigned i, uint8_t* out32) {
    const FrRoots& R = (curve == ZKMI_CURVE_BN128) ? fr_roots<Bn254Fr>() : fr_roots<Bls12381Fr>();
    if ((int)i > R.s) return fail(ZKMI_ERR_INVALID, "root index exceeds the 2-adicity of Fr");
    memcpy(out32, R.w[i].v, 32);
    return ZKMI_OK;
}
int fr_coset_inc(int curve, unsigned power, uint8_t* out32) {
    const FrRoots& R = (curve == ZKMI_CURVE_BN128) ? fr_roots<Bn254Fr>() : fr_roots<Bls12381Fr>();
    if ((int)power > R.s) return fail(ZKMI_ERR_UNSUPPORTED, "domain exceeds the 2-adicity of Fr");
    const HE& v = ((int)power == R.s) ? R.shift : R.w[power + 1];
    memcpy(out32, v.v, 32);
    return ZKMI_OK;
}
int apply_key_dev_dispatch(int curve, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc) {
    if (curve == ZKMI_CURVE_BN128) return apply_key_run<Bn254Fr>(d_in, d_out, n, first, inc);
    if (curve == ZKMI_CURVE_BLS12381) return apply_key_run<Bls12381Fr>(d_in, d_out, n, first, inc);
    return fail(ZKMI_ERR_INVALID, "unknown curve");
}

template <class C> static int fr_batch_run(int op, const void* d_in, void* d_out, size_t n) {
    Ctx& cx = ctx();
    if (!n) return ZKMI_OK;
    ZK_HIP(hipEventRecord(cx.ev0, cx.stream));
    if (op == ZKMI_BATCH_TO_MONTGOMERY || op == ZKMI_BATCH_FROM_MONTGOMERY) {
        hipLaunchKernelGGL((k_fr_convert<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, cx.stream, (const uint32_t*)d_in, (uint32_t*)d_out, n, op == ZKMI_BATCH_TO_MONTGOMERY ? 0 : 1);
    } else if (op == ZKMI_BATCH_INVERSE) {
        constexpr int CHUNK = 32;
        const void* src = d_in;
        if (d_in == d_out) {       // the kernel keeps prefix products in `out` and re-reads `in`
            void* tmp; ZK_TRY(ws_get("fr.inv_tmp", n * 32, &tmp));
            ZK_HIP(hipMemcpyAsync(tmp, d_in, n * 32, hipMemcpyDeviceToDevice, cx.stream));
            src = tmp;
        }
        const size_t lanes = (n + CHUNK - 1) / CHUNK;
        hipLaunchKernelGGL((k_batch_inverse<C, CHUNK>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, cx.stream, (const uint32_t*)src, (uint32_t*)d_out, n);
    } else return fail(ZKMI_ERR_INVALID, "unknown batch op");
    ZK_HIP(hipEventRecord(cx.ev1, cx.stream));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}
int fr_batch_dev_dispatch(int curve, int op, const void* d_in, void* d_out, size_t n) {
    if (curve == ZKMI_CURVE_BN128) return fr_batch_run<Bn254Fr>(op, d_in, d_out, n);
    if (curve == ZKMI_CURVE_BLS12381) return fr_batch_run<Bls12381Fr>(op, d_in, d_out, n);
    return fail(ZKMI_ERR_INVALID, "unknown curve");
}
int join_abc_dev_dispatch(int curve, const void* a, const void* b, const void* c, void* out, size_t n) {
    Ctx& cx = ctx();
    if (!n) return ZKMI_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    ZK_HIP(hipEventRecord(cx.ev0, cx.stream));
    if (curve == ZKMI_CURVE_BN128) hipLaunchKernelGGL((k_join_abc<Bn254Fr>), dim3(blocks), dim3(256), 0, cx.stream, (const uint32_t*)a, (const uint32_t*)b, (const uint32_t*)c, (uint32_t*)out, n);
    else if (curve == ZKMI_CURVE_BLS12381) hipLaunchKernelGGL((k_join_abc<Bls12381Fr>), dim3(blocks), dim3(256), 0, cx.stream, (const uint32_t*)a, (const uint32_t*)b, (const uint32_t*)c, (uint32_t*)out, n);
    else return fail(ZKMI_ERR_INVALID, "unknown curve");
    ZK_HIP(hipEventRecord(cx.ev1, cx.stream));
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

}  // namespace zkmi
