// snarkjs_amd/csrc/plonk.hip — device kernels for the per-element loops of the PLONK prover.
//
// In the reference these are single-threaded JavaScript loops calling Fr.mul / Fr.add on 32-byte slices
// (src/plonk_prove.js, src/mul_z.js, src/polynomial/polynomial.js); here each is a data-parallel kernel over Fr:
//   computeWirePolynomials gather  (plonk_prove.js:267-283)   k_plonk_gather
//   computeZ                       (plonk_prove.js:361-455)   k_plonk_z_factors + multiplicative scan + batch inverse
//   computeT + MulZ.mul2/mul4      (plonk_prove.js:516-628, mul_z.js:49-148)   k_plonk_t29 (plonk29.cuh: 29-bit limbs; one lane per evaluation point), k_plonk_t (32-bit limbs)
//   Polynomial.add/sub/mulScalar   (polynomial.js:218-284)    k_poly_axpy / k_poly_scale
//   Polynomial.evaluate (Horner)   (polynomial.js:174-184)    k_poly_eval_partial + k_poly_sum   (parallel reduction)
//   Polynomial.divZh               (polynomial.js:592-615)    k_poly_div_zh          (stride-n recurrence, one lane per residue)
//   Polynomial.divByZerofier(1,b)  (polynomial.js:617-674)    power weighting + additive scan (the linear recurrence
//                                                              q_i = (q_{i-1} - c_i)/b  solved as a prefix sum)
// All buffers are device pointers to little-endian Montgomery Fr elements, the reference's own representation.
#include <string.h>
#include <algorithm>
#include <type_traits>
#include "host_field.hpp"
#include "ntt.cuh"
#include "plonk29.cuh"
#include "zkmi_common.hpp"

namespace zkmi {

typedef host::HField<4> HFr;
typedef host::HFp<4> HE;

// base^e = lo[e & (2^lb - 1)] * hi[e >> lb]   (struct PowTab: plonk29.cuh)
template <class C> ZK_DEV Fp<C> pow_tab(const PowTab& t, uint64_t e) {
    Fp<C> a = fp_load<C>(t.lo + (size_t)(e & ((1ull << t.lb) - 1)) * C::N);
    Fp<C> b = fp_load<C>(t.hi + (size_t)(e >> t.lb) * C::N);
    return fp_mul(a, b);
}
// host: table for exponents < 2^log_count, into the named scratch buffer
// The bases are roots of unity of the key's domain: the same few tables are asked for by every proof, so they are cached by name
// (the named scratch buffer keeps the device copy; the cache remembers which base / size it was built for).
struct PowTabKey { HE base; unsigned log_count; uint32_t* d; };
static std::map<std::string, PowTabKey>& pow_tab_cache() { static std::map<std::string, PowTabKey> m; return m; }
static int build_pow_tab(const HFr& F, const HE& base, unsigned log_count, const char* name, PowTab* out) {
    Ctx& cx = ctx();
    const unsigned lb = (log_count + 1) / 2, hb = log_count - lb;
    const size_t nlo = (size_t)1 << lb, nhi = (size_t)1 << hb;
    uint32_t* d;
    ZK_TRY(ws_get(name, (nlo + nhi) * 32, (void**)&d));
    const std::string key = std::string(cx.pipe ? "P1:" : "") + name;            // one device copy per pipeline slot (ws_get), one cache entry each
    auto it = pow_tab_cache().find(key);
    if (it == pow_tab_cache().end() || !(it->second.base == base) || it->second.log_count != log_count || it->second.d != d) {
        std::vector<HE> t(nlo + nhi);
        t[0] = F.One();
        for (size_t i = 1; i < nlo; i++) t[i] = F.mul(t[i - 1], base);
        const HE step = F.mul(t[nlo - 1], base);
        t[nlo] = F.One();
        for (size_t i = 1; i < nhi; i++) t[nlo + i] = F.mul(t[nlo + i - 1], step);
        ZK_HIP(hipMemcpyAsync(d, t.data(), t.size() * 32, hipMemcpyHostToDevice, cx.stream));
        ZK_HIP(hipStreamSynchronize(cx.stream));              // `t` is a stack-owned staging buffer
        pow_tab_cache()[key] = PowTabKey{base, log_count, d};
    }
    out->lo = d; out->hi = d + nlo * 8; out->lb = lb;
    return ZKMI_OK;
}
static inline unsigned clog2(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }

// A field element (or a few) as a KERNEL ARGUMENT: per-call constants — challenges, blinding factors, evaluation points — travel in the launch packet instead of
// through an upload of their own (r06: every upload was one more copy launch on the stream, 80 of a PLONK proof's 376).
struct alignas(16) FrK { uint32_t v[8]; };
static inline FrK frk(const uint8_t* p) { FrK k; memcpy(k.v, p, 32); return k; }
static inline FrK frk(const HE& e) { FrK k; memcpy(k.v, e.v, 32); return k; }
template <class C> ZK_DEV Fp<C> frk_load(const FrK& k) { return fp_load<C>(k.v); }
// power table of a point that changes with every call (xi, xi w, their inverses), built ON the device: entry t < nlo is base^t, entry nlo + t is (base^nlo)^t, each by
// square-and-multiply (at most 2 log2 products per lane, one launch, no host arithmetic and no wait — the host-built table of build_pow_tab costs 2^(L/2+1) host
// products, an upload and a stream synchronisation, which is right for the roots of unity it caches and wrong for a point used once)
constexpr int POWTAB_MAX = 8;
struct PowTabBuildArgs { FrK base[POWTAB_MAX]; uint32_t* out[POWTAB_MAX]; };
template <class C> __global__ void k_pow_tab_build(PowTabBuildArgs a, uint32_t lb, uint32_t nlo, uint32_t nhi) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;                     // several tables of one size per launch (blockIdx.y)
    if (t >= nlo + nhi) return;
    Fp<C> b = frk_load<C>(a.base[q]);
    uint32_t e = t;
    if (t >= nlo) { for (uint32_t k = 0; k < lb; k++) b = fp_mul(b, b); e = t - nlo; }
    Fp<C> r = fp_one<C>();
    while (e) { if (e & 1u) r = fp_mul(r, b); b = fp_mul(b, b); e >>= 1; }
    fp_store<C>(a.out[q] + (size_t)t * 8, r);
}
// `count` tables for exponents < 2^log_count, into the scratch buffers name.0, name.1, ...
template <class C> static int build_pow_tabs_dyn(const HE* bases, int count, unsigned log_count, const std::string& name, PowTab* out) {
    Ctx& cx = ctx();
    if (count < 1 || count > POWTAB_MAX) return fail(ZKMI_ERR_INVALID, "power tables: 1..8 per launch");
    const unsigned lb = (log_count + 1) / 2, hb = log_count - lb;
    const uint32_t nlo = 1u << lb, nhi = 1u << hb;
    PowTabBuildArgs a = {};
    for (int q = 0; q < count; q++) {
        const std::string nm = name + "." + std::to_string(q);
        uint32_t* d;
        ZK_TRY(ws_get(nm, ((size_t)nlo + nhi) * 32, (void**)&d));
        pow_tab_cache().erase(std::string(cx.pipe ? "P1:" : "") + nm);          // the named buffer no longer holds what build_pow_tab may have cached under this name
        a.base[q] = frk(bases[q]); a.out[q] = d;
        out[q].lo = d; out[q].hi = d + (size_t)nlo * 8; out[q].lb = lb;
    }
    hipLaunchKernelGGL((k_pow_tab_build<C>), dim3((nlo + nhi + 255) / 256, (unsigned)count), dim3(256), 0, cx.stream, a, lb, nlo, nhi);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

// ---- scans over Fr (multiplicative or additive), 3 launches: chunk totals, scan of totals, chunk scan with offset -----------
constexpr int SCAN_T = 256, SCAN_K = 8, SCAN_CHUNK = SCAN_T * SCAN_K;
template <class C, bool MUL> ZK_DEV Fp<C> scan_op(const Fp<C>& a, const Fp<C>& b) { return MUL ? fp_mul(a, b) : fp_add(a, b); }
template <class C, bool MUL> ZK_DEV Fp<C> scan_id() { return MUL ? fp_one<C>() : fp_zero<C>(); }

// block-wide inclusive scan of one value per thread (Hillis-Steele in LDS); returns the inclusive value of this thread
template <class C, bool MUL> ZK_DEV Fp<C> block_scan(Fp<C> v, uint32_t* lds) {
    const uint32_t t = threadIdx.x;
    fp_store<C>(lds + t * 8, v);
    __syncthreads();
    for (int d = 1; d < SCAN_T; d <<= 1) {
        Fp<C> o = t >= (uint32_t)d ? fp_load<C>(lds + (t - d) * 8) : scan_id<C, MUL>();
        __syncthreads();
        v = scan_op<C, MUL>(o, v);
        fp_store<C>(lds + t * 8, v);
        __syncthreads();
    }
    return v;
}
template <class C, bool MUL> __global__ void __launch_bounds__(SCAN_T)
k_scan_totals(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[SCAN_T * 8];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_K;
    Fp<C> acc = scan_id<C, MUL>();
    for (int k = 0; k < SCAN_K; k++) if (base + k < n) acc = scan_op<C, MUL>(acc, fp_load<C>(in + (base + k) * 8));
    Fp<C> inc = block_scan<C, MUL>(acc, lds);
    if (threadIdx.x == SCAN_T - 1) fp_store<C>(part + (size_t)blockIdx.x * 8, inc);
}
// exclusive scan of the chunk totals by ONE block
template <class C, bool MUL> __global__ void __launch_bounds__(SCAN_T)
k_scan_parts(uint32_t* __restrict__ part, uint32_t nparts) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[SCAN_T * 8];
    __shared__ __attribute__((aligned(16))) uint32_t carry_s[8];
    Fp<C> carry = scan_id<C, MUL>();
    for (uint32_t b0 = 0; b0 < nparts; b0 += SCAN_T) {
        const uint32_t i = b0 + threadIdx.x;
        Fp<C> v = i < nparts ? fp_load<C>(part + (size_t)i * 8) : scan_id<C, MUL>();
        Fp<C> inc = block_scan<C, MUL>(v, lds);
        // exclusive = carry op (inclusive of the previous thread)
        Fp<C> prev = threadIdx.x ? fp_load<C>(lds + (threadIdx.x - 1) * 8) : scan_id<C, MUL>();
        if (i < nparts) fp_store<C>(part + (size_t)i * 8, scan_op<C, MUL>(carry, prev));
        if (threadIdx.x == SCAN_T - 1) fp_store<C>(carry_s, scan_op<C, MUL>(carry, inc));
        __syncthreads();
        carry = fp_load<C>(carry_s);
        __syncthreads();
    }
}
// out[i] = inclusive scan up to i (in place allowed)
template <class C, bool MUL> __global__ void __launch_bounds__(SCAN_T)
k_scan_final(const uint32_t* __restrict__ in, size_t n, const uint32_t* __restrict__ part, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[SCAN_T * 8];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_K;
    Fp<C> v[SCAN_K];
    Fp<C> acc = scan_id<C, MUL>();
    for (int k = 0; k < SCAN_K; k++) {
        v[k] = base + k < n ? fp_load<C>(in + (base + k) * 8) : scan_id<C, MUL>();
        acc = scan_op<C, MUL>(acc, v[k]);
    }
    block_scan<C, MUL>(acc, lds);
    Fp<C> run = scan_op<C, MUL>(fp_load<C>(part + (size_t)blockIdx.x * 8), threadIdx.x ? fp_load<C>(lds + (threadIdx.x - 1) * 8) : scan_id<C, MUL>());
    for (int k = 0; k < SCAN_K; k++) {
        run = scan_op<C, MUL>(run, v[k]);
        if (base + k < n) fp_store<C>(out + (base + k) * 8, run);
    }
}
template <class C, bool MUL> static int scan_inclusive(const uint32_t* in, size_t n, uint32_t* out) {
    Ctx& cx = ctx();
    if (!n) return ZKMI_OK;
    const uint32_t nparts = (uint32_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    uint32_t* part;
    ZK_TRY(ws_get("plonk.scanpart", (size_t)nparts * 32, (void**)&part));
    hipLaunchKernelGGL((k_scan_totals<C, MUL>), dim3(nparts), dim3(SCAN_T), 0, cx.stream, in, n, part);
    hipLaunchKernelGGL((k_scan_parts<C, MUL>), dim3(1), dim3(SCAN_T), 0, cx.stream, part, nparts);
    hipLaunchKernelGGL((k_scan_final<C, MUL>), dim3(nparts), dim3(SCAN_T), 0, cx.stream, in, n, part, out);
    ZK_HIP(hipGetLastError());
    return ZKMI_OK;
}

// ---- wires ---------------------------------------------------------------------------------------------------------------
template <class C> __global__ void k_plonk_gather(const uint32_t* __restrict__ wit, uint32_t n_wit, const uint32_t* __restrict__ internal, uint32_t n_add,
                                                 const uint32_t* __restrict__ ma, const uint32_t* __restrict__ mb, const uint32_t* __restrict__ mc,
                                                 uint32_t n_constraints, uint32_t domain, uint32_t* __restrict__ A, uint32_t* __restrict__ B, uint32_t* __restrict__ Cc, int to_mont) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= domain) return;
    const uint32_t* maps[3] = {ma, mb, mc};
    uint32_t* outs[3] = {A, B, Cc};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Fp<C> v = fp_zero<C>();
        if (i < n_constraints) {
            const uint32_t id = maps[k][i];                              // getWitness (:207-215)
            if (id < n_wit) v = fp_load<C>(wit + (size_t)id * 8);
            else if (id < n_wit + n_add) v = fp_load<C>(internal + (size_t)(id - n_wit) * 8);
            if (to_mont) v = fp_to_mont(v);                              // Fr.batchToMontgomery (:278) in the same pass
        }
        fp_store<C>(outs[k] + (size_t)i * 8, v);
    }
}

// ---- calculateAdditions (plonk_prove.js:174-204, fflonk_prove.js:269-300) -------------------------------------------------------------------
// internal[i] = factor1_i * getWitness(id1_i) + factor2_i * getWitness(id2_i), i = 0 .. nAdditions-1, in index order in the reference: an
// operand may be an internal signal created EARLIER (id - nWitness < i), so the loop is a dependency DAG. In the keys the reference writes the DAG is
// SHALLOW: reduceCoefs (src/plonk_setup.js:176-212) takes two terms from the FRONT of the queue and appends their sum at the BACK, i.e. it folds a
// k-term linear combination as a balanced tree of depth ceil(log2 k) (17 for 10^5 terms), not as a chain. The latency of this kernel is proportional to
// the depth (one release / acquire hop through the L2 per level, a few microseconds each): a hand-made key with a chain 10^5 deep would take ~0.3 s per
// proof here — correct (the tests run chains 6 000 deep), slow, and nothing plonk.setup emits. One launch computes all of it: a lane owns addition i and polls a ready flag per
// internal operand; blocks take a ticket when they START, so a lane only ever waits for lanes of blocks that started before its own (resident
// or finished: forward progress without assuming an order of block dispatch). The poll loop's condition is WAVE-uniform (ballot) and the
// result is published INSIDE the loop: a lane that published keeps iterating, masked, until its whole wave is done — with a per-lane exit
// the compiler may sink the publish behind the loop's reconvergence point, where it waits for the very lanes that wait for it.
// Factors are Montgomery, signals normal form: the Montgomery product of the two is the normal-form product, as in the reference
// (Fr.mul(factor, witness) on the zkey's bytes). Records are 72 bytes (u32 id1, u32 id2, 2 x 32): word loads. An operand id at or beyond
// the addition's own slot reads what the reference reads there — its zero-initialised buffer, i.e. 0 — and ids >= nVars read Fr.zero (:213).
// a field element at an 8-byte aligned address (the factors of a 72-byte record): four 8-byte loads
template <class C> ZK_DEV Fp<C> fp_load_u2(const uint32_t* p) {
    static_assert(C::N == 8, "Fr is eight words on both curves");
    Fp<C> a;
    const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint2 v = q[k]; a.l[2 * k] = v.x; a.l[2 * k + 1] = v.y; }
    return a;
}
template <class C> __global__ void __launch_bounds__(256)
k_plonk_additions(const uint32_t* __restrict__ rec, uint32_t n_add, const uint32_t* __restrict__ wit, uint32_t n_wit, uint32_t* internal, uint32_t* flags, uint32_t* ticket) {
    __shared__ uint32_t bid_s;
    if (threadIdx.x == 0) bid_s = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t i = bid_s * 256u + threadIdx.x;
    const bool active = i < n_add;
    Fp<C> f[2], w[2];
    bool ready[2] = {true, true};
    uint32_t dep[2] = {0, 0};
    if (active) {
        const uint32_t* r = rec + (size_t)i * 18;
        const uint2 ids = *reinterpret_cast<const uint2*>(r);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t id = k ? ids.y : ids.x;
            f[k] = fp_load_u2<C>(r + 2 + 8 * k);
            w[k] = fp_zero<C>();
            if (id < n_wit) w[k] = fp_load<C>(wit + (size_t)id * 8);
            else if (id - n_wit < i) { ready[k] = false; dep[k] = id - n_wit; }       // an earlier internal signal: wait for it
        }
    }
    bool done = !active;
    while (__ballot(!done) != 0ull) {
        bool progressed = false;
        if (!done) {
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (!ready[k] && __hip_atomic_load(flags + dep[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    // published with a release store of the flag behind plain vector stores: the acquire above orders these loads behind it
                    const uint4* src = reinterpret_cast<const uint4*>(internal + (size_t)dep[k] * 8);
                    const uint4 lo = src[0], hi = src[1];
                    w[k].l[0] = lo.x; w[k].l[1] = lo.y; w[k].l[2] = lo.z; w[k].l[3] = lo.w; w[k].l[4] = hi.x; w[k].l[5] = hi.y; w[k].l[6] = hi.z; w[k].l[7] = hi.w;
                    ready[k] = true;
                }
            if (ready[0] && ready[1]) {
                const Fp<C> v = fp_add(fp_mul(f[0], w[0]), fp_mul(f[1], w[1]));
                uint4* dst = reinterpret_cast<uint4*>(internal + (size_t)i * 8);
                dst[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
                dst[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
                __hip_atomic_store(flags + i, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
                progressed = true;
            }
        }
        if (__ballot(progressed) == 0ull) __builtin_amdgcn_s_sleep(1);
    }
}

// ---- computeZ -------------------------------------------------------------------------------------------------------------
// constants block (device): enum PK_* in plonk29.cuh
template <class C> ZK_DEV Fp<C> kc(const uint32_t* k, int i) { return fp_load<C>(k + (size_t)i * 8); }

template <class C> __global__ void __launch_bounds__(256)
k_plonk_z_factors(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, const uint32_t* __restrict__ Cc, const uint32_t* __restrict__ s1, const uint32_t* __restrict__ s2,
                  const uint32_t* __restrict__ s3, uint32_t domain, const uint32_t* __restrict__ k, PowTab wt, uint32_t* __restrict__ num, uint32_t* __restrict__ den) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= domain) return;
    const Fp<C> beta = kc<C>(k, PK_BETA), gamma = kc<C>(k, PK_GAMMA);
    const Fp<C> a = fp_load<C>(A + (size_t)i * 8), b = fp_load<C>(B + (size_t)i * 8), c = fp_load<C>(Cc + (size_t)i * 8);
    const Fp<C> betaw = fp_mul(beta, pow_tab<C>(wt, i));
    Fp<C> n1 = fp_add(fp_add(a, betaw), gamma);
    Fp<C> n2 = fp_add(fp_add(b, fp_mul(kc<C>(k, PK_K1), betaw)), gamma);
    Fp<C> n3 = fp_add(fp_add(c, fp_mul(kc<C>(k, PK_K2), betaw)), gamma);
    fp_store<C>(num + (size_t)i * 8, fp_mul(n1, fp_mul(n2, n3)));
    Fp<C> d1 = fp_add(fp_add(a, fp_mul(fp_load<C>(s1 + (size_t)i * 4 * 8), beta)), gamma);      // sigma evaluations sampled at stride 4 (:401-408)
    Fp<C> d2 = fp_add(fp_add(b, fp_mul(fp_load<C>(s2 + (size_t)i * 4 * 8), beta)), gamma);
    Fp<C> d3 = fp_add(fp_add(c, fp_mul(fp_load<C>(s3 + (size_t)i * 4 * 8), beta)), gamma);
    fp_store<C>(den + (size_t)i * 8, fp_mul(d1, fp_mul(d2, d3)));
}
// Z[i] = P_num[i] * inv(P_den[i]) with P[i] = prod_{j<i} (exclusive) and P[0] = total product; incN/incD = inclusive scans, invD = 1/incD
template <class C> __global__ void k_plonk_z_finish(const uint32_t* __restrict__ incN, const uint32_t* __restrict__ invD, uint32_t domain, uint32_t* __restrict__ Z) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= domain) return;
    const uint32_t j = i ? i - 1 : domain - 1;
    fp_store<C>(Z + (size_t)i * 8, fp_mul(fp_load<C>(incN + (size_t)j * 8), fp_load<C>(invD + (size_t)j * 8)));
}

// ---- computeT ---------------------------------------------------------------------------------------------------------------
// struct PlonkTArgs: plonk29.cuh
// CALLS: the field product as a CALL (16 argument registers, no stack traffic): k_plonk_t is 63 - 82 KB of straight-line code per part against
// a 64 KB instruction cache, 2.8 x slower on a slow-fetch box (field29.cuh: Compact; r03: 2.8 / 3.0 ms per part there against 1.0 ms)
template <class C> __device__ __attribute__((noinline)) Fp<C> fp_mul_call(Fp<C> a, Fp<C> b) { return fp_mul(a, b); }
template <class C, bool CALLS> ZK_DEV Fp<C> fp_mulx(const Fp<C>& a, const Fp<C>& b) {
    if constexpr (CALLS) return fp_mul_call<C>(a, b); else return fp_mul(a, b);
}
template <class C, bool CALLS = false> struct MulZ {
    Fp<C> Z1, Z2, Z3;
    bool p;
    // mul_z.js:49-71
    ZK_DEV void mul2(const Fp<C>& a, const Fp<C>& b, const Fp<C>& ap, const Fp<C>& bp, Fp<C>& r, Fp<C>& rz) const {
        r = fp_mulx<C, CALLS>(a, b);
        rz = fp_add(fp_mulx<C, CALLS>(a, bp), fp_mulx<C, CALLS>(ap, b));
        if (p) rz = fp_add(rz, fp_mulx<C, CALLS>(Z1, fp_mulx<C, CALLS>(ap, bp)));
    }
    // mul_z.js:103-148: the product (a + ap Z)(b + bp Z)(c + cp Z)(d + dp Z) reduced by the blinding terms. The reference expands it into 8
    // pair products and 15 triple / quadruple sums (27 field multiplications). The same field elements follow from two quadratics
    // (A0 + A1 Z + A2 Z^2)(B0 + B1 Z + B2 Z^2), A = (a b, a bp + ap b, ap bp), B likewise: Karatsuba on the pairs (3 + 3) and on the
    // quadratics (6) + the three weights = 15 multiplications where all coefficients are needed (3 of 4 points), 9 where only Z^1 is.
    // Field arithmetic is exact: the results are the reference's, bit for bit (tests: stages against the oracle's literal expansion).
    ZK_DEV void mul4(const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d, const Fp<C>& ap, const Fp<C>& bp, const Fp<C>& cp, const Fp<C>& dp, Fp<C>& r, Fp<C>& rz) const {
        const Fp<C> A0 = fp_mulx<C, CALLS>(a, b), B0 = fp_mulx<C, CALLS>(c, d);
        r = fp_mulx<C, CALLS>(A0, B0);
        if (!p) {
            const Fp<C> u = fp_add(fp_mulx<C, CALLS>(a, bp), fp_mulx<C, CALLS>(ap, b)), v = fp_add(fp_mulx<C, CALLS>(c, dp), fp_mulx<C, CALLS>(cp, d));
            rz = fp_add(fp_mulx<C, CALLS>(u, B0), fp_mulx<C, CALLS>(A0, v));                                          // Z^1 only
            return;
        }
        const Fp<C> A2 = fp_mulx<C, CALLS>(ap, bp), B2 = fp_mulx<C, CALLS>(cp, dp);
        const Fp<C> A1 = fp_sub(fp_sub(fp_mulx<C, CALLS>(fp_add(a, ap), fp_add(b, bp)), A0), A2);          // a bp + ap b
        const Fp<C> B1 = fp_sub(fp_sub(fp_mulx<C, CALLS>(fp_add(c, cp), fp_add(d, dp)), B0), B2);          // c dp + cp d
        const Fp<C> P1 = fp_mulx<C, CALLS>(A1, B1), P2 = fp_mulx<C, CALLS>(A2, B2);
        const Fp<C> P01 = fp_mulx<C, CALLS>(fp_add(A0, A1), fp_add(B0, B1)), P02 = fp_mulx<C, CALLS>(fp_add(A0, A2), fp_add(B0, B2)), P12 = fp_mulx<C, CALLS>(fp_add(A1, A2), fp_add(B1, B2));
        const Fp<C> z1 = fp_sub(fp_sub(P01, r), P1);                                            // Z^1: A0 B1 + A1 B0
        const Fp<C> z2 = fp_add(fp_sub(fp_sub(P02, r), P2), P1);                                // Z^2: A0 B2 + A1 B1 + A2 B0
        const Fp<C> z3 = fp_sub(fp_sub(P12, P1), P2);                                           // Z^3: A1 B2 + A2 B1
        rz = fp_add(fp_add(z1, fp_mulx<C, CALLS>(Z1, z2)), fp_add(fp_mulx<C, CALLS>(Z2, z3), fp_mulx<C, CALLS>(Z3, P2)));         // Z^4: A2 B2
    }
};
// One lane per extended evaluation point. The numerator is a sum of four terms with little in common besides the wire values, and
// the live set of all four together exceeds 256 VGPRs (1 wave per SIMD: 5.5 ms at 2^20, or 3.7 ms with scratch at 3 waves), so it is
// evaluated in three launches that accumulate into t / tz:  PART 0: e1 + e4 (store),  PART 1: += e2,  PART 2: -= e3.
template <class C, int PART, bool CALLS = false> __global__ void __launch_bounds__(256, 2) k_plonk_t(PlonkTArgs g, PowTab w4) {
    const uint32_t n4 = 4 * g.domain;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    auto ld = [&](const uint32_t* p, size_t idx) { return fp_load<C>(p + idx * 8); };
    const uint32_t* k = g.k;
    const Fp<C> w = pow_tab<C>(w4, i);                                     // w = Fr.w[power+2]^i
    auto bl = [&](int j) { return kc<C>(k, PK_B1 + j - 1); };              // challenges.b[j]
    const Fp<C> a = ld(g.a, i), b = ld(g.b, i), c = ld(g.c, i);
    const Fp<C> ap = fp_add(bl(2), fp_mulx<C, CALLS>(bl(1), w)), bp = fp_add(bl(4), fp_mulx<C, CALLS>(bl(3), w)), cp = fp_add(bl(6), fp_mulx<C, CALLS>(bl(5), w));
    MulZ<C, CALLS> mz;
    mz.p = (i & 3) != 0;
    mz.Z1 = kc<C>(k, PK_Z1 + (i & 3)); mz.Z2 = kc<C>(k, PK_Z2 + (i & 3)); mz.Z3 = kc<C>(k, PK_Z3 + (i & 3));
    uint32_t* pt = g.t + (size_t)i * 8;
    uint32_t* ptz = g.tz + (size_t)i * 8;
    if constexpr (PART == 0) {
        // e1 := a b qM + a qL + b qR + c qO + PI + qC ;  e4 := alpha^2 (z - 1) L1
        Fp<C> pi = fp_zero<C>();
        for (uint32_t j = 0; j < g.n_public; j++)
            pi = fp_sub(pi, fp_mulx<C, CALLS>(ld(g.lagrange, (size_t)j * 5 * g.domain + g.domain + i), ld(g.pub_a, j)));
        Fp<C> e1, e1z;
        mz.mul2(a, b, ap, bp, e1, e1z);
        const Fp<C> qm = ld(g.qm, i), ql = ld(g.ql, i), qr = ld(g.qr, i), qo = ld(g.qo, i);
        e1 = fp_mulx<C, CALLS>(e1, qm); e1z = fp_mulx<C, CALLS>(e1z, qm);
        e1 = fp_add(e1, fp_mulx<C, CALLS>(a, ql)); e1z = fp_add(e1z, fp_mulx<C, CALLS>(ap, ql));
        e1 = fp_add(e1, fp_mulx<C, CALLS>(b, qr)); e1z = fp_add(e1z, fp_mulx<C, CALLS>(bp, qr));
        e1 = fp_add(e1, fp_mulx<C, CALLS>(c, qo)); e1z = fp_add(e1z, fp_mulx<C, CALLS>(cp, qo));
        e1 = fp_add(fp_add(e1, pi), ld(g.qc, i));
        const Fp<C> z = ld(g.z, i), alpha2 = kc<C>(k, PK_ALPHA2);
        const Fp<C> zp = fp_add(fp_mulx<C, CALLS>(fp_add(fp_mulx<C, CALLS>(bl(7), w), bl(8)), w), bl(9));
        const Fp<C> l1 = ld(g.lagrange, (size_t)g.domain + i);
        fp_store<C>(pt, fp_add(e1, fp_mulx<C, CALLS>(fp_mulx<C, CALLS>(fp_sub(z, kc<C>(k, PK_ONE)), l1), alpha2)));
        fp_store<C>(ptz, fp_add(e1z, fp_mulx<C, CALLS>(fp_mulx<C, CALLS>(zp, l1), alpha2)));
    } else if constexpr (PART == 3) {
        // PART 1 and PART 2 in one pass (the factored mul4 leaves room for both): t += alpha (e2 - e3), the wire values, w, the blinding
        // evaluations and t / tz touched once
        const Fp<C> beta = kc<C>(k, PK_BETA), gamma = kc<C>(k, PK_GAMMA), alpha = kc<C>(k, PK_ALPHA);
        Fp<C> e, ez;
        {
            const Fp<C> zp = fp_add(fp_mulx<C, CALLS>(fp_add(fp_mulx<C, CALLS>(bl(7), w), bl(8)), w), bl(9));
            const Fp<C> betaw = fp_mulx<C, CALLS>(beta, w);
            mz.mul4(fp_add(fp_add(a, betaw), gamma), fp_add(fp_add(b, fp_mulx<C, CALLS>(betaw, kc<C>(k, PK_K1))), gamma), fp_add(fp_add(c, fp_mulx<C, CALLS>(betaw, kc<C>(k, PK_K2))), gamma), ld(g.z, i), ap, bp,
                    cp, zp, e, ez);
        }
        {
            Fp<C> e3, e3z;
            const Fp<C> wW = fp_mulx<C, CALLS>(w, kc<C>(k, PK_WN));
            const Fp<C> zWp = fp_add(fp_mulx<C, CALLS>(fp_add(fp_mulx<C, CALLS>(bl(7), wW), bl(8)), wW), bl(9));
            mz.mul4(fp_add(fp_add(a, fp_mulx<C, CALLS>(beta, ld(g.s1, i))), gamma), fp_add(fp_add(b, fp_mulx<C, CALLS>(beta, ld(g.s2, i))), gamma), fp_add(fp_add(c, fp_mulx<C, CALLS>(beta, ld(g.s3, i))), gamma),
                    ld(g.z, (n4 + 4 + i) % n4), ap, bp, cp, zWp, e3, e3z);
            e = fp_sub(e, e3); ez = fp_sub(ez, e3z);
        }
        fp_store<C>(pt, fp_add(fp_load<C>(pt), fp_mulx<C, CALLS>(e, alpha)));
        fp_store<C>(ptz, fp_add(fp_load<C>(ptz), fp_mulx<C, CALLS>(ez, alpha)));
    } else {
        const Fp<C> beta = kc<C>(k, PK_BETA), gamma = kc<C>(k, PK_GAMMA), alpha = kc<C>(k, PK_ALPHA);
        Fp<C> e, ez;
        if constexpr (PART == 1) {
            // e2 := alpha (a + beta X + gamma)(b + beta k1 X + gamma)(c + beta k2 X + gamma) z
            const Fp<C> zp = fp_add(fp_mulx<C, CALLS>(fp_add(fp_mulx<C, CALLS>(bl(7), w), bl(8)), w), bl(9));
            const Fp<C> betaw = fp_mulx<C, CALLS>(beta, w);
            mz.mul4(fp_add(fp_add(a, betaw), gamma), fp_add(fp_add(b, fp_mulx<C, CALLS>(betaw, kc<C>(k, PK_K1))), gamma), fp_add(fp_add(c, fp_mulx<C, CALLS>(betaw, kc<C>(k, PK_K2))), gamma), ld(g.z, i), ap, bp,
                    cp, zp, e, ez);
            fp_store<C>(pt, fp_add(fp_load<C>(pt), fp_mulx<C, CALLS>(e, alpha)));
            fp_store<C>(ptz, fp_add(fp_load<C>(ptz), fp_mulx<C, CALLS>(ez, alpha)));
        } else {
            // e3 := alpha (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma) z(Xw)
            const Fp<C> wW = fp_mulx<C, CALLS>(w, kc<C>(k, PK_WN));
            const Fp<C> zWp = fp_add(fp_mulx<C, CALLS>(fp_add(fp_mulx<C, CALLS>(bl(7), wW), bl(8)), wW), bl(9));
            mz.mul4(fp_add(fp_add(a, fp_mulx<C, CALLS>(beta, ld(g.s1, i))), gamma), fp_add(fp_add(b, fp_mulx<C, CALLS>(beta, ld(g.s2, i))), gamma), fp_add(fp_add(c, fp_mulx<C, CALLS>(beta, ld(g.s3, i))), gamma),
                    ld(g.z, (n4 + 4 + i) % n4), ap, bp, cp, zWp, e, ez);
            fp_store<C>(pt, fp_sub(fp_load<C>(pt), fp_mulx<C, CALLS>(e, alpha)));
            fp_store<C>(ptz, fp_sub(fp_load<C>(ptz), fp_mulx<C, CALLS>(ez, alpha)));
        }
    }
}

// The same numerator on 9 x 29-bit limbs (plonk29.cuh): same three launches, same buffers, the same bytes out. F = the field, or Compact<field>
// with the products behind calls.
template <class F, int PART> __global__ void __launch_bounds__(256, 2) k_plonk_t29(PlonkTArgs g, PowTab w4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * g.domain) return;
    plonk_t29_point<Fp29<F>, PART>(g, w4, i);
}

// ---- FFLONK quotient numerators (src/fflonk_prove.js) ------------------------------------------------------------------------
// T0 (:415-504): q_L a + q_R b + q_M a b + q_O c + q_C + PI over the 4n extended points
struct FflonkT0Args { const uint32_t *a, *b, *c, *ql, *qr, *qm, *qo, *qc, *lagrange, *pub_a; uint32_t domain, n_public; uint32_t* t0; };
template <class C> __global__ void __launch_bounds__(256) k_fflonk_t0(FflonkT0Args g) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * g.domain) return;
    auto ld = [&](const uint32_t* p, size_t idx) { return fp_load<C>(p + idx * 8); };
    const Fp<C> a = ld(g.a, i), b = ld(g.b, i), c = ld(g.c, i);
    Fp<C> pi = fp_zero<C>();
    for (uint32_t j = 0; j < g.n_public; j++) pi = fp_sub(pi, fp_mul(ld(g.lagrange, (size_t)j * 5 * g.domain + g.domain + i), ld(g.pub_a, j)));
    const Fp<C> e1 = fp_mul(a, ld(g.ql, i)), e2 = fp_mul(b, ld(g.qr, i)), e3 = fp_mul(fp_mul(a, b), ld(g.qm, i)), e4 = fp_mul(c, ld(g.qo, i));
    fp_store<C>(g.t0 + (size_t)i * 8, fp_add(e1, fp_add(e2, fp_add(e3, fp_add(e4, fp_add(ld(g.qc, i), pi))))));
}
// T1 (:667-718): (z - 1) L_1 and z' L_1 over 2n points (every second extended evaluation); k: [0..2] = b7, b8, b9, [3] = one
template <class C> __global__ void __launch_bounds__(256)
k_fflonk_t1(const uint32_t* __restrict__ z4, const uint32_t* __restrict__ lagrange, uint32_t domain, const uint32_t* __restrict__ k, PowTab w2n, uint32_t* __restrict__ t1, uint32_t* __restrict__ t1z) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * domain) return;
    const Fp<C> om = pow_tab<C>(w2n, i), om2 = fp_sqr(om);
    const Fp<C> z = fp_load<C>(z4 + (size_t)2 * i * 8);
    const Fp<C> zp = fp_add(fp_add(fp_mul(kc<C>(k, 0), om2), fp_mul(kc<C>(k, 1), om)), kc<C>(k, 2));
    const Fp<C> l1 = fp_load<C>(lagrange + ((size_t)domain + 2 * i) * 8);
    fp_store<C>(t1 + (size_t)i * 8, fp_mul(fp_sub(z, kc<C>(k, 3)), l1));
    fp_store<C>(t1z + (size_t)i * 8, fp_mul(zp, l1));
}
// T2 (:720-815): permutation argument numerator and its blinding part over 4n points;
// k: [0] beta [1] gamma [2] k1 [3] k2 [4] w_n [5..7] b7, b8, b9
struct FflonkT2Args { const uint32_t *a, *b, *c, *z, *s1, *s2, *s3, *k; uint32_t domain; uint32_t *t2, *t2z; };
template <class C> __global__ void __launch_bounds__(256) k_fflonk_t2(FflonkT2Args g, PowTab w4) {
    const uint32_t n4 = 4 * g.domain;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    auto ld = [&](const uint32_t* p, size_t idx) { return fp_load<C>(p + idx * 8); };
    const uint32_t* k = g.k;
    const Fp<C> om = pow_tab<C>(w4, i), om2 = fp_sqr(om), omW = fp_mul(om, kc<C>(k, 4)), omW2 = fp_sqr(omW);
    const Fp<C> a = ld(g.a, i), b = ld(g.b, i), c = ld(g.c, i), z = ld(g.z, i), zW = ld(g.z, (n4 + 4 + i) % n4);
    const Fp<C> beta = kc<C>(k, 0), gamma = kc<C>(k, 1);
    const Fp<C> zp = fp_add(fp_add(fp_mul(kc<C>(k, 5), om2), fp_mul(kc<C>(k, 6), om)), kc<C>(k, 7));
    const Fp<C> zWp = fp_add(fp_add(fp_mul(kc<C>(k, 5), omW2), fp_mul(kc<C>(k, 6), omW)), kc<C>(k, 7));
    const Fp<C> betaX = fp_mul(beta, om);
    const Fp<C> e11 = fp_add(fp_add(a, betaX), gamma), e12 = fp_add(fp_add(b, fp_mul(betaX, kc<C>(k, 2))), gamma), e13 = fp_add(fp_add(c, fp_mul(betaX, kc<C>(k, 3))), gamma);
    const Fp<C> p1 = fp_mul(fp_mul(e11, e12), e13);
    const Fp<C> e21 = fp_add(fp_add(a, fp_mul(beta, ld(g.s1, i))), gamma), e22 = fp_add(fp_add(b, fp_mul(beta, ld(g.s2, i))), gamma), e23 = fp_add(fp_add(c, fp_mul(beta, ld(g.s3, i))), gamma);
    const Fp<C> p2 = fp_mul(fp_mul(e21, e22), e23);
    fp_store<C>(g.t2 + (size_t)i * 8, fp_sub(fp_mul(p1, z), fp_mul(p2, zW)));
    fp_store<C>(g.t2z + (size_t)i * 8, fp_sub(fp_mul(p1, zp), fp_mul(p2, zWp)));
}
// highest index of a non-zero element (0 when all vanish): Polynomial.degree (polynomial.js:163-172). One ballot per wave finds the wave's highest
// non-zero lane; the blocks walk the array from its END, so that the first waves to finish already hold the answer and the rest see a larger value and
// skip their atomic (r04: one atomicMax per non-zero element on one address took 190 us per call at 2^20 - 2^22 elements, 15 calls per FFLONK proof).
static __global__ void __launch_bounds__(256) k_poly_degree(const uint32_t* __restrict__ p, size_t n, unsigned long long* __restrict__ deg) {
    const size_t i = (size_t)(gridDim.x - 1 - blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t o = 0;
    if (i < n) {
        const uint4* q = reinterpret_cast<const uint4*>(p + i * 8);
        const uint4 a = q[0], b = q[1];
        o = a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w;
    }
    const unsigned long long m = __ballot(o != 0);
    if (m == 0 || (threadIdx.x & 63) != 0) return;
    const unsigned long long cand = (unsigned long long)i + (unsigned long long)(63 - __clzll((long long)m));        // lanes of a wave hold consecutive indices
    if (cand > __atomic_load_n(deg, __ATOMIC_RELAXED)) atomicMax(deg, cand);
}

// ---- polynomial ops ---------------------------------------------------------------------------------------------------------
// y[i] = y[i] +/- (k ? k*x[i] : x[i]),  i < nx
// blindCoefficients (polynomial.js:68-93): p[n+i] += f_i, p[i] -= f_i;  addScalar (:286-290): p[0] += f_0 (n = 0, sub = 0)
constexpr int BLIND_MAX = 8;                             // blinding factors of one call (the reference uses 2 or 3; addScalar is count = 1)
struct BlindArgs { FrK f[BLIND_MAX]; };
// mode 0: p[n+i] += f_i (addScalar: n = 0, count = 1); 1: and p[i] -= f_i (blindCoefficients on a zero-padded copy); 2: p[n+i] = f_i, p[i] -= f_i (blindCoefficients IN PLACE:
// the buffer has room for n + count coefficients and nothing was written behind n yet)
template <class C> __global__ void k_poly_blind(uint32_t* __restrict__ p, size_t n, BlindArgs a, int count, int mode) {
    const int i = threadIdx.x;
    if (i >= count) return;
    const Fp<C> v = frk_load<C>(a.f[i]);
    fp_store<C>(p + (n + i) * 8, mode == 2 ? v : fp_add(fp_load<C>(p + (n + i) * 8), v));
    if (mode) fp_store<C>(p + (size_t)i * 8, fp_sub(fp_load<C>(p + (size_t)i * 8), v));
}
template <class C> __global__ void k_poly_axpy(uint32_t* __restrict__ y, const uint32_t* __restrict__ x, size_t nx, FrK k, int has_k, int subtract) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    Fp<C> b = fp_load<C>(x + i * 8);
    if (has_k) b = fp_mul(b, frk_load<C>(k));
    const Fp<C> a = fp_load<C>(y + i * 8);
    fp_store<C>(y + i * 8, subtract ? fp_sub(a, b) : fp_add(a, b));
}
template <class C> __global__ void k_poly_scale(uint32_t* __restrict__ p, size_t n, FrK k) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fp_store<C>(p + i * 8, fp_mul(fp_load<C>(p + i * 8), frk_load<C>(k)));
}
// out[i] = sum_j k_j p_j[i] (i < len_j) + (i == 0 ? constant : 0): a whole chain of Polynomial.add / sub / mulScalar / addScalar (polynomial.js:218-290) in ONE pass over the operands.
// Fr arithmetic is exact, so the order of the additions does not show in the result: the linearisation polynomial and the opening numerators of round 5
// (plonk_prove.js:769-866: 18 add / sub calls, 2 mulScalar, 5 copies of selector polynomials) come out bit-identical from one launch each.
constexpr int LINCOMB_MAX = 16;
struct LincombArgs { const uint32_t* p[LINCOMB_MAX]; uint64_t len[LINCOMB_MAX]; FrK k[LINCOMB_MAX]; FrK constant; uint32_t n, has_k, has_const; };
template <class C> __global__ void __launch_bounds__(256) k_poly_lincomb(LincombArgs a, uint32_t* __restrict__ out, size_t out_len) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_len) return;
    Fp<C> acc = (i == 0 && a.has_const) ? frk_load<C>(a.constant) : fp_zero<C>();
    for (uint32_t j = 0; j < a.n; j++) {
        if (i >= a.len[j]) continue;
        Fp<C> x = fp_load<C>(a.p[j] + i * 8);
        if ((a.has_k >> j) & 1u) x = fp_mul(x, frk_load<C>(a.k[j]));
        acc = fp_add(acc, x);
    }
    fp_store<C>(out + i * 8, acc);
}
// the split of T into T1 | T2 | T3 with the blinding of round 3 (plonk_prove.js:649-672): T1 = T[0, n) + b10 X^n, T2 = T[n, 2n) - b10 + b11 X^n, T3 = T[2n, 3n+6) - b11
template <class C> __global__ void k_plonk_split_t(const uint32_t* __restrict__ t, size_t t_len, uint32_t n, FrK b10, FrK b11, uint32_t* __restrict__ t1, uint32_t* __restrict__ t2, uint32_t* __restrict__ t3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + 6) return;
    if (i <= n) {
        Fp<C> a = i < n ? fp_load<C>(t + (size_t)i * 8) : frk_load<C>(b10);
        fp_store<C>(t1 + (size_t)i * 8, a);
        Fp<C> b = i < n ? fp_load<C>(t + ((size_t)n + i) * 8) : frk_load<C>(b11);
        if (i == 0) b = fp_sub(b, frk_load<C>(b10));
        fp_store<C>(t2 + (size_t)i * 8, b);
    }
    Fp<C> c = 2 * (size_t)n + i < t_len ? fp_load<C>(t + (2 * (size_t)n + i) * 8) : fp_zero<C>();
    if (i == 0) c = fp_sub(c, frk_load<C>(b11));
    fp_store<C>(t3 + (size_t)i * 8, c);
}
// Fr.batchFromMontgomery of up to four arrays in one launch (blockIdx.y = array): the scalars of a round's commitments
struct ConvertMultiArgs { const uint32_t* in[4]; uint32_t* out[4]; uint64_t n[4]; };
template <class C> __global__ void k_fr_convert_multi(ConvertMultiArgs a, int op) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= a.n[j]) return;
    const Fp<C> x = fp_load<C>(a.in[j] + i * 8);
    fp_store<C>(a.out[j] + i * 8, op == 0 ? fp_to_mont(x) : fp_from_mont(x));
}
static __global__ void k_poly_any_nonzero(const uint32_t* __restrict__ p, size_t n_words, uint32_t* __restrict__ flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words && p[i]) atomicOr(flag, 1u);
}
// divZh: c[i] <- -c[i] (i < n); c[i] <- c[i-n] - c[i] (i >= n, using the UPDATED c[i-n]); one lane per residue class.
// bad: set when a coefficient that must vanish (i > n*(ext-1) - ext) does not.
template <class C> __global__ void k_poly_div_zh(uint32_t* __restrict__ c, size_t len, uint32_t n, uint32_t ext, uint32_t* __restrict__ bad) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || j >= len) return;
    Fp<C> prev = fp_neg(fp_load<C>(c + (size_t)j * 8));
    fp_store<C>(c + (size_t)j * 8, prev);
    for (size_t i = (size_t)j + n; i < len; i += n) {
        prev = fp_sub(prev, fp_load<C>(c + i * 8));
        fp_store<C>(c + i * 8, prev);
        if (i > (size_t)n * (ext - 1) - ext && !fp_is_zero(prev)) atomicAdd(bad, 1u);
    }
}
// Horner by chunks: partial[b] = sum over the block's coefficients c_i x^i
constexpr int EV_K = 16, EVAL_MAX = 8;
// up to EVAL_MAX evaluations in one launch (blockIdx.y = which): round 4 of PLONK evaluates six polynomials at two points (plonk_prove.js:686-708)
struct EvalArgs { const uint32_t* c[EVAL_MAX]; uint64_t n[EVAL_MAX]; FrK x[EVAL_MAX]; PowTab xt[EVAL_MAX]; uint32_t np[EVAL_MAX], off[EVAL_MAX]; };
template <class C> __global__ void __launch_bounds__(256)
k_poly_eval_partial(EvalArgs a, uint32_t* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 8];
    const uint32_t q = blockIdx.y;
    if (blockIdx.x >= a.np[q]) return;
    const uint32_t* __restrict__ c = a.c[q];
    const size_t n = a.n[q];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * EV_K;
    const Fp<C> xv = frk_load<C>(a.x[q]);
    Fp<C> acc = fp_zero<C>();
    for (int k = EV_K - 1; k >= 0; k--) { acc = fp_mul(acc, xv); if (base + k < n) acc = fp_add(acc, fp_load<C>(c + (base + k) * 8)); }
    if (base < n) acc = fp_mul(acc, pow_tab<C>(a.xt[q], base)); else acc = fp_zero<C>();
    fp_store<C>(lds + threadIdx.x * 8, acc);
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (threadIdx.x < (uint32_t)d) { acc = fp_add(acc, fp_load<C>(lds + (threadIdx.x + d) * 8)); fp_store<C>(lds + threadIdx.x * 8, acc); }
        __syncthreads();
    }
    if (threadIdx.x == 0) fp_store<C>(part + ((size_t)a.off[q] + blockIdx.x) * 8, acc);
}
// out[q] = sum of the partials of evaluation q (one block each)
template <class C> __global__ void __launch_bounds__(256) k_poly_sum(EvalArgs a, const uint32_t* __restrict__ part, uint32_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 8];
    const uint32_t q = blockIdx.x, np = a.np[q];
    part += (size_t)a.off[q] * 8;
    Fp<C> acc = fp_zero<C>();
    for (uint32_t i = threadIdx.x; i < np; i += 256) acc = fp_add(acc, fp_load<C>(part + (size_t)i * 8));
    fp_store<C>(lds + threadIdx.x * 8, acc);
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (threadIdx.x < (uint32_t)d) { acc = fp_add(acc, fp_load<C>(lds + (threadIdx.x + d) * 8)); fp_store<C>(lds + threadIdx.x * 8, acc); }
        __syncthreads();
    }
    if (threadIdx.x == 0) fp_store<C>(out + (size_t)q * 8, acc);
}
// divByZerofier(1, beta): u_i = -c_i * beta^i * (1/beta); S = inclusive prefix sums of u; q_i = S_i * (1/beta)^i
// (chain k of residue `off` modulo `stride`: element index off + k*stride; stride = 1 for the PLONK openings)
template <class C> __global__ void k_dz_weight(const uint32_t* __restrict__ c, size_t m, size_t off, size_t stride, PowTab bt, FrK inv_beta, uint32_t* __restrict__ u) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    fp_store<C>(u + k * 8, fp_neg(fp_mul(fp_mul(fp_load<C>(c + (off + k * stride) * 8), pow_tab<C>(bt, k)), frk_load<C>(inv_beta))));
}
template <class C> __global__ void k_dz_unweight(const uint32_t* __restrict__ s, size_t m, size_t off, size_t stride, PowTab it, uint32_t* __restrict__ q) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    fp_store<C>(q + (off + k * stride) * 8, fp_mul(fp_load<C>(s + k * 8), pow_tab<C>(it, k)));
}
// many short chains (n large): one lane per residue, sequential along the chain; bad != 0 if a coefficient that must vanish does not
template <class C> __global__ void k_dz_chain(uint32_t* __restrict__ c, size_t len, uint32_t n, FrK inv_beta, uint32_t* __restrict__ bad) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || r >= len) return;
    const Fp<C> ib = frk_load<C>(inv_beta);
    Fp<C> prev = fp_neg(fp_mul(ib, fp_load<C>(c + (size_t)r * 8)));
    fp_store<C>(c + (size_t)r * 8, prev);
    for (size_t i = (size_t)r + n; i < len; i += n) {
        prev = fp_mul(ib, fp_sub(prev, fp_load<C>(c + i * 8)));
        fp_store<C>(c + i * 8, prev);
        if (i + n + 1 > len && !fp_is_zero(prev)) atomicAdd(bad, 1u);           // i > len - n - 1
    }
}
// CPolynomial.getPolynomial (cpolynomial.js:53-73): out[i*n + j] = P_j[i], i < len_j
struct InterleaveArgs { const uint32_t* p[16]; uint32_t len[16]; uint32_t n; };
template <class C> __global__ void k_cpoly_interleave(InterleaveArgs a, uint32_t* __restrict__ out, size_t out_len) {
    size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out_len) return;
    const uint32_t j = (uint32_t)(o % a.n);
    const size_t i = o / a.n;
    Fp<C> v = fp_zero<C>();
    if (a.p[j] && i < a.len[j]) v = fp_load<C>(a.p[j] + i * 8);
    fp_store<C>(out + o * 8, v);
}

// ---- host drivers (templated on the Fr configuration) -----------------------------------------------------------------------------
template <class C> struct PlonkOps {
    static HFr F() { return HFr::from_cfg<C>(); }
    static HE he(const uint8_t* p) { HE e; memcpy(e.v, p, 32); return e; }

    static int gather(const void* w, uint32_t nw, const void* in, uint32_t na, const void* ma, const void* mb, const void* mc, uint32_t ncon, uint32_t dom, void* A, void* B, void* Cc,
                      int to_mont = 0) {
        hipLaunchKernelGGL((k_plonk_gather<C>), dim3((dom + 255) / 256), dim3(256), 0, ctx().stream, (const uint32_t*)w, nw, (const uint32_t*)in, na, (const uint32_t*)ma, (const uint32_t*)mb,
                           (const uint32_t*)mc, ncon, dom, (uint32_t*)A, (uint32_t*)B, (uint32_t*)Cc, to_mont);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int additions(const void* rec, uint32_t na, const void* w, uint32_t nw, void* internal) {
        if (!na) return ZKMI_OK;
        Ctx& cx = ctx();
        uint32_t* flags;
        ZK_TRY(ws_get("plonk.addflags", ((size_t)na + 1) * 4, (void**)&flags));
        ZK_HIP(hipMemsetAsync(flags, 0, ((size_t)na + 1) * 4, cx.stream));                 // ready flags + the block ticket behind them
        hipLaunchKernelGGL((k_plonk_additions<C>), dim3((na + 255) / 256), dim3(256), 0, cx.stream, (const uint32_t*)rec, na, (const uint32_t*)w, nw, (uint32_t*)internal, flags, flags + na);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    // Per-call constants (challenges, blinding factors) go through a ring of pinned host / device slots: the copy is truly
    // asynchronous and the host never waits for the stream. A slot is reused only after the work that was queued up to the NEXT
    // upload has completed (event recorded at that point), i.e. after the kernels that read it.
    static constexpr int RING = 64, SLOT_BYTES = 2048;
    struct ConstRing { uint8_t* h = nullptr; uint8_t* d = nullptr; hipEvent_t ev[RING] = {}; bool used[RING] = {}; int next = 0, prev = -1; };
    static ConstRing& ring() { static ConstRing r[2]; return r[ctx().pipe & 1]; }        // one ring per pipeline slot: the slot events are recorded on that slot's stream
    static int upload_consts(const std::vector<HE>& v, const char* name, uint32_t** d) {
        (void)name;
        ConstRing& r = ring();
        hipStream_t st = ctx().stream;
        if (v.size() * 32 > (size_t)SLOT_BYTES) return fail(ZKMI_ERR_INVALID, "plonk: constants block too large");
        if (!r.h) {
            ZK_HIP(hipHostMalloc((void**)&r.h, (size_t)RING * SLOT_BYTES, hipHostMallocDefault));
            ZK_HIP(hipMalloc((void**)&r.d, (size_t)RING * SLOT_BYTES));
            for (int i = 0; i < RING; i++) ZK_HIP(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
        }
        if (r.prev >= 0) { ZK_HIP(hipEventRecord(r.ev[r.prev], st)); r.used[r.prev] = true; }
        const int slot = r.next;
        r.next = (r.next + 1) % RING;
        if (r.used[slot]) ZK_HIP(hipEventSynchronize(r.ev[slot]));
        memcpy(r.h + (size_t)slot * SLOT_BYTES, v.data(), v.size() * 32);
        ZK_HIP(hipMemcpyAsync(r.d + (size_t)slot * SLOT_BYTES, r.h + (size_t)slot * SLOT_BYTES, v.size() * 32, hipMemcpyHostToDevice, st));
        r.prev = slot;
        *d = (uint32_t*)(r.d + (size_t)slot * SLOT_BYTES);
        return ZKMI_OK;
    }
    // check = false: enqueue only — the caller reads Z[0] at its next synchronisation point (two proofs in flight: the host must not wait here)
    static int compute_z(const void* A, const void* B, const void* Cc, const void* s1, const void* s2, const void* s3, uint32_t dom, const uint8_t* beta, const uint8_t* gamma, const uint8_t* k1,
                         const uint8_t* k2, const uint8_t* w_n, void* Z, bool check = true) {
        Ctx& cx = ctx();
        const HFr Fh = F();
        std::vector<HE> kv(PK_COUNT, Fh.zero());
        kv[PK_BETA] = he(beta); kv[PK_GAMMA] = he(gamma); kv[PK_K1] = he(k1); kv[PK_K2] = he(k2);
        uint32_t* dk;
        ZK_TRY(upload_consts(kv, "plonk.kz", &dk));
        PowTab wt;
        ZK_TRY(build_pow_tab(Fh, he(w_n), clog2(dom), "plonk.powz", &wt));
        uint32_t *num, *den;
        ZK_TRY(ws_get("plonk.znum", (size_t)dom * 32, (void**)&num));
        ZK_TRY(ws_get("plonk.zden", (size_t)dom * 32, (void**)&den));
        const unsigned blocks = (dom + 255) / 256;
        hipLaunchKernelGGL((k_plonk_z_factors<C>), dim3(blocks), dim3(256), 0, cx.stream, (const uint32_t*)A, (const uint32_t*)B, (const uint32_t*)Cc, (const uint32_t*)s1, (const uint32_t*)s2,
                           (const uint32_t*)s3, dom, dk, wt, num, den);
        ZK_TRY((scan_inclusive<C, true>(num, dom, num)));
        ZK_TRY((scan_inclusive<C, true>(den, dom, den)));
        ZK_TRY(fr_batch_dev_dispatch(std::is_same<C, Bn254Fr>::value ? ZKMI_CURVE_BN128 : ZKMI_CURVE_BLS12381, ZKMI_BATCH_INVERSE, den, den, dom));   // Fr.batchInverse (:420)
        hipLaunchKernelGGL((k_plonk_z_finish<C>), dim3(blocks), dim3(256), 0, cx.stream, num, den, dom, (uint32_t*)Z);
        ZK_HIP(hipGetLastError());
        if (!check) return ZKMI_OK;
        HE z0;
        ZK_HIP(hipMemcpyAsync(z0.v, Z, 32, hipMemcpyDeviceToHost, cx.stream));
        ZK_HIP(hipStreamSynchronize(cx.stream));
        if (!(z0 == Fh.One())) return fail(ZKMI_ERR_INVALID, "Copy constraints does not match");     // :437-439
        return ZKMI_OK;
    }
    static int compute_t(const zkmi_plonk_evals* ev, uint32_t dom, uint32_t n_public, const uint8_t* blind, const uint8_t* beta, const uint8_t* gamma, const uint8_t* alpha, const uint8_t* k1,
                         const uint8_t* k2, const uint8_t* w_n, const uint8_t* w_4n, const uint8_t* w_2, void* T, void* Tz) {
        Ctx& cx = ctx();
        const HFr Fh = F();
        std::vector<HE> kv(PK_COUNT, Fh.zero());
        kv[PK_BETA] = he(beta); kv[PK_GAMMA] = he(gamma); kv[PK_K1] = he(k1); kv[PK_K2] = he(k2); kv[PK_ALPHA] = he(alpha); kv[PK_ALPHA2] = Fh.sqr(he(alpha)); kv[PK_WN] = he(w_n);
        for (int j = 0; j < 11; j++) kv[PK_B1 + j] = he(blind + 32 * j);
        kv[PK_ONE] = Fh.One(); kv[PK_NALPHA] = Fh.neg(he(alpha));
        // MulZ constants (mul_z.js:21-47), w2 = Fr.w[2]
        const HE w2 = he(w_2), one = Fh.One(), two = Fh.from_u64(2), m1 = Fh.neg(one), m2 = Fh.neg(two);
        kv[PK_Z1 + 1] = Fh.add(m1, w2); kv[PK_Z1 + 2] = m2; kv[PK_Z1 + 3] = Fh.sub(m1, w2);
        kv[PK_Z2 + 1] = Fh.mul(m2, w2); kv[PK_Z2 + 2] = Fh.from_u64(4); kv[PK_Z2 + 3] = Fh.neg(Fh.mul(m2, w2));
        kv[PK_Z3 + 1] = Fh.add(two, Fh.mul(two, w2)); kv[PK_Z3 + 2] = Fh.neg(Fh.from_u64(8)); kv[PK_Z3 + 3] = Fh.sub(two, Fh.mul(two, w2));
        // the 29-bit kernels read every constant in R'-form as well (x 2^261 = 32 x in R-form): second half of the block
        const HE k32 = Fh.from_u64(32);
        static_assert(2 * PK_COUNT * 32 <= SLOT_BYTES, "both forms of the constants block share one ring slot");
        kv.resize(2 * PK_COUNT, Fh.zero());
        for (int j = 0; j < PK_COUNT; j++) kv[PK_COUNT + j] = Fh.mul(kv[j], k32);
        uint32_t* dk;
        ZK_TRY(upload_consts(kv, "plonk.kt", &dk));
        PowTab w4;
        ZK_TRY(build_pow_tab(Fh, he(w_4n), clog2((size_t)4 * dom), "plonk.powt", &w4));
        PlonkTArgs g;
        g.a = (const uint32_t*)ev->a; g.b = (const uint32_t*)ev->b; g.c = (const uint32_t*)ev->c; g.z = (const uint32_t*)ev->z;
        g.qm = (const uint32_t*)ev->qm; g.ql = (const uint32_t*)ev->ql; g.qr = (const uint32_t*)ev->qr; g.qo = (const uint32_t*)ev->qo; g.qc = (const uint32_t*)ev->qc;
        g.s1 = (const uint32_t*)ev->s1; g.s2 = (const uint32_t*)ev->s2; g.s3 = (const uint32_t*)ev->s3;
        g.lagrange = (const uint32_t*)ev->lagrange; g.pub_a = (const uint32_t*)ev->pub_a; g.k = dk; g.k29 = dk + (size_t)PK_COUNT * 8;
        g.domain = dom; g.n_public = n_public; g.t = (uint32_t*)T; g.tz = (uint32_t*)Tz;
        const dim3 grid((4 * dom + 255) / 256);
        // default: three launches (the merged e2 + e3 kernel measured the same on a healthy box and is twice the code); ZKMI_PLONK_T_PARTS=2 merges
        static const bool three = !(getenv("ZKMI_PLONK_T_PARTS") && atoi(getenv("ZKMI_PLONK_T_PARTS")) == 2);
        // Products CALLED (8 - 12 KB per part instead of 60 - 85 KB): the default on every box — same speed as the inlined kernels where
        // instruction fetch is healthy (38.4 / 37.9 against 38.0 / 38.3 proofs/s, same box), 2.6 x faster where it is not; an explicit
        // ZKMI_COMPACT_CODE mask without bit 4 selects the inlined kernels
        static const bool calls = getenv("ZKMI_COMPACT_CODE") ? (compact_code() & 16) != 0 : true;
        // Default (r04): the 29-bit-limb kernels of plonk29.cuh with the products inlined — 41 - 52 KB per part, inside the instruction cache, 0.66 + 0.75 +
        // 0.80 ms at 2^20 with two proofs in flight against 0.96 + 1.27 + 1.30 ms for the 32-bit kernels below (same box, same run: 39.5 / 39.9 against
        // 37.9 / 38.4 proofs/s; with the products behind calls 36.8: the operands of the three- and four-product sums travel through the stack).
        // ZKMI_PLONK_T29: 1 = that, 2 = products behind calls, 0 = the 32-bit-limb kernels (products called; also chosen by bit 4 of the ZKMI_COMPACT_CODE mask or of the box probe)
        const char* t29_env = getenv("ZKMI_PLONK_T29");                 // read per call (once per proof): the tests switch it
        // no variable: the inlined 29-bit kernels, except on a box whose instruction fetch beyond the cache the library's probe found slow (compact_code():
        // all bits set) — they were never measured on such a box, the called 32-bit kernels below were (r03: same speed there as on a healthy one)
        const int t29 = t29_env ? atoi(t29_env) : ((compact_code() & 16) ? 0 : 1);
        if (t29 == 1) {
            hipLaunchKernelGGL((k_plonk_t29<C, 0>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t29<C, 1>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t29<C, 2>), grid, dim3(256), 0, cx.stream, g, w4);
        } else if (t29 == 2) {
            hipLaunchKernelGGL((k_plonk_t29<Compact<C>, 0>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t29<Compact<C>, 1>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t29<Compact<C>, 2>), grid, dim3(256), 0, cx.stream, g, w4);
        } else if (calls) {
            hipLaunchKernelGGL((k_plonk_t<C, 0, true>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t<C, 1, true>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t<C, 2, true>), grid, dim3(256), 0, cx.stream, g, w4);
        } else if (three) {
            hipLaunchKernelGGL((k_plonk_t<C, 0>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t<C, 1>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t<C, 2>), grid, dim3(256), 0, cx.stream, g, w4);
        } else {
            hipLaunchKernelGGL((k_plonk_t<C, 0>), grid, dim3(256), 0, cx.stream, g, w4);
            hipLaunchKernelGGL((k_plonk_t<C, 3>), grid, dim3(256), 0, cx.stream, g, w4);
        }
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int fflonk_t0(const zkmi_plonk_evals* ev, uint32_t dom, uint32_t n_public, void* t0) {
        FflonkT0Args g;
        g.a = (const uint32_t*)ev->a; g.b = (const uint32_t*)ev->b; g.c = (const uint32_t*)ev->c;
        g.ql = (const uint32_t*)ev->ql; g.qr = (const uint32_t*)ev->qr; g.qm = (const uint32_t*)ev->qm; g.qo = (const uint32_t*)ev->qo; g.qc = (const uint32_t*)ev->qc;
        g.lagrange = (const uint32_t*)ev->lagrange; g.pub_a = (const uint32_t*)ev->pub_a; g.domain = dom; g.n_public = n_public; g.t0 = (uint32_t*)t0;
        hipLaunchKernelGGL((k_fflonk_t0<C>), dim3((4 * dom + 255) / 256), dim3(256), 0, ctx().stream, g);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int fflonk_t1(const void* z4, const void* lagrange, uint32_t dom, const uint8_t* b789, const uint8_t* w_2n, void* t1, void* t1z) {
        const HFr Fh = F();
        std::vector<HE> kv = {he(b789), he(b789 + 32), he(b789 + 64), Fh.One()};
        uint32_t* dk;
        ZK_TRY(upload_consts(kv, "plonk.kt1", &dk));
        PowTab w2;
        ZK_TRY(build_pow_tab(Fh, he(w_2n), clog2((size_t)2 * dom), "plonk.powt1", &w2));
        hipLaunchKernelGGL((k_fflonk_t1<C>), dim3((2 * dom + 255) / 256), dim3(256), 0, ctx().stream, (const uint32_t*)z4, (const uint32_t*)lagrange, dom, dk, w2, (uint32_t*)t1, (uint32_t*)t1z);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int fflonk_t2(const zkmi_plonk_evals* ev, uint32_t dom, const uint8_t* b789, const uint8_t* beta, const uint8_t* gamma, const uint8_t* k1, const uint8_t* k2, const uint8_t* w_n,
                         const uint8_t* w_4n, void* t2, void* t2z) {
        const HFr Fh = F();
        std::vector<HE> kv = {he(beta), he(gamma), he(k1), he(k2), he(w_n), he(b789), he(b789 + 32), he(b789 + 64)};
        uint32_t* dk;
        ZK_TRY(upload_consts(kv, "plonk.kt2", &dk));
        PowTab w4;
        ZK_TRY(build_pow_tab(Fh, he(w_4n), clog2((size_t)4 * dom), "plonk.powt", &w4));
        FflonkT2Args g;
        g.a = (const uint32_t*)ev->a; g.b = (const uint32_t*)ev->b; g.c = (const uint32_t*)ev->c; g.z = (const uint32_t*)ev->z;
        g.s1 = (const uint32_t*)ev->s1; g.s2 = (const uint32_t*)ev->s2; g.s3 = (const uint32_t*)ev->s3; g.k = dk; g.domain = dom; g.t2 = (uint32_t*)t2; g.t2z = (uint32_t*)t2z;
        hipLaunchKernelGGL((k_fflonk_t2<C>), dim3((4 * dom + 255) / 256), dim3(256), 0, ctx().stream, g, w4);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int axpy(void* y, const void* x, size_t nx, const uint8_t* k, int subtract) {
        if (!nx) return ZKMI_OK;
        FrK kk = {};
        if (k) kk = frk(k);
        hipLaunchKernelGGL((k_poly_axpy<C>), dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, ctx().stream, (uint32_t*)y, (const uint32_t*)x, nx, kk, k ? 1 : 0, subtract);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    // mode: k_poly_blind
    static int blind(void* p, size_t n, const uint8_t* factors, int count, int mode) {
        if (count < 1 || count > BLIND_MAX) return fail(ZKMI_ERR_INVALID, "poly_blind: 1..8 factors");
        if (mode && n < (size_t)count) return fail(ZKMI_ERR_INVALID, "poly_blind: polynomial shorter than the blinding factors");
        BlindArgs a = {};
        for (int i = 0; i < count; i++) a.f[i] = frk(factors + 32 * i);
        hipLaunchKernelGGL((k_poly_blind<C>), dim3(1), dim3(64), 0, ctx().stream, (uint32_t*)p, n, a, count, mode);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int scale(void* p, size_t n, const uint8_t* k) {
        if (!n) return ZKMI_OK;
        hipLaunchKernelGGL((k_poly_scale<C>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, (uint32_t*)p, n, frk(k));
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int lincomb(void* out, size_t out_len, const zkmi_poly_term* terms, int count, const uint8_t* constant) {
        if (count < 0 || count > LINCOMB_MAX) return fail(ZKMI_ERR_INVALID, "poly_lincomb: at most 16 terms");
        if (!out_len) return ZKMI_OK;
        LincombArgs a = {};
        a.n = (uint32_t)count;
        for (int j = 0; j < count; j++) {
            if (terms[j].len > out_len) return fail(ZKMI_ERR_INVALID, "poly_lincomb: a term is longer than the output");
            if (terms[j].len && !terms[j].d_p) return fail(ZKMI_ERR_INVALID, "null argument");
            a.p[j] = (const uint32_t*)terms[j].d_p; a.len[j] = terms[j].len;
            if (terms[j].has_k) { a.k[j] = frk(terms[j].k); a.has_k |= 1u << j; }
        }
        if (constant) { a.constant = frk(constant); a.has_const = 1; }
        hipLaunchKernelGGL((k_poly_lincomb<C>), dim3((unsigned)((out_len + 255) / 256)), dim3(256), 0, ctx().stream, a, (uint32_t*)out, out_len);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int split_t(const void* t, size_t t_len, uint32_t dom, const uint8_t* b10, const uint8_t* b11, void* t1, void* t2, void* t3) {
        hipLaunchKernelGGL((k_plonk_split_t<C>), dim3((dom + 6 + 255) / 256), dim3(256), 0, ctx().stream, (const uint32_t*)t, t_len, dom, frk(b10), frk(b11), (uint32_t*)t1, (uint32_t*)t2,
                           (uint32_t*)t3);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int convert_multi(const void* const* in, void* const* out, const size_t* n, int count, int op) {
        if (count < 1 || count > 4) return fail(ZKMI_ERR_INVALID, "fr_convert_multi: 1..4 arrays");
        ConvertMultiArgs a = {};
        size_t mx = 0;
        for (int j = 0; j < count; j++) { a.in[j] = (const uint32_t*)in[j]; a.out[j] = (uint32_t*)out[j]; a.n[j] = n[j]; mx = std::max(mx, n[j]); }
        if (!mx) return ZKMI_OK;
        hipLaunchKernelGGL((k_fr_convert_multi<C>), dim3((unsigned)((mx + 255) / 256), (unsigned)count), dim3(256), 0, ctx().stream, a, op);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    // `count` evaluations, ONE wait: partial sums of all of them in one launch, one block per evaluation for the totals, one read-back. Points that repeat share a power table.
    static int evaluate_multi(const void* const* polys, const size_t* lens, const uint8_t* xs, int count, uint8_t* out) {
        Ctx& cx = ctx();
        if (count < 1 || count > EVAL_MAX) return fail(ZKMI_ERR_INVALID, "poly_evaluate_multi: 1..8 evaluations");
        EvalArgs a = {};
        uint32_t tot = 0, np_max = 0;
        size_t n_max = 1;
        for (int q = 0; q < count; q++) n_max = std::max(n_max, lens[q]);
        HE pts[EVAL_MAX];
        int which[EVAL_MAX], n_pts = 0;
        for (int q = 0; q < count; q++) {
            a.c[q] = (const uint32_t*)polys[q]; a.n[q] = lens[q]; a.x[q] = frk(xs + 32 * q);
            a.np[q] = (uint32_t)((lens[q] + 256 * EV_K - 1) / (256 * EV_K)); a.off[q] = tot;
            tot += a.np[q]; np_max = std::max(np_max, a.np[q]);
            const HE x = he(xs + 32 * q);
            which[q] = -1;
            for (int r = 0; r < n_pts && which[q] < 0; r++) if (pts[r] == x) which[q] = r;
            if (which[q] < 0) { which[q] = n_pts; pts[n_pts++] = x; }
        }
        PowTab tabs[EVAL_MAX];
        ZK_TRY((build_pow_tabs_dyn<C>(pts, n_pts, std::max(1u, clog2(n_max)), "plonk.powe", tabs)));       // one launch for the distinct points
        for (int q = 0; q < count; q++) a.xt[q] = tabs[which[q]];
        uint32_t* part;
        ZK_TRY(ws_get("plonk.evpart", ((size_t)tot + EVAL_MAX) * 32, (void**)&part));
        uint32_t* res = part + (size_t)tot * 8;
        if (np_max) hipLaunchKernelGGL((k_poly_eval_partial<C>), dim3(np_max, (unsigned)count), dim3(256), 0, cx.stream, a, part);
        hipLaunchKernelGGL((k_poly_sum<C>), dim3((unsigned)count), dim3(256), 0, cx.stream, a, part, res);
        ZK_HIP(hipMemcpyAsync(out, res, (size_t)count * 32, hipMemcpyDeviceToHost, cx.stream));
        ZK_HIP(hipStreamSynchronize(cx.stream));
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
    static int evaluate(const void* p, size_t n, const uint8_t* x, uint8_t* out) {
        if (!n) { memset(out, 0, 32); return ZKMI_OK; }
        return evaluate_multi(&p, &n, x, 1, out);
    }
    static int div_zh(void* p, size_t len, uint32_t dom, uint32_t ext) {
        Ctx& cx = ctx();
        uint32_t* bad;
        ZK_TRY(ws_get("plonk.bad", 16, (void**)&bad));
        ZK_HIP(hipMemsetAsync(bad, 0, 16, cx.stream));
        hipLaunchKernelGGL((k_poly_div_zh<C>), dim3((dom + 255) / 256), dim3(256), 0, cx.stream, (uint32_t*)p, len, dom, ext, bad);
        uint32_t nbad = 0;
        ZK_HIP(hipMemcpyAsync(&nbad, bad, 4, hipMemcpyDeviceToHost, cx.stream));
        ZK_HIP(hipStreamSynchronize(cx.stream));
        ZK_HIP(hipGetLastError());
        if (nbad) return fail(ZKMI_ERR_INVALID, "Polynomial is not divisible");            // polynomial.js:607-611
        return ZKMI_OK;
    }
    // check = false: enqueue only. The divisibility test of the reference (polynomial.js:665-669) is "the n highest coefficients of the quotient vanish": the caller reads them at its
    // next synchronisation point (zkmi_poly_is_zero_dev on p[len - n, len)) — two proofs in flight: the host must not wait here.
    static int div_by_zerofier(void* p, size_t len, uint32_t n, const uint8_t* beta, bool check = true) {
        Ctx& cx = ctx();
        if (n == 0) return fail(ZKMI_ERR_INVALID, "divByZerofier: n must be positive");
        if (!len) return ZKMI_OK;
        const HFr Fh = F();
        const HE b = he(beta), ib = Fh.inv(b);
        uint32_t* bad;
        const size_t chain = (len + n - 1) / n;                    // longest chain
        if (n >= 64 || chain <= 64) {
            // many short chains: one lane per residue class
            ZK_TRY(ws_get("plonk.bad", 16, (void**)&bad));
            ZK_HIP(hipMemsetAsync(bad, 0, 16, cx.stream));
            hipLaunchKernelGGL((k_dz_chain<C>), dim3((n + 255) / 256), dim3(256), 0, cx.stream, (uint32_t*)p, len, n, frk(ib), bad);
            ZK_HIP(hipGetLastError());
            if (!check) return ZKMI_OK;
            uint32_t nbad = 0;
            ZK_HIP(hipMemcpyAsync(&nbad, bad, 4, hipMemcpyDeviceToHost, cx.stream));
            ZK_HIP(hipStreamSynchronize(cx.stream));
            if (nbad) return fail(ZKMI_ERR_INVALID, "Polynomial is not divisible");
            return ZKMI_OK;
        }
        // few long chains: each residue class r is the linear recurrence q_k = (q_{k-1} - c_k)/beta, solved as a prefix sum
        const HE both[2] = {b, ib};
        PowTab tabs[2];
        ZK_TRY((build_pow_tabs_dyn<C>(both, 2, std::max(1u, clog2(chain)), "plonk.powb", tabs)));          // beta^k and beta^-k in one launch
        const PowTab bt = tabs[0], it = tabs[1];
        uint32_t* u;
        ZK_TRY(ws_get("plonk.dzu", chain * 32, (void**)&u));
        for (uint32_t r = 0; r < n && r < len; r++) {
            const size_t m = (len - r + n - 1) / n;
            const unsigned blocks = (unsigned)((m + 255) / 256);
            hipLaunchKernelGGL((k_dz_weight<C>), dim3(blocks), dim3(256), 0, cx.stream, (const uint32_t*)p, m, (size_t)r, (size_t)n, bt, frk(ib), u);
            ZK_TRY((scan_inclusive<C, false>(u, m, u)));
            hipLaunchKernelGGL((k_dz_unweight<C>), dim3(blocks), dim3(256), 0, cx.stream, u, m, (size_t)r, (size_t)n, it, (uint32_t*)p);
        }
        ZK_HIP(hipGetLastError());
        if (!check) return ZKMI_OK;
        // the n highest coefficients must vanish (polynomial.js:665-669)
        const size_t tail = std::min<size_t>(n, len);
        std::vector<HE> last(tail);
        ZK_HIP(hipMemcpyAsync(last.data(), (uint8_t*)p + (len - tail) * 32, tail * 32, hipMemcpyDeviceToHost, cx.stream));
        ZK_HIP(hipStreamSynchronize(cx.stream));
        for (const HE& e : last) if (!e.is_zero()) return fail(ZKMI_ERR_INVALID, "Polynomial is not divisible");
        return ZKMI_OK;
    }
    static int interleave(const void* const* polys, const size_t* lens, int n, void* out, size_t out_len) {
        if (n < 1 || n > 16) return fail(ZKMI_ERR_UNSUPPORTED, "CPolynomial: 1..16 component polynomials");
        InterleaveArgs a;
        a.n = (uint32_t)n;
        for (int j = 0; j < 16; j++) { a.p[j] = j < n ? (const uint32_t*)polys[j] : nullptr; a.len[j] = j < n ? (uint32_t)lens[j] : 0; }
        if (out_len) hipLaunchKernelGGL((k_cpoly_interleave<C>), dim3((unsigned)((out_len + 255) / 256)), dim3(256), 0, ctx().stream, a, (uint32_t*)out, out_len);
        ZK_HIP(hipGetLastError());
        return ZKMI_OK;
    }
};

int fr_convert_multi_dispatch(int curve, int op, const void* const* d_in, void* const* d_out, const size_t* ns, int count) {
    if (op != ZKMI_BATCH_TO_MONTGOMERY && op != ZKMI_BATCH_FROM_MONTGOMERY) return fail(ZKMI_ERR_UNSUPPORTED, "fr_batch_multi: the two Montgomery conversions only");
    const int o = op == ZKMI_BATCH_TO_MONTGOMERY ? 0 : 1;
    if (curve == ZKMI_CURVE_BN128) return PlonkOps<Bn254Fr>::convert_multi(d_in, d_out, ns, count, o);
    if (curve == ZKMI_CURVE_BLS12381) return PlonkOps<Bls12381Fr>::convert_multi(d_in, d_out, ns, count, o);
    return fail(ZKMI_ERR_INVALID, "unknown curve");
}

}  // namespace zkmi

using namespace zkmi;
#define PLONK_DISPATCH(curve, call)                                                     \
    do {                                                                                \
        ZK_TRY(require_ctx());                                                          \
        if ((curve) == ZKMI_CURVE_BN128) return PlonkOps<Bn254Fr>::call;                \
        if ((curve) == ZKMI_CURVE_BLS12381) return PlonkOps<Bls12381Fr>::call;          \
        return fail(ZKMI_ERR_INVALID, "unknown curve");                                 \
    } while (0)

// ---- Keccak-256 (host): the Fiat-Shamir transcript of src/Keccak256Transcript.js (@noble/hashes keccak_256: original 0x01 padding,
// rate 136) — a few hundred bytes per challenge, so it stays on the host like in the reference
static void keccak_f1600(uint64_t st[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
                                    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
                                    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                                    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            const uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63));
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            const int j = PIL[i];
            const uint64_t b = st[j];
            st[j] = (t << ROT[i]) | (t >> (64 - ROT[i]));
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}

extern "C" {

int zkmi_keccak256(const uint8_t* data, size_t len, uint8_t* out32) {
    if ((!data && len) || !out32) return fail(ZKMI_ERR_INVALID, "keccak256: null argument");
    constexpr size_t RATE = 136;
    uint64_t st[25] = {0};
    uint8_t block[RATE];
    size_t off = 0;
    for (;;) {
        const size_t take = std::min(RATE, len - off);
        const bool last = take < RATE;
        memset(block, 0, RATE);
        if (take) memcpy(block, data + off, take);
        if (last) { block[take] ^= 0x01; block[RATE - 1] ^= 0x80; }
        for (size_t i = 0; i < RATE / 8; i++) { uint64_t v; memcpy(&v, block + 8 * i, 8); st[i] ^= v; }
        keccak_f1600(st);
        off += take;
        if (last) break;
    }
    memcpy(out32, st, 32);
    return ZKMI_OK;
}

int zkmi_plonk_gather_wires_dev(int curve, const void* d_witness, uint32_t n_witness, const void* d_internal, uint32_t n_additions, const void* d_map_a, const void* d_map_b,
                                const void* d_map_c, uint32_t n_constraints, uint32_t domain, void* d_a, void* d_b, void* d_c) {
    PLONK_DISPATCH(curve, gather(d_witness, n_witness, d_internal, n_additions, d_map_a, d_map_b, d_map_c, n_constraints, domain, d_a, d_b, d_c));
}
int zkmi_plonk_additions_dev(int curve, const void* d_additions, uint32_t n_additions, const void* d_witness, uint32_t n_witness, void* d_internal) {
    if (n_additions && (!d_additions || !d_witness || !d_internal)) return fail(ZKMI_ERR_INVALID, "zkmi_plonk_additions_dev: null buffer");
    if ((uint64_t)n_additions + n_witness > 0xffffffffull) return fail(ZKMI_ERR_INVALID, "zkmi_plonk_additions_dev: more than 2^32 signals");
    PLONK_DISPATCH(curve, additions(d_additions, n_additions, d_witness, n_witness, d_internal));
}
int zkmi_plonk_compute_z_dev(int curve, const void* d_a, const void* d_b, const void* d_c, const void* d_s1e, const void* d_s2e, const void* d_s3e, uint32_t domain, const uint8_t* beta,
                             const uint8_t* gamma, const uint8_t* k1, const uint8_t* k2, const uint8_t* w_n, void* d_z) {
    PLONK_DISPATCH(curve, compute_z(d_a, d_b, d_c, d_s1e, d_s2e, d_s3e, domain, beta, gamma, k1, k2, w_n, d_z));
}
int zkmi_plonk_compute_z_enqueue(int curve, const void* d_a, const void* d_b, const void* d_c, const void* d_s1e, const void* d_s2e, const void* d_s3e, uint32_t domain, const uint8_t* beta,
                                 const uint8_t* gamma, const uint8_t* k1, const uint8_t* k2, const uint8_t* w_n, void* d_z) {
    PLONK_DISPATCH(curve, compute_z(d_a, d_b, d_c, d_s1e, d_s2e, d_s3e, domain, beta, gamma, k1, k2, w_n, d_z, false));
}
int zkmi_plonk_compute_t_dev(int curve, const zkmi_plonk_evals* ev, uint32_t domain, uint32_t n_public, const uint8_t* blind11, const uint8_t* beta, const uint8_t* gamma,
                             const uint8_t* alpha, const uint8_t* k1, const uint8_t* k2, const uint8_t* w_n, const uint8_t* w_4n, const uint8_t* w_2, void* d_t, void* d_tz) {
    if (!ev) return fail(ZKMI_ERR_INVALID, "null evaluations");
    PLONK_DISPATCH(curve, compute_t(ev, domain, n_public, blind11, beta, gamma, alpha, k1, k2, w_n, w_4n, w_2, d_t, d_tz));
}
int zkmi_poly_axpy_dev(int curve, void* d_y, const void* d_x, size_t nx, const uint8_t* k, int subtract) { PLONK_DISPATCH(curve, axpy(d_y, d_x, nx, k, subtract)); }
int zkmi_poly_scale_dev(int curve, void* d_p, size_t n, const uint8_t* k) { PLONK_DISPATCH(curve, scale(d_p, n, k)); }
int zkmi_poly_blind_dev(int curve, void* d_p, size_t n, const uint8_t* factors, int count) { PLONK_DISPATCH(curve, blind(d_p, n, factors, count, 1)); }
int zkmi_poly_blind_tail_dev(int curve, void* d_p, size_t n, const uint8_t* factors, int count) { PLONK_DISPATCH(curve, blind(d_p, n, factors, count, 2)); }
int zkmi_poly_lincomb_dev(int curve, void* d_out, size_t out_len, const zkmi_poly_term* terms, int count, const uint8_t* constant) {
    if (!d_out || (count && !terms)) return fail(ZKMI_ERR_INVALID, "null argument");
    PLONK_DISPATCH(curve, lincomb(d_out, out_len, terms, count, constant));
}
int zkmi_poly_evaluate_multi_dev(int curve, const void* const* d_polys, const size_t* lens, const uint8_t* xs, int count, uint8_t* out) {
    if (!d_polys || !lens || !xs || !out) return fail(ZKMI_ERR_INVALID, "null argument");
    PLONK_DISPATCH(curve, evaluate_multi(d_polys, lens, xs, count, out));
}
int zkmi_plonk_split_t_dev(int curve, const void* d_t, size_t t_len, uint32_t domain, const uint8_t* b10, const uint8_t* b11, void* d_t1, void* d_t2, void* d_t3) {
    if (!d_t || !b10 || !b11 || !d_t1 || !d_t2 || !d_t3) return fail(ZKMI_ERR_INVALID, "null argument");
    PLONK_DISPATCH(curve, split_t(d_t, t_len, domain, b10, b11, d_t1, d_t2, d_t3));
}
int zkmi_fr_batch_multi_dev(int curve, int op, const void* const* d_in, void* const* d_out, const size_t* ns, int count) {
    if (!d_in || !d_out || !ns) return fail(ZKMI_ERR_INVALID, "null argument");
    ZK_TRY(require_ctx());
    return fr_convert_multi_dispatch(curve, op, d_in, d_out, ns, count);
}
int zkmi_plonk_gather_wires_mont_dev(int curve, const void* d_witness, uint32_t n_witness, const void* d_internal, uint32_t n_additions, const void* d_map_a, const void* d_map_b,
                                     const void* d_map_c, uint32_t n_constraints, uint32_t domain, void* d_a, void* d_b, void* d_c) {
    PLONK_DISPATCH(curve, gather(d_witness, n_witness, d_internal, n_additions, d_map_a, d_map_b, d_map_c, n_constraints, domain, d_a, d_b, d_c, 1));
}
int zkmi_poly_add_scalar_dev(int curve, void* d_p, const uint8_t* value) { PLONK_DISPATCH(curve, blind(d_p, 0, value, 1, 0)); }
int zkmi_poly_evaluate_dev(int curve, const void* d_p, size_t n, const uint8_t* x, uint8_t* out) { PLONK_DISPATCH(curve, evaluate(d_p, n, x, out)); }
int zkmi_poly_is_zero_dev(int curve, const void* d_p, size_t n, int* all_zero) {
    ZK_TRY(require_ctx());
    (void)curve;
    if (!all_zero) return fail(ZKMI_ERR_INVALID, "null argument");
    *all_zero = 1;
    if (!n) return ZKMI_OK;
    uint32_t* flag;
    ZK_TRY(ws_get("plonk.bad", 16, (void**)&flag));
    ZK_HIP(hipMemsetAsync(flag, 0, 16, ctx().stream));
    hipLaunchKernelGGL(k_poly_any_nonzero, dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, ctx().stream, (const uint32_t*)d_p, n * 8, flag);
    uint32_t f = 0;
    ZK_HIP(hipMemcpyAsync(&f, flag, 4, hipMemcpyDeviceToHost, ctx().stream));
    ZK_HIP(hipStreamSynchronize(ctx().stream));
    *all_zero = f ? 0 : 1;
    return ZKMI_OK;
}
int zkmi_poly_degree_dev(int curve, const void* d_p, size_t n, size_t* degree) {
    ZK_TRY(require_ctx());
    (void)curve;
    if (!degree) return fail(ZKMI_ERR_INVALID, "null argument");
    *degree = 0;
    if (!n) return ZKMI_OK;
    unsigned long long* d;
    ZK_TRY(ws_get("plonk.bad", 16, (void**)&d));
    ZK_HIP(hipMemsetAsync(d, 0, 16, ctx().stream));
    hipLaunchKernelGGL(k_poly_degree, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx().stream, (const uint32_t*)d_p, n, d);
    unsigned long long h = 0;
    ZK_HIP(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx().stream));
    ZK_HIP(hipStreamSynchronize(ctx().stream));
    *degree = (size_t)h;
    return ZKMI_OK;
}
int zkmi_fflonk_t0_dev(int curve, const zkmi_plonk_evals* ev, uint32_t domain, uint32_t n_public, void* d_t0) {
    if (!ev) return fail(ZKMI_ERR_INVALID, "null evaluations");
    PLONK_DISPATCH(curve, fflonk_t0(ev, domain, n_public, d_t0));
}
int zkmi_fflonk_t1_dev(int curve, const void* d_z4, const void* d_lagrange, uint32_t domain, const uint8_t* b789, const uint8_t* w_2n, void* d_t1, void* d_t1z) {
    PLONK_DISPATCH(curve, fflonk_t1(d_z4, d_lagrange, domain, b789, w_2n, d_t1, d_t1z));
}
int zkmi_fflonk_t2_dev(int curve, const zkmi_plonk_evals* ev, uint32_t domain, const uint8_t* b789, const uint8_t* beta, const uint8_t* gamma, const uint8_t* k1, const uint8_t* k2,
                       const uint8_t* w_n, const uint8_t* w_4n, void* d_t2, void* d_t2z) {
    if (!ev) return fail(ZKMI_ERR_INVALID, "null evaluations");
    PLONK_DISPATCH(curve, fflonk_t2(ev, domain, b789, beta, gamma, k1, k2, w_n, w_4n, d_t2, d_t2z));
}
int zkmi_poly_div_zh_dev(int curve, void* d_p, size_t len, uint32_t domain, uint32_t extensions) { PLONK_DISPATCH(curve, div_zh(d_p, len, domain, extensions)); }
int zkmi_poly_div_by_zerofier_dev(int curve, void* d_p, size_t len, uint32_t n, const uint8_t* beta) { PLONK_DISPATCH(curve, div_by_zerofier(d_p, len, n, beta)); }
int zkmi_poly_div_by_zerofier_enqueue(int curve, void* d_p, size_t len, uint32_t n, const uint8_t* beta) { PLONK_DISPATCH(curve, div_by_zerofier(d_p, len, n, beta, false)); }
int zkmi_cpoly_interleave_dev(int curve, const void* const* d_polys, const size_t* lens, int n, void* d_out, size_t out_len) {
    if (!d_polys || !lens) return fail(ZKMI_ERR_INVALID, "null argument");
    PLONK_DISPATCH(curve, interleave(d_polys, lens, n, d_out, out_len));
}

}  // extern "C"
