// snarkjs_amd/csrc/ntt.cuh — radix-2 NTT / iNTT over Fr for gfx950, multi-pass with LDS-resident tiles.
//
// Replaces ffjavascript's engine_fft (host bit-reversal + <=2^14-point fftMix blocks + log-many full-array fftJoin
// round trips through postMessage, build/snarkjs.min.js:1@215859/@216834) and wasmcurves' frm_fftMix/_fftJoin/_fftFinal
// (@103755).  Same function:  X[k] = sum_j x[j] w^(jk),  w = Fr.w[log n],  natural order in and out, inverse scaled
// by 1/n — but computed in p = ceil(log n / 8) passes over HBM; each pass is a 2^l-point decimation-in-frequency
// transform of 512-element tiles held in LDS (Stockham-style: the digit transposition that makes the output
// natural-order is folded into the last pass, there is no separate bit-reversal pass).
//
// Index algebra (n = N1·N2·…·Np, Ni = 2^li):  j = sum_i j_i·S_i with S_i = prod_{m>i} N_m (j_1 most significant),
// k = sum_i k_i·P_i with P_i = prod_{m<i} N_m (k_1 least significant).  Then
//   w^(jk) = prod_i  w_{Ni}^(j_i k_i) · w^(j_i·S_i·K_{i-1}),   K_{i-1} = sum_{m<i} k_m P_m,
// i.e. pass i multiplies each input by the "row factor" w^(j_i S_i K_{i-1}) and does a length-Ni DFT over j_i.
// Passes 1..p-1 work in place on tiles {all j_i} x {CH consecutive lower indices} (coalesced CH·32-byte segments);
// pass p reads contiguous j_p runs for CH consecutive k_1 and writes natural-order positions K_{p-1} + P_p·k_p
// (again CH·32-byte segments).  Optional fused pre-scale x[i]·first·inc^i (Fr.batchApplyKey, @211529) rides on the
// per-pass row factors.
#pragma once
#include "field.cuh"

namespace zkmi {

constexpr int NTT_MAX_PASSES = 4;
#ifndef ZKMI_NTT_TILE_LOG
#define ZKMI_NTT_TILE_LOG 9
#endif
// 512 elements = 16 KiB of LDS per workgroup -> 8 workgroups (8 waves per SIMD) per CU: the butterflies are latency-bound carry
// chains, occupancy matters more than tile size (2^20 transform: 0.150 ms; 1024-element tiles 0.163 ms; 2048-element tiles 0.180 ms;
// A/B on one box, PLONK's 2^22 transforms unchanged)
constexpr int NTT_TILE_LOG = ZKMI_NTT_TILE_LOG;
constexpr int NTT_THREADS = 256;

struct NttPassArgs {
    uint32_t log_n;
    uint32_t n_pass, pass;                // pass is 0-based
    uint32_t l[NTT_MAX_PASSES];           // digit widths, l[0] = most significant input digit
    uint32_t log_ch;                      // log2 columns per tile
    uint32_t log_lb;                      // split of the power tables: w^e = T_lo[e & (2^lb-1)] * T_hi[e >> lb]
    const uint32_t* T_lo;
    const uint32_t* T_hi;                 // for the last pass of an inverse transform this table carries the 1/n factor
    const uint32_t* LT;                   // local twiddles w_{Ni}^k, k < Ni/2
    const uint32_t* rowinc;               // per-pass pre-scale factors (Ni entries) or nullptr
    const uint32_t* scale;                // single constant applied on load (1/n for single-pass inverse) or nullptr
    uint64_t in_len;                      // first pass only: elements the caller's input holds (0 = all 2^log_n); the rest reads as zero (Evaluations.fromPolynomial's zero padding)
    uint64_t in_bs, out_bs;               // batched launches (gridDim.y transforms of the same plan): words between the inputs / outputs of consecutive members
};

template <class C> ZK_DEV Fp<C> ntt_pow(const NttPassArgs& a, uint64_t e) {
    Fp<C> lo = fp_load<C>(a.T_lo + (size_t)(e & ((1ull << a.log_lb) - 1)) * C::N);
    Fp<C> hi = fp_load<C>(a.T_hi + (size_t)(e >> a.log_lb) * C::N);
    return fp_mul(lo, hi);
}

// LDS element storage: two 16-byte planes so that consecutive elements are consecutive 16-byte slots
// (ds_read_b128/ds_write_b128 of 16 consecutive lanes then cover all 64 banks exactly once).
template <class C> ZK_DEV Fp<C> lds_get(const uint4* p0, const uint4* p1, uint32_t e) {
    static_assert(C::N == 8, "Fr has 8 limbs");
    uint4 a = p0[e], b = p1[e];
    Fp<C> r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
template <class C> ZK_DEV void lds_put(uint4* p0, uint4* p1, uint32_t e, const Fp<C>& v) {
    p0[e] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p1[e] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// K_{i-1} from the memory-order flattened index u of (k_1..k_{i-1}) (k_1 most significant in u, least in K)
ZK_DEV uint64_t ntt_digit_reverse(const NttPassArgs& a, uint64_t u, uint32_t first_digit, uint32_t n_digits) {
    uint64_t K = 0;
    uint32_t shift_out = 0, rem_bits = 0;
    for (uint32_t m = 0; m < n_digits; m++) rem_bits += a.l[first_digit + m];
    for (uint32_t m = 0; m < n_digits; m++) {
        uint32_t w = a.l[first_digit + m];
        rem_bits -= w;
        uint64_t d = (u >> rem_bits) & ((1ull << w) - 1);
        K |= d << shift_out;
        shift_out += w;
    }
    return K;
}

// radix-2 DIF stages on an LDS tile. ROWMAJOR: element (row j, col c) at j*CH + c (strided passes);
// otherwise at c*(N+1) + j (last pass; +1 pad keeps the transposed store conflict-free).
template <class C, bool ROWMAJOR> ZK_DEV void ntt_tile_stages(uint4* p0, uint4* p1, const uint4* lt0, const uint4* lt1, uint32_t l, uint32_t log_ch) {
    const uint32_t half_elems = 1u << (l + log_ch - 1);
    const uint32_t N = 1u << l;
    for (int s = (int)l - 1; s >= 0; s--) {
        const uint32_t h = 1u << s;
        for (uint32_t b = threadIdx.x; b < half_elems; b += NTT_THREADS) {
            uint32_t c, pr;
            if (ROWMAJOR) { c = b & ((1u << log_ch) - 1); pr = b >> log_ch; }
            else { pr = b & ((N >> 1) - 1); c = b >> (l - 1); }
            uint32_t jl = pr & (h - 1);
            uint32_t j = ((pr >> s) << (s + 1)) | jl;
            uint32_t e0 = ROWMAJOR ? (j << log_ch) + c : c * (N + 1) + j;
            uint32_t e1 = ROWMAJOR ? ((j + h) << log_ch) + c : c * (N + 1) + j + h;
            Fp<C> x = lds_get<C>(p0, p1, e0), y = lds_get<C>(p0, p1, e1);
            Fp<C> sum = fp_add(x, y), diff = fp_sub(x, y);
            if (s > 0) diff = fp_mul(diff, lds_get<C>(lt0, lt1, jl << (l - 1 - s)));
            lds_put<C>(p0, p1, e0, sum);
            lds_put<C>(p0, p1, e1, diff);
        }
        __syncthreads();
    }
}

// ---- passes 1 .. p-1 (in place over the FFT digit, columns contiguous in memory) ------------------------------------
template <class C> __global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass_strided(const uint32_t* in, uint32_t* out, NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    in += (size_t)blockIdx.y * a.in_bs; out += (size_t)blockIdx.y * a.out_bs;
    const uint32_t l = a.l[a.pass], N = 1u << l, CH = 1u << a.log_ch, E = N << a.log_ch;
    uint4* p0 = lds4;                 // E
    uint4* p1 = p0 + E;               // E
    uint4* lt0 = p1 + E;              // N/2 local twiddles
    uint4* lt1 = lt0 + (N >> 1);
    uint4* rf0 = lt1 + (N >> 1);      // N row factors
    uint4* rf1 = rf0 + N;
    uint32_t log_S = 0;
    for (uint32_t m = a.pass + 1; m < a.n_pass; m++) log_S += a.l[m];
    const uint64_t tiles_per_u = (1ull << log_S) >> a.log_ch;
    const uint64_t u = blockIdx.x / tiles_per_u, q0 = (blockIdx.x % tiles_per_u) << a.log_ch;
    const uint64_t base = (u << (l + log_S)) + q0;
    const bool has_fac = (a.pass > 0) || (a.rowinc != nullptr);
    if (has_fac) {
        const uint64_t K = a.pass > 0 ? ntt_digit_reverse(a, u, 0, a.pass) : 0;
        for (uint32_t j = threadIdx.x; j < N; j += NTT_THREADS) {
            Fp<C> f;
            if (a.pass > 0) {
                uint64_t e = (((uint64_t)j * K) << log_S) & ((1ull << a.log_n) - 1);
                f = ntt_pow<C>(a, e);
                if (a.rowinc) f = fp_mul(f, fp_load<C>(a.rowinc + (size_t)j * C::N));
            } else f = fp_load<C>(a.rowinc + (size_t)j * C::N);
            lds_put<C>(rf0, rf1, j, f);
        }
    }
    for (uint32_t k = threadIdx.x; k < (N >> 1); k += NTT_THREADS) lds_put<C>(lt0, lt1, k, fp_load<C>(a.LT + (size_t)k * C::N));
    __syncthreads();
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        uint32_t c = idx & (CH - 1), j = idx >> a.log_ch;
        const uint64_t src = base + ((uint64_t)j << log_S) + c;
        Fp<C> x = (a.pass == 0 && a.in_len && src >= a.in_len) ? fp_zero<C>() : fp_load<C>(in + src * C::N);
        if (has_fac) x = fp_mul(x, lds_get<C>(rf0, rf1, j));
        lds_put<C>(p0, p1, idx, x);
    }
    __syncthreads();
    ntt_tile_stages<C, true>(p0, p1, lt0, lt1, l, a.log_ch);
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        uint32_t c = idx & (CH - 1), k = idx >> a.log_ch;
        uint32_t j = __brev(k) >> (32 - l);
        fp_store<C>(out + (base + ((uint64_t)k << log_S) + c) * C::N, lds_get<C>(p0, p1, (j << a.log_ch) + c));
    }
}

// ---- last pass: contiguous j_p runs in, natural order out -------------------------------------------------------------
template <class C> __global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass_last(const uint32_t* in, uint32_t* out, NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
    in += (size_t)blockIdx.y * a.in_bs; out += (size_t)blockIdx.y * a.out_bs;
    const uint32_t l = a.l[a.pass], N = 1u << l, CH = 1u << a.log_ch, E = N << a.log_ch, PL = (N + 1) << a.log_ch;
    uint4* p0 = lds4;                 // (N+1)*CH
    uint4* p1 = p0 + PL;
    uint4* lt0 = p1 + PL;
    uint4* lt1 = lt0 + (N >> 1);
    const bool multi = a.n_pass > 1;
    const uint32_t l1 = a.l[0];
    uint64_t r = 0, c0 = 0, Krest = 0;
    uint32_t log_S1 = 0;
    if (multi) {
        const uint64_t tiles_per_r = (1ull << l1) >> a.log_ch;
        r = blockIdx.x / tiles_per_r;
        c0 = (blockIdx.x % tiles_per_r) << a.log_ch;
        Krest = ntt_digit_reverse(a, r, 1, a.n_pass - 2);
        log_S1 = a.log_n - l1;
    }
    for (uint32_t k = threadIdx.x; k < (N >> 1); k += NTT_THREADS) lds_put<C>(lt0, lt1, k, fp_load<C>(a.LT + (size_t)k * C::N));
    Fp<C> sc;
    if (a.scale) sc = fp_load<C>(a.scale);
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        uint32_t j = idx & (N - 1), c = idx >> l;
        uint64_t addr = multi ? (((c0 + c) << log_S1) + (r << l) + j) : j;
        Fp<C> x = (!multi && a.in_len && addr >= a.in_len) ? fp_zero<C>() : fp_load<C>(in + addr * C::N);
        if (multi) {
            uint64_t K = (c0 + c) + (Krest << l1);
            x = fp_mul(x, ntt_pow<C>(a, (uint64_t)j * K));
        }
        if (a.rowinc) x = fp_mul(x, fp_load<C>(a.rowinc + (size_t)j * C::N));
        if (a.scale) x = fp_mul(x, sc);
        lds_put<C>(p0, p1, c * (N + 1) + j, x);
    }
    __syncthreads();
    ntt_tile_stages<C, false>(p0, p1, lt0, lt1, l, a.log_ch);
    for (uint32_t idx = threadIdx.x; idx < E; idx += NTT_THREADS) {
        uint32_t c = idx & (CH - 1), k = idx >> a.log_ch;
        uint32_t j = __brev(k) >> (32 - l);
        uint64_t K = (c0 + c) + (Krest << l1);
        uint64_t addr = multi ? (K + ((uint64_t)k << (a.log_n - l))) : k;
        fp_store<C>(out + addr * C::N, lds_get<C>(p0, p1, c * (N + 1) + j));
    }
}

// ---- element-wise batch kernels ------------------------------------------------------------------------------------------
// op 0: x*R (toMontgomery), 1: x*R^-1 (fromMontgomery)
template <class C> __global__ void k_fr_convert(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n, int op) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<C> x = fp_load<C>(in + i * C::N);
    fp_store<C>(out + i * C::N, op == 0 ? fp_to_mont(x) : fp_from_mont(x));
}
// out[i] = from_mont(a[i]*b[i] - c[i])   (joinABC, src/groth16_prove.js:320-374)
template <class C> __global__ void k_join_abc(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, const uint32_t* __restrict__ c,
                                             uint32_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp<C> t = fp_sub(fp_mul(fp_load<C>(a + i * C::N), fp_load<C>(b + i * C::N)), fp_load<C>(c + i * C::N));
    fp_store<C>(out + i * C::N, fp_from_mont(t));
}
// out[i] = in[i]*first*inc^i. Each block owns blockDim*PER consecutive elements; lane t handles i0+t, i0+t+T, ...
// stepping its factor by inc^T (host supplies inc^T in `step`).
template <class C, int PER> __global__ void k_apply_key(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n,
                                                       const uint32_t* __restrict__ first, const uint32_t* __restrict__ inc, const uint32_t* __restrict__ step) {
    size_t i0 = (size_t)blockIdx.x * blockDim.x * PER + threadIdx.x;
    if (i0 >= n) return;
    Fp<C> g = fp_load<C>(inc), f = fp_load<C>(first);
    // f *= inc^i0 (square-and-multiply over the bits of i0)
    Fp<C> base = g;
    for (size_t e = i0; e; e >>= 1) { if (e & 1) f = fp_mul(f, base); base = fp_sqr(base); }
    Fp<C> st = fp_load<C>(step);
#pragma unroll 1
    for (int m = 0; m < PER; m++) {
        size_t i = i0 + (size_t)m * blockDim.x;
        if (i >= n) break;
        fp_store<C>(out + i * C::N, fp_mul(fp_load<C>(in + i * C::N), f));
        f = fp_mul(f, st);
    }
}
// element-wise inverse with Montgomery's trick over CHUNK consecutive elements per lane; 0 -> 0
template <class C, int CHUNK> __global__ void __launch_bounds__(128)
k_batch_inverse(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * CHUNK;
    if (i0 >= n) return;
    const int cnt = (int)((n - i0) < (size_t)CHUNK ? (n - i0) : (size_t)CHUNK);
    // forward: out[i] = product of the non-zero elements before i
    Fp<C> acc = fp_one<C>();
    for (int k = 0; k < cnt; k++) {
        Fp<C> x = fp_load<C>(in + (i0 + k) * C::N);
        fp_store<C>(out + (i0 + k) * C::N, acc);
        if (!fp_is_zero(x)) acc = fp_mul(acc, x);
    }
    Fp<C> inv = fp_inv(acc);
    for (int k = cnt - 1; k >= 0; k--) {
        Fp<C> x = fp_load<C>(in + (i0 + k) * C::N);
        Fp<C> pre = fp_load<C>(out + (i0 + k) * C::N);
        if (fp_is_zero(x)) { fp_store<C>(out + (i0 + k) * C::N, x); continue; }
        fp_store<C>(out + (i0 + k) * C::N, fp_mul(inv, pre));
        inv = fp_mul(inv, x);
    }
}

}  // namespace zkmi
