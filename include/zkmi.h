/*
 * include/zkmi.h — C-ABI of the MI355X proving backend (libzkmi.so).
 *
 * This is the drop-in boundary for the snarkjs prover hot path: one entry point per bulk operation
 * that snarkjs issues through ffjavascript's curve object.  Plain pointers and sizes only; the N-API addon
 * (snarkjs_amd/napi/zkmi_napi.c) and the Python ctypes mirror (snarkjs_amd/__init__.py) are thin wrappers.
 * Each declaration cites the reference interface it replaces; `min.js:1@<col>` = column in line 1 of
 * /root/reference/build/snarkjs.min.js (ffjavascript 0.3.1 / wasmcurves 0.2.2 exist in the reference tree only there).
 *
 * Conventions (identical to the reference's in-memory formats):
 *   - field elements: little-endian, n8 bytes (n8r = 32; n8q = 32 BN254 / 48 BLS12-381), Montgomery form unless
 *     stated; scalars handed to the MSM are plain little-endian integers of `scalar_bytes` bytes, NOT reduced mod r.
 *   - affine points (x,y), all-zero bytes = point at infinity; G2 coordinates are (c0,c1) pairs.
 *   - results of MSMs are Jacobian (X,Y,Z), any representative of the group element (snarkjs normalises with
 *     toAffine before serialising); the zero point is returned as all-zero bytes.
 *   - every function returns 0 on success, non-zero on error; zkmi_last_error() describes the last failure on the
 *     calling thread.  Inputs are never modified.  There is NO CPU fallback: without a HIP device every compute
 *     entry point fails with ZKMI_ERR_NO_DEVICE.
 *   - "pages": a logical buffer given as an array of (pointer,length) host segments, because ffjavascript's
 *     BigBuffer (min.js:1@183423) keeps > 1 GiB buffers as a list of <= 1 GiB Uint8Arrays.
 */
#ifndef ZKMI_H
#define ZKMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKMI_CURVE_BN128 0      /* BN254; src/curves.js:9-53 name "bn128" */
#define ZKMI_CURVE_BLS12381 1

#define ZKMI_OK 0
#define ZKMI_ERR_NO_DEVICE 1
#define ZKMI_ERR_INVALID 2
#define ZKMI_ERR_HIP 3
#define ZKMI_ERR_UNSUPPORTED 4

/* Fr batch operations: zkmi_fr_batch(op) */
#define ZKMI_BATCH_TO_MONTGOMERY 0    /* Fr.batchToMontgomery   (helper before min.js:1@185893; src/plonk_prove.js:278) */
#define ZKMI_BATCH_FROM_MONTGOMERY 1  /* Fr.batchFromMontgomery (src/polynomial/polynomial.js:973)                     */
#define ZKMI_BATCH_INVERSE 2          /* Fr.batchInverse        (min.js:1@188677; src/plonk_prove.js:420), 0 -> 0       */

typedef struct zkmi_pages {
    const uint8_t* const* ptr;   /* n_pages segment pointers */
    const size_t* len;           /* n_pages segment lengths in bytes */
    int n_pages;
} zkmi_pages;

/* ---- library / device ------------------------------------------------------------------------------------------- */
/* Bind the calling process to HIP device `device` (>= 0) and create the library context. Idempotent. */
int zkmi_init(int device);
int zkmi_device_count(void);
const char* zkmi_last_error(void);
const char* zkmi_version(void);
/* Use an externally owned hipStream_t for all subsequent launches (NULL = the library's own stream). */
int zkmi_set_stream(void* hip_stream);
/* Block until all work queued by the library has finished. */
int zkmi_synchronize(void);

/* Device memory owned by the library's pool — lets a host without its own HIP binding (the Node addon) keep
 * data resident between calls. */
int zkmi_dev_alloc(size_t bytes, void** d_ptr);
int zkmi_dev_free(void* d_ptr);
int zkmi_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int zkmi_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);

/* ---- G.multiExpAffine ------------------------------------------------------------------------------------------- */
/* curve.G1.multiExpAffine / curve.G2.multiExpAffine (min.js:1@214996 -> @214651 -> _multiExpChunk @213360; kernel
 * g1m_/g2m_multiexpAffine_chunk @75966).  group = 1 | 2.  n bases of 2*group*n8q bytes, n scalars of scalar_bytes.
 * base_cache_key != 0: the base table is uploaded once and kept resident under that key (zkey sections are static
 * per circuit, src/groth16_prove.js:84-100); release with zkmi_release_bases.
 * out_jacobian: 3*group*n8q bytes. */
int zkmi_msm(int curve, int group, zkmi_pages bases, zkmi_pages scalars, size_t n, size_t scalar_bytes,
             uint64_t base_cache_key, uint8_t* out_jacobian);
int zkmi_release_bases(uint64_t base_cache_key);
/* Same with bases and scalars already resident in device memory (bench.py, fused pipelines). */
int zkmi_msm_dev(int curve, int group, const void* d_bases, const void* d_scalars, size_t n, size_t scalar_bytes,
                 uint8_t* out_jacobian);
/* Device time (ms, HIP events on the library stream) of the bucket-accumulation kernel of the last MSM that used job
 * slot `slot`: zkmi_msm / zkmi_msm_dev use slot 0; zkmi_groth16_prove uses 0..4 = A, B1, B2, C, H. -1 if never run. */
double zkmi_msm_accum_ms(int slot);
/* Window width used for n terms (tuning knob; 0 restores the built-in table). */
int zkmi_msm_set_window_bits(int c);

/* ---- Fr.fft / Fr.ifft ------------------------------------------------------------------------------------------- */
/* curve.Fr.fft / curve.Fr.ifft (min.js:1@215859; kernels frm_fftMix/_fftJoin/_fftFinal @103755):
 *   X[k] = sum_j x[j] w^(jk), w = Fr.w[log_n], natural order in and out; inverse includes the 1/n scaling.
 * n = 2^log_n elements of 32 bytes; log_n must be <= Fr.s (28 BN254, 32 BLS12-381) — the reference's extra
 * n = 2^(s+1) coset case (@216148) is rejected with ZKMI_ERR_UNSUPPORTED.
 * Optional fused pre-scale x[i] <- x[i]*first*inc^i (= Fr.batchApplyKey in front, src/groth16_prove.js:66-76):
 * pass prescale_first = prescale_inc = NULL for a plain transform. */
int zkmi_ntt(int curve, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, unsigned log_n,
             int inverse, const uint8_t* prescale_first, const uint8_t* prescale_inc);
int zkmi_ntt_dev(int curve, const void* d_in, void* d_out, unsigned log_n, int inverse, const uint8_t* prescale_first,
                 const uint8_t* prescale_inc);

/* ---- Fr batch operations ---------------------------------------------------------------------------------------- */
/* curve.Fr.batchApplyKey(buf, first, inc) (min.js:1@211529, kernel frm_batchApplyKey @128060): out[i] = in[i]*first*inc^i */
int zkmi_fr_batch_apply_key(int curve, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages,
                            size_t n, const uint8_t* first, const uint8_t* inc);
int zkmi_fr_batch_apply_key_dev(int curve, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc);
/* op = ZKMI_BATCH_* */
int zkmi_fr_batch(int curve, int op, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, size_t n);
int zkmi_fr_batch_dev(int curve, int op, const void* d_in, void* d_out, size_t n);
/* joinABC (src/groth16_prove.js:320-374: qap_joinABC min.js:1@123664 + frm_batchFromMontgomery):
 * out[i] = fromMontgomery(a[i]*b[i] - c[i]) */
int zkmi_groth16_join_abc(int curve, zkmi_pages a, zkmi_pages b, zkmi_pages c, uint8_t* const* out_ptr,
                          const size_t* out_len, int n_out_pages, size_t n);
int zkmi_groth16_join_abc_dev(int curve, const void* d_a, const void* d_b, const void* d_c, void* d_out, size_t n);

/* ---- fused Groth16 prover ---------------------------------------------------------------------- */
/* groth16Prove (src/groth16_prove.js:28-144) from already-read sections: buildABC1 (:147-187) as a device kernel,
 * 3 x (ifft -> coset scale -> fft), joinABC, 4 G1 MSMs + 1 G2 MSM, blinding (:103-120) and toAffine (:130-132).
 * All section pointers are HOST pointers in the zkey/wtns on-disk layout (src/zkey_utils.js:229-259):
 *   coeffs = zkey section 4; bases_a/b1/c/h = sections 5,6,8,9 (G1 affine); bases_b2 = section 7 (G2 affine);
 *   witness = wtns section 2 (n_vars x 32 B, normal form); vk_* = header points (affine);
 *   r_mont / s_mont = the two Fr.random() draws (:103-104), Montgomery form.
 * zkey_cache_key != 0 keeps the five base tables and the coefficient table resident across calls.
 * Outputs are affine Montgomery: pi_a, pi_c 2*n8q bytes, pi_b 4*n8q bytes. */
typedef struct zkmi_groth16_zkey {
    int curve;
    uint32_t n_vars, n_public, domain_size;
    const uint8_t* coeffs; size_t coeffs_len;
    const uint8_t *bases_a, *bases_b1, *bases_b2, *bases_c, *bases_h;
    const uint8_t *vk_alpha_1, *vk_beta_1, *vk_beta_2, *vk_delta_1, *vk_delta_2;
} zkmi_groth16_zkey;
/* Upload the proving key under `zkey_cache_key` (!= 0): base tables + CSR form of the coefficient section. */
int zkmi_groth16_load(const zkmi_groth16_zkey* zkey, uint64_t zkey_cache_key);
/* One proof. zkey_cache_key != 0: the key is loaded on first use (zkey may be NULL afterwards) and stays resident;
 * zkey_cache_key == 0: load, prove, release. `witness` is a HOST pointer (n_vars x 32 B, normal form). */
int zkmi_groth16_prove(const zkmi_groth16_zkey* zkey, uint64_t zkey_cache_key, const uint8_t* witness,
                       const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);
/* Same with the witness already in device memory and the key loaded (bench.py: inputs resident in HBM). */
int zkmi_groth16_prove_dev(uint64_t zkey_cache_key, const void* d_witness, const uint8_t* r_mont, const uint8_t* s_mont,
                           uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);
int zkmi_groth16_release(uint64_t zkey_cache_key);
/* Device time (ms, HIP events) of the stages of the last proof, in order: buildABC, 6 NTTs, joinABC, sort(witness),
 * bucket accumulation of MSM A, B1, B2, C, sort(H scalars), accumulation of MSM H, batched bucket reductions.
 * Writes min(n, ZKMI_GROTH16_STAGES) values. */
#define ZKMI_GROTH16_STAGES 11
int zkmi_groth16_stage_ms(double* out, int n);

/* ---- utilities --------------------------------------------------------------------------------------------------- */
/* Synthetic base table of SURVEY.md §8d: P_i = (f*g^i mod r)*G written to device memory as affine Montgomery points
 * (what G.batchApplyKey(G repeated n, Fr.e(f), Fr.e(g)) returns).  For benchmarks and tests. */
int zkmi_gen_geometric_bases_dev(int curve, int group, size_t n, uint64_t f, uint64_t g, void* d_out);
/* G.toAffine on host for one Jacobian point (tiny; used by bindings to normalise results). */
int zkmi_to_affine(int curve, int group, const uint8_t* jacobian, uint8_t* affine);
/* G.add on host for two Jacobian points (O(1)): folds the per-GPU partial results of a sharded MSM
 * (reference: host-side `G.add` over chunk results, min.js:1@214651). Needs no device. */
int zkmi_point_add(int curve, int group, const uint8_t* a_jacobian, const uint8_t* b_jacobian, uint8_t* out_jacobian);
/* Wall-clock-free device timing of the last call of each kind, in milliseconds (HIP events on the library stream). */
double zkmi_last_kernel_ms(void);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI_H */
