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
 *   - ONE CALLER PER PROCESS. The library keeps one context per process (one process per GPU): the active pipeline slot, its streams
 *     and its scratch buffers are process-global state, and an entry point may swap them while it runs. Calls must therefore be
 *     serialised by the host: the N-API addon takes a mutex around every entry point (main thread and libuv pool threads alike),
 *     the Python mirror a threading.RLock (snarkjs_amd/zkmi.py). zkmi_last_error() is per thread.
 *   - statistics, calibration probes, synthetic-base generators and tuning knobs (no reference counterpart) are declared in
 *     zkmi_diag.h, not here.
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
/* stream-ordered device-to-device copy / fill on the library stream */
int zkmi_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes);
int zkmi_memset_dev(void* d_dst, int value, size_t bytes);

/* ---- G.multiExpAffine ------------------------------------------------------------------------------------------- */
/* curve.G1.multiExpAffine / curve.G2.multiExpAffine (min.js:1@214996 -> @214651 -> _multiExpChunk @213360; kernel
 * g1m_/g2m_multiexpAffine_chunk @75966).  group = 1 | 2.  n bases of 2*group*n8q bytes, n scalars of scalar_bytes.
 * base_cache_key & ZKMI_BASES_CACHE ALLOWS the library to keep these bases resident (zkey sections / SRS slices are static per circuit,
 * src/groth16_prove.js:84-100); its value carries no identity. The cache is content-addressed: the library hashes the whole base
 * buffer on every call (128 bits per 64 KiB chunk, a fast NON-cryptographic mix: accidental changes — another zkey, an edited section —
 * are told apart, deliberately constructed colliding buffers are not; callers that take bases from an untrusted party pass key 0). The
 * pre-computed window table of a buffer is built on its SECOND sight, an MSM over a prefix of a resident buffer re-uses its
 * table, tables that cannot fit are never built (plain bases instead), and least-recently-used tables are evicted under a
 * byte budget (env ZKMI_BASE_CACHE_BYTES, default 64 GiB). zkmi_release_bases drops every cached table.
 * base_cache_key is a set of permission bits (any other bit is ignored):
 *   ZKMI_BASES_CACHE (1)      resident tables allowed; identity = the FULL content hash, on every call: the result is always a function of the
 *                             bytes passed, like the reference's (min.js:1@213360) — an in-place edit of a resident buffer is seen by the next call.
 *   ZKMI_BASES_IMMUTABLE (2)  (with bit 1) the caller PROMISES that a buffer it passes again at the same first-page address with the same length
 *                             still holds the same bytes (a zkey section it never writes to). Only then a buffer that has been checked byte for
 *                             byte against a resident table is re-checked by SAMPLE (its first and last whole chunk and 30 chunks at positions that
 *                             change from call to call; 2 MB instead of 64 MB hashed at 2^20 points) and in full on every 32nd sight. A caller that
 *                             breaks the promise can be handed the old buffer's result for up to 31 calls. Never assumed by default (r04 did; r05:
 *                             opt-in): register(curve, {immutableBases: true}) in js/register.js, cache_key=3 in the Python mirror.
 * ZKMI_BASE_HASH_FULL=1 in the environment forces the full hash whatever the caller promised.
 * out_jacobian: 3*group*n8q bytes. */
#define ZKMI_BASES_CACHE 1ull
#define ZKMI_BASES_IMMUTABLE 2ull
int zkmi_msm(int curve, int group, zkmi_pages bases, zkmi_pages scalars, size_t n, size_t scalar_bytes,
             uint64_t base_cache_key, uint8_t* out_jacobian);
int zkmi_release_bases(uint64_t base_cache_key);
/* Same with bases and scalars already resident in device memory (bench.py, fused pipelines). */
int zkmi_msm_dev(int curve, int group, const void* d_bases, const void* d_scalars, size_t n, size_t scalar_bytes,
                 uint8_t* out_jacobian);
/* Resident bases with pre-computed window tables T[k][i] = 2^(c*k) * P_i (static per zkey / SRS: src/groth16_prove.js:84-100,
 * src/polynomial/polynomial.js:970-977 slices the same PTau for every commitment). Build once from n device-resident affine
 * points; each MSM then uses the first k <= n bases with k scalars of at most 32 bytes. */
int zkmi_msm_table_build(int curve, int group, const void* d_bases, size_t n, uint64_t* handle);
int zkmi_msm_table_dev(uint64_t handle, const void* d_scalars, size_t k, size_t scalar_bytes, uint8_t* out_jacobian);
/* up to 4 independent MSMs against one table in a single call (PLONK commits A,B,C / T1,T2,T3 / Wxi,Wxiw per round): their
 * latency-bound bucket reductions share one set of launches. out_jacobians: count x 3*group*n8q bytes. */
int zkmi_msm_table_multi_dev(uint64_t handle, const void* const* d_scalars, const size_t* ks, int count, size_t scalar_bytes,
                             uint8_t* out_jacobians);
/* The same call in two halves (r06), for a host that drives two proofs from one thread (zkmi_pipeline_select): _enqueue_dev puts the digit sorts, accumulations and bucket
 * reductions of the MSMs on the ACTIVE pipeline slot's streams and returns at once; _collect (same slot, same table, same count) waits for them, folds the window sums and writes
 * count x 3*group*n8q bytes. Between the two the host can enqueue the other proof's work — its accumulations then run underneath this call's latency-bound reduction tail.
 * d_scalars must stay valid until the collect; one enqueued call per pipeline slot (an enqueued call that is never collected — its proof was abandoned — is waited for and
 * dropped by the slot's next enqueue). */
int zkmi_msm_table_multi_enqueue_dev(uint64_t handle, const void* const* d_scalars, const size_t* ks, int count, size_t scalar_bytes);
int zkmi_msm_table_multi_collect(uint64_t handle, int count, uint8_t* out_jacobians);
/* Polynomial.multiExponentiation (src/polynomial/polynomial.js:970-977) for the commitments of one round: the `count` coefficient arrays are MONTGOMERY Fr elements; their
 * batchFromMontgomery (one launch for all of them, into scratch memory of the active pipeline slot) and the MSMs are enqueued as by zkmi_msm_table_multi_enqueue_dev. The
 * polynomials are read by the conversion launch only, which is stream-ordered before anything the caller enqueues on the slot afterwards. Collect with zkmi_msm_table_multi_collect. */
int zkmi_msm_table_multi_enqueue_mont_dev(uint64_t handle, const void* const* d_polys, const size_t* ks, int count);
int zkmi_msm_table_release(uint64_t handle);
/* Curve, group and number of resident points of a table (any pointer may be NULL): a binding sizes the result buffers of the two calls above
 * — 3*group*n8q bytes per MSM — from the TABLE instead of trusting its caller. */
int zkmi_msm_table_info(uint64_t handle, int* curve, int* group, size_t* n);

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
/* Evaluations.fromPolynomial (src/polynomial/evaluations.js:30-37): the transform of a polynomial of in_len <= 2^log_n coefficients, zero-padded to 2^log_n — the padding is
 * never written: the first pass reads zeros beyond in_len. d_out: 2^log_n elements (d_in == d_out allowed when that buffer is 2^log_n long). */
int zkmi_ntt_padded_dev(int curve, const void* d_in, size_t in_len, void* d_out, unsigned log_n, int inverse);

/* ---- Fr batch operations ---------------------------------------------------------------------------------------- */
/* curve.Fr.batchApplyKey(buf, first, inc) (min.js:1@211529, kernel frm_batchApplyKey @128060): out[i] = in[i]*first*inc^i */
int zkmi_fr_batch_apply_key(int curve, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages,
                            size_t n, const uint8_t* first, const uint8_t* inc);
int zkmi_fr_batch_apply_key_dev(int curve, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc);
/* op = ZKMI_BATCH_* */
int zkmi_fr_batch(int curve, int op, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, size_t n);
int zkmi_fr_batch_dev(int curve, int op, const void* d_in, void* d_out, size_t n);
/* the two Montgomery conversions on up to four arrays in one launch (ns[i] elements each; in place allowed) */
int zkmi_fr_batch_multi_dev(int curve, int op, const void* const* d_in, void* const* d_out, const size_t* ns, int count);
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
    /* byte lengths of sections 5, 6, 7, 8, 9 as found in the file: checked against nVars / nPublic / domainSize of the header before
     * anything is read (a truncated or malformed zkey fails with ZKMI_ERR_INVALID instead of reading past the caller's buffers; the
     * reference gets the same protection from bounds-checked JS buffers). The vk_* points are 2*n8q (G1) / 4*n8q (G2) bytes. */
    size_t bases_a_len, bases_b1_len, bases_b2_len, bases_c_len, bases_h_len;
} zkmi_groth16_zkey;
/* Upload the proving key under `zkey_cache_key` (!= 0): base tables + the coefficient section as a length-sorted sliced layout. */
int zkmi_groth16_load(const zkmi_groth16_zkey* zkey, uint64_t zkey_cache_key);
/* The same with every bulk section given as PAGES: what the reference holds after binFileUtils.readSection (src/groth16_prove.js:29-33, :57-59,
 * :84-100) — a Uint8Array below 2^30 bytes, a BigBuffer of <= 1 GiB pages from there on (sections 4-9 of a 2^24-constraint key are 1 - 2 GB
 * each; one Node buffer ends at 2 GiB - 1). Section lengths are the page totals; they are checked against the header like the flat form's.
 * A page whose POINTER is NULL is a gap of `len` bytes the caller did not read: zkmi_groth16_load_shard_paged only touches the byte ranges of
 * its own variables / H bases, so a shard process reads just those from the file (js/groth16_shards.js) and passes the rest as gaps; a load
 * that needs a byte inside a gap fails with ZKMI_ERR_INVALID. The coefficient section is needed whole by every loader. */
typedef struct zkmi_groth16_zkey_paged {
    int curve;
    uint32_t n_vars, n_public, domain_size;
    zkmi_pages coeffs, bases_a, bases_b1, bases_b2, bases_c, bases_h;      /* zkey sections 4, 5, 6, 7, 8, 9 */
    const uint8_t *vk_alpha_1, *vk_beta_1, *vk_beta_2, *vk_delta_1, *vk_delta_2;
} zkmi_groth16_zkey_paged;
int zkmi_groth16_load_paged(const zkmi_groth16_zkey_paged* zkey, uint64_t zkey_cache_key);
int zkmi_groth16_load_shard_paged(const zkmi_groth16_zkey_paged* zkey, uint64_t zkey_cache_key, uint32_t var_lo, uint32_t var_hi, uint32_t h_lo, uint32_t h_hi);
/* zkmi_groth16_prove (below) with the paged descriptor: same key semantics (load on first use, descriptor checked against a resident key, key 0 =
 * load, prove, release). */
int zkmi_groth16_prove_paged(const zkmi_groth16_zkey_paged* zkey, uint64_t zkey_cache_key, const uint8_t* witness, size_t witness_len,
                             const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);
/* buildABC1 (src/groth16_prove.js:147-187) alone, on a resident key: A_T, B_T, C_T = A_T * B_T (domain x 32 bytes each, Montgomery) from the
 * witness in device memory (normal form); any output may be NULL. Rows are balanced by length: a 10^5-term row of a real circuit is spread
 * over thousands of lanes (csrc/groth16.hip). zkmi_last_kernel_ms() = the device time of the three kernels. */
int zkmi_groth16_build_abc_dev(uint64_t zkey_cache_key, const void* d_witness, void* d_a, void* d_b, void* d_c);
/* One proof. zkey_cache_key != 0: the key is loaded on first use (zkey may be NULL afterwards) and stays resident; a descriptor
 * given together with an already resident key must describe the same circuit (curve, nVars, nPublic, domainSize, nCoef), else the
 * call fails with ZKMI_ERR_INVALID — release the key first to replace it. zkey_cache_key == 0: load, prove, release.
 * `witness` is a HOST pointer to wtns section 2 (normal form), witness_len its byte length: must be n_vars x 32
 * ("Invalid witness length", src/groth16_prove.js:45-47). */
int zkmi_groth16_prove(const zkmi_groth16_zkey* zkey, uint64_t zkey_cache_key, const uint8_t* witness, size_t witness_len,
                       const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);
/* Same with the witness already in device memory and the key loaded (bench.py: inputs resident in HBM). */
int zkmi_groth16_prove_dev(uint64_t zkey_cache_key, const void* d_witness, const uint8_t* r_mont, const uint8_t* s_mont,
                           uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);
/* Two proofs in flight on one GPU (throughput mode): zkmi_groth16_submit_dev enqueues the whole device part of a proof into
 * pipeline slot 0 or 1 and returns at once; zkmi_groth16_collect waits for that slot, folds the window sums and applies the
 * blinding + toAffine. While proof k sits in its latency-bound tail (bucket reductions: few waves, long dependency chains; result
 * copies; host folds) the throughput-bound front of proof k+1 (buildABC, NTTs, accumulations) already runs. Each slot owns its
 * streams, scratch and work buffers; d_witness must stay valid until the slot is collected. prove_dev == submit(0) + collect(0). */
int zkmi_groth16_submit_dev(uint64_t zkey_cache_key, const void* d_witness, int slot);
/* Host-orchestrated provers (plonk.prove, fflonk.prove: the rounds are driven by the host between transcript hashes) with two proofs in flight
 * from ONE host thread: every library call works on the ACTIVE pipeline slot (0 | 1) — its own stream and events, scratch buffers, pool of
 * zkmi_dev_alloc blocks and ring of per-call constants — so the host alternates between two proofs, switching slots at its blocking calls, and
 * the GPU always holds the queued work of the other proof (snarkjs_amd/plonk.py: prove_many). Resident keys and window tables are shared. */
int zkmi_pipeline_select(int slot);
int zkmi_pipeline_active(void);
/* The same for a witness in HOST memory (wtns section 2, witness_len = n_vars x 32): it crosses PCIe on the slot's own stream into the
 * slot's own buffer, i.e. underneath the kernels of the proof in the other slot — the throughput mode of a host that keeps witnesses in
 * host memory (js/groth16_native.js: proveMany). `witness` may be re-used as soon as the call returns. */
int zkmi_groth16_submit(uint64_t zkey_cache_key, const uint8_t* witness, size_t witness_len, int slot);
int zkmi_groth16_collect(uint64_t zkey_cache_key, int slot, const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);
int zkmi_groth16_release(uint64_t zkey_cache_key);
/* Multi-GPU proof (BASELINE configs[2]: MSMs sharded across the GPUs of a node). Every rank loads the shard of the
 * key that holds the witness-side bases of the variables [var_lo, var_hi) (sections 5-8) and the H bases [h_lo, h_hi) (section 9);
 * the section pointers of `zkey` are those of the FULL sections, the library slices them. zkmi_groth16_sums_dev runs the device
 * part of a proof on the FULL witness (buildABC, NTT chain, joinABC are replicated: they need no exchange) and returns this
 * shard's five partial MSM sums  jA | jB1 | jB2 | jC | jH  (Jacobian, 3*n8q bytes each, jB2 6*n8q: 7*3*n8q bytes in total).
 * The caller adds the sums of all ranks (one all-gather of < 1 KB per rank + zkmi_point_add) and zkmi_groth16_finish applies the
 * blinding and toAffine (:103-132, host, O(1)). With the full key, sums + finish == zkmi_groth16_prove_dev. */
int zkmi_groth16_load_shard(const zkmi_groth16_zkey* zkey, uint64_t zkey_cache_key, uint32_t var_lo, uint32_t var_hi, uint32_t h_lo, uint32_t h_hi);
int zkmi_groth16_sums_dev(uint64_t zkey_cache_key, const void* d_witness, uint8_t* sums);
/* Chain-parallel multi-GPU proof: the three iNTT -> coset -> NTT chains of src/groth16_prove.js:64-76 are independent until joinABC
 * (:79), so each runs on a different rank. zkmi_groth16_chains_dev runs buildABC (cheap, every owning rank) and the chains selected by
 * chain_mask (bit 0: A, 1: B, 2: C) on the full domain and writes their outputs (domain x 32 B, Montgomery) to d_a / d_b / d_c
 * (NULL for unselected chains). The owner of a chain sends rank j the slice [h_lo_j, h_hi_j) of its output (point-to-point over
 * xGMI, domain*32 bytes leave each owner in total); rank j joins its three slices (zkmi_groth16_join_abc_dev) into ITS H-MSM scalars and
 * zkmi_groth16_sums_h_dev runs the five MSMs of its key shard with them — no rank repeats another rank's transforms. */
int zkmi_groth16_chains_dev(uint64_t zkey_cache_key, const void* d_witness, unsigned chain_mask, void* d_a, void* d_b, void* d_c);
/* The shard's MSMs in two halves, so that no rank idles while transforms run elsewhere and slices travel: zkmi_groth16_sums_w_dev enqueues
 * the witness-side half (digit sorts of the witness, bucket accumulations B2, B1, A, C, the G2 bucket reduction — they need the witness
 * only) and returns at once; zkmi_groth16_sums_h_dev then enqueues the H half (digit sort of d_h_scalars, accumulation H, the batched G1
 * bucket reductions), waits and returns the sums. Without a preceding _sums_w_dev, _sums_h_dev runs both halves (r02 behaviour). */
int zkmi_groth16_sums_w_dev(uint64_t zkey_cache_key, const void* d_witness);
int zkmi_groth16_sums_h_dev(uint64_t zkey_cache_key, const void* d_witness, const void* d_h_scalars, uint8_t* sums);
int zkmi_groth16_finish(uint64_t zkey_cache_key, const uint8_t* sums, const uint8_t* r_mont, const uint8_t* s_mont, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);

/* ---- PLONK prover: the per-element loops of src/plonk_prove.js and src/polynomial/polynomial.js as device kernels ---------
 * (SURVEY.md 8a rows a10-a12). All buffers are DEVICE pointers to Montgomery Fr elements; 32-byte constants are host
 * pointers (Montgomery). Polynomial lengths are in elements. */
/* Fr.w[i] (min.js:1@185893), needs no device */
int zkmi_fr_root(int curve, unsigned i, uint8_t* out32);
/* computeWirePolynomials gather (plonk_prove.js:267-283): A/B/C[i] = getWitness(map[i]) for i < n_constraints, 0 up to domain.
 * d_witness: n_witness elements (normal form, witness[0] already zeroed, :94-96); d_internal: the n_additions internal
 * signals (:174-204); maps are u32 arrays. Output is in the witness's (normal) form: run batchToMontgomery next (:285-287). */
int zkmi_plonk_gather_wires_dev(int curve, const void* d_witness, uint32_t n_witness, const void* d_internal, uint32_t n_additions,
                                const void* d_map_a, const void* d_map_b, const void* d_map_c, uint32_t n_constraints, uint32_t domain,
                                void* d_a, void* d_b, void* d_c);
/* the same with Fr.batchToMontgomery (:278) applied in the same pass: A/B/C come out in Montgomery form */
int zkmi_plonk_gather_wires_mont_dev(int curve, const void* d_witness, uint32_t n_witness, const void* d_internal, uint32_t n_additions,
                                     const void* d_map_a, const void* d_map_b, const void* d_map_c, uint32_t n_constraints, uint32_t domain,
                                     void* d_a, void* d_b, void* d_c);
/* calculateAdditions (src/plonk_prove.js:174-204, src/fflonk_prove.js:269-300): the internal signals of a PLONK / FFLONK key,
 *   internal[i] = factor1_i * getWitness(id1_i) + factor2_i * getWitness(id2_i),   getWitness as in :207-215 (n_vars = n_witness + n_additions),
 * computed on the device in ONE launch although an addition may read internal signals created before it (a dependency DAG of any depth; time grows
 * with the depth — a few microseconds per level — which is ceil(log2 k) for a k-term combination in keys written by plonk.setup, src/plonk_setup.js:176-212). d_additions: the zkey's additions section as it lies in the file (n_additions records of 72 bytes: u32 id1, u32 id2, factor1,
 * factor2 in Montgomery form). d_witness as for zkmi_plonk_gather_wires_dev. d_internal: n_additions elements, normal form (the
 * reference's buffInternalWitness). Stream-ordered: returns once enqueued. */
int zkmi_plonk_additions_dev(int curve, const void* d_additions, uint32_t n_additions, const void* d_witness, uint32_t n_witness, void* d_internal);
/* computeZ (plonk_prove.js:361-455): grand-product evaluations Z[0..domain) from the wire buffers (Montgomery) and the 4n
 * sigma evaluations (sampled at stride 4). Fails with "Copy constraints does not match" if Z[0] != 1. w_n = Fr.w[power]. */
int zkmi_plonk_compute_z_dev(int curve, const void* d_a, const void* d_b, const void* d_c, const void* d_s1e, const void* d_s2e,
                             const void* d_s3e, uint32_t domain, const uint8_t* beta, const uint8_t* gamma, const uint8_t* k1,
                             const uint8_t* k2, const uint8_t* w_n, void* d_z);
/* The same without the wait and without the Z[0] check: the caller compares Z[0] with one at its next synchronisation point and raises the
 * reference's "Copy constraints does not match" there (two proofs in flight from one host thread: the host must not block per kernel). */
int zkmi_plonk_compute_z_enqueue(int curve, const void* d_a, const void* d_b, const void* d_c, const void* d_s1e, const void* d_s2e,
                                 const void* d_s3e, uint32_t domain, const uint8_t* beta, const uint8_t* gamma, const uint8_t* k1,
                                 const uint8_t* k2, const uint8_t* w_n, void* d_z);
/* computeT (plonk_prove.js:516-628 with MulZ.mul2/mul4, mul_z.js:49-148): T and Tz over the 4n extended evaluation points.
 * lagrange = zkey section 13 on the device (per public input: n coefficients then 4n evaluations); pub_a = buffers.A. */
typedef struct zkmi_plonk_evals {
    const void *a, *b, *c, *z, *qm, *ql, *qr, *qo, *qc, *s1, *s2, *s3;     /* 4n evaluations each */
    const void* lagrange;
    const void* pub_a;
} zkmi_plonk_evals;
int zkmi_plonk_compute_t_dev(int curve, const zkmi_plonk_evals* ev, uint32_t domain, uint32_t n_public, const uint8_t* blind11 /* b1..b11 */,
                             const uint8_t* beta, const uint8_t* gamma, const uint8_t* alpha, const uint8_t* k1, const uint8_t* k2,
                             const uint8_t* w_n, const uint8_t* w_4n, const uint8_t* w_2, void* d_t, void* d_tz);
/* FFLONK quotient numerators (src/fflonk_prove.js): T0 (:415-504) over 4n points from a,b,c,ql,qr,qm,qo,qc,lagrange,pub_a of
 * `ev`; T1 / T1z (:667-718) over 2n points from the 4n evaluations of z and the Lagrange section (b789 = b7,b8,b9; w_2n =
 * Fr.w[power+1]); T2 / T2z (:720-815) over 4n points from a,b,c,z,s1,s2,s3 of `ev`. */
int zkmi_fflonk_t0_dev(int curve, const zkmi_plonk_evals* ev, uint32_t domain, uint32_t n_public, void* d_t0);
int zkmi_fflonk_t1_dev(int curve, const void* d_z4, const void* d_lagrange, uint32_t domain, const uint8_t* b789, const uint8_t* w_2n,
                       void* d_t1, void* d_t1z);
int zkmi_fflonk_t2_dev(int curve, const zkmi_plonk_evals* ev, uint32_t domain, const uint8_t* b789, const uint8_t* beta, const uint8_t* gamma,
                       const uint8_t* k1, const uint8_t* k2, const uint8_t* w_n, const uint8_t* w_4n, void* d_t2, void* d_t2z);
/* Polynomial.degree (polynomial.js:163-172): highest index of a non-zero coefficient, 0 if none */
int zkmi_poly_degree_dev(int curve, const void* d_p, size_t n, size_t* degree);
/* Keccak-256 with the original 0x01 padding (@noble/hashes keccak_256 as used by src/Keccak256Transcript.js:18-62); host only,
 * needs no device: the Fiat-Shamir transcript of the PLONK / FFLONK provers */
int zkmi_keccak256(const uint8_t* data, size_t len, uint8_t* out32);
/* Polynomial.add / sub with optional blinding value (polynomial.js:218-276): y[i] = y[i] +/- k*x[i], i < nx (k NULL = 1) */
int zkmi_poly_axpy_dev(int curve, void* d_y, const void* d_x, size_t nx, const uint8_t* k, int subtract);
/* Polynomial.mulScalar (:278-284) */
int zkmi_poly_scale_dev(int curve, void* d_p, size_t n, const uint8_t* k);
/* blindCoefficients (polynomial.js:68-93) on a buffer of n + count elements whose tail is zero: p[n+i] += f_i, p[i] -= f_i.
 * factors: count x 32 bytes (host, Montgomery), count <= 8. No host synchronisation. Per-call constants of every function of this section (factors, k, x, beta) travel as
 * kernel arguments: none of them costs an upload of its own. */
int zkmi_poly_blind_dev(int curve, void* d_p, size_t n, const uint8_t* factors, int count);
/* the same IN PLACE on a buffer with room for n + count elements of which only the first n were written: p[n+i] = f_i, p[i] -= f_i */
int zkmi_poly_blind_tail_dev(int curve, void* d_p, size_t n, const uint8_t* factors, int count);
/* A chain of Polynomial.add / sub / mulScalar / addScalar (polynomial.js:218-290) as ONE pass: out[i] = sum_j k_j p_j[i] (i < len_j) + (i == 0 ? constant : 0), i < out_len.
 * has_k = 0: k_j = 1 (pass -k for a subtraction), constant NULL = none, count <= 16, every len_j <= out_len; out may be one of the operands. Exact field arithmetic: the result is
 * the one the reference's sequence of calls leaves, whatever their order (src/plonk_prove.js:769-866 builds R and Wxi with 18 such calls). A term is 56 bytes without host
 * pointers, so that a binding can fill an array of them in one flat buffer (device pointer and length as 64-bit little-endian integers). */
typedef struct zkmi_poly_term { const void* d_p; uint64_t len; uint8_t k[32]; uint32_t has_k; uint32_t reserved; } zkmi_poly_term;
int zkmi_poly_lincomb_dev(int curve, void* d_out, size_t out_len, const zkmi_poly_term* terms, int count, const uint8_t* constant);
/* addScalar (polynomial.js:286-290): p[0] += value. No host synchronisation. */
int zkmi_poly_add_scalar_dev(int curve, void* d_p, const uint8_t* value);
/* Polynomial.evaluate (Horner, :174-184) as a parallel reduction; out = 32 bytes (host) */
int zkmi_poly_evaluate_dev(int curve, const void* d_p, size_t n, const uint8_t* x, uint8_t* out);
/* count <= 8 evaluations with ONE wait: polynomial q (lens[q] coefficients) at xs[32 q .. 32 q + 32) -> out[32 q ..] (round 4 of PLONK: six evaluations at two points,
 * src/plonk_prove.js:686-708). The power table of a point is built on the device; points that repeat share it. */
int zkmi_poly_evaluate_multi_dev(int curve, const void* const* d_polys, const size_t* lens, const uint8_t* xs, int count, uint8_t* out);
/* *all_zero = 1 iff p[0..n) are all zero (degree checks, plonk_prove.js:298-306, :645-647) */
int zkmi_poly_is_zero_dev(int curve, const void* d_p, size_t n, int* all_zero);
/* Polynomial.divZh(domainSize, extensions) (:592-615), in place; "Polynomial is not divisible" on a non-zero tail */
int zkmi_poly_div_zh_dev(int curve, void* d_p, size_t len, uint32_t domain, uint32_t extensions);
/* Polynomial.divByZerofier(n, beta) (:617-674): division by X^n - beta, in place (PLONK openings use n = 1, FFLONK n > 1) */
int zkmi_poly_div_by_zerofier_dev(int curve, void* d_p, size_t len, uint32_t n, const uint8_t* beta);
/* the same, enqueued only (no wait, no error from the division itself): the reference's divisibility test is "the n highest coefficients of the quotient are zero"
 * (:665-669) — the caller makes it at its next synchronisation point with zkmi_poly_is_zero_dev(p + len - n, n) and raises "Polynomial is not divisible" itself. */
int zkmi_poly_div_by_zerofier_enqueue(int curve, void* d_p, size_t len, uint32_t n, const uint8_t* beta);
/* Round 3 of PLONK (src/plonk_prove.js:649-672): T (t_len >= 3*domain + 6 coefficients, fewer read as zero) split into T1 = T[0, n) + b10 X^n (n + 1 coefficients),
 * T2 = T[n, 2n) - b10 + b11 X^n (n + 1), T3 = T[2n, 3n + 6) - b11 (n + 6): one launch instead of three zero fills, three copies and five single-coefficient reads / writes */
int zkmi_plonk_split_t_dev(int curve, const void* d_t, size_t t_len, uint32_t domain, const uint8_t* b10, const uint8_t* b11, void* d_t1, void* d_t2, void* d_t3);
/* CPolynomial.getPolynomial (src/polynomial/cpolynomial.js:53-73): out[i*n + j] = P_j[i] for i < lens[j] (d_polys[j] may be
 * NULL), zero elsewhere; n <= 16 component polynomials; out_len elements are written. */
int zkmi_cpoly_interleave_dev(int curve, const void* const* d_polys, const size_t* lens, int n, void* d_out, size_t out_len);

/* ---- group-element FFTs and G.batchApplyKey (ceremony side) -------------------------------------------------
 * curve.G1/G2.fft / .ifft (engine_fft for groups, min.js:1@215859: the Fr butterflies with "multiply by a twiddle" = G.timesFr) and, through
 * it, G.lagrangeEvaluations for 2^k <= 2^Fr.s points (src/powersoftau_preparephase2.js:87). n = 2^log_n affine points in (2*group*n8q bytes
 * each, Montgomery, all-zero = infinity), n affine points out, natural order; the inverse includes the factor 1/n. */
int zkmi_group_fft(int curve, int group, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, unsigned log_n, int inverse);
int zkmi_group_fft_dev(int curve, int group, const void* d_in, void* d_out, unsigned log_n, int inverse);
/* curve.G1/G2.batchApplyKey(buff, first, inc) (engine_applykey, min.js:1@211529; src/mpc_applykey.js:44-70): out_i = (first * inc^i) * P_i,
 * affine in and out; first / inc are Montgomery Fr elements (host pointers). */
int zkmi_group_batch_apply_key(int curve, int group, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, size_t n,
                               const uint8_t* first, const uint8_t* inc);
int zkmi_group_batch_apply_key_dev(int curve, int group, const void* d_in, void* d_out, size_t n, const uint8_t* first, const uint8_t* inc);

/* curve.G1/G2.batchLEMtoU / batchUtoLEM / batchLEMtoC / batchCtoLEM (engine_batchconvert over wasmcurves' g1m_/g2m_batch*, min.js:1@128060;
 * callers src/powersoftau_import.js:159,221, src/powersoftau_contribute.js:145,176, src/powersoftau_export_challenge.js:72,
 * src/powersoftau_verify.js:358, src/mpc_applykey.js:64-70, src/zkey_export_bellman.js:36-83, src/zkey_new.js:103-115,373).
 *   LEM = affine little-endian Montgomery (2*group*n8q bytes, all-zero = infinity), U = affine big-endian normal form (same size, Fq2 as
 *   c1 || c0, all-zero = infinity), C = x alone big-endian (group*n8q bytes; first byte |= 0x80 when y > (p-1)/2, 0x40 = infinity).
 * n points in, n points out. C_TO_LEM recovers y by a square root and returns ZKMI_ERR_INVALID when some x has no point on the curve. */
#define ZKMI_CONV_LEM_TO_U 0
#define ZKMI_CONV_U_TO_LEM 1
#define ZKMI_CONV_LEM_TO_C 2
#define ZKMI_CONV_C_TO_LEM 3
int zkmi_group_convert(int curve, int group, int kind, zkmi_pages in, uint8_t* const* out_ptr, const size_t* out_len, int n_out_pages, size_t n);
int zkmi_group_convert_dev(int curve, int group, int kind, const void* d_in, void* d_out, size_t n);


/* ---- utilities --------------------------------------------------------------------------------------------------- */
/* Page-lock caller-owned host memory for device transfers (hipHostRegister): shared-memory regions through which the processes of a
 * multi-GPU proof exchange chain outputs (js/groth16_shards.js). Optional: transfers from unregistered memory work, slower. */
int zkmi_host_register(void* host_ptr, size_t bytes);
int zkmi_host_unregister(void* host_ptr);
/* ---- GPU-to-GPU exchange between the processes of a multi-GPU proof (snarkjs_amd/csrc/peer.hip) --------------------------------------
 * The reference spreads one multiExp / one proof over its workers by handing them chunk buffers and folding the results on the host
 * (ffjavascript engine_multiexp, min.js:1@214651; worker dispatch @207729). Here a worker is a process that owns one GPU; the bulk data of a
 * sharded Groth16 proof (the three chain outputs, domain x 32 bytes each) moves device to device: the owner of a chain exports the buffer
 * that holds its output, every other process opens the handle once and pulls ITS slice per proof — over xGMI between two GPUs, inside HBM
 * when two processes share a device. Control (who is ready, the 7 x 3 x n8q-byte partial sums) stays on the host's own channel.
 *   zkmi_ipc_export: handle (ZKMI_IPC_HANDLE_BYTES, plain bytes: send them over any channel) for a device pointer of this process
 *                    (zkmi_dev_alloc or any hipMalloc'ed range; interior pointers allowed).
 *   zkmi_ipc_open:   device pointer in THIS process for a handle (peer access is enabled on first use); *bytes_visible (optional) = bytes
 *                    from the pointer to the end of the exported allocation. A handle exported by the calling process resolves to the
 *                    original pointer. zkmi_ipc_close drops the mapping (the exporter keeps the memory).
 *   zkmi_peer_copy:  d_dst <- d_src for `bytes` bytes, complete on return. The copies run on a stream of their own, NOT behind what the library
 *                    stream has queued (a shard process has its witness-side accumulations there when the slices arrive).
 *                    zkmi_peer_copy_async: queued only; zkmi_peer_fence() makes the library stream wait (an event, no host wait) for every
 *                    copy queued so far, so that the next call (zkmi_groth16_join_abc_dev) sees the data. The exporter must have finished
 *                    writing (its zkmi_groth16_chains_dev returned) before a peer reads, and must not overwrite the buffer until every peer
 *                    has finished its copy: the host orders this (js/groth16_shards.js: a chain is announced when complete; the next proof
 *                    starts only after every worker has delivered its sums). */
#define ZKMI_IPC_HANDLE_BYTES 96
int zkmi_ipc_export(const void* d_ptr, uint8_t* handle);
int zkmi_ipc_open(const uint8_t* handle, void** d_ptr, size_t* bytes_visible);
int zkmi_ipc_close(void* d_ptr);
int zkmi_peer_copy(void* d_dst, const void* d_src, size_t bytes);
int zkmi_peer_copy_async(void* d_dst, const void* d_src, size_t bytes);
int zkmi_peer_fence(void);
/* Curve of a resident Groth16 key (ZKMI_CURVE_*), -1 when the key is not loaded: bindings size their output buffers from the KEY, not from
 * a caller-supplied curve id. */
int zkmi_groth16_key_curve(uint64_t key);
/* Error recovery for the split proof (zkmi_groth16_sums_w_dev / _sums_h_dev, _submit / _collect): waits for the device, then forgets any
 * half-enqueued or in-flight proof of this key in both pipeline slots, so that the next proof starts clean after a failed exchange. */
int zkmi_groth16_reset(uint64_t key);
/* G.toAffine on host for one Jacobian point (tiny; used by bindings to normalise results). */
int zkmi_to_affine(int curve, int group, const uint8_t* jacobian, uint8_t* affine);
/* G.add on host for two Jacobian points (O(1)): folds the per-GPU partial results of a sharded MSM
 * (reference: host-side `G.add` over chunk results, min.js:1@214651). Needs no device. */
int zkmi_point_add(int curve, int group, const uint8_t* a_jacobian, const uint8_t* b_jacobian, uint8_t* out_jacobian);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI_H */
