/*
 * include/zkmi_diag.h — diagnostics of libzkmi.so: timing and statistics read-outs, calibration probes of the box a run landed on,
 * synthetic-base generators for tests and benchmarks, tuning knobs. None of these has a counterpart in the reference; bench.py, the
 * tests and tools/ use them, a snarkjs binding does not need them. The drop-in boundary is include/zkmi.h.
 *
 * Like the boundary itself the library is single-caller per process (zkmi.h, conventions): serialise these calls with the others.
 */
#ifndef ZKMI_DIAG_H
#define ZKMI_DIAG_H
#include "zkmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* resident tables / their bytes / buffers seen once (no table yet) — for tests and diagnostics; any pointer may be NULL */
int zkmi_base_cache_stats(uint64_t* n_tables, uint64_t* table_bytes, uint64_t* n_seen);

/* Device time (ms, HIP events on the library stream) of the bucket-accumulation kernel of the last MSM that used job
 * slot `slot`: zkmi_msm / zkmi_msm_dev use slot 0; zkmi_groth16_prove uses 0..4 = A, B1, B2, C, H. -1 if never run. */
double zkmi_msm_accum_ms(int slot);
/* Diagnostics for the roofline accounting (bench.py int_alu): with zkmi_msm_stats(1) every bucket-accumulation launch is followed by a
 * small kernel that counts the mixed additions the launch performed (list entries of the digit sort whose base is not skipped and not the
 * point at infinity); zkmi_msm_accum_additions(slot) returns the count of the last MSM that used the job slot once the stream has been
 * synchronised (-1: never counted). Off by default; costs one pass over the sorted lists (~20 us at 2^20 terms) per MSM. */
int zkmi_msm_stats(int enable);
double zkmi_msm_accum_additions(int slot);

/* Window width used for n terms (tuning knob; 0 restores the built-in table). */
int zkmi_msm_set_window_bits(int c);

/* Device time (ms, HIP events) of the stages of the last proof, in order: buildABC, 6 NTTs, joinABC, sort(witness),
 * bucket accumulation of MSM B2, B1 (+ the second witness sort), A, C, sort(H scalars), accumulation of MSM H, batched G1 bucket reductions (the B2
 * reduction runs on a second stream underneath the G1 accumulations).
 * Writes min(n, ZKMI_GROTH16_STAGES) values. */
#define ZKMI_GROTH16_STAGES 11
int zkmi_groth16_stage_ms(double* out, int n);

/* Two short probes of the device a run landed on, for reading benchmark lines (no reference counterpart): Montgomery products per second on
 * 29-bit limbs (two dependent chains per lane, 8 workgroups per CU: about 150 G products/s on a healthy MI355X) and 16 dependent random 128-byte
 * gathers per lane over a 2 GiB table (about 6 TB/s on a healthy box). */
int zkmi_calibrate_box(double* mul29_gmul_per_s, double* gather128_gb_per_s);
/* Third probe: the same product chain as straight-line loops of ~17 KB and ~210 KB of code at two waves per SIMD; big / small < 1 is the cost of
 * instruction fetch beyond the 64 KB instruction cache on this box (the accumulation loops of the 14-limb curve are that large). */
int zkmi_calibrate_code_fetch(double* small_loop_gmul_per_s, double* big_loop_gmul_per_s);
/* Which MSM kernels run their compact instantiation (products called instead of inlined: loops that fit the instruction cache) on this box:
 * bit 0 G1 accumulation of the 14-limb curve, 1 its G2 accumulation, 2 G1 row/column sums, 3 Fq2 row/column sums by the generic kernel, 4 PLONK's quotient
 * numerator by the 32-bit kernels with called products; decided once from the probe above — bits 1, 2, 3 when big / small < 0.85 (r05: the loops behind bits 0
 * and 4 fit the instruction cache since r04 and measured faster inlined on such a box) — unless ZKMI_COMPACT_CODE=<mask> is set. Results are bit-identical
 * either way. -1: no device. */
int zkmi_compact_code(void);

/* Synthetic base table of SURVEY.md §8d: P_i = (f*g^i mod r)*G written to device memory as affine Montgomery points
 * (what G.batchApplyKey(G repeated n, Fr.e(f), Fr.e(g)) returns).  For benchmarks and tests. */
int zkmi_gen_geometric_bases_dev(int curve, int group, size_t n, uint64_t f, uint64_t g, void* d_out);
/* P_i = k_i * G for n caller-supplied scalars (device, 32-byte little-endian integers, normal form), affine Montgomery points out:
 * the base sections of synthetic VALID proving keys built from a known trapdoor (src/zkey_new.js:182-201, :338-502
 * compute the same points from a ptau file). For tests and benchmarks. */
int zkmi_gen_bases_from_scalars_dev(int curve, int group, const void* d_scalars, size_t n, void* d_out);

/* Wall-clock-free device timing of the last call of each kind, in milliseconds (HIP events on the library stream). */
double zkmi_last_kernel_ms(void);
/* zkmi_msm_dev calls since start-up that ran on the saturated-limb path because the scratch copy of the bases (n x 2*group*n8q bytes) could not be allocated
 * (out of memory ONLY: any other failure is reported by the call itself). A non-zero count explains a slow standalone MSM. */
unsigned long long zkmi_msm_dev_fallbacks(void);
/* Shape of the resident coefficient layout of a Groth16 key (csrc/groth16.hip: buildABC as a length-sorted sliced layout with split rows):
 * out[0..5) = coefficient records, segments (<= 32 terms each), rows cut into several segments, partial-sum slots, padded terms held. */
int zkmi_groth16_coef_layout(uint64_t zkey_cache_key, uint64_t* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI_DIAG_H */
