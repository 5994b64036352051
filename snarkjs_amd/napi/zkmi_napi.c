/*
 * snarkjs_amd/napi/zkmi_napi.c — thin N-API (C) binding of libzkmi.so (include/zkmi.h) for Node.js.
 *
 * One exported function per C-ABI entry point; no arithmetic here.  Buffers cross as Uint8Array or as an array of
 * Uint8Array pages (ffjavascript's BigBuffer.buffers, reference bundle build/snarkjs.min.js:1@183423).
 * snarkjs_amd/js/register.js turns these into the curve.G1.multiExpAffine / curve.Fr.fft ... surface snarkjs calls.
 *
 * Build (snarkjs_amd/build.py):  gcc -shared -fPIC -I/usr/include/node zkmi_napi.c -o zkmi_napi.node -L.. -lzkmi
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <node_api.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/zkmi.h"

#define MAX_PAGES 256

/* The library serves one caller at a time (include/zkmi.h): every entry point is taken under this lock, on the main thread and on
 * the libuv pool threads of the *Async functions alike. */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
#define ZK_CALL(expr) ({ pthread_mutex_lock(&g_lock); int rc__ = (expr); pthread_mutex_unlock(&g_lock); rc__; })

#define NAPI_OK(call)                                                              \
    do {                                                                           \
        if ((call) != napi_ok) { napi_throw_error(env, NULL, "zkmi: N-API call failed: " #call); return NULL; } \
    } while (0)

static napi_value throw_zkmi(napi_env env, int rc) {
    char msg[512];
    const char* e = zkmi_last_error();
    snprintf(msg, sizeof msg, "zkmi error %d: %s", rc, e ? e : "");
    napi_throw_error(env, NULL, msg);
    return NULL;
}

typedef struct {
    const uint8_t* ptr[MAX_PAGES];
    size_t len[MAX_PAGES];
    int n;
    size_t total;
} pages_t;

/* v: Uint8Array | Array<Uint8Array>; with gaps != 0 an array element may also be a NUMBER = a gap of that many bytes the caller did not read
 * (pointer NULL: include/zkmi.h, zkmi_groth16_zkey_paged); a BigBuffer of ffjavascript is passed as its .buffers array by the JS side */
static int get_pages_ex(napi_env env, napi_value v, pages_t* out, int gaps) {
    bool is_arr = false, is_ta = false;
    out->n = 0; out->total = 0;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok) return -1;
    if (is_ta) {
        napi_typedarray_type t; size_t len; void* data; napi_value ab; size_t off;
        if (napi_get_typedarray_info(env, v, &t, &len, &data, &ab, &off) != napi_ok || t != napi_uint8_array) return -1;
        out->ptr[0] = (const uint8_t*)data; out->len[0] = len; out->n = 1; out->total = len;
        return 0;
    }
    if (napi_is_array(env, v, &is_arr) != napi_ok || !is_arr) return -1;
    uint32_t k = 0;
    if (napi_get_array_length(env, v, &k) != napi_ok || k > MAX_PAGES) return -1;
    for (uint32_t i = 0; i < k; i++) {
        napi_value e; napi_typedarray_type t; size_t len; void* data; napi_value ab; size_t off; napi_valuetype vt;
        if (napi_get_element(env, v, i, &e) != napi_ok || napi_typeof(env, e, &vt) != napi_ok) return -1;
        if (vt == napi_number) {
            double g;
            if (!gaps || napi_get_value_double(env, e, &g) != napi_ok || g < 0 || g > 9007199254740992.0) return -1;
            out->ptr[i] = NULL; out->len[i] = (size_t)g; out->total += (size_t)g;
            continue;
        }
        if (napi_get_typedarray_info(env, e, &t, &len, &data, &ab, &off) != napi_ok || t != napi_uint8_array) return -1;
        /* an empty typed array may report a NULL data pointer: keep the page, never as a gap */
        out->ptr[i] = len ? (const uint8_t*)data : (const uint8_t*)""; out->len[i] = len; out->total += len;
    }
    out->n = (int)k;
    return 0;
}
static int get_pages(napi_env env, napi_value v, pages_t* out) { return get_pages_ex(env, v, out, 0); }
static zkmi_pages as_zk(const pages_t* p) { zkmi_pages z; z.ptr = p->ptr; z.len = p->len; z.n_pages = p->n; return z; }

static int get_i32(napi_env env, napi_value v, int32_t* o) { return napi_get_value_int32(env, v, o) == napi_ok ? 0 : -1; }
static int get_f64(napi_env env, napi_value v, double* o) { return napi_get_value_double(env, v, o) == napi_ok ? 0 : -1; }
/* optional 32-byte element: null/undefined -> NULL */
static int get_opt32(napi_env env, napi_value v, const uint8_t** o) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return -1;
    if (t == napi_null || t == napi_undefined) { *o = NULL; return 0; }
    pages_t p;
    if (get_pages(env, v, &p) || p.n != 1 || p.len[0] != 32) return -1;
    *o = p.ptr[0];
    return 0;
}
static napi_value new_u8(napi_env env, size_t n, uint8_t** data) {
    napi_value ab, ta;
    void* d;
    if (napi_create_arraybuffer(env, n, &d, &ab) != napi_ok) return NULL;
    if (napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta) != napi_ok) return NULL;
    *data = (uint8_t*)d;
    return ta;
}
#define ARGS(N)                                                           \
    size_t argc = N; napi_value argv[N];                                  \
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));        \
    if (argc < N) { napi_throw_type_error(env, NULL, "zkmi: too few arguments"); return NULL; }
#define BAD_ARG() do { napi_throw_type_error(env, NULL, "zkmi: bad argument"); return NULL; } while (0)

/* A Groth16 zkey descriptor as the addon holds it: the paged C struct plus the page arrays its zkmi_pages point into (a copy must re-point them:
 * zkey_desc_fix). Sections 4 - 9 arrive as Uint8Array | Array<Uint8Array | gapBytes> — what binFileUtils.readSection returns (a Uint8Array, or a
 * BigBuffer's .buffers from 2^30 bytes on; src/groth16_prove.js:57-59). */
typedef struct zkey_desc {
    zkmi_groth16_zkey_paged z;
    pages_t sec[6];                        /* coeffs, A, B1, B2, C, H */
} zkey_desc;
static void zkey_desc_fix(zkey_desc* d) {
    zkmi_pages* dst[6] = {&d->z.coeffs, &d->z.bases_a, &d->z.bases_b1, &d->z.bases_b2, &d->z.bases_c, &d->z.bases_h};
    for (int i = 0; i < 6; i++) { dst[i]->ptr = d->sec[i].ptr; dst[i]->len = d->sec[i].len; dst[i]->n_pages = d->sec[i].n; }
}
static zkey_desc* zkey_desc_dup(const zkey_desc* d) {
    zkey_desc* c = (zkey_desc*)malloc(sizeof *c);
    if (!c) return NULL;
    memcpy(c, d, sizeof *c);
    zkey_desc_fix(c);
    return c;
}

/* ---- asynchronous calls (napi_create_async_work; SURVEY.md 8 b) -----------------------------------------------------------------
 * msmAsync / nttAsync / groth16ProveAsync take the arguments of msm / ntt / groth16Prove and return a Promise: arguments are parsed
 * on the main thread, the library call runs on a libuv pool thread (the Node event loop keeps turning), the promise is settled
 * back on the main thread. Input and output buffers stay referenced until then; the caller must not write to them meanwhile. */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref keep[8]; int n_keep;          /* arguments and result kept alive while the job runs */
    napi_ref result;                       /* value the promise resolves to (NULL: undefined) */
    int kind;                              /* 0 msm, 1 ntt, 2 groth16Prove, 3 groth16Submit, 4 groth16Collect, 5 groth16Load, 6 msmTableMultiDev, 7 synchronize */
    double handle; const void* dptr[4]; size_t dk[4]; int dcnt;      /* kind 6: table handle, device scalars, term counts */
    int32_t slot;
    int32_t curve, group, logn, inverse;
    pages_t a, b;
    double n, sb, key;
    uint8_t first[32], inc[32]; bool has_first, has_inc;
    struct zkey_desc* zk; bool has_zk;      /* heap copy of the paged descriptor (freed in job_complete) */
    uint8_t *o0, *o1, *o2;
    int rc; char err[400];
} job_t;

static int job_run(job_t* j) {
    switch (j->kind) {
    case 0: return zkmi_msm(j->curve, j->group, as_zk(&j->a), as_zk(&j->b), (size_t)j->n, (size_t)j->sb, (uint64_t)j->key, j->o0);
    case 1: return zkmi_ntt(j->curve, as_zk(&j->a), (uint8_t* const*)j->b.ptr, j->b.len, j->b.n, (unsigned)j->logn, j->inverse, j->has_first ? j->first : NULL, j->has_inc ? j->inc : NULL);
    case 2: return zkmi_groth16_prove_paged(j->has_zk ? &j->zk->z : NULL, (uint64_t)j->key, j->a.ptr[0], j->a.len[0], j->first, j->inc, j->o0, j->o1, j->o2);
    case 3: return zkmi_groth16_submit((uint64_t)j->key, j->a.ptr[0], j->a.len[0], j->slot);
    case 4: return zkmi_groth16_collect((uint64_t)j->key, j->slot, j->first, j->inc, j->o0, j->o1, j->o2);
    case 5: return zkmi_groth16_load_paged(&j->zk->z, (uint64_t)j->key);
    default: {
        /* the round drivers of plonk.prove / fflonk.prove (js/plonk_native.js: proveAsync): the call that makes the host WAIT — the commitments of a
         * round, or the queued work before a read-back — runs here on a pool thread, in the pipeline slot of the proof it belongs to */
        const int prev = zkmi_pipeline_active();
        int rc = zkmi_pipeline_select(j->slot);
        if (!rc) rc = j->kind == 6 ? zkmi_msm_table_multi_dev((uint64_t)j->handle, j->dptr, j->dk, j->dcnt, (size_t)j->sb, j->o0) : zkmi_synchronize();
        (void)zkmi_pipeline_select(prev);
        return rc;
    }
    }
}
static void job_execute(napi_env env, void* data) {
    (void)env;
    job_t* j = (job_t*)data;
    pthread_mutex_lock(&g_lock);
    j->rc = job_run(j);
    if (j->rc) { const char* e = zkmi_last_error(); snprintf(j->err, sizeof j->err, "zkmi error %d: %s", j->rc, e ? e : ""); }   /* thread-local: read it here */
    pthread_mutex_unlock(&g_lock);
}
static void job_complete(napi_env env, napi_status status, void* data) {
    job_t* j = (job_t*)data;
    napi_value v = NULL, msg, err;
    if (status == napi_ok && j->rc == 0) {
        if (j->result) napi_get_reference_value(env, j->result, &v); else napi_get_undefined(env, &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_create_string_utf8(env, status == napi_ok ? j->err : "zkmi: asynchronous call cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
    }
    for (int i = 0; i < j->n_keep; i++) napi_delete_reference(env, j->keep[i]);
    if (j->result) napi_delete_reference(env, j->result);
    napi_delete_async_work(env, j->work);
    free(j->zk);
    free(j);
}
/* queue `j` (heap, filled by the caller); keeps argv[0..argc) and `result` alive; returns the promise */
static napi_value job_queue(napi_env env, job_t* j, const char* name, napi_value* argv, size_t argc, napi_value result) {
    napi_value promise, rname;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) { free(j->zk); free(j); napi_throw_error(env, NULL, "zkmi: cannot create a promise"); return NULL; }
    for (size_t i = 0; i < argc && j->n_keep < 8; i++) {
        napi_valuetype t;
        if (napi_typeof(env, argv[i], &t) == napi_ok && t == napi_object && napi_create_reference(env, argv[i], 1, &j->keep[j->n_keep]) == napi_ok) j->n_keep++;
    }
    if (result) napi_create_reference(env, result, 1, &j->result);
    napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rname);
    if (napi_create_async_work(env, NULL, rname, job_execute, job_complete, j, &j->work) != napi_ok || napi_queue_async_work(env, j->work) != napi_ok) {
        napi_value msg, err;
        napi_create_string_utf8(env, "zkmi: cannot queue asynchronous work", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
        for (int i = 0; i < j->n_keep; i++) napi_delete_reference(env, j->keep[i]);
        if (j->result) napi_delete_reference(env, j->result);
        free(j->zk);
        free(j);
    }
    return promise;
}

/* init(device) */
static napi_value js_init(napi_env env, napi_callback_info info) {
    ARGS(1);
    int32_t dev;
    if (get_i32(env, argv[0], &dev)) BAD_ARG();
    int rc = ZK_CALL(zkmi_init(dev));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_device_count(napi_env env, napi_callback_info info) {
    napi_value v;
    NAPI_OK(napi_create_int32(env, zkmi_device_count(), &v));
    return v;
}
static napi_value js_version(napi_env env, napi_callback_info info) {
    napi_value v;
    NAPI_OK(napi_create_string_utf8(env, zkmi_version(), NAPI_AUTO_LENGTH, &v));
    return v;
}
/* msm(curve, group, bases, scalars, n, scalarBytes, cacheKey) -> Uint8Array(3*group*n8q); msmAsync(...) -> Promise of the same */
static napi_value msm_impl(napi_env env, napi_callback_info info, bool async) {
    ARGS(7);
    int32_t curve, group; double n, sb, key;
    pages_t b, s;
    if (get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &group) || get_pages(env, argv[2], &b) || get_pages(env, argv[3], &s) ||
        get_f64(env, argv[4], &n) || get_f64(env, argv[5], &sb) || get_f64(env, argv[6], &key)) BAD_ARG();
    if (group != 1 && group != 2) BAD_ARG();
    uint8_t* out;
    napi_value res = new_u8(env, (size_t)3 * group * (curve == ZKMI_CURVE_BN128 ? 32 : 48), &out);
    if (!res) BAD_ARG();
    if (async) {
        job_t* j = (job_t*)calloc(1, sizeof *j);
        if (!j) BAD_ARG();
        j->kind = 0; j->curve = curve; j->group = group; j->a = b; j->b = s; j->n = n; j->sb = sb; j->key = key; j->o0 = out;
        return job_queue(env, j, "zkmi.msm", argv, 7, res);
    }
    int rc = ZK_CALL(zkmi_msm(curve, group, as_zk(&b), as_zk(&s), (size_t)n, (size_t)sb, (uint64_t)key, out));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_msm(napi_env env, napi_callback_info info) { return msm_impl(env, info, false); }
static napi_value js_msm_async(napi_env env, napi_callback_info info) { return msm_impl(env, info, true); }
static napi_value js_release_bases(napi_env env, napi_callback_info info) {
    ARGS(1);
    double key;
    if (get_f64(env, argv[0], &key)) BAD_ARG();
    (void)ZK_CALL(zkmi_release_bases((uint64_t)key));          /* under the lock: an asynchronous job may be running on a pool thread */
    return NULL;
}
/* ntt(curve, in, out, logN, inverse, first|null, inc|null): out is preallocated by the caller (same container type);
 * nttAsync(...) -> Promise<undefined>, `out` is filled when it resolves */
static napi_value ntt_impl(napi_env env, napi_callback_info info, bool async) {
    ARGS(7);
    int32_t curve, logn, inverse;
    pages_t in, out;
    const uint8_t *first, *inc;
    if (get_i32(env, argv[0], &curve) || get_pages(env, argv[1], &in) || get_pages(env, argv[2], &out) || get_i32(env, argv[3], &logn) ||
        get_i32(env, argv[4], &inverse) || get_opt32(env, argv[5], &first) || get_opt32(env, argv[6], &inc)) BAD_ARG();
    if (async) {
        job_t* j = (job_t*)calloc(1, sizeof *j);
        if (!j) BAD_ARG();
        j->kind = 1; j->curve = curve; j->a = in; j->b = out; j->logn = logn; j->inverse = inverse;
        if (first) { memcpy(j->first, first, 32); j->has_first = true; }
        if (inc) { memcpy(j->inc, inc, 32); j->has_inc = true; }
        return job_queue(env, j, "zkmi.ntt", argv, 7, NULL);
    }
    int rc = ZK_CALL(zkmi_ntt(curve, as_zk(&in), (uint8_t* const*)out.ptr, out.len, out.n, (unsigned)logn, inverse, first, inc));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_ntt(napi_env env, napi_callback_info info) { return ntt_impl(env, info, false); }
static napi_value js_ntt_async(napi_env env, napi_callback_info info) { return ntt_impl(env, info, true); }
/* frBatch(curve, op, in, out, n) */
static napi_value js_fr_batch(napi_env env, napi_callback_info info) {
    ARGS(5);
    int32_t curve, op; double n;
    pages_t in, out;
    if (get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &op) || get_pages(env, argv[2], &in) || get_pages(env, argv[3], &out) || get_f64(env, argv[4], &n)) BAD_ARG();
    int rc = ZK_CALL(zkmi_fr_batch(curve, op, as_zk(&in), (uint8_t* const*)out.ptr, out.len, out.n, (size_t)n));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* applyKey(curve, in, out, n, first, inc) */
static napi_value js_apply_key(napi_env env, napi_callback_info info) {
    ARGS(6);
    int32_t curve; double n;
    pages_t in, out;
    const uint8_t *first, *inc;
    if (get_i32(env, argv[0], &curve) || get_pages(env, argv[1], &in) || get_pages(env, argv[2], &out) || get_f64(env, argv[3], &n) ||
        get_opt32(env, argv[4], &first) || get_opt32(env, argv[5], &inc) || !first || !inc) BAD_ARG();
    int rc = ZK_CALL(zkmi_fr_batch_apply_key(curve, as_zk(&in), (uint8_t* const*)out.ptr, out.len, out.n, (size_t)n, first, inc));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* groupFft(curve, group, in, out, logN, inverse): G.fft / G.ifft over affine points, out preallocated by the caller */
static napi_value js_group_fft(napi_env env, napi_callback_info info) {
    ARGS(6);
    int32_t curve, group, logn, inverse;
    pages_t in, out;
    if (get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &group) || get_pages(env, argv[2], &in) || get_pages(env, argv[3], &out) ||
        get_i32(env, argv[4], &logn) || get_i32(env, argv[5], &inverse)) BAD_ARG();
    int rc = ZK_CALL(zkmi_group_fft(curve, group, as_zk(&in), (uint8_t* const*)out.ptr, out.len, out.n, (unsigned)logn, inverse));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* groupApplyKey(curve, group, in, out, n, first, inc): G.batchApplyKey over affine points */
static napi_value js_group_apply_key(napi_env env, napi_callback_info info) {
    ARGS(7);
    int32_t curve, group; double n;
    pages_t in, out;
    const uint8_t *first, *inc;
    if (get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &group) || get_pages(env, argv[2], &in) || get_pages(env, argv[3], &out) || get_f64(env, argv[4], &n) ||
        get_opt32(env, argv[5], &first) || get_opt32(env, argv[6], &inc) || !first || !inc) BAD_ARG();
    int rc = ZK_CALL(zkmi_group_batch_apply_key(curve, group, as_zk(&in), (uint8_t* const*)out.ptr, out.len, out.n, (size_t)n, first, inc));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* groupConvert(curve, group, kind, in, out, n): G.batchLEMtoU (0) / batchUtoLEM (1) / batchLEMtoC (2) / batchCtoLEM (3) */
static napi_value js_group_convert(napi_env env, napi_callback_info info) {
    ARGS(6);
    int32_t curve, group, kind; double n;
    pages_t in, out;
    if (get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &group) || get_i32(env, argv[2], &kind) || get_pages(env, argv[3], &in) || get_pages(env, argv[4], &out) ||
        get_f64(env, argv[5], &n)) BAD_ARG();
    int rc = ZK_CALL(zkmi_group_convert(curve, group, kind, as_zk(&in), (uint8_t* const*)out.ptr, out.len, out.n, (size_t)n));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* joinABC(curve, a, b, c, out, n) */
static napi_value js_join_abc(napi_env env, napi_callback_info info) {
    ARGS(6);
    int32_t curve; double n;
    pages_t a, b, c, out;
    if (get_i32(env, argv[0], &curve) || get_pages(env, argv[1], &a) || get_pages(env, argv[2], &b) || get_pages(env, argv[3], &c) ||
        get_pages(env, argv[4], &out) || get_f64(env, argv[5], &n)) BAD_ARG();
    int rc = ZK_CALL(zkmi_groth16_join_abc(curve, as_zk(&a), as_zk(&b), as_zk(&c), (uint8_t* const*)out.ptr, out.len, out.n, (size_t)n));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* toAffine(curve, group, jacobian) -> Uint8Array */
static napi_value js_to_affine(napi_env env, napi_callback_info info) {
    ARGS(3);
    int32_t curve, group;
    pages_t j;
    if (get_i32(env, argv[0], &curve) || get_i32(env, argv[1], &group) || get_pages(env, argv[2], &j) || j.n != 1) BAD_ARG();
    const size_t q = curve == ZKMI_CURVE_BN128 ? 32 : 48;
    if ((group != 1 && group != 2) || j.len[0] != 3 * group * q) BAD_ARG();
    uint8_t* out;
    napi_value res = new_u8(env, 2 * group * q, &out);
    if (!res) BAD_ARG();
    int rc = ZK_CALL(zkmi_to_affine(curve, group, j.ptr[0], out));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
/* groth16Prove({curve,nVars,nPublic,domainSize,coeffs,A,B1,B2,C,H,alpha1,beta1,beta2,delta1,delta2} | null, key, witness, r, s)
 *   -> {pi_a, pi_b, pi_c} (affine Montgomery bytes). A section is a Uint8Array or an array of Uint8Array pages (ffjavascript BigBuffer.buffers). */
static int get_named_u8(napi_env env, napi_value obj, const char* name, const uint8_t** p, size_t* len) {
    napi_value v; pages_t pg;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok || get_pages(env, v, &pg) || pg.n != 1) return -1;
    *p = pg.ptr[0]; if (len) *len = pg.len[0];
    return 0;
}
static int get_named_u32(napi_env env, napi_value obj, const char* name, uint32_t* o) {
    napi_value v;
    return (napi_get_named_property(env, obj, name, &v) == napi_ok && napi_get_value_uint32(env, v, o) == napi_ok) ? 0 : -1;
}
/* {curve,nVars,nPublic,domainSize,coeffs,A,B1,B2,C,H,alpha1,beta1,beta2,delta1,delta2} -> zkey_desc (pointers into the typed arrays); the six
 * sections may be single Uint8Arrays or page arrays, base sections also with gaps (numbers) for a shard load */
static int get_named_pages(napi_env env, napi_value obj, const char* name, pages_t* pg, int gaps) {
    napi_value v;
    return (napi_get_named_property(env, obj, name, &v) == napi_ok && get_pages_ex(env, v, pg, gaps) == 0) ? 0 : -1;
}
static int get_zkey_desc(napi_env env, napi_value obj, zkey_desc* d) {
    uint32_t c;
    memset(d, 0, sizeof *d);
    zkmi_groth16_zkey_paged* zk = &d->z;
    if (get_named_u32(env, obj, "curve", &c) || get_named_u32(env, obj, "nVars", &zk->n_vars) || get_named_u32(env, obj, "nPublic", &zk->n_public) ||
        get_named_u32(env, obj, "domainSize", &zk->domain_size) || get_named_pages(env, obj, "coeffs", &d->sec[0], 0) ||
        get_named_pages(env, obj, "A", &d->sec[1], 1) || get_named_pages(env, obj, "B1", &d->sec[2], 1) || get_named_pages(env, obj, "B2", &d->sec[3], 1) ||
        get_named_pages(env, obj, "C", &d->sec[4], 1) || get_named_pages(env, obj, "H", &d->sec[5], 1)) return -1;
    if (c != ZKMI_CURVE_BN128 && c != ZKMI_CURVE_BLS12381) return -1;
    zk->curve = (int)c;
    zkey_desc_fix(d);
    /* header points: 2*n8q (G1) / 4*n8q (G2) bytes each */
    const size_t q = c == ZKMI_CURVE_BN128 ? 32 : 48;
    size_t l1, l2, l3, l4, l5;
    if (get_named_u8(env, obj, "alpha1", &zk->vk_alpha_1, &l1) || get_named_u8(env, obj, "beta1", &zk->vk_beta_1, &l2) || get_named_u8(env, obj, "beta2", &zk->vk_beta_2, &l3) ||
        get_named_u8(env, obj, "delta1", &zk->vk_delta_1, &l4) || get_named_u8(env, obj, "delta2", &zk->vk_delta_2, &l5) || l1 < 2 * q || l2 < 2 * q || l3 < 4 * q || l4 < 2 * q || l5 < 4 * q) return -1;
    return 0;
}
/* {pi_a, pi_b, pi_c} with fresh Uint8Arrays of 2 / 4 / 2 x n8q bytes */
static napi_value new_proof_obj(napi_env env, int curve, uint8_t** pa, uint8_t** pb, uint8_t** pc) {
    const size_t q = curve == ZKMI_CURVE_BN128 ? 32 : 48;
    napi_value va = new_u8(env, 2 * q, pa), vb = new_u8(env, 4 * q, pb), vc = new_u8(env, 2 * q, pc), res;
    if (!va || !vb || !vc || napi_create_object(env, &res) != napi_ok) return NULL;
    if (napi_set_named_property(env, res, "pi_a", va) != napi_ok || napi_set_named_property(env, res, "pi_b", vb) != napi_ok || napi_set_named_property(env, res, "pi_c", vc) != napi_ok) return NULL;
    return res;
}
static int key_curve_mismatch(napi_env env, double key, int32_t curve);
static napi_value groth16_impl(napi_env env, napi_callback_info info, bool async) {
    ARGS(5);
    zkey_desc zk, *pzk = NULL;
    napi_valuetype t;
    NAPI_OK(napi_typeof(env, argv[0], &t));
    double key;
    pages_t w;
    const uint8_t *r, *s;
    int curve = 0;
    if (t == napi_object) {
        if (get_zkey_desc(env, argv[0], &zk)) BAD_ARG();
        curve = zk.z.curve; pzk = &zk;
    }
    if (get_f64(env, argv[1], &key) || get_pages(env, argv[2], &w) || w.n != 1 || get_opt32(env, argv[3], &r) || get_opt32(env, argv[4], &s) || !r || !s) BAD_ARG();
    if (!pzk) {            /* key already resident: the caller passes the curve id in place of the zkey object */
        int32_t c;
        if (get_i32(env, argv[0], &c)) BAD_ARG();
        curve = c;
    }
    if ((curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || key_curve_mismatch(env, key, curve)) { if (curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) BAD_ARG(); return NULL; }
    uint8_t *pa, *pb, *pc;
    napi_value res = new_proof_obj(env, curve, &pa, &pb, &pc);
    if (!res) BAD_ARG();
    if (async) {           /* the zkey sections are referenced through argv[0] (the descriptor object holds the typed arrays) */
        job_t* j = (job_t*)calloc(1, sizeof *j);
        if (!j) BAD_ARG();
        j->kind = 2; j->key = key; j->a = w; j->o0 = pa; j->o1 = pb; j->o2 = pc;
        memcpy(j->first, r, 32); memcpy(j->inc, s, 32);
        if (pzk) { j->zk = zkey_desc_dup(pzk); if (!j->zk) { free(j); BAD_ARG(); } j->has_zk = true; }
        return job_queue(env, j, "zkmi.groth16Prove", argv, 5, res);
    }
    int rc = ZK_CALL(zkmi_groth16_prove_paged(pzk ? &pzk->z : NULL, (uint64_t)key, w.ptr[0], w.len[0], r, s, pa, pb, pc));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_groth16_prove(napi_env env, napi_callback_info info) { return groth16_impl(env, info, false); }
static napi_value js_groth16_prove_async(napi_env env, napi_callback_info info) { return groth16_impl(env, info, true); }
static napi_value js_groth16_release(napi_env env, napi_callback_info info) {
    ARGS(1);
    double key;
    if (get_f64(env, argv[0], &key)) BAD_ARG();
    (void)ZK_CALL(zkmi_groth16_release((uint64_t)key));        /* waits for the lock: never frees a key under a running job */
    return NULL;
}

/* ---- resident keys, the two-slot pipeline and the multi-GPU shard API: one typed wrapper per C entry point --------------------------------
 * groth16Load(desc, key) / groth16LoadAsync(desc, key) -> undefined | Promise; groth16LoadShard(desc, key, varLo, varHi, hLo, hHi) */
static napi_value load_impl(napi_env env, napi_callback_info info, bool async) {
    ARGS(2);
    double key;
    zkey_desc zk;
    if (get_zkey_desc(env, argv[0], &zk) || get_f64(env, argv[1], &key) || key < 1) BAD_ARG();
    if (async) {
        job_t* j = (job_t*)calloc(1, sizeof *j);
        if (!j) BAD_ARG();
        j->kind = 5; j->key = key; j->zk = zkey_desc_dup(&zk); if (!j->zk) { free(j); BAD_ARG(); } j->has_zk = true;
        return job_queue(env, j, "zkmi.groth16Load", argv, 2, NULL);
    }
    int rc = ZK_CALL(zkmi_groth16_load_paged(&zk.z, (uint64_t)key));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_groth16_load(napi_env env, napi_callback_info info) { return load_impl(env, info, false); }
static napi_value js_groth16_load_async(napi_env env, napi_callback_info info) { return load_impl(env, info, true); }
static napi_value js_groth16_load_shard(napi_env env, napi_callback_info info) {
    ARGS(6);
    double key, a, b, c, d;
    zkey_desc zk;
    if (get_zkey_desc(env, argv[0], &zk) || get_f64(env, argv[1], &key) || key < 1 || get_f64(env, argv[2], &a) || get_f64(env, argv[3], &b) || get_f64(env, argv[4], &c) ||
        get_f64(env, argv[5], &d) || a < 0 || b < a || c < 0 || d < c || b > 4294967295.0 || d > 4294967295.0) BAD_ARG();
    int rc = ZK_CALL(zkmi_groth16_load_shard_paged(&zk.z, (uint64_t)key, (uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)d));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* groth16Submit(key, witness, slot) / groth16SubmitAsync: the witness crosses PCIe on the slot's stream and the device part of the proof is
 * enqueued (zkmi_groth16_submit); groth16Collect(curve, key, slot, r, s) / groth16CollectAsync -> {pi_a, pi_b, pi_c} waits for the slot. */
static napi_value submit_impl(napi_env env, napi_callback_info info, bool async) {
    ARGS(3);
    double key; int32_t slot; pages_t w;
    if (get_f64(env, argv[0], &key) || get_pages(env, argv[1], &w) || w.n != 1 || get_i32(env, argv[2], &slot) || (slot != 0 && slot != 1)) BAD_ARG();
    if (async) {
        job_t* j = (job_t*)calloc(1, sizeof *j);
        if (!j) BAD_ARG();
        j->kind = 3; j->key = key; j->a = w; j->slot = slot;
        return job_queue(env, j, "zkmi.groth16Submit", argv, 3, NULL);
    }
    int rc = ZK_CALL(zkmi_groth16_submit((uint64_t)key, w.ptr[0], w.len[0], slot));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_groth16_submit(napi_env env, napi_callback_info info) { return submit_impl(env, info, false); }
static napi_value js_groth16_submit_async(napi_env env, napi_callback_info info) { return submit_impl(env, info, true); }
/* The library writes results in the format of the RESIDENT KEY's curve: a caller-supplied curve id that disagrees with it would size the
 * output arrays wrongly (curve 0 with a BLS12-381 key: 96 / 192 / 1008 bytes written into 64 / 128 / 672). Checked before every such call. */
static int key_curve_mismatch(napi_env env, double key, int32_t curve) {
    const int kc = ZK_CALL(zkmi_groth16_key_curve((uint64_t)key));   /* -1: not resident (the call itself will say so) */
    if (kc >= 0 && kc != curve) { napi_throw_type_error(env, NULL, "zkmi: the curve argument differs from the curve of the resident key"); return 1; }
    return 0;
}
static napi_value collect_impl(napi_env env, napi_callback_info info, bool async) {
    ARGS(5);
    int32_t curve, slot; double key;
    const uint8_t *r, *s;
    if (get_i32(env, argv[0], &curve) || get_f64(env, argv[1], &key) || get_i32(env, argv[2], &slot) || (slot != 0 && slot != 1) || get_opt32(env, argv[3], &r) ||
        get_opt32(env, argv[4], &s) || !r || !s || (curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381)) BAD_ARG();
    if (key_curve_mismatch(env, key, curve)) return NULL;
    uint8_t *pa, *pb, *pc;
    napi_value res = new_proof_obj(env, curve, &pa, &pb, &pc);
    if (!res) BAD_ARG();
    if (async) {
        job_t* j = (job_t*)calloc(1, sizeof *j);
        if (!j) BAD_ARG();
        j->kind = 4; j->key = key; j->slot = slot; j->o0 = pa; j->o1 = pb; j->o2 = pc;
        memcpy(j->first, r, 32); memcpy(j->inc, s, 32);
        return job_queue(env, j, "zkmi.groth16Collect", argv, 5, res);
    }
    int rc = ZK_CALL(zkmi_groth16_collect((uint64_t)key, slot, r, s, pa, pb, pc));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_groth16_collect(napi_env env, napi_callback_info info) { return collect_impl(env, info, false); }
static napi_value js_groth16_collect_async(napi_env env, napi_callback_info info) { return collect_impl(env, info, true); }

/* device memory for the shard driver: devAlloc(bytes) -> pointer (a Number: device addresses fit 53 bits), devFree(ptr),
 * memcpyH2D(ptr, Uint8Array), memcpyD2H(Uint8Array, ptr) */
static int get_dptr(napi_env env, napi_value v, void** o) {
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok || d < 0 || d > 9007199254740992.0) return -1;
    *o = (void*)(uintptr_t)d;
    return 0;
}
static napi_value js_dev_alloc(napi_env env, napi_callback_info info) {
    ARGS(1);
    double bytes; void* p = NULL; napi_value v;
    if (get_f64(env, argv[0], &bytes) || bytes < 0) BAD_ARG();
    int rc = ZK_CALL(zkmi_dev_alloc((size_t)bytes, &p));
    if (rc) return throw_zkmi(env, rc);
    NAPI_OK(napi_create_double(env, (double)(uintptr_t)p, &v));
    return v;
}
static napi_value js_dev_free(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* p;
    if (get_dptr(env, argv[0], &p)) BAD_ARG();
    int rc = ZK_CALL(zkmi_dev_free(p));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_memcpy_h2d(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* p; pages_t h;
    if (get_dptr(env, argv[0], &p) || get_pages(env, argv[1], &h) || h.n != 1) BAD_ARG();
    int rc = h.len[0] ? ZK_CALL(zkmi_memcpy_h2d(p, h.ptr[0], h.len[0])) : 0;
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_memcpy_d2h(napi_env env, napi_callback_info info) {
    ARGS(2);
    void* p; pages_t h;
    if (get_pages(env, argv[0], &h) || h.n != 1 || get_dptr(env, argv[1], &p)) BAD_ARG();
    int rc = h.len[0] ? ZK_CALL(zkmi_memcpy_d2h((void*)h.ptr[0], p, h.len[0])) : 0;
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* groth16ChainsDev(key, dWitness, chainMask, dA|0, dB|0, dC|0); groth16SumsWDev(key, dWitness);
 * groth16SumsHDev(curve, key, dWitness, dH) / groth16SumsDev(curve, key, dWitness) -> Uint8Array(7*3*n8q) partial sums;
 * groth16Finish(curve, key, sums, r, s) -> {pi_a, pi_b, pi_c}; joinABCDev(curve, dA, dB, dC, dOut, n); pointAdd(curve, group, a, b) -> Uint8Array */
static napi_value js_groth16_chains_dev(napi_env env, napi_callback_info info) {
    ARGS(6);
    double key; int32_t mask; void *w, *a, *b, *c;
    if (get_f64(env, argv[0], &key) || get_dptr(env, argv[1], &w) || get_i32(env, argv[2], &mask) || mask < 0 || mask > 7 || get_dptr(env, argv[3], &a) || get_dptr(env, argv[4], &b) ||
        get_dptr(env, argv[5], &c)) BAD_ARG();
    int rc = ZK_CALL(zkmi_groth16_chains_dev((uint64_t)key, w, (unsigned)mask, a, b, c));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_groth16_sums_w_dev(napi_env env, napi_callback_info info) {
    ARGS(2);
    double key; void* w;
    if (get_f64(env, argv[0], &key) || get_dptr(env, argv[1], &w)) BAD_ARG();
    int rc = ZK_CALL(zkmi_groth16_sums_w_dev((uint64_t)key, w));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value sums_impl(napi_env env, napi_callback_info info, bool with_h) {
    size_t argc = 4; napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < (with_h ? 4u : 3u)) { napi_throw_type_error(env, NULL, "zkmi: too few arguments"); return NULL; }
    int32_t curve; double key; void *w, *h = NULL;
    if (get_i32(env, argv[0], &curve) || (curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || get_f64(env, argv[1], &key) || get_dptr(env, argv[2], &w) ||
        (with_h && get_dptr(env, argv[3], &h))) BAD_ARG();
    if (key_curve_mismatch(env, key, curve)) return NULL;
    uint8_t* out;
    napi_value res = new_u8(env, (size_t)21 * (curve == ZKMI_CURVE_BN128 ? 32 : 48), &out);
    if (!res) BAD_ARG();
    int rc = with_h ? ZK_CALL(zkmi_groth16_sums_h_dev((uint64_t)key, w, h, out)) : ZK_CALL(zkmi_groth16_sums_dev((uint64_t)key, w, out));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_groth16_sums_h_dev(napi_env env, napi_callback_info info) { return sums_impl(env, info, true); }
static napi_value js_groth16_sums_dev(napi_env env, napi_callback_info info) { return sums_impl(env, info, false); }
static napi_value js_groth16_finish(napi_env env, napi_callback_info info) {
    ARGS(5);
    int32_t curve; double key; pages_t sm; const uint8_t *r, *s;
    if (get_i32(env, argv[0], &curve) || (curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || get_f64(env, argv[1], &key) || get_pages(env, argv[2], &sm) || sm.n != 1 ||
        sm.len[0] != (size_t)21 * (curve == ZKMI_CURVE_BN128 ? 32 : 48) || get_opt32(env, argv[3], &r) || get_opt32(env, argv[4], &s) || !r || !s) BAD_ARG();
    if (key_curve_mismatch(env, key, curve)) return NULL;
    uint8_t *pa, *pb, *pc;
    napi_value res = new_proof_obj(env, curve, &pa, &pb, &pc);
    if (!res) BAD_ARG();
    int rc = ZK_CALL(zkmi_groth16_finish((uint64_t)key, sm.ptr[0], r, s, pa, pb, pc));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_join_abc_dev(napi_env env, napi_callback_info info) {
    ARGS(6);
    int32_t curve; double n; void *a, *b, *c, *o;
    if (get_i32(env, argv[0], &curve) || get_dptr(env, argv[1], &a) || get_dptr(env, argv[2], &b) || get_dptr(env, argv[3], &c) || get_dptr(env, argv[4], &o) || get_f64(env, argv[5], &n) || n < 0) BAD_ARG();
    int rc = ZK_CALL(zkmi_groth16_join_abc_dev(curve, a, b, c, o, (size_t)n));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_point_add(napi_env env, napi_callback_info info) {
    ARGS(4);
    int32_t curve, group; pages_t a, b;
    if (get_i32(env, argv[0], &curve) || (curve != ZKMI_CURVE_BN128 && curve != ZKMI_CURVE_BLS12381) || get_i32(env, argv[1], &group) || (group != 1 && group != 2) ||
        get_pages(env, argv[2], &a) || get_pages(env, argv[3], &b) || a.n != 1 || b.n != 1) BAD_ARG();
    const size_t len = (size_t)3 * group * (curve == ZKMI_CURVE_BN128 ? 32 : 48);
    if (a.len[0] != len || b.len[0] != len) BAD_ARG();
    uint8_t* out;
    napi_value res = new_u8(env, len, &out);
    if (!res) BAD_ARG();
    int rc = zkmi_point_add(curve, group, a.ptr[0], b.ptr[0], out);        /* host only, no library state */
    if (rc) return throw_zkmi(env, rc);
    return res;
}

/* msmTableDev(handle, dScalars, k, scalarBytes) -> Uint8Array(3*group*n8q); msmTableMultiDev(handle, [dScalars...], [k...], scalarBytes) ->
 * Uint8Array(count * 3*group*n8q): the result buffers are sized from the TABLE (zkmi_msm_table_info), never from a caller-supplied length. */
static napi_value js_msm_table_dev(napi_env env, napi_callback_info info) {
    ARGS(4);
    double h, k; void* sc; int32_t sb; int curve, group;
    if (get_f64(env, argv[0], &h) || get_dptr(env, argv[1], &sc) || get_f64(env, argv[2], &k) || k < 0 || get_i32(env, argv[3], &sb)) BAD_ARG();
    int rc = ZK_CALL(zkmi_msm_table_info((uint64_t)h, &curve, &group, NULL));
    if (rc) return throw_zkmi(env, rc);
    uint8_t* out;
    napi_value res = new_u8(env, (size_t)3 * group * (curve == ZKMI_CURVE_BN128 ? 32 : 48), &out);
    if (!res) BAD_ARG();
    rc = ZK_CALL(zkmi_msm_table_dev((uint64_t)h, sc, (size_t)k, (size_t)sb, out));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_msm_table_multi_dev(napi_env env, napi_callback_info info) {
    ARGS(4);
    double h; int32_t sb; int curve, group; uint32_t cnt = 0, cnt2 = 0;
    bool a0 = false, a1 = false;
    if (get_f64(env, argv[0], &h) || napi_is_array(env, argv[1], &a0) != napi_ok || !a0 || napi_is_array(env, argv[2], &a1) != napi_ok || !a1 ||
        napi_get_array_length(env, argv[1], &cnt) != napi_ok || napi_get_array_length(env, argv[2], &cnt2) != napi_ok || cnt != cnt2 || cnt < 1 || cnt > 4 ||
        get_i32(env, argv[3], &sb)) BAD_ARG();
    const void* ptrs[4]; size_t ks[4];
    for (uint32_t i = 0; i < cnt; i++) {
        napi_value e; void* p; double k;
        if (napi_get_element(env, argv[1], i, &e) != napi_ok || get_dptr(env, e, &p) || napi_get_element(env, argv[2], i, &e) != napi_ok || get_f64(env, e, &k) || k < 0) BAD_ARG();
        ptrs[i] = p; ks[i] = (size_t)k;
    }
    int rc = ZK_CALL(zkmi_msm_table_info((uint64_t)h, &curve, &group, NULL));
    if (rc) return throw_zkmi(env, rc);
    uint8_t* out;
    napi_value res = new_u8(env, (size_t)cnt * 3 * group * (curve == ZKMI_CURVE_BN128 ? 32 : 48), &out);
    if (!res) BAD_ARG();
    rc = ZK_CALL(zkmi_msm_table_multi_dev((uint64_t)h, ptrs, ks, (int)cnt, (size_t)sb, out));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
/* msmTableMultiEnqueueDev(handle, [dScalars...], [k...], scalarBytes): the MSMs of a round are put on the active pipeline slot's streams, nothing waits;
 * msmTableMultiCollect(handle, count) -> Uint8Array(count * 3*group*n8q) waits for them (zkmi_msm_table_multi_enqueue_dev / _collect) */
static napi_value js_msm_table_multi_enqueue_dev(napi_env env, napi_callback_info info) {
    ARGS(4);
    double h; int32_t sb; uint32_t cnt = 0, cnt2 = 0;
    bool a0 = false, a1 = false;
    if (get_f64(env, argv[0], &h) || napi_is_array(env, argv[1], &a0) != napi_ok || !a0 || napi_is_array(env, argv[2], &a1) != napi_ok || !a1 ||
        napi_get_array_length(env, argv[1], &cnt) != napi_ok || napi_get_array_length(env, argv[2], &cnt2) != napi_ok || cnt != cnt2 || cnt < 1 || cnt > 4 ||
        get_i32(env, argv[3], &sb)) BAD_ARG();
    const void* ptrs[4]; size_t ks[4];
    for (uint32_t i = 0; i < cnt; i++) {
        napi_value e; void* p; double k;
        if (napi_get_element(env, argv[1], i, &e) != napi_ok || get_dptr(env, e, &p) || napi_get_element(env, argv[2], i, &e) != napi_ok || get_f64(env, e, &k) || k < 0) BAD_ARG();
        ptrs[i] = p; ks[i] = (size_t)k;
    }
    int rc = ZK_CALL(zkmi_msm_table_multi_enqueue_dev((uint64_t)h, ptrs, ks, (int)cnt, (size_t)sb));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* msmTableMultiEnqueueMontDev(handle, [dPolys...], [k...]): the same for Montgomery coefficient arrays — their batchFromMontgomery (one launch for the round) and the MSMs
 * (zkmi_msm_table_multi_enqueue_mont_dev); collected by msmTableMultiCollect */
static napi_value js_msm_table_multi_enqueue_mont_dev(napi_env env, napi_callback_info info) {
    ARGS(3);
    double h; uint32_t cnt = 0, cnt2 = 0;
    bool a0 = false, a1 = false;
    if (get_f64(env, argv[0], &h) || napi_is_array(env, argv[1], &a0) != napi_ok || !a0 || napi_is_array(env, argv[2], &a1) != napi_ok || !a1 ||
        napi_get_array_length(env, argv[1], &cnt) != napi_ok || napi_get_array_length(env, argv[2], &cnt2) != napi_ok || cnt != cnt2 || cnt < 1 || cnt > 4) BAD_ARG();
    const void* ptrs[4]; size_t ks[4];
    for (uint32_t i = 0; i < cnt; i++) {
        napi_value e; void* p; double k;
        if (napi_get_element(env, argv[1], i, &e) != napi_ok || get_dptr(env, e, &p) || napi_get_element(env, argv[2], i, &e) != napi_ok || get_f64(env, e, &k) || k < 0) BAD_ARG();
        ptrs[i] = p; ks[i] = (size_t)k;
    }
    int rc = ZK_CALL(zkmi_msm_table_multi_enqueue_mont_dev((uint64_t)h, ptrs, ks, (int)cnt));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_msm_table_multi_collect(napi_env env, napi_callback_info info) {
    ARGS(2);
    double h; int32_t cnt; int curve, group;
    if (get_f64(env, argv[0], &h) || get_i32(env, argv[1], &cnt) || cnt < 1 || cnt > 4) BAD_ARG();
    int rc = ZK_CALL(zkmi_msm_table_info((uint64_t)h, &curve, &group, NULL));
    if (rc) return throw_zkmi(env, rc);
    uint8_t* out;
    napi_value res = new_u8(env, (size_t)cnt * 3 * group * (curve == ZKMI_CURVE_BN128 ? 32 : 48), &out);
    if (!res) BAD_ARG();
    rc = ZK_CALL(zkmi_msm_table_multi_collect((uint64_t)h, cnt, out));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
/* msmTableMultiDevAsync(handle, [dScalars...], [k...], scalarBytes, slot) -> Promise<Uint8Array>; synchronizeAsync(slot) -> Promise<undefined>: the two calls on
 * which a host-orchestrated prover waits, on a libuv pool thread (the event loop keeps turning), each in pipeline slot `slot` */
static napi_value js_msm_table_multi_dev_async(napi_env env, napi_callback_info info) {
    ARGS(5);
    double h; int32_t sb, slot; int curve, group; uint32_t cnt = 0, cnt2 = 0;
    bool a0 = false, a1 = false;
    if (get_f64(env, argv[0], &h) || napi_is_array(env, argv[1], &a0) != napi_ok || !a0 || napi_is_array(env, argv[2], &a1) != napi_ok || !a1 ||
        napi_get_array_length(env, argv[1], &cnt) != napi_ok || napi_get_array_length(env, argv[2], &cnt2) != napi_ok || cnt != cnt2 || cnt < 1 || cnt > 4 ||
        get_i32(env, argv[3], &sb) || get_i32(env, argv[4], &slot) || (slot != 0 && slot != 1)) BAD_ARG();
    job_t* j = (job_t*)calloc(1, sizeof *j);
    if (!j) BAD_ARG();
    for (uint32_t i = 0; i < cnt; i++) {
        napi_value e; void* p; double k;
        if (napi_get_element(env, argv[1], i, &e) != napi_ok || get_dptr(env, e, &p) || napi_get_element(env, argv[2], i, &e) != napi_ok || get_f64(env, e, &k) || k < 0) { free(j); BAD_ARG(); }
        j->dptr[i] = p; j->dk[i] = (size_t)k;
    }
    int rc = ZK_CALL(zkmi_msm_table_info((uint64_t)h, &curve, &group, NULL));
    if (rc) { free(j); return throw_zkmi(env, rc); }
    uint8_t* out;
    napi_value res = new_u8(env, (size_t)cnt * 3 * group * (curve == ZKMI_CURVE_BN128 ? 32 : 48), &out);
    if (!res) { free(j); BAD_ARG(); }
    j->kind = 6; j->handle = h; j->dcnt = (int)cnt; j->sb = sb; j->slot = slot; j->o0 = out;
    return job_queue(env, j, "zkmi.msmTableMultiDev", argv, 0, res);
}
static napi_value js_synchronize_async(napi_env env, napi_callback_info info) {
    ARGS(1);
    int32_t slot;
    if (get_i32(env, argv[0], &slot) || (slot != 0 && slot != 1)) BAD_ARG();
    job_t* j = (job_t*)calloc(1, sizeof *j);
    if (!j) BAD_ARG();
    j->kind = 7; j->slot = slot;
    return job_queue(env, j, "zkmi.synchronize", argv, 0, NULL);
}
/* GPU-to-GPU exchange between shard processes (include/zkmi.h: zkmi_ipc_*): ipcExport(ptr) -> Uint8Array(ZKMI_IPC_HANDLE_BYTES);
 * ipcOpen(handle) -> pointer in this process; ipcClose(ptr); peerCopy(dDst, dSrc, bytes) (complete on return); groth16Reset(key);
 * groth16KeyCurve(key) -> 0 | 1 | -1 */
static napi_value js_ipc_export(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* p;
    if (get_dptr(env, argv[0], &p)) BAD_ARG();
    uint8_t* out;
    napi_value res = new_u8(env, ZKMI_IPC_HANDLE_BYTES, &out);
    if (!res) BAD_ARG();
    int rc = ZK_CALL(zkmi_ipc_export(p, out));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_ipc_open(napi_env env, napi_callback_info info) {
    ARGS(1);
    pages_t h; void* p = NULL; napi_value v;
    if (get_pages(env, argv[0], &h) || h.n != 1 || h.len[0] != ZKMI_IPC_HANDLE_BYTES) BAD_ARG();
    int rc = ZK_CALL(zkmi_ipc_open(h.ptr[0], &p, NULL));
    if (rc) return throw_zkmi(env, rc);
    NAPI_OK(napi_create_double(env, (double)(uintptr_t)p, &v));
    return v;
}
static napi_value js_ipc_close(napi_env env, napi_callback_info info) {
    ARGS(1);
    void* p;
    if (get_dptr(env, argv[0], &p)) BAD_ARG();
    int rc = ZK_CALL(zkmi_ipc_close(p));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_peer_copy(napi_env env, napi_callback_info info) {
    ARGS(3);
    void *d, *s; double bytes;
    if (get_dptr(env, argv[0], &d) || get_dptr(env, argv[1], &s) || get_f64(env, argv[2], &bytes) || bytes < 0) BAD_ARG();
    int rc = ZK_CALL(zkmi_peer_copy(d, s, (size_t)bytes));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
/* peerCopyAsync(dDst, dSrc, bytes): queued on the copy stream; peerFence(): the library stream waits for everything queued so far */
static napi_value js_peer_copy_async(napi_env env, napi_callback_info info) {
    ARGS(3);
    void *d, *s; double bytes;
    if (get_dptr(env, argv[0], &d) || get_dptr(env, argv[1], &s) || get_f64(env, argv[2], &bytes) || bytes < 0) BAD_ARG();
    int rc = ZK_CALL(zkmi_peer_copy_async(d, s, (size_t)bytes));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_peer_fence(napi_env env, napi_callback_info info) {
    (void)info;
    int rc = ZK_CALL(zkmi_peer_fence());
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_groth16_reset(napi_env env, napi_callback_info info) {
    ARGS(1);
    double key;
    if (get_f64(env, argv[0], &key)) BAD_ARG();
    int rc = ZK_CALL(zkmi_groth16_reset((uint64_t)key));
    if (rc) return throw_zkmi(env, rc);
    return NULL;
}
static napi_value js_groth16_key_curve(napi_env env, napi_callback_info info) {
    ARGS(1);
    double key; napi_value v;
    if (get_f64(env, argv[0], &key)) BAD_ARG();
    NAPI_OK(napi_create_int32(env, ZK_CALL(zkmi_groth16_key_curve((uint64_t)key)), &v));
    return v;
}

/* call(name, ...args) -> 0 — table-driven binding of the device-resident PLONK / FFLONK entry points of include/zkmi.h (every parameter an
 * integer, a device pointer or a host buffer). Only the entry points listed below can be reached, and every argument is checked against the
 * kind the C prototype expects before the call is made:
 *   i  integer (a JS number)                       d  device pointer (a JS number; null / 0 where the prototype allows NULL)
 *   bN host buffer (typed array) of at least N bytes; b@K: at least as many bytes as the integer argument K; b*K:N: at least N bytes per unit of the integer argument K
 *   (arrays of `count` records); BN / B@K: the same, or null
 * Used by js/plonk_native.js and js/fflonk_native.js, which drive the device-resident provers from Node the way snarkjs_amd/plonk.py does
 * from Python. Throws Error(zkmi_last_error()) on a non-zero return. */
typedef int (*zk_fn16)(uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t,
                       uintptr_t, uintptr_t, uintptr_t, uintptr_t);
static const struct { const char* name; const char* sig; } g_call_table[] = {
    {"zkmi_fr_root", "i i b32"},
    {"zkmi_dev_alloc", "i b8"}, {"zkmi_dev_free", "d"}, {"zkmi_memcpy_h2d", "d b@2 i"}, {"zkmi_memcpy_d2h", "b@2 d i"}, {"zkmi_memcpy_d2d", "d d i"}, {"zkmi_memset_dev", "d i i"},
    {"zkmi_fr_batch_dev", "i i d d i"}, {"zkmi_ntt_dev", "i d d i i B32 B32"},
    {"zkmi_msm_table_build", "i i d i b8"}, {"zkmi_msm_table_release", "i"},
    {"zkmi_plonk_gather_wires_dev", "i d i d i d d d i i d d d"},
    {"zkmi_plonk_additions_dev", "i d i d i d"},
    {"zkmi_plonk_compute_z_dev", "i d d d d d d i b32 b32 b32 b32 b32 d"}, {"zkmi_plonk_compute_z_enqueue", "i d d d d d d i b32 b32 b32 b32 b32 d"},
    {"zkmi_pipeline_select", "i"}, {"zkmi_synchronize", ""},
    {"zkmi_plonk_compute_t_dev", "i b112 i i b352 b32 b32 b32 b32 b32 b32 b32 b32 d d"},
    {"zkmi_fflonk_t0_dev", "i b112 i i d"}, {"zkmi_fflonk_t1_dev", "i d d i b96 b32 d d"}, {"zkmi_fflonk_t2_dev", "i b112 i b96 b32 b32 b32 b32 b32 b32 d d"},
    {"zkmi_poly_degree_dev", "i d i b8"}, {"zkmi_keccak256", "B@1 i b32"},
    {"zkmi_poly_axpy_dev", "i d d i B32 i"}, {"zkmi_poly_scale_dev", "i d i b32"}, {"zkmi_poly_blind_dev", "i d i b32 i"}, {"zkmi_poly_add_scalar_dev", "i d b32"},
    {"zkmi_poly_evaluate_dev", "i d i b32 b32"}, {"zkmi_poly_is_zero_dev", "i d i b4"}, {"zkmi_poly_div_zh_dev", "i d i i i"}, {"zkmi_poly_div_by_zerofier_dev", "i d i i b32"},
    {"zkmi_cpoly_interleave_dev", "i b8 b8 i d i"}, {"zkmi_to_affine", "i i b96 b64"},
    /* r06: fused forms (one launch where the calls above take a chain of them) */
    {"zkmi_plonk_gather_wires_mont_dev", "i d i d i d d d i i d d d"}, {"zkmi_ntt_padded_dev", "i d i d i i"}, {"zkmi_poly_blind_tail_dev", "i d i b*4:32 i"},
    {"zkmi_poly_lincomb_dev", "i d i b*4:56 i B32"}, {"zkmi_poly_evaluate_multi_dev", "i b*4:8 b*4:8 b*4:32 i b*4:32"}, {"zkmi_poly_div_by_zerofier_enqueue", "i d i i b32"},
    {"zkmi_plonk_split_t_dev", "i d i i b32 b32 d d d"},
};
static void* zk_lib_handle(void) {
    static void* h = NULL;
    if (!h) {
        Dl_info di;
        if (dladdr((void*)&zkmi_init, &di) && di.dli_fname) h = dlopen(di.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    }
    return h;
}
static napi_value js_call(napi_env env, napi_callback_info info) {
    size_t argc = 17; napi_value argv[17];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 1) { napi_throw_type_error(env, NULL, "zkmi.call: function name expected"); return NULL; }
    char name[96]; size_t nl = 0;
    if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &nl) != napi_ok) BAD_ARG();
    const char* sig = NULL;
    for (size_t i = 0; i < sizeof g_call_table / sizeof g_call_table[0]; i++) if (!strcmp(name, g_call_table[i].name)) sig = g_call_table[i].sig;
    if (!sig) { napi_throw_error(env, NULL, "zkmi.call: not an entry point of the checked binding table"); return NULL; }
    void* h = zk_lib_handle();
    zk_fn16 fn = h ? (zk_fn16)dlsym(h, name) : NULL;
    if (!fn) { napi_throw_error(env, NULL, "zkmi.call: no such entry point"); return NULL; }
    uintptr_t a[16] = {0};
    size_t blen[16] = {0};
    struct { int arg; int ref; size_t mul; } at_checks[16];
    int n_at = 0;
    size_t k = 0;
    for (const char* c = sig; *c; k++) {
        const char kind = *c++;
        size_t min_len = 0, mul = 1; int ref = -1;
        if (*c == '@') { c++; ref = (int)strtol(c, (char**)&c, 10); }
        else if (*c == '*') { c++; ref = (int)strtol(c, (char**)&c, 10); if (*c == ':') { c++; mul = (size_t)strtol(c, (char**)&c, 10); } }
        else if (*c >= '0' && *c <= '9') min_len = (size_t)strtol(c, (char**)&c, 10);
        while (*c == ' ') c++;
        if (k + 1 >= argc || k >= 16) { napi_throw_type_error(env, NULL, "zkmi.call: too few arguments for this entry point"); return NULL; }
        napi_valuetype t;
        NAPI_OK(napi_typeof(env, argv[k + 1], &t));
        const bool is_null = t == napi_null || t == napi_undefined;
        if (kind == 'i' || kind == 'd') {
            if (is_null && kind == 'd') { a[k] = 0; continue; }
            double d;
            if (t != napi_number || napi_get_value_double(env, argv[k + 1], &d) != napi_ok || (kind == 'd' && (d < 0 || d > 9007199254740992.0))) BAD_ARG();
            a[k] = (uintptr_t)(int64_t)d;
        } else {                                            /* b / B */
            if (is_null) { if (kind != 'B') BAD_ARG(); a[k] = 0; blen[k] = 0; if (ref >= 0) { at_checks[n_at].arg = (int)k; at_checks[n_at].mul = mul; at_checks[n_at++].ref = ref; } continue; }
            bool is_ta = false;
            if (napi_is_typedarray(env, argv[k + 1], &is_ta) != napi_ok || !is_ta) BAD_ARG();
            napi_typedarray_type tt; size_t len; void* data; napi_value ab; size_t off;
            NAPI_OK(napi_get_typedarray_info(env, argv[k + 1], &tt, &len, &data, &ab, &off));
            const size_t esz = (tt == napi_uint8_array || tt == napi_int8_array || tt == napi_uint8_clamped_array) ? 1 : (tt == napi_uint16_array || tt == napi_int16_array) ? 2 :
                               (tt == napi_uint32_array || tt == napi_int32_array || tt == napi_float32_array) ? 4 : 8;
            blen[k] = len * esz;
            if (blen[k] < min_len) { napi_throw_type_error(env, NULL, "zkmi.call: a host buffer is shorter than the entry point requires"); return NULL; }
            if (ref >= 0) { at_checks[n_at].arg = (int)k; at_checks[n_at].mul = mul; at_checks[n_at++].ref = ref; }
            a[k] = (uintptr_t)data;
        }
    }
    if (k + 1 != argc) { napi_throw_type_error(env, NULL, "zkmi.call: too many arguments for this entry point"); return NULL; }
    for (int i = 0; i < n_at; i++)
        if ((uint64_t)blen[at_checks[i].arg] < (uint64_t)a[at_checks[i].ref] * at_checks[i].mul) { napi_throw_type_error(env, NULL, "zkmi.call: a host buffer is shorter than its length argument"); return NULL; }
    int rc = ZK_CALL(fn(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]));
    if (rc) return throw_zkmi(env, rc);
    napi_value z;
    NAPI_OK(napi_create_int32(env, 0, &z));
    return z;
}

/* ---- shared host memory between the processes of a multi-GPU proof (js/groth16_shards.js): shmMap(name, bytes, create) -> Uint8Array over
 * a POSIX shared-memory object (the chain outputs travel owner GPU -> shared pages -> the other GPUs; north star: host side stays Node.js);
 * shmUnlink(name). The mapping is page-locked for the device (zkmi_host_register) when a device is bound; it lives until the array is
 * collected. */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
typedef struct { void* p; size_t len; int registered; } shm_t;
static void shm_finalize(napi_env env, void* data, void* hint) {
    (void)env; (void)data;
    shm_t* m = (shm_t*)hint;
    if (m->registered) (void)ZK_CALL(zkmi_host_unregister(m->p));
    munmap(m->p, m->len);
    free(m);
}
static napi_value js_shm_map(napi_env env, napi_callback_info info) {
    ARGS(3);
    char name[128]; size_t nl = 0; double bytes; bool create = false;
    if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &nl) != napi_ok || name[0] != '/' || get_f64(env, argv[1], &bytes) || bytes < 1 ||
        napi_get_value_bool(env, argv[2], &create) != napi_ok) BAD_ARG();
    int fd = shm_open(name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) { napi_throw_error(env, NULL, "zkmi.shmMap: shm_open failed"); return NULL; }
    if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); napi_throw_error(env, NULL, "zkmi.shmMap: ftruncate failed"); return NULL; }
    void* p = mmap(NULL, (size_t)bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { napi_throw_error(env, NULL, "zkmi.shmMap: mmap failed"); return NULL; }
    shm_t* m = (shm_t*)calloc(1, sizeof *m);
    if (!m) { munmap(p, (size_t)bytes); BAD_ARG(); }
    m->p = p; m->len = (size_t)bytes;
    m->registered = ZK_CALL(zkmi_host_register(p, (size_t)bytes)) == 0;      /* without a device the pages simply stay pageable */
    napi_value ab, ta;
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, shm_finalize, m, &ab) != napi_ok || napi_create_typedarray(env, napi_uint8_array, (size_t)bytes, ab, 0, &ta) != napi_ok) {
        shm_finalize(env, p, m);
        napi_throw_error(env, NULL, "zkmi.shmMap: cannot wrap the mapping");
        return NULL;
    }
    return ta;
}
static napi_value js_shm_unlink(napi_env env, napi_callback_info info) {
    ARGS(1);
    char name[128]; size_t nl = 0;
    if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &nl) != napi_ok || name[0] != '/') BAD_ARG();
    shm_unlink(name);
    return NULL;
}

static napi_value module_init(napi_env env, napi_value exports) {
    static const struct { const char* name; napi_callback fn; } fns[] = {
        {"init", js_init}, {"deviceCount", js_device_count}, {"version", js_version}, {"msm", js_msm}, {"releaseBases", js_release_bases},
        {"ntt", js_ntt}, {"frBatch", js_fr_batch}, {"applyKey", js_apply_key}, {"joinABC", js_join_abc}, {"toAffine", js_to_affine}, {"groupFft", js_group_fft}, {"groupApplyKey", js_group_apply_key}, {"groupConvert", js_group_convert},
        {"groth16Prove", js_groth16_prove}, {"groth16ProveAsync", js_groth16_prove_async}, {"msmAsync", js_msm_async}, {"nttAsync", js_ntt_async}, {"groth16Release", js_groth16_release}, {"call", js_call},
        {"groth16Load", js_groth16_load}, {"groth16LoadAsync", js_groth16_load_async}, {"groth16LoadShard", js_groth16_load_shard}, {"groth16Submit", js_groth16_submit},
        {"groth16SubmitAsync", js_groth16_submit_async}, {"groth16Collect", js_groth16_collect}, {"groth16CollectAsync", js_groth16_collect_async},
        {"devAlloc", js_dev_alloc}, {"devFree", js_dev_free}, {"memcpyH2D", js_memcpy_h2d}, {"memcpyD2H", js_memcpy_d2h},
        {"groth16ChainsDev", js_groth16_chains_dev}, {"groth16SumsWDev", js_groth16_sums_w_dev}, {"groth16SumsHDev", js_groth16_sums_h_dev}, {"groth16SumsDev", js_groth16_sums_dev},
        {"groth16Finish", js_groth16_finish}, {"joinABCDev", js_join_abc_dev}, {"pointAdd", js_point_add}, {"shmMap", js_shm_map}, {"shmUnlink", js_shm_unlink},
        {"msmTableDev", js_msm_table_dev}, {"msmTableMultiDev", js_msm_table_multi_dev}, {"msmTableMultiDevAsync", js_msm_table_multi_dev_async}, {"msmTableMultiEnqueueDev", js_msm_table_multi_enqueue_dev}, {"msmTableMultiEnqueueMontDev", js_msm_table_multi_enqueue_mont_dev}, {"msmTableMultiCollect", js_msm_table_multi_collect}, {"synchronizeAsync", js_synchronize_async}, {"ipcExport", js_ipc_export}, {"ipcOpen", js_ipc_open}, {"ipcClose", js_ipc_close},
        {"peerCopy", js_peer_copy}, {"peerCopyAsync", js_peer_copy_async}, {"peerFence", js_peer_fence}, {"groth16Reset", js_groth16_reset}, {"groth16KeyCurve", js_groth16_key_curve},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, module_init)
