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

#define MAX_PAGES 64

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

/* v: Uint8Array | Array<Uint8Array> */
static int get_pages(napi_env env, napi_value v, pages_t* out) {
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
        napi_value e; napi_typedarray_type t; size_t len; void* data; napi_value ab; size_t off;
        if (napi_get_element(env, v, i, &e) != napi_ok) return -1;
        if (napi_get_typedarray_info(env, e, &t, &len, &data, &ab, &off) != napi_ok || t != napi_uint8_array) return -1;
        out->ptr[i] = (const uint8_t*)data; out->len[i] = len; out->total += len;
    }
    out->n = (int)k;
    return 0;
}
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

/* ---- asynchronous calls (napi_create_async_work; SURVEY.md 8 b) -----------------------------------------------------------------
 * msmAsync / nttAsync / groth16ProveAsync take the arguments of msm / ntt / groth16Prove and return a Promise: arguments are parsed
 * on the main thread, the library call runs on a libuv pool thread (the Node event loop keeps turning), the promise is settled
 * back on the main thread. Input and output buffers stay referenced until then; the caller must not write to them meanwhile. */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref keep[8]; int n_keep;          /* arguments and result kept alive while the job runs */
    napi_ref result;                       /* value the promise resolves to (NULL: undefined) */
    int kind;                              /* 0 msm, 1 ntt, 2 groth16Prove */
    int32_t curve, group, logn, inverse;
    pages_t a, b;
    double n, sb, key;
    uint8_t first[32], inc[32]; bool has_first, has_inc;
    zkmi_groth16_zkey zk; bool has_zk;
    uint8_t *o0, *o1, *o2;
    int rc; char err[400];
} job_t;

static int job_run(job_t* j) {
    switch (j->kind) {
    case 0: return zkmi_msm(j->curve, j->group, as_zk(&j->a), as_zk(&j->b), (size_t)j->n, (size_t)j->sb, (uint64_t)j->key, j->o0);
    case 1: return zkmi_ntt(j->curve, as_zk(&j->a), (uint8_t* const*)j->b.ptr, j->b.len, j->b.n, (unsigned)j->logn, j->inverse, j->has_first ? j->first : NULL, j->has_inc ? j->inc : NULL);
    default: return zkmi_groth16_prove(j->has_zk ? &j->zk : NULL, (uint64_t)j->key, j->a.ptr[0], j->a.len[0], j->first, j->inc, j->o0, j->o1, j->o2);
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
    free(j);
}
/* queue `j` (heap, filled by the caller); keeps argv[0..argc) and `result` alive; returns the promise */
static napi_value job_queue(napi_env env, job_t* j, const char* name, napi_value* argv, size_t argc, napi_value result) {
    napi_value promise, rname;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok) { free(j); napi_throw_error(env, NULL, "zkmi: cannot create a promise"); return NULL; }
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
    zkmi_release_bases((uint64_t)key);
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
 *   -> {pi_a, pi_b, pi_c} (affine Montgomery bytes). Sections must be single Uint8Arrays (< 2 GiB each). */
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
static napi_value groth16_impl(napi_env env, napi_callback_info info, bool async) {
    ARGS(5);
    zkmi_groth16_zkey zk, *pzk = NULL;
    napi_valuetype t;
    NAPI_OK(napi_typeof(env, argv[0], &t));
    double key;
    pages_t w;
    const uint8_t *r, *s;
    int curve = 0;
    if (t == napi_object) {
        uint32_t c;
        memset(&zk, 0, sizeof zk);
        if (get_named_u32(env, argv[0], "curve", &c) || get_named_u32(env, argv[0], "nVars", &zk.n_vars) || get_named_u32(env, argv[0], "nPublic", &zk.n_public) ||
            get_named_u32(env, argv[0], "domainSize", &zk.domain_size) || get_named_u8(env, argv[0], "coeffs", &zk.coeffs, &zk.coeffs_len) ||
            get_named_u8(env, argv[0], "A", &zk.bases_a, &zk.bases_a_len) || get_named_u8(env, argv[0], "B1", &zk.bases_b1, &zk.bases_b1_len) ||
            get_named_u8(env, argv[0], "B2", &zk.bases_b2, &zk.bases_b2_len) || get_named_u8(env, argv[0], "C", &zk.bases_c, &zk.bases_c_len) ||
            get_named_u8(env, argv[0], "H", &zk.bases_h, &zk.bases_h_len) ||
            get_named_u8(env, argv[0], "alpha1", &zk.vk_alpha_1, NULL) || get_named_u8(env, argv[0], "beta1", &zk.vk_beta_1, NULL) ||
            get_named_u8(env, argv[0], "beta2", &zk.vk_beta_2, NULL) || get_named_u8(env, argv[0], "delta1", &zk.vk_delta_1, NULL) ||
            get_named_u8(env, argv[0], "delta2", &zk.vk_delta_2, NULL)) BAD_ARG();
        zk.curve = (int)c; curve = zk.curve; pzk = &zk;
    }
    if (get_f64(env, argv[1], &key) || get_pages(env, argv[2], &w) || w.n != 1 || get_opt32(env, argv[3], &r) || get_opt32(env, argv[4], &s) || !r || !s) BAD_ARG();
    if (!pzk) {            /* key already resident: the caller passes the curve id in place of the zkey object */
        int32_t c;
        if (get_i32(env, argv[0], &c)) BAD_ARG();
        curve = c;
    }
    const size_t q = curve == ZKMI_CURVE_BN128 ? 32 : 48;
    uint8_t *pa, *pb, *pc;
    napi_value va = new_u8(env, 2 * q, &pa), vb = new_u8(env, 4 * q, &pb), vc = new_u8(env, 2 * q, &pc), res;
    if (!va || !vb || !vc) BAD_ARG();
    if (pzk) {             /* header points: 2*n8q (G1) / 4*n8q (G2) bytes each */
        size_t l1, l2, l3, l4, l5; const uint8_t* d;
        if (get_named_u8(env, argv[0], "alpha1", &d, &l1) || get_named_u8(env, argv[0], "beta1", &d, &l2) || get_named_u8(env, argv[0], "beta2", &d, &l3) ||
            get_named_u8(env, argv[0], "delta1", &d, &l4) || get_named_u8(env, argv[0], "delta2", &d, &l5) || l1 < 2 * q || l2 < 2 * q || l3 < 4 * q || l4 < 2 * q || l5 < 4 * q) BAD_ARG();
    }
    NAPI_OK(napi_create_object(env, &res));
    NAPI_OK(napi_set_named_property(env, res, "pi_a", va));
    NAPI_OK(napi_set_named_property(env, res, "pi_b", vb));
    NAPI_OK(napi_set_named_property(env, res, "pi_c", vc));
    if (async) {           /* the zkey sections are referenced through argv[0] (the descriptor object holds the typed arrays) */
        job_t* j = (job_t*)calloc(1, sizeof *j);
        if (!j) BAD_ARG();
        j->kind = 2; j->key = key; j->a = w; j->o0 = pa; j->o1 = pb; j->o2 = pc;
        memcpy(j->first, r, 32); memcpy(j->inc, s, 32);
        if (pzk) { j->zk = zk; j->has_zk = true; }
        return job_queue(env, j, "zkmi.groth16Prove", argv, 5, res);
    }
    int rc = ZK_CALL(zkmi_groth16_prove(pzk, (uint64_t)key, w.ptr[0], w.len[0], r, s, pa, pb, pc));
    if (rc) return throw_zkmi(env, rc);
    return res;
}
static napi_value js_groth16_prove(napi_env env, napi_callback_info info) { return groth16_impl(env, info, false); }
static napi_value js_groth16_prove_async(napi_env env, napi_callback_info info) { return groth16_impl(env, info, true); }
static napi_value js_groth16_release(napi_env env, napi_callback_info info) {
    ARGS(1);
    double key;
    if (get_f64(env, argv[0], &key)) BAD_ARG();
    zkmi_groth16_release((uint64_t)key);
    return NULL;
}

/* call(name, ...args) -> 0 — generic binding of any `int zkmi_*(...)` entry point of include/zkmi.h whose parameters are all
 * integers or pointers (every *_dev function is): numbers pass as 64-bit integers (device pointers fit a double's 53 bits),
 * typed arrays as their data pointer, null/undefined as NULL. Used by js/plonk_native.js, which drives the device-resident
 * PLONK prover from Node the way snarkjs_amd/plonk.py does from Python. Throws Error(zkmi_last_error()) on a non-zero return. */
typedef int (*zk_fn16)(uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t, uintptr_t,
                       uintptr_t, uintptr_t, uintptr_t, uintptr_t);
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
    if (napi_get_value_string_utf8(env, argv[0], name, sizeof name, &nl) != napi_ok || strncmp(name, "zkmi_", 5)) BAD_ARG();
    void* h = zk_lib_handle();
    zk_fn16 fn = h ? (zk_fn16)dlsym(h, name) : NULL;
    if (!fn) { napi_throw_error(env, NULL, "zkmi.call: no such entry point"); return NULL; }
    uintptr_t a[16] = {0};
    for (size_t i = 1; i < argc && i <= 16; i++) {
        napi_valuetype t;
        NAPI_OK(napi_typeof(env, argv[i], &t));
        if (t == napi_number) { double d; NAPI_OK(napi_get_value_double(env, argv[i], &d)); a[i - 1] = (uintptr_t)(int64_t)d; }
        else if (t == napi_null || t == napi_undefined) a[i - 1] = 0;
        else {
            bool is_ta = false;
            if (napi_is_typedarray(env, argv[i], &is_ta) != napi_ok || !is_ta) BAD_ARG();
            napi_typedarray_type tt; size_t len; void* data; napi_value ab; size_t off;
            NAPI_OK(napi_get_typedarray_info(env, argv[i], &tt, &len, &data, &ab, &off));
            a[i - 1] = (uintptr_t)data;
        }
    }
    int rc = ZK_CALL(fn(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]));
    if (rc) return throw_zkmi(env, rc);
    napi_value z;
    NAPI_OK(napi_create_int32(env, 0, &z));
    return z;
}

static napi_value module_init(napi_env env, napi_value exports) {
    static const struct { const char* name; napi_callback fn; } fns[] = {
        {"init", js_init}, {"deviceCount", js_device_count}, {"version", js_version}, {"msm", js_msm}, {"releaseBases", js_release_bases},
        {"ntt", js_ntt}, {"frBatch", js_fr_batch}, {"applyKey", js_apply_key}, {"joinABC", js_join_abc}, {"toAffine", js_to_affine}, {"groupFft", js_group_fft}, {"groupApplyKey", js_group_apply_key}, {"groupConvert", js_group_convert},
        {"groth16Prove", js_groth16_prove}, {"groth16ProveAsync", js_groth16_prove_async}, {"msmAsync", js_msm_async}, {"nttAsync", js_ntt_async}, {"groth16Release", js_groth16_release}, {"call", js_call},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}
NAPI_MODULE(NODE_GYP_MODULE_NAME, module_init)
