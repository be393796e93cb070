/*
 * fdd_cpu_shim.c -- TEST INFRASTRUCTURE, never shipped, never loaded by the
 * product.  A stand-in for libfdd_hip.so that implements the same C-ABI
 * (include/fdd_hip.h) on HOST memory by forwarding every kernel entry to the
 * CPU oracle (oracle/fdd_oracle.h).  It exists for one purpose: to let the
 * multi-rank logic of the C++ host layer (rank partition, boundary-first
 * numbering, interface-slot exchange, scalar all-reduces, communication
 * callbacks) run in world_size-2 `gloo` tests on a machine without a GPU
 * (tests/test_cpu_multirank.py).  The product's loader
 * (polynomial_..._amd/lib.py) only ever opens libfdd_hip.so and fails loudly
 * without it.  In effect this is the reference's OCCA "Serial" mode.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fdd_hip.h"
#include "fdd_oracle.h"

static __thread char g_err[256] = "";
#define REQ(c)                                                                  \
    do                                                                          \
    {                                                                           \
        if (!(c))                                                               \
        {                                                                       \
            snprintf(g_err, sizeof(g_err), "invalid argument: %s", #c);         \
            return FDD_ERR_INVALID_ARGUMENT;                                    \
        }                                                                       \
    } while (0)

const char *fdd_version(void) { return "fdd_cpu_shim (oracle-backed, tests only)"; }
const char *fdd_last_error(void) { return g_err; }
int fdd_device_count(int *count) { *count = 1; return 0; }
int fdd_set_device(int device) { (void)device; return 0; }
int fdd_get_device(int *device) { *device = 0; return 0; }
int fdd_device_name(char *buf, size_t n) { snprintf(buf, n, "cpu-shim"); return 0; }
int fdd_malloc(void **ptr, size_t bytes) { *ptr = bytes ? calloc(1, bytes) : NULL; return (bytes && !*ptr) ? 2 : 0; }
int fdd_free(void *ptr) { free(ptr); return 0; }
int fdd_memcpy_h2d(void *d, const void *s, size_t b, void *st) { (void)st; if (b) memmove(d, s, b); return 0; }
int fdd_memcpy_d2h(void *d, const void *s, size_t b, void *st) { (void)st; if (b) memmove(d, s, b); return 0; }
int fdd_fetch_scalars(void *d, const void *s, size_t b, void *st) { (void)st; if (b) memmove(d, s, b); return 0; }
int fdd_memcpy_d2d(void *d, const void *s, size_t b, void *st) { (void)st; if (b) memmove(d, s, b); return 0; }
int fdd_memset(void *d, int v, size_t b, void *st) { (void)st; if (b) memset(d, v, b); return 0; }
int fdd_stream_create(void **s) { *s = NULL; return 0; }
int fdd_stream_destroy(void *s) { (void)s; return 0; }
int fdd_stream_sync(void *s) { (void)s; return 0; }
int fdd_device_sync(void) { return 0; }

/* no graphs on the CPU: capture is refused, callers fall back to eager launches */
int fdd_graph_begin_capture(void *s) { (void)s; snprintf(g_err, sizeof(g_err), "no graph capture in the CPU shim"); return FDD_ERR_UNSUPPORTED; }
int fdd_graph_end_capture(void *s, void **g) { (void)s; *g = NULL; return FDD_ERR_UNSUPPORTED; }
int fdd_graph_launch(void *g, void *s) { (void)g; (void)s; return FDD_ERR_UNSUPPORTED; }
int fdd_graph_destroy(void *g) { (void)g; return 0; }
int fdd_event_create(void **e) { *e = calloc(1, sizeof(struct timespec)); return 0; }
int fdd_event_destroy(void *e) { free(e); return 0; }
int fdd_event_record(void *e, void *s) { (void)s; clock_gettime(CLOCK_MONOTONIC, (struct timespec *)e); return 0; }
int fdd_event_elapsed_ms(float *ms, void *a, void *b)
{
    struct timespec *x = (struct timespec *)a, *y = (struct timespec *)b;
    *ms = (float)((y->tv_sec - x->tv_sec) * 1e3 + (y->tv_nsec - x->tv_nsec) * 1e-6);
    return 0;
}

/* ---- csr ---- */
int fdd_csr_multiply(double *Au, const int *p, const int *c, const double *v, const double *u, int n, void *s) { (void)s; REQ(n >= 0); orc_csr_multiply(Au, p, c, v, u, n); return 0; }
int fdd_csr_multiply_range(double *Au, const int *p, const int *c, const double *v, const double *u, int r0, int r1, void *s) { (void)s; REQ(r0 >= 0 && r1 >= r0); orc_csr_multiply_range(Au, p, c, v, u, r0, r1); return 0; }
int fdd_csr_multiply_weight(double *Au, const int *p, const int *c, const double *v, const double *u, const double *w, int n, void *s) { (void)s; REQ(n >= 0); orc_csr_multiply_weight(Au, p, c, v, u, w, n); return 0; }

struct fdd_csr_plan { int num_rows, num_cols, num_nnz; };
int fdd_csr_plan_create(fdd_csr_plan **plan, const int *ptr, int rows, int cols, int nnz)
{
    (void)ptr;
    *plan = (fdd_csr_plan *)calloc(1, sizeof(fdd_csr_plan));
    (*plan)->num_rows = rows; (*plan)->num_cols = cols; (*plan)->num_nnz = nnz;
    return 0;
}
int fdd_csr_plan_destroy(fdd_csr_plan *plan) { free(plan); return 0; }
int fdd_csr_plan_num_blocks(const fdd_csr_plan *plan, int *nb) { (void)plan; *nb = 0; return 0; }
int fdd_csr_plan_kind(const fdd_csr_plan *plan, int *kind) { (void)plan; *kind = 0; return 0; }
int fdd_csr_plan_pipelined(const fdd_csr_plan *plan, int *pipelined) { (void)plan; *pipelined = 0; return 0; }
int fdd_csr_plan_sell_info(const fdd_csr_plan *plan, int *slices, int *compact_slices) { (void)plan; *slices = 0; *compact_slices = 0; return 0; }
int fdd_csr_plan_set_unit_values(fdd_csr_plan *plan, int unit) { (void)plan; (void)unit; return 0; }
int fdd_csr_plan_attach_sell(fdd_csr_plan *plan, const int *ph, const int *p, const int *c, const void *v, double mp, int *attached, void *s) { (void)plan; (void)ph; (void)p; (void)c; (void)v; (void)mp; (void)s; *attached = 0; return 0; }
int fdd_csr_plan_multiply(const fdd_csr_plan *plan, double *Au, const int *p, const int *c, const double *v, const double *u, const double *w, void *s)
{
    (void)s;
    if (plan->num_rows == 0 || plan->num_cols == 0) return 0;
    if (w) orc_csr_multiply_weight(Au, p, c, v, u, w, plan->num_rows);
    else orc_csr_multiply(Au, p, c, v, u, plan->num_rows);
    return 0;
}

/* ---- math ---- */
int fdd_set_to_value(double *u, double a, int n, int off, void *s) { (void)s; REQ(n >= 0); orc_set_to_value(u, a, n, off); return 0; }
int fdd_invert_vector_elements(double *u, int n, void *s) { (void)s; REQ(n >= 0); orc_invert_vector_elements(u, n); return 0; }
int fdd_vector_vector_addition(double *uv, double a, const double *u, double b, const double *v, int n, void *s) { (void)s; REQ(n >= 0); orc_vector_vector_addition(uv, a, u, b, v, n); return 0; }
int fdd_vector_scaling(double *au, double a, const double *u, int n, void *s) { (void)s; REQ(n >= 0); orc_vector_scaling(au, a, u, n); return 0; }

/* ---- domain ---- */
int fdd_dom_stiffness_matrix_1(double *const GDu[3], const double *u, const double *D, const double *const G[6], int np, int N, int dim, void *s) { (void)s; orc_dom_stiffness_matrix_1(GDu, u, D, G, np, N, dim); return 0; }
int fdd_dom_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *D, int np, int N, int dim, void *s) { (void)s; orc_dom_stiffness_matrix_2(Au, GDu, D, np, N, dim); return 0; }

static int fused(double *Au, const double *u, const double *D, const double *const G[6], const int *eo, int ne, int N)
{
    int n3 = (N + 1) * (N + 1) * (N + 1);
    double *w[3];
    for (int k = 0; k < 3; k++) w[k] = (double *)malloc(sizeof(double) * (size_t)n3);
    for (int e = 0; e < ne; e++)
    {
        size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        const double *Ge[6];
        for (int g = 0; g < 6; g++) Ge[g] = G[g] + o;
        orc_dom_stiffness_matrix_1(w, u + o, D, Ge, n3, N, 3);
        orc_dom_stiffness_matrix_2(Au + o, (const double *const *)w, D, n3, N, 3);
    }
    for (int k = 0; k < 3; k++) free(w[k]);
    return 0;
}
int fdd_dom_stiffness_matrix(double *Au, const double *u, const double *D, const double *const G[6], int ne, int N, void *s) { (void)s; return fused(Au, u, D, G, NULL, ne, N); }
int fdd_sub_stiffness_matrix(double *Au, const double *u, const double *D, const double *const G[6], const int *eo, int ne, int N, void *s) { (void)s; return fused(Au, u, D, G, eo, ne, N); }
int fdd_stiffness_matrix_2d(double *Au, const double *u, const double *D, const double *const G[6], const int *eo, int ne, int N, void *s)
{
    (void)s;
    int n2 = (N + 1) * (N + 1);
    double *w[3] = {(double *)malloc(sizeof(double) * (size_t)n2), (double *)malloc(sizeof(double) * (size_t)n2), NULL};
    for (int e = 0; e < ne; e++)
    {
        size_t o = eo ? (size_t)eo[e] : (size_t)e * n2;
        const double *Ge[6];
        for (int g = 0; g < 6; g++) Ge[g] = G[g] ? G[g] + o : NULL;
        orc_dom_stiffness_matrix_1(w, u + o, D, Ge, n2, N, 2);
        orc_dom_stiffness_matrix_2(Au + o, (const double *const *)w, D, n2, N, 2);
    }
    free(w[0]); free(w[1]);
    return 0;
}
int fdd_sub_stiffness_matrix_gather(double *Au, const double *v, const int *pd, const double *D, const double *const G[6], const int *eo, int ne, int N, void *s)
{
    (void)s;
    int n3 = (N + 1) * (N + 1) * (N + 1);
    double *u = (double *)malloc(sizeof(double) * (size_t)n3);
    double *w[3];
    for (int k = 0; k < 3; k++) w[k] = (double *)malloc(sizeof(double) * (size_t)n3);
    for (int e = 0; e < ne; e++)
    {
        size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        const double *Ge[6];
        for (int g = 0; g < 6; g++) Ge[g] = G[g] + o;
        for (int q = 0; q < n3; q++) u[q] = (pd[o + q] >= 0) ? v[pd[o + q]] : 0.0;
        orc_dom_stiffness_matrix_1(w, u, D, Ge, n3, N, 3);
        orc_dom_stiffness_matrix_2(Au + o, (const double *const *)w, D, n3, N, 3);
    }
    for (int k = 0; k < 3; k++) free(w[k]);
    free(u);
    return 0;
}
int fdd_stiffness_matrix_mfma(double *Au, const double *u, const double *D, const double *const G[6], const int *eo, int ne, int N, void *s) { (void)s; return fused(Au, u, D, G, eo, ne, N); }
int fdd_dom_initialize_arrays(double *u, double *r, const double *f, int n, void *s) { (void)s; orc_dom_initialize_arrays(u, r, f, n); return 0; }

size_t fdd_reduce_workspace_doubles(void) { return 2 * (size_t)FDD_REDUCE_MAX_BLOCKS; }

/* reductions: the oracle's 128-wide tree + in-order block sum */
#define NB(n) (((n) + ORC_BLOCK_SIZE - 1) / ORC_BLOCK_SIZE)
int fdd_dom_residual_norm(double *out, double *ws, const double *r, const double *q, const double *m, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc((size_t)NB(n) + 1, sizeof(double));
    orc_dom_residual_norm(b, r, q, m, n, NB(n)); out[0] = orc_block_sum(b, NB(n)); free(b); return 0;
}
int fdd_dom_projection_inner_products(double *out, double *ws, const double *z, const double *r, const double *p, const double *q, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc(2 * (size_t)NB(n) + 2, sizeof(double));
    orc_dom_projection_inner_products(b, z, r, p, q, n, NB(n));
    out[0] = orc_block_sum(b, NB(n)); out[1] = orc_block_sum(b + NB(n), NB(n)); free(b); return 0;
}
int fdd_dom_inner_product_flexible(double *out, double *ws, const double *r, const double *r1, const double *z, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc((size_t)NB(n) + 1, sizeof(double));
    orc_dom_inner_product_flexible(b, r, r1, z, n, NB(n)); out[0] = orc_block_sum(b, NB(n)); free(b); return 0;
}
int fdd_dom_inner_product(double *out, double *ws, const double *u, const double *v, const double *m, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc((size_t)NB(n) + 1, sizeof(double));
    orc_dom_inner_product(b, u, v, m, n, NB(n)); out[0] = orc_block_sum(b, NB(n)); free(b); return 0;
}
int fdd_dom_solution_and_residual_update(double *u, double *r1, const double *r, const double *p, const double *q, double a, int n, void *s) { (void)s; orc_dom_solution_and_residual_update(u, r1, r, p, q, a, n); return 0; }
int fdd_dom_residual_and_search_update(double *p, double *r, const double *z, const double *r1, double b, int n, void *s) { (void)s; orc_dom_residual_and_search_update(p, r, z, r1, b, n); return 0; }
int fdd_dom_solution_and_residual_update_dev(double *u, double *r1, const double *r, const double *p, const double *q, const double *num, const double *den, int n, void *s) { (void)s; orc_dom_solution_and_residual_update(u, r1, r, p, q, num[0] / den[0], n); return 0; }
int fdd_dom_residual_and_search_update_dev(double *p, double *r, const double *z, const double *r1, const double *num, const double *den, int n, void *s) { (void)s; orc_dom_residual_and_search_update(p, r, z, r1, num[0] / den[0], n); return 0; }

/* ---- subdomain ---- */
int fdd_sub_stiffness_matrix_1(double *const GDu[3], const double *u, const double *const *Dp, const int *off, const int *vert, const int *lev, const int *pd, int nl, const double *const G[6], int np, int dim, void *s) { (void)s; (void)nl; orc_sub_stiffness_matrix_1(GDu, u, Dp, off, vert, lev, pd, G, np, dim); return 0; }
int fdd_sub_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *const *Dp, const int *off, const int *vert, const int *lev, const int *pd, int nl, int np, int dim, void *s) { (void)s; (void)nl; orc_sub_stiffness_matrix_2(Au, GDu, Dp, off, vert, lev, pd, np, dim); return 0; }
int fdd_sub_inner_product(double *out, double *ws, const double *u, const double *v, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc((size_t)NB(n) + 1, sizeof(double));
    orc_sub_inner_product(b, u, v, n, NB(n)); out[0] = orc_block_sum(b, NB(n)); free(b); return 0;
}
int fdd_sub_weighted_inner_product(double *out, double *ws, const double *u, const double *v, const double *w, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc((size_t)NB(n) + 1, sizeof(double));
    orc_sub_weighted_inner_product(b, u, v, w, n, NB(n)); out[0] = orc_block_sum(b, NB(n)); free(b); return 0;
}
int fdd_sub_projection_inner_products(double *out, double *ws, const double *z, const double *r, const double *p, const double *q, const double *w, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc(2 * (size_t)NB(n) + 2, sizeof(double));
    orc_sub_projection_inner_products(b, z, r, p, q, w, n, NB(n));
    out[0] = orc_block_sum(b, NB(n)); out[1] = orc_block_sum(b + NB(n), NB(n)); free(b); return 0;
}
int fdd_sub_search_update_inner_product(double *out, double *ws, const double *r, const double *r1, const double *z, const double *w, int n, void *s)
{
    (void)s; (void)ws;
    double *b = (double *)calloc((size_t)NB(n) + 1, sizeof(double));
    orc_sub_search_update_inner_product(b, r, r1, z, w, n, NB(n)); out[0] = orc_block_sum(b, NB(n)); free(b); return 0;
}
int fdd_sub_initialize_arrays(double *u, double *r, const double *f, int n, void *s) { (void)s; orc_sub_initialize_arrays(u, r, f, n); return 0; }
int fdd_sub_solution_and_residual_update(double *u, double *r1, const double *r, const double *p, const double *q, double a, int n, void *s) { (void)s; orc_sub_solution_and_residual_update(u, r1, r, p, q, a, n); return 0; }
int fdd_sub_residual_and_search_update(double *p, double *r, const double *z, const double *r1, double b, int n, void *s) { (void)s; orc_sub_residual_and_search_update(p, r, z, r1, b, n); return 0; }
int fdd_sub_copy_f64_f64(double *u, const double *v, int n, void *s) { (void)s; orc_sub_copy_f64_f64(u, v, n); return 0; }
int fdd_sub_copy_f32_f64(float *u, const double *v, int n, void *s) { (void)s; orc_sub_copy_f32_f64(u, v, n); return 0; }
int fdd_sub_copy_f64_f32(double *u, const float *v, int n, void *s) { (void)s; orc_sub_copy_f64_f32(u, v, n); return 0; }
int fdd_sub_restriction_1(double *Ju, const double *J, const double *u, int np, int nf, int nc, int dim, void *s) { (void)s; orc_sub_restriction_1(Ju, J, u, np, nf, nc, dim); return 0; }
int fdd_sub_restriction_2(double *Ju, const double *J, const double *u, int np, int nf, int nc, int dim, void *s) { (void)s; orc_sub_restriction_2(Ju, J, u, np, nf, nc, dim); return 0; }
int fdd_sub_restriction_3(double *Ju, const double *J, const double *u, int np, int nf, int nc, void *s) { (void)s; orc_sub_restriction_3(Ju, J, u, np, nf, nc); return 0; }
int fdd_sub_restriction_2d(double *uc, const double *J, const double *uf, int ne, int nf, int nc, void *s)
{
    (void)s;
    double *t = (double *)malloc(sizeof(double) * (size_t)ne * nf * nc + 8);
    orc_sub_restriction_1(t, J, uf, ne * nf * nc, nf, nc, 2);
    orc_sub_restriction_2(uc, J, t, ne * nc * nc, nf, nc, 2);
    free(t);
    return 0;
}
int fdd_sub_restriction(double *uc, const double *J, const double *uf, int ne, int nf, int nc, void *s)
{
    (void)s;
    double *w1 = (double *)malloc(sizeof(double) * (size_t)ne * nf * nf * nc);
    double *w2 = (double *)malloc(sizeof(double) * (size_t)ne * nf * nc * nc);
    orc_sub_restriction_1(w1, J, uf, ne * nf * nf * nc, nf, nc, 3);
    orc_sub_restriction_2(w2, J, w1, ne * nf * nc * nc, nf, nc, 3);
    orc_sub_restriction_3(uc, J, w2, ne * nc * nc * nc, nf, nc);
    free(w1); free(w2);
    return 0;
}

/* ---- AMG ---- */
int fdd_amg_vector_set_to_value(double *d, double v, int n, void *s) { (void)s; orc_amg_vector_set_to_value(d, v, n); return 0; }
int fdd_amg_main_scaled_residual(double *Sr, double *w, const double *f, const double *S, double a, int n, void *s) { (void)s; orc_amg_main_scaled_residual(Sr, w, f, S, a, n); return 0; }
int fdd_amg_main_polynomial_evaluation(double *w, double *v, const double *r, const double *D, double a, int n, void *s) { (void)s; orc_amg_main_polynomial_evaluation(w, v, r, D, a, n); return 0; }
int fdd_amg_main_update_field(double *u, const double *w, const double *D, int n, void *s) { (void)s; orc_amg_main_update_field(u, w, D, n); return 0; }
int fdd_amg_vector_multiplication(double *uv, const double *u, const double *v, int n, void *s) { (void)s; orc_amg_vector_multiplication(uv, u, v, n); return 0; }
int fdd_amg_matvec(double *y, const int *p, const int *c, const double *v, const double *x, double a, double b, int n, void *s) { (void)s; orc_amg_matvec(y, p, c, v, x, a, b, n); return 0; }
int fdd_amg_dot(double *out, double *ws, const double *x, const double *y, int n, void *s) { return fdd_sub_inner_product(out, ws, x, y, n, s); }
int fdd_csr_plan_matvec(const fdd_csr_plan *plan, double *y, const int *p, const int *c, const double *v, const double *x, double a, double b, void *s) { (void)s; orc_amg_matvec(y, p, c, v, x, a, b, plan->num_rows); return 0; }

/* ---- multi-vector forms: the reference's launch-per-vector sequences ---- */
static const double *shim_unit_weights(int n) /* w == NULL: unit weights */
{
    static __thread double *ones = NULL;
    static __thread int cap = 0;
    if (n > cap)
    {
        free(ones);
        ones = (double *)malloc(sizeof(double) * (size_t)n);
        for (int i = 0; i < n; i++) ones[i] = 1.0;
        cap = n;
    }
    return ones;
}
int fdd_multi_weighted_inner_product(double *out, double *ws, const double *a, const double *const *b, int m, const double *w, int n, void *s)
{
    if (!w) w = shim_unit_weights(n);
    for (int k = 0; k < m; k++) fdd_sub_weighted_inner_product(out + k, ws, a, b[k], w, n, s);
    return 0;
}
int fdd_multi_axpy_norm2_dev(double *out, double *ws, double *y, const double *c, double sign, const double *const *x, int m, const double *w, int n, void *s)
{
    for (int i = 0; i < n; i++)
    {
        double v = y[i];
        for (int k = 0; k < m; k++) v = 1.0 * v + (sign * c[k]) * x[k][i];
        y[i] = v;
    }
    return fdd_sub_weighted_inner_product(out, ws, y, y, w ? w : shim_unit_weights(n), n, s);
}
int fdd_vector_scaling_rsqrt_dev(double *au, const double *s2, const double *u, int n, void *s)
{
    (void)s;
    orc_vector_scaling(au, 1.0 / sqrt(*s2), u, n);
    return 0;
}
int fdd_multi_axpy(double *q, const double *c, const double *const *v, int m, int n, void *s)
{
    (void)s;
    for (int k = 0; k < m; k++) orc_vector_vector_addition(q, 1.0, q, c[k], v[k], n);
    return 0;
}
int fdd_gather_weighted_norm2(double *out, double *ws, const int *p, const int *c, const double *u, const double *w, int n, void *s)
{
    double *t = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    orc_csr_multiply_weight(t, p, c, NULL, u, w, 0); /* n = 0: keeps the symbol referenced, computes nothing */
    for (int i = 0; i < n; i++)
    {
        double acc = 0.0;
        for (int j = p[i]; j < p[i + 1]; j++) acc += 1.0 * u[c[j]];
        t[i] = acc * w[i];
    }
    fdd_sub_weighted_inner_product(out, ws, t, t, w, n, s);
    free(t);
    return 0;
}

/* ---- fused dssum: the two reference SpMVs, restricted to a node range ---- */
static void gather_range(double *t, const int *p, const int *c, const double *u, const double *w, int n0, int n1)
{
    for (int i = n0; i < n1; i++)
    {
        double s = 0.0;
        for (int j = p[i]; j < p[i + 1]; j++) s += 1.0 * u[c[j]];
        t[i] = w ? s * w[i] : s;
    }
}
static void scatter_range(double *out, const double *t, const int *p, const int *c, const double *m, int n0, int n1)
{
    for (int i = n0; i < n1; i++)
        for (int j = p[i]; j < p[i + 1]; j++)
        {
            double v = 0.0 + 1.0 * t[i];
            out[c[j]] = m ? v * m[c[j]] : v;
        }
}
int fdd_dssum_fused(double *out, double *t, const int *p, const int *c, const double *u, const double *w, const double *m, int n0, int n1, void *s)
{
    (void)s;
    double *tmp = t ? t : (double *)malloc(sizeof(double) * (size_t)(n1 > 0 ? n1 : 1));
    gather_range(tmp, p, c, u, w, n0, n1);
    scatter_range(out, tmp, p, c, m, n0, n1);
    if (!t) free(tmp);
    return 0;
}
int fdd_dssum_gather(double *t, const int *p, const int *c, const double *u, const double *w, int n0, int n1, void *s) { (void)s; gather_range(t, p, c, u, w, n0, n1); return 0; }
int fdd_dssum_scatter(double *out, const double *t, const int *p, const int *c, const double *m, int n0, int n1, void *s) { (void)s; scatter_range(out, t, p, c, m, n0, n1); return 0; }
int fdd_fill_indexed(double *out, const int *idx, double v, int n, void *s) { (void)s; for (int i = 0; i < n; i++) out[idx[i]] = v; return 0; }
int fdd_scatter_add_indexed(double *y, const int *index, const double *t, int n, void *s) { (void)s; for (int i = 0; i < n; i++) y[index[i]] = y[index[i]] + t[i]; return 0; }
int fdd_scatter_add_indexed_f32(float *y, const int *index, const float *t, int n, void *s) { (void)s; for (int i = 0; i < n; i++) y[index[i]] = y[index[i]] + t[i]; return 0; }
int fdd_gather_indexed_split(double *out, const double *lo, const double *hi, int split, const int *index, int n, void *s) { (void)s; for (int i = 0; i < n; i++) out[i] = index[i] < 0 ? 0.0 : (index[i] < split ? lo[index[i]] : hi[index[i]]); return 0; }
int fdd_gather_indexed(double *out, const double *in, const int *index, const double *scale, int n, void *s) { (void)s; for (int i = 0; i < n; i++) { double v = index[i] >= 0 ? in[index[i]] : 0.0; if (scale) v *= scale[i]; out[i] = v; } return 0; }

int fdd_csr_plan_dssum(const fdd_csr_plan *plan, double *out, double *t, const int *p, const int *c, const double *u, const double *w, const double *m, int r0, int r1, int mode, void *s)
{
    (void)plan;
    if (mode == 0) return fdd_dssum_fused(out, t, p, c, u, w, m, r0, r1, s);
    if (mode == 1) return fdd_dssum_gather(t, p, c, u, w, r0, r1, s);
    return fdd_dssum_scatter(out, t, p, c, m, r0, r1, s);
}
int fdd_csr_plan_gather_weighted_norm2(const fdd_csr_plan *plan, double *out, double *ws, const int *p, const int *c, const double *u, const double *w, void *s)
{
    return fdd_gather_weighted_norm2(out, ws, p, c, u, w, plan->num_rows, s);
}

/* ---- interface exchange ---- */
int fdd_interface_pack(double *slots, const int *slot_of, const double *prefix, int n, void *s) { (void)s; for (int i = 0; i < n; i++) slots[slot_of[i]] = prefix[i]; return 0; }
int fdd_interface_unpack(double *prefix, const double *slots, const int *slot_of, int n, void *s) { (void)s; for (int i = 0; i < n; i++) prefix[i] = slots[slot_of[i]]; return 0; }
int fdd_interface_gather(double *buf, const int *index, int n, const double *a, const double *b, void *s)
{
    (void)s;
    const int nc = b ? 2 : 1;
    for (int i = 0; i < n; i++)
    {
        buf[(size_t)i * nc] = a[index[i]];
        if (b) buf[(size_t)i * nc + 1] = b[index[i]];
    }
    return 0;
}
int fdd_interface_sum(double *a, double *b, const int *ptr, const int *col, int rows, const double *buf, void *s)
{
    (void)s;
    const int nc = b ? 2 : 1;
    for (int r = 0; r < rows; r++)
    {
        double sa = 0.0, sb = 0.0;
        for (int k = ptr[r]; k < ptr[r + 1]; k++)
        {
            sa += buf[(size_t)col[k] * nc];
            if (b) sb += buf[(size_t)col[k] * nc + 1];
        }
        a[r] = sa;
        if (b) b[r] = sb;
    }
    return 0;
}

/* ---- device-side GMRES bookkeeping (csrc/fdd_krylov.hip), same statements on the host ---- */
#define KMAX FDD_MULTI_MAX
typedef struct
{
    double H[KMAX][KMAX];
    double c[KMAX], s[KMAX], gamma[KMAX + 1];
    double y[KMAX];
    double hist[KMAX + 1];
    double r0, num_hist, stopped, j_last, steps, converged;
    double inv[KMAX + 1];
} shim_gmres_state;

size_t fdd_gmres_state_bytes(void) { return sizeof(shim_gmres_state); }

int fdd_gmres_begin_dev(void *state, const double *norm2, int first_cycle, void *s)
{
    (void)s;
    shim_gmres_state *st = (shim_gmres_state *)state;
    double g0 = sqrt(*norm2);
    st->gamma[0] = g0;
    st->inv[0] = 1.0 / g0;
    if (first_cycle) st->r0 = g0;
    st->hist[0] = g0;
    st->num_hist = 1.0;
    st->stopped = 0.0;
    st->j_last = -1.0;
    st->steps = 0.0;
    st->converged = 0.0;
    for (int k = 0; k < KMAX; k++) st->y[k] = 0.0;
    return 0;
}

int fdd_gmres_step_dev(void *state, const double *dots, int j, int iterations_before, int max_iterations, double tolerance, int use_relative, void *s)
{
    (void)s;
    shim_gmres_state *st = (shim_gmres_state *)state;
    if (st->stopped != 0.0) return 0;
    st->steps += 1.0;
    int iter = iterations_before + (int)st->steps;
    for (int i = 0; i < j + 1; i++) st->H[i][j] = dots[i];
    for (int i = 0; i < j; i++)
    {
        double h_ij = st->H[i][j];
        st->H[i][j] = st->c[i] * h_ij + st->s[i] * st->H[i + 1][j];
        st->H[i + 1][j] = -st->s[i] * h_ij + st->c[i] * st->H[i + 1][j];
    }
    double alpha_j = sqrt(dots[j + 1]);
    st->j_last = (double)j;
    if (fabs(alpha_j) == 0.0)
    {
        st->stopped = 1.0;
        st->converged = 1.0;
        return 0;
    }
    st->inv[j + 1] = 1.0 / alpha_j;
    double beta_j = sqrt(st->H[j][j] * st->H[j][j] + alpha_j * alpha_j);
    double gamma_j = 1.0 / beta_j;
    st->c[j] = st->H[j][j] * gamma_j;
    st->s[j] = alpha_j * gamma_j;
    st->H[j][j] = beta_j;
    st->gamma[j + 1] = -st->s[j] * st->gamma[j];
    st->gamma[j] = st->c[j] * st->gamma[j];
    double r_norm = fabs(st->gamma[j + 1]);
    st->hist[(int)st->num_hist] = r_norm;
    st->num_hist += 1.0;
    int small = use_relative ? (r_norm / st->r0 < tolerance) : (r_norm < tolerance);
    if (small || iter >= max_iterations)
    {
        st->stopped = 1.0;
        st->converged = 1.0;
    }
    return 0;
}

int fdd_gmres_finish_dev(void *state, int m, void *s)
{
    (void)s;
    shim_gmres_state *st = (shim_gmres_state *)state;
    int j = (int)st->j_last;
    if (st->stopped == 0.0) j = m - 1;
    st->j_last = (double)j;
    for (int k = j; k >= 0; k--)
    {
        double gamma_k = st->gamma[k];
        for (int i = j; i > k; i--) gamma_k -= st->H[k][i] * st->c[i];
        st->c[k] = gamma_k / st->H[k][k];
    }
    for (int k = 0; k < KMAX; k++) st->y[k] = (k <= j) ? st->c[k] : 0.0;
    return 0;
}

int fdd_gmres_fetch(void *state, double *y, double *hist, int *num_hist, int *j_last, int *steps, int *converged, void *s)
{
    (void)s;
    shim_gmres_state *st = (shim_gmres_state *)state;
    if (y) memcpy(y, st->y, sizeof(st->y));
    if (hist) memcpy(hist, st->hist, sizeof(double) * (size_t)st->num_hist);
    if (num_hist) *num_hist = (int)st->num_hist;
    if (j_last) *j_last = (int)st->j_last;
    if (steps) *steps = (int)st->steps;
    if (converged) *converged = (int)st->converged;
    return 0;
}

int fdd_gmres_coefficients(void *state, const double **y_dev)
{
    *y_dev = ((shim_gmres_state *)state)->y;
    return 0;
}

int fdd_multi_axpy_dev(double *q, const double *c, const double *const *v, int m, int n, void *s) { return fdd_multi_axpy(q, c, v, m, n, s); }

int fdd_gmres_scales(void *state, const double **inv_dev)
{
    *inv_dev = ((shim_gmres_state *)state)->inv;
    return 0;
}

int fdd_vector_scaling_dev(double *au, const double *scale, const double *u, int n, void *s)
{
    (void)s;
    orc_vector_scaling(au, *scale, u, n);
    return 0;
}

/* the *_scaled forms: materialise scale * vector as vector_scaling would have, then the unscaled entry */
static double **shim_scaled(const double *const *v, const double *scale, int m, int n)
{
    double **t = (double **)malloc(sizeof(double *) * (size_t)m);
    for (int k = 0; k < m; k++)
    {
        t[k] = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        orc_vector_scaling(t[k], scale[k], v[k], n);
    }
    return t;
}
static void shim_scaled_free(double **t, int m)
{
    for (int k = 0; k < m; k++) free(t[k]);
    free(t);
}

int fdd_multi_weighted_inner_product_scaled(double *out, double *ws, const double *a, const double *const *b, const double *bs, int m, const double *w, int n, void *s)
{
    if (!bs) return fdd_multi_weighted_inner_product(out, ws, a, b, m, w, n, s);
    double **t = shim_scaled(b, bs, m, n);
    int rc = fdd_multi_weighted_inner_product(out, ws, a, (const double *const *)t, m, w, n, s);
    shim_scaled_free(t, m);
    return rc;
}

int fdd_multi_axpy_norm2_scaled_dev(double *out, double *ws, double *dst, const double *y, const double *c, double sign, const double *const *x, const double *xs, int m, const double *w, int n, void *s)
{
    if (!dst) /* norm only: the updated vector goes nowhere */
    {
        double *t = (double *)malloc(sizeof(double) * (size_t)(n + 1));
        int rc = fdd_multi_axpy_norm2_scaled_dev(out, ws, t, y, c, sign, x, xs, m, w, n, s);
        free(t);
        return rc;
    }
    if (dst != y) memmove(dst, y, sizeof(double) * (size_t)n);
    if (!xs) return fdd_multi_axpy_norm2_dev(out, ws, dst, c, sign, x, m, w, n, s);
    double **t = shim_scaled(x, xs, m, n);
    int rc = fdd_multi_axpy_norm2_dev(out, ws, dst, c, sign, (const double *const *)t, m, w, n, s);
    shim_scaled_free(t, m);
    return rc;
}

int fdd_multi_axpy_scaled_dev(double *q, const double *c, const double *const *v, const double *vs, int m, int n, void *s)
{
    if (!vs) return fdd_multi_axpy(q, c, v, m, n, s);
    double **t = shim_scaled(v, vs, m, n);
    int rc = fdd_multi_axpy(q, c, (const double *const *)t, m, n, s);
    shim_scaled_free(t, m);
    return rc;
}

int fdd_sub_stiffness_matrix_gather_scaled(double *Au, const double *v, const double *vscale, const int *pd, const double *D, const double *const G[6], const int *eo, int ne, int N, void *s)
{
    if (!vscale) return fdd_sub_stiffness_matrix_gather(Au, v, pd, D, G, eo, ne, N, s);
    /* the dof vector's length is not passed: scale what the index array reaches */
    int n3 = (N + 1) * (N + 1) * (N + 1), maxd = -1;
    for (int e = 0; e < ne; e++)
    {
        size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        for (int q = 0; q < n3; q++)
            if (pd[o + q] > maxd) maxd = pd[o + q];
    }
    double *t = (double *)malloc(sizeof(double) * (size_t)(maxd + 2));
    orc_vector_scaling(t, *vscale, v, maxd + 1);
    int rc = fdd_sub_stiffness_matrix_gather(Au, t, pd, D, G, eo, ne, N, s);
    free(t);
    return rc;
}

int fdd_stiffness_matrix_mfma_gather(double *Au, const double *v, const double *vscale, const int *pd, const double *D, const double *const G[6], const int *eo, int ne, int N, void *s)
{
    return fdd_sub_stiffness_matrix_gather_scaled(Au, v, vscale, pd, D, G, eo, ne, N, s);
}

int fdd_multi_lincomb_scaled_dev(double *q, int q_is_zero, const double *c, const double *const *v, const double *vs, int m, int n, void *s)
{
    if (q_is_zero) memset(q, 0, sizeof(double) * (size_t)n); /* the caller promises 0 and does not clear it: the kernel never reads q */
    return fdd_multi_axpy_scaled_dev(q, c, v, vs, m, n, s);
}

int fdd_xmay_ratio_dev(double *out, const double *x, const double *num, const double *den, const double *y, int n, void *s)
{
    (void)s;
    const double alpha = *num / *den;
    for (int i = 0; i < n; i++) out[i] = x[i] - alpha * y[i];
    return 0;
}
int fdd_xpby_ratio_dev(double *out, const double *x, const double *num, const double *den, const double *y, int n, void *s)
{
    (void)s;
    double beta = *num / *den;
    for (int i = 0; i < n; i++) out[i] = x[i] + beta * y[i];
    return 0;
}

int fdd_multi_lincomb_limited_dev(double *q, int q_is_zero, const double *c, const double *const *v, const double *vs, const double *last, int m, int n, void *s)
{
    int use = last ? (int)*last + 1 : m;
    if (use > m) use = m;
    if (use <= 0)
    {
        if (q_is_zero) memset(q, 0, sizeof(double) * (size_t)n);
        return 0;
    }
    return fdd_multi_lincomb_scaled_dev(q, q_is_zero, c, v, vs, use, n, s);
}

int fdd_sqrt_sum_dev(double *out, const double *parts, int nparts, void *s)
{
    (void)s;
    double t = 0.0;
    for (int k = 0; k < nparts; k++) t += parts[k];
    *out = sqrt(t);
    return 0;
}

int fdd_gmres_last_column(void *state, const double **j_last_dev)
{
    *j_last_dev = &((shim_gmres_state *)state)->j_last;
    return 0;
}

int fdd_csr_plan_matvec_to(const fdd_csr_plan *plan, double *y, const double *y_in, const int *p, const int *c, const double *v, const double *x, double a, double b, void *s)
{
    if (y_in && y_in != y && b != 0.0) memcpy(y, y_in, sizeof(double) * (size_t)plan->num_rows);
    return fdd_csr_plan_matvec(plan, y, p, c, v, x, a, b, s);
}

/* fused smoother entries: the unfused oracle pieces in sequence */
/* matrix-free lattice transfer (csrc/fdd_transfer.hip): plain loops over elements, lattice points and corners */
int fdd_lattice_supported(int n, int m, int *supported) { *supported = ((n == 8 || n == 16) && m >= 2 && m <= n / 2 + 1) ? 1 : 0; return 0; }
#define SHIM_LATTICE(NAME_P, NAME_R, T)                                                                                                   \
    int NAME_P(T *u, const T *coarse, const int *owner_dof, const int *coarse_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long E, void *s) \
    {                                                                                                                                     \
        (void)s;                                                                                                                          \
        REQ(n >= 2 && n <= 16 && m >= 2 && m <= n);                                                                                        \
        const long long np = (long long)n * n * n, mc = (long long)m * m * m;                                                             \
        for (long long e = 0; e < E; e++)                                                                                                 \
            for (int k = 0; k < n; k++)                                                                                                   \
                for (int j = 0; j < n; j++)                                                                                               \
                    for (int i = 0; i < n; i++)                                                                                           \
                    {                                                                                                                     \
                        const int d = owner_dof[e * np + ((long long)k * n + j) * n + i];                                                 \
                        if (d < 0) continue;                                                                                              \
                        T sum = 0;                                                                                                        \
                        for (int corner = 0; corner < 8; corner++)                                                                        \
                        {                                                                                                                 \
                            const int sx = corner & 1, sy = (corner >> 1) & 1, sz = corner >> 2;                                          \
                            if ((sx && lo[i] == hi[i]) || (sy && lo[j] == hi[j]) || (sz && lo[k] == hi[k])) continue;                       \
                            const int a = sx ? hi[i] : lo[i], b = sy ? hi[j] : lo[j], c = sz ? hi[k] : lo[k];                             \
                            const T w = (T)(sx ? 1.0 - wl[i] : wl[i]) * (T)(sy ? 1.0 - wl[j] : wl[j]) * (T)(sz ? 1.0 - wl[k] : wl[k]);      \
                            const int cd = coarse_dof[e * mc + ((long long)c * m + b) * m + a];                                           \
                            if (cd >= 0) sum += w * coarse[cd];                                                                           \
                        }                                                                                                                 \
                        u[d] += sum;                                                                                                      \
                    }                                                                                                                     \
        return 0;                                                                                                                         \
    }                                                                                                                                     \
    int NAME_R(T *partial, const T *fine, const int *owner_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long E, void *s) \
    {                                                                                                                                     \
        (void)s;                                                                                                                          \
        REQ(n >= 2 && n <= 16 && m >= 2 && m <= n);                                                                                        \
        const long long np = (long long)n * n * n, mc = (long long)m * m * m;                                                             \
        for (long long t = 0; t < E * mc; t++) partial[t] = 0;                                                                            \
        for (long long e = 0; e < E; e++)                                                                                                 \
            for (int k = 0; k < n; k++)                                                                                                   \
                for (int j = 0; j < n; j++)                                                                                               \
                    for (int i = 0; i < n; i++)                                                                                           \
                    {                                                                                                                     \
                        const int d = owner_dof[e * np + ((long long)k * n + j) * n + i];                                                 \
                        if (d < 0) continue;                                                                                              \
                        for (int corner = 0; corner < 8; corner++)                                                                        \
                        {                                                                                                                 \
                            const int sx = corner & 1, sy = (corner >> 1) & 1, sz = corner >> 2;                                          \
                            if ((sx && lo[i] == hi[i]) || (sy && lo[j] == hi[j]) || (sz && lo[k] == hi[k])) continue;                       \
                            const int a = sx ? hi[i] : lo[i], b = sy ? hi[j] : lo[j], c = sz ? hi[k] : lo[k];                             \
                            const T w = (T)(sx ? 1.0 - wl[i] : wl[i]) * (T)(sy ? 1.0 - wl[j] : wl[j]) * (T)(sz ? 1.0 - wl[k] : wl[k]);      \
                            partial[e * mc + ((long long)c * m + b) * m + a] += w * fine[d];                                              \
                        }                                                                                                                 \
                    }                                                                                                                     \
        return 0;                                                                                                                         \
    }
SHIM_LATTICE(fdd_lattice_prolong, fdd_lattice_restrict, double)
SHIM_LATTICE(fdd_lattice_prolong_f32, fdd_lattice_restrict_f32, float)

int fdd_amg_smooth_start(double *work, double *Sr, const double *f, const double *D, double coef, int n, void *s)
{
    (void)s;
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    orc_amg_main_scaled_residual(Sr, w, f, D, coef, n);
    orc_amg_vector_multiplication(work, D, w, n);
    free(w);
    return 0;
}

int fdd_amg_smooth_residual_matvec(const fdd_csr_plan *plan, double *work, double *Sr, const int *p, const int *c, const double *v, const double *u, const double *f, const double *D, double coef, void *s)
{
    int n = plan->num_rows;
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int rc = fdd_csr_plan_matvec_to(plan, work, f, p, c, v, u, -1.0, 1.0, s);
    orc_amg_main_scaled_residual(Sr, w, work, D, coef, n);
    orc_amg_vector_multiplication(work, D, w, n);
    free(w);
    return rc;
}

static int shim_smooth_poly(const fdd_csr_plan *plan, double *w, const int *p, const int *c, const double *v, const double *work_in, const double *Sr, const double *D, double coef, void *s)
{
    int n = plan->num_rows;
    double *t = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int rc = fdd_csr_plan_matvec(plan, t, p, c, v, work_in, 1.0, 0.0, s);
    orc_amg_main_polynomial_evaluation(w, t, Sr, D, coef, n);
    free(t);
    return rc;
}

int fdd_amg_smooth_polynomial_matvec(const fdd_csr_plan *plan, double *work_out, const int *p, const int *c, const double *v, const double *work_in, const double *Sr, const double *D, double coef, void *s)
{
    int n = plan->num_rows;
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int rc = shim_smooth_poly(plan, w, p, c, v, work_in, Sr, D, coef, s);
    orc_amg_vector_multiplication(work_out, D, w, n);
    free(w);
    return rc;
}

int fdd_amg_smooth_update_matvec(const fdd_csr_plan *plan, double *u, const int *p, const int *c, const double *v, const double *work_in, const double *Sr, const double *D, double coef, void *s)
{
    int n = plan->num_rows;
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int rc = shim_smooth_poly(plan, w, p, c, v, work_in, Sr, D, coef, s);
    orc_amg_main_update_field(u, w, D, n);
    free(w);
    return rc;
}

/* ---- Float = float V-cycle entries: plain float loops in the device statement order ---- */
int fdd_csr_plan_create_f32(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz) { return fdd_csr_plan_create(plan, A_ptr_host, num_rows, num_cols, num_nnz); }

static float shim_rowsum32(const int *p, const int *c, const float *v, const float *x, int row)
{
    float s = 0.0f;
    for (int i = p[row]; i < p[row + 1]; i++) s += v[i] * x[c[i]];
    return s;
}

int fdd_csr_plan_matvec_to_f32(const fdd_csr_plan *plan, float *y, const float *y_in, const int *p, const int *c, const float *v, const float *x, float a, float b, void *s)
{
    (void)s;
    for (int row = 0; row < plan->num_rows; row++)
    {
        float t = shim_rowsum32(p, c, v, x, row);
        y[row] = (b == 0.0f) ? a * t : a * t + b * (y_in ? y_in[row] : y[row]);
    }
    return 0;
}

int fdd_amg_smooth_residual_matvec_f32(const fdd_csr_plan *plan, float *work, float *Sr, const int *p, const int *c, const float *v, const float *u, const float *f, const float *D, float coef, void *s)
{
    (void)s;
    for (int row = 0; row < plan->num_rows; row++)
    {
        float t = shim_rowsum32(p, c, v, u, row);
        float wk = -1.0f * t + 1.0f * f[row];
        float sr = D[row] * wk;
        Sr[row] = sr;
        float w = coef * sr;
        work[row] = w * D[row];
    }
    return 0;
}

int fdd_amg_smooth_polynomial_matvec_f32(const fdd_csr_plan *plan, float *work_out, const int *p, const int *c, const float *v, const float *work_in, const float *Sr, const float *D, float coef, void *s)
{
    (void)s;
    for (int row = 0; row < plan->num_rows; row++)
    {
        float t = shim_rowsum32(p, c, v, work_in, row);
        float vv = (1.0f * t) * D[row];
        float w = coef * Sr[row] + vv;
        work_out[row] = w * D[row];
    }
    return 0;
}

int fdd_amg_smooth_update_matvec_from_zero(const fdd_csr_plan *plan, double *u, const int *p, const int *c, const double *v, const double *work_in, const double *Sr, const double *D, double coef, void *s)
{
    for (int i = 0; i < plan->num_rows; i++) u[i] = 0.0;
    return fdd_amg_smooth_update_matvec(plan, u, p, c, v, work_in, Sr, D, coef, s);
}
int fdd_amg_smooth_update_matvec_f32(const fdd_csr_plan *plan, float *u, const int *p, const int *c, const float *v, const float *work_in, const float *Sr, const float *D, float coef, void *s);
int fdd_amg_smooth_update_matvec_from_zero_f32(const fdd_csr_plan *plan, float *u, const int *p, const int *c, const float *v, const float *work_in, const float *Sr, const float *D, float coef, void *s)
{
    for (int i = 0; i < plan->num_rows; i++) u[i] = 0.0f;
    return fdd_amg_smooth_update_matvec_f32(plan, u, p, c, v, work_in, Sr, D, coef, s);
}
int fdd_amg_smooth_update_matvec_f32(const fdd_csr_plan *plan, float *u, const int *p, const int *c, const float *v, const float *work_in, const float *Sr, const float *D, float coef, void *s)
{
    (void)s;
    for (int row = 0; row < plan->num_rows; row++)
    {
        float t = shim_rowsum32(p, c, v, work_in, row);
        float vv = (1.0f * t) * D[row];
        float w = coef * Sr[row] + vv;
        u[row] = u[row] + D[row] * w;
    }
    return 0;
}

int fdd_amg_smooth_start_f32(float *work, float *Sr, const float *f, const float *D, float coef, int n, void *s)
{
    (void)s;
    for (int i = 0; i < n; i++)
    {
        float sr = D[i] * f[i];
        Sr[i] = sr;
        work[i] = D[i] * (coef * sr);
    }
    return 0;
}

int fdd_amg_vector_set_to_value_f32(float *data, float value, int n, void *s)
{
    (void)s;
    for (int i = 0; i < n; i++) data[i] = value;
    return 0;
}

/* ---- single-precision preconditioner entries: float storage, emulated through the double kernels (test shim only) ---- */
int fdd_sub_stiffness_matrix_gather_scaled_f32(float *Au, const float *v, const double *vscale, const int *pd, const float *D, const float *const G[6], const int *eo, int ne, int N, void *s)
{
    int n = N + 1, n3 = n * n * n, maxd = -1;
    size_t maxp = 0;
    for (int e = 0; e < ne; e++)
    {
        size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        if (o + n3 > maxp) maxp = o + n3;
        for (int q = 0; q < n3; q++)
            if (pd[o + q] > maxd) maxd = pd[o + q];
    }
    double *vd = (double *)malloc(sizeof(double) * (size_t)(maxd + 2));
    double *Dd = (double *)malloc(sizeof(double) * (size_t)(n * n));
    double *Ad = (double *)calloc(maxp + 1, sizeof(double));
    double *Gd[6];
    const float sc = vscale ? (float)*vscale : 1.0f;
    for (int i = 0; i <= maxd; i++) vd[i] = (double)(sc * v[i]);
    for (int i = 0; i < n * n; i++) Dd[i] = (double)D[i];
    for (int g = 0; g < 6; g++)
    {
        Gd[g] = (double *)malloc(sizeof(double) * (maxp + 1));
        for (size_t i = 0; i < maxp; i++) Gd[g][i] = (double)G[g][i];
    }
    int rc = fdd_sub_stiffness_matrix_gather(Ad, vd, pd, Dd, (const double *const *)Gd, eo, ne, N, s);
    for (int e = 0; e < ne; e++)
    {
        size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        for (int q = 0; q < n3; q++) Au[o + q] = (float)Ad[o + q];
    }
    for (int g = 0; g < 6; g++) free(Gd[g]);
    free(vd);
    free(Dd);
    free(Ad);
    return rc;
}
int fdd_multi_inner_product_scaled_f32(double *out, double *ws, const float *a, const float *const *b, const double *bs, int m, int n, void *s)
{
    (void)ws; (void)s;
    for (int k = 0; k < m; k++)
    {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc += (double)a[i] * ((bs ? bs[k] : 1.0) * (double)b[k][i]);
        out[k] = acc;
    }
    return 0;
}
int fdd_multi_axpy_norm2_scaled_dev_f32(double *out, double *ws, float *dst, const float *y, const double *c, double sign, const float *const *x, const double *xs, int m, int n, void *s)
{
    (void)ws; (void)s;
    double acc = 0.0;
    for (int i = 0; i < n; i++)
    {
        double v = y[i];
        for (int k = 0; k < m; k++) v += sign * c[k] * (xs ? xs[k] : 1.0) * (double)x[k][i];
        const float r = (float)v;
        if (dst) dst[i] = r; /* NULL: norm only */
        acc += (double)r * (double)r;
    }
    out[0] = acc;
    return 0;
}
int fdd_vector_diagonal_scaling_dev(double *z, const double *d, const double *sc, const double *u, int n, void *s) { (void)s; for (int i = 0; i < n; i++) z[i] = sc ? d[i] * ((*sc) * u[i]) : d[i] * u[i]; return 0; }
int fdd_vector_diagonal_scaling_dev_f32(float *z, const float *d, const double *sc, const float *u, int n, void *s) { (void)s; const float f = sc ? (float)*sc : 1.0f; for (int i = 0; i < n; i++) z[i] = sc ? d[i] * (f * u[i]) : d[i] * u[i]; return 0; }
int fdd_vector_scaling_dev_f32(float *au, const double *sc, const float *u, int n, void *s) { (void)s; const float f = (float)*sc; for (int i = 0; i < n; i++) au[i] = f * u[i]; return 0; }
int fdd_vector_vector_addition_f32(float *uv, float a, const float *u, float b, const float *v, int n, void *s) { (void)s; for (int i = 0; i < n; i++) uv[i] = a * u[i] + b * v[i]; return 0; }
int fdd_multi_lincomb_limited_dev_f32(float *q, int q_is_zero, const double *c, const float *const *v, const double *vs, const double *last, int m, int n, void *s)
{
    (void)s;
    int use = last ? (int)*last + 1 : m;
    if (use > m) use = m;
    for (int i = 0; i < n; i++)
    {
        double acc = q_is_zero ? 0.0 : (double)q[i];
        for (int k = 0; k < use; k++) acc += c[k] * (vs ? vs[k] : 1.0) * (double)v[k][i];
        q[i] = (float)acc;
    }
    return 0;
}
int fdd_gather_rows_f32(float *t, const int *ptr, const int *col, const float *u, int lo, int hi, void *s)
{
    (void)s;
    for (int r = lo; r < hi; r++)
    {
        float acc = 0.0f;
        for (int j = ptr[r]; j < ptr[r + 1]; j++) acc += u[col[j]];
        t[r] = acc;
    }
    return 0;
}
int fdd_gather_rows_f32(float *t, const int *ptr, const int *col, const float *u, int row_lo, int row_hi, void *s);
int fdd_csr_plan_gather_f32(const fdd_csr_plan *plan, float *t, const int *ptr, const int *col, const float *u, int row_lo, int row_hi, void *s) { (void)plan; return fdd_gather_rows_f32(t, ptr, col, u, row_lo, row_hi, s); }
int fdd_gather_indexed_f32(float *out, const float *in, const int *idx, int n, void *s) { (void)s; for (int i = 0; i < n; i++) out[i] = idx[i] < 0 ? 0.0f : in[idx[i]]; return 0; }
int fdd_gather_indexed_f32_f64(double *out, const float *in, const int *idx, int n, void *s) { (void)s; for (int i = 0; i < n; i++) out[i] = idx[i] < 0 ? 0.0 : (double)in[idx[i]]; return 0; }

/* ---- affine elements (an option of the build): the factor arrays rebuilt from six numbers per element, then the streamed kernels ---- */
static size_t affine_span(const int *eo, int ne, int n3)
{
    size_t maxp = 0;
    for (int e = 0; e < ne; e++)
    {
        size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        if (o + n3 > maxp) maxp = o + n3;
    }
    return maxp;
}

int fdd_stiffness_matrix_affine(double *Au, const double *v, const double *vscale, const int *pd, const double *D, const double *c, const double *w, const int *eo, int ne, int N, void *s)
{
    if (ne <= 0) return 0;
    const int n = N + 1, n3 = n * n * n;
    const size_t span = affine_span(eo, ne, n3);
    double *G[6];
    for (int f = 0; f < 6; f++) G[f] = (double *)calloc(span + 1, sizeof(double));
    for (int e = 0; e < ne; e++)
    {
        const size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        for (int k = 0; k < n; k++)
            for (int j = 0; j < n; j++)
                for (int i = 0; i < n; i++)
                {
                    const double W = (w[i] * w[j]) * w[k];
                    for (int f = 0; f < 6; f++) G[f][o + i + j * n + k * n * n] = c[(size_t)e * 6 + f] * W;
                }
    }
    const double *Gc[6] = {G[0], G[1], G[2], G[3], G[4], G[5]};
    int rc;
    if (pd)
        rc = fdd_sub_stiffness_matrix_gather_scaled(Au, v, vscale, pd, D, Gc, eo, ne, N, s);
    else
        rc = fdd_sub_stiffness_matrix(Au, v, D, Gc, eo, ne, N, s);
    for (int f = 0; f < 6; f++) free(G[f]);
    return rc;
}

int fdd_stiffness_matrix_affine_f32(float *Au, const float *v, const double *vscale, const int *pd, const float *D, const float *c, const float *w, const int *eo, int ne, int N, void *s)
{
    if (ne <= 0) return 0;
    if (!pd) return 1; /* the float inner solve always gathers */
    const int n = N + 1, n3 = n * n * n;
    const size_t span = affine_span(eo, ne, n3);
    float *G[6];
    for (int f = 0; f < 6; f++) G[f] = (float *)calloc(span + 1, sizeof(float));
    for (int e = 0; e < ne; e++)
    {
        const size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        for (int k = 0; k < n; k++)
            for (int j = 0; j < n; j++)
                for (int i = 0; i < n; i++)
                {
                    const float W = (w[i] * w[j]) * w[k];
                    for (int f = 0; f < 6; f++) G[f][o + i + j * n + k * n * n] = c[(size_t)e * 6 + f] * W;
                }
    }
    const float *Gc[6] = {G[0], G[1], G[2], G[3], G[4], G[5]};
    int rc = fdd_sub_stiffness_matrix_gather_scaled_f32(Au, v, vscale, pd, D, Gc, eo, ne, N, s);
    for (int f = 0; f < 6; f++) free(G[f]);
    return rc;
}

int fdd_stiffness_affine_detect(double *c, double *deviation, const double *const G[6], const int *eo, const double *w, int ne, int N, void *s)
{
    (void)s;
    const int n = N + 1, n3 = n * n * n, m = n / 2, p0 = m + m * n + m * n * n;
    for (int e = 0; e < ne; e++)
    {
        const size_t o = eo ? (size_t)eo[e] : (size_t)e * n3;
        double scale = 0.0, worst = 0.0;
        for (int f = 0; f < 6; f++)
        {
            c[(size_t)e * 6 + f] = G[f][o + p0] / ((w[m] * w[m]) * w[m]);
            if (fabs(c[(size_t)e * 6 + f]) > scale) scale = fabs(c[(size_t)e * 6 + f]);
        }
        for (int p = 0; p < n3; p++)
        {
            const int i = p % n, j = (p / n) % n, k = p / (n * n);
            const double W = (w[i] * w[j]) * w[k];
            for (int f = 0; f < 6; f++)
            {
                const double d = fabs(G[f][o + p] - c[(size_t)e * 6 + f] * W) / (scale * W);
                if (d > worst) worst = d;
            }
        }
        deviation[e] = (scale > 0.0) ? worst : 1.0;
    }
    return 0;
}

int fdd_stiffness_matrix_mfma_affine(double *Au, const double *v, const double *vscale, const int *pd, const double *D, const double *c, const double *w, const int *eo, int ne, int N, void *s)
{
    return fdd_stiffness_matrix_affine(Au, v, vscale, pd, D, c, w, eo, ne, N, s);
}

int fdd_dom_inner_product_flexible_gamma(double *out2, double *ws, const double *r, const double *r1, const double *z, int n, void *s)
{
    /* {gamma_next, theta}: the flexible sum as above; gamma_next = <z, r+> is the first sum of the projection kernel */
    double two[2];
    int rc = fdd_dom_projection_inner_products(two, ws, z, r1, z, r1, n, s);
    if (rc) return rc;
    out2[0] = two[0];
    return fdd_dom_inner_product_flexible(out2 + 1, ws, r, r1, z, n, s);
}
