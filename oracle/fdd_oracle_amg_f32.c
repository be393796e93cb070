/*
 * fdd_oracle_amg_f32.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * The low-order AMG V-cycle of fdd_oracle_amg.c with `Float = float`
 * (AMG/config.hpp:4, swept by run.py:157): matrices, vectors, Chebyshev
 * coefficients and every operation of the cycle in single precision, the
 * residual cast down on entry and the correction cast up on exit
 * (subdomain.tpp:4008,4142 copy Float data; the casts are subdomain.okl:268-282).
 * Statement order of the DEVICE branches:
 *   scaled_residual        subdomain.tpp:34-39  (work = f; work = -A u + work; main_scaled_residual)
 *   polynomial_evaluation  subdomain.tpp:62-67  (work = D*w; v = A work; main_polynomial_evaluation)
 *   update_field           subdomain.tpp:79-82
 *   kernels                AMG/kernels.cu:11-94, SpMV y = alpha*A*x + beta*y AMG/csr_matrix.cpp:112-134
 *
 * The hierarchy is an input (HYPRE is not available: fdd_oracle_amg.c); its
 * double arrays are rounded to float here.  Coarsest level: dense Gaussian
 * elimination in double on the rounded matrix, result rounded to float
 * (hypre_GaussElimSolve works in HYPRE_Real = double).  Parity unpinned by
 * upstream, as for the double V-cycle.
 *
 * See fdd_oracle.h for who may use this file.
 */
#include "fdd_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
    int rows, cols;
    int *ptr, *col;
    float *val;
} csr32;

typedef struct
{
    csr32 A, P, R;
    float *D_val, *coefs;
    float *f, *u, *r, *v, *w, *work;
} level32;

struct orc_amg32
{
    int num_levels, cheby_order, num_vcycles;
    level32 *lev;
    double *coarse_dense;
};

static void *xcalloc32(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    if (!p)
    {
        fprintf(stderr, "fdd_oracle: out of memory\n");
        abort();
    }
    return p;
}

static void csr32_copy(csr32 *dst, int rows, int cols, const int *ptr, const int *col, const double *val)
{
    int nnz = ptr[rows];
    dst->rows = rows;
    dst->cols = cols;
    dst->ptr = (int *)xcalloc32((size_t)rows + 1, sizeof(int));
    dst->col = (int *)xcalloc32((size_t)nnz, sizeof(int));
    dst->val = (float *)xcalloc32((size_t)nnz, sizeof(float));
    memcpy(dst->ptr, ptr, ((size_t)rows + 1) * sizeof(int));
    memcpy(dst->col, col, (size_t)nnz * sizeof(int));
    for (int i = 0; i < nnz; i++) dst->val[i] = (float)val[i];
}

/* R = P^T with rows in ascending column order of P's rows (as CSR_Matrix::transpose builds it) */
static void csr32_transpose(csr32 *dst, const csr32 *src)
{
    int nnz = src->ptr[src->rows];
    dst->rows = src->cols;
    dst->cols = src->rows;
    dst->ptr = (int *)xcalloc32((size_t)dst->rows + 1, sizeof(int));
    dst->col = (int *)xcalloc32((size_t)nnz, sizeof(int));
    dst->val = (float *)xcalloc32((size_t)nnz, sizeof(float));
    for (int i = 0; i < nnz; i++) dst->ptr[src->col[i] + 1]++;
    for (int r = 0; r < dst->rows; r++) dst->ptr[r + 1] += dst->ptr[r];
    int *next = (int *)xcalloc32((size_t)dst->rows, sizeof(int));
    memcpy(next, dst->ptr, (size_t)dst->rows * sizeof(int));
    for (int r = 0; r < src->rows; r++)
        for (int i = src->ptr[r]; i < src->ptr[r + 1]; i++)
        {
            int k = next[src->col[i]]++;
            dst->col[k] = r;
            dst->val[k] = src->val[i];
        }
    free(next);
}

static void csr32_free(csr32 *m)
{
    free(m->ptr);
    free(m->col);
    free(m->val);
}

/* AMG/csr_matrix.cpp:112-134 in float; beta == 0 does not read y */
static void matvec32(float *y, const csr32 *A, const float *x, float alpha, float beta)
{
    for (int row = 0; row < A->rows; row++)
    {
        float s = 0.0f;
        for (int i = A->ptr[row]; i < A->ptr[row + 1]; i++) s += A->val[i] * x[A->col[i]];
        y[row] = (beta == 0.0f) ? alpha * s : alpha * s + beta * y[row];
    }
}

orc_amg32 *orc_amg32_create(int num_levels, int cheby_order, int num_vcycles)
{
    orc_amg32 *a = (orc_amg32 *)xcalloc32(1, sizeof(orc_amg32));
    a->num_levels = num_levels;
    a->cheby_order = cheby_order;
    a->num_vcycles = num_vcycles;
    a->lev = (level32 *)xcalloc32((size_t)num_levels, sizeof(level32));
    return a;
}

void orc_amg32_set_level(orc_amg32 *a, int l, int n, const int *A_ptr, const int *A_col, const double *A_val, const double *D_val, const double *coefs, int n_coarse, const int *P_ptr, const int *P_col, const double *P_val)
{
    level32 *L = &a->lev[l];
    csr32_copy(&L->A, n, n, A_ptr, A_col, A_val);
    if (P_ptr)
    {
        csr32_copy(&L->P, n, n_coarse, P_ptr, P_col, P_val);
        csr32_transpose(&L->R, &L->P);
    }
    L->D_val = (float *)xcalloc32((size_t)n, sizeof(float));
    for (int i = 0; i < n; i++) L->D_val[i] = (float)D_val[i];
    L->coefs = (float *)xcalloc32((size_t)a->cheby_order, sizeof(float));
    for (int i = 0; i < a->cheby_order; i++) L->coefs[i] = (float)coefs[i];
    float **vecs[6] = {&L->f, &L->u, &L->r, &L->v, &L->w, &L->work};
    for (int k = 0; k < 6; k++) *vecs[k] = (float *)xcalloc32((size_t)n, sizeof(float));
    if (l == a->num_levels - 1)
    {
        a->coarse_dense = (double *)xcalloc32((size_t)n * n, sizeof(double));
        for (int r = 0; r < n; r++)
            for (int i = L->A.ptr[r]; i < L->A.ptr[r + 1]; i++) a->coarse_dense[(size_t)r * n + L->A.col[i]] += (double)L->A.val[i];
    }
}

void orc_amg32_destroy(orc_amg32 *a)
{
    if (!a) return;
    for (int l = 0; l < a->num_levels; l++)
    {
        level32 *L = &a->lev[l];
        csr32_free(&L->A);
        csr32_free(&L->P);
        csr32_free(&L->R);
        free(L->D_val);
        free(L->coefs);
        free(L->f);
        free(L->u);
        free(L->r);
        free(L->v);
        free(L->w);
        free(L->work);
    }
    free(a->lev);
    free(a->coarse_dense);
    free(a);
}

/* Chebyshev smoother, device branches of subdomain.tpp:19-83 in float */
static void smooth32(orc_amg32 *a, int l)
{
    level32 *L = &a->lev[l];
    int n = L->A.rows, c = a->cheby_order;
    memcpy(L->work, L->f, (size_t)n * sizeof(float)); /* :34 */
    matvec32(L->work, &L->A, L->u, -1.0f, 1.0f);       /* :35 */
    for (int i = 0; i < n; i++)                        /* main_scaled_residual, AMG/kernels.cu:25-41 */
    {
        L->r[i] = L->D_val[i] * L->work[i];
        L->w[i] = L->coefs[c - 1] * L->r[i];
    }
    for (int p = c - 2; p >= 0; p--)
    {
        for (int i = 0; i < n; i++) L->work[i] = L->D_val[i] * L->w[i]; /* vector_multiplication, :62 */
        matvec32(L->v, &L->A, L->work, 1.0f, 0.0f);                     /* :63 */
        for (int i = 0; i < n; i++)                                     /* main_polynomial_evaluation, AMG/kernels.cu:43-59 */
        {
            L->v[i] *= L->D_val[i];
            L->w[i] = L->coefs[p] * L->r[i] + L->v[i];
        }
    }
    for (int i = 0; i < n; i++) L->u[i] += L->D_val[i] * L->w[i]; /* main_update_field, AMG/kernels.cu:61-76 */
}

static void gauss_solve32(const double *A, const float *b, float *x, int n)
{
    double *M = (double *)xcalloc32((size_t)n * n, sizeof(double));
    double *y = (double *)xcalloc32((size_t)n, sizeof(double));
    double *z = (double *)xcalloc32((size_t)n, sizeof(double));
    memcpy(M, A, (size_t)n * n * sizeof(double));
    for (int i = 0; i < n; i++) y[i] = (double)b[i];
    for (int k = 0; k < n; k++)
        for (int i = k + 1; i < n; i++)
        {
            double m = M[(size_t)i * n + k] / M[(size_t)k * n + k];
            if (m != 0.0)
            {
                for (int j = k + 1; j < n; j++) M[(size_t)i * n + j] -= m * M[(size_t)k * n + j];
                y[i] -= m * y[k];
            }
        }
    for (int i = n - 1; i >= 0; i--)
    {
        double s = y[i];
        for (int j = i + 1; j < n; j++) s -= M[(size_t)i * n + j] * z[j];
        z[i] = s / M[(size_t)i * n + i];
    }
    for (int i = 0; i < n; i++) x[i] = (float)z[i];
    free(M);
    free(y);
    free(z);
}

/* subdomain.tpp:4011-4139 with Float = float */
void orc_amg32_vcycle(orc_amg32 *a, double *u0, const double *f0)
{
    int L = a->num_levels, n0 = a->lev[0].A.rows;
    for (int i = 0; i < n0; i++) a->lev[0].f[i] = (float)f0[i];
    for (int i = 0; i < n0; i++) a->lev[0].u[i] = 0.0f;
    for (int iter = 0; iter < a->num_vcycles; iter++)
    {
        for (int l = 0; l < L - 1; l++)
        {
            level32 *lv = &a->lev[l];
            if (l > 0)
                for (int i = 0; i < lv->A.rows; i++) lv->u[i] = 0.0f;
            smooth32(a, l);
            memcpy(lv->v, lv->f, (size_t)lv->A.rows * sizeof(float));
            matvec32(lv->v, &lv->A, lv->u, -1.0f, 1.0f);
            matvec32(a->lev[l + 1].f, &lv->R, lv->v, 1.0f, 0.0f);
        }
        gauss_solve32(a->coarse_dense, a->lev[L - 1].f, a->lev[L - 1].u, a->lev[L - 1].A.rows);
        for (int l = L - 1; l > 0; l--)
        {
            level32 *fine = &a->lev[l - 1];
            matvec32(fine->u, &fine->P, a->lev[l].u, 1.0f, 1.0f);
            smooth32(a, l - 1);
        }
    }
    for (int i = 0; i < n0; i++) u0[i] = (double)a->lev[0].u[i];
}
