/*
 * fdd_oracle_amg.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Serial restatement of the low-order AMG V-cycle preconditioner of the FDD
 * subdomain solve, given a hierarchy:
 *   Subdomain::low_order_preconditioner      subdomain.tpp:3987-4159
 *   scaled_residual / polynomial_evaluation / update_field ("host" branches)
 *                                             subdomain.tpp:19-83
 *   amg::CSR_Matrix::matvec (host loop)       AMG/csr_matrix.cpp:112-134
 *
 * The reference takes the hierarchy (A_l, the Chebyshev diagonal scaling
 * D_val_l, the Chebyshev coefficients, P_l and R_l = P_l^T) from HYPRE
 * BoomerAMG (subdomain.tpp:3474-3549) and solves the coarsest level with
 * hypre_GaussElimSolve (subdomain.tpp:4080-4088).  HYPRE is not available, so
 * the hierarchy is an INPUT here (tests build one; documented deviation,
 * SURVEY.md 8(c)); what is pinned is the V-cycle arithmetic given a hierarchy.
 * The coarsest level is plain dense Gaussian elimination without pivoting.
 *
 * See fdd_oracle.h for who may use this file and the parity-pin statement.
 */
#include "fdd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
    orc_csr A;
    orc_csr P; /* rows of this level x rows of the next (coarser) level */
    orc_csr R; /* P^T */
    double *D_val;
    double *coefs;
    double *f, *u, *r, *v, *w;
} amg_level;

struct orc_amg
{
    int num_levels;
    int cheby_order;
    int num_vcycles;
    amg_level *lev;
    double *coarse_dense; /* row-major copy of the coarsest A */
};

static void *xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    if (!p)
    {
        fprintf(stderr, "fdd_oracle: out of memory\n");
        abort();
    }
    return p;
}

static void csr_copy(orc_csr *dst, int rows, int cols, const int *ptr, const int *col, const double *val)
{
    int nnz = ptr[rows];
    dst->num_rows = rows;
    dst->num_cols = cols;
    dst->num_nnz = nnz;
    dst->ptr = (int *)xcalloc((size_t)rows + 1, sizeof(int));
    dst->col = (int *)xcalloc((size_t)nnz, sizeof(int));
    dst->val = (double *)xcalloc((size_t)nnz, sizeof(double));
    memcpy(dst->ptr, ptr, ((size_t)rows + 1) * sizeof(int));
    memcpy(dst->col, col, (size_t)nnz * sizeof(int));
    memcpy(dst->val, val, (size_t)nnz * sizeof(double));
}

orc_amg *orc_amg_create(int num_levels, int cheby_order, int num_vcycles)
{
    orc_amg *a = (orc_amg *)xcalloc(1, sizeof(orc_amg));
    a->num_levels = num_levels;
    a->cheby_order = cheby_order; /* clamped 1..4 in the reference (subdomain.tpp:3477-3478) */
    if (a->cheby_order < 1) a->cheby_order = 1;
    if (a->cheby_order > 4) a->cheby_order = 4;
    a->num_vcycles = num_vcycles;
    a->lev = (amg_level *)xcalloc((size_t)num_levels, sizeof(amg_level));
    return a;
}

/* level l: A (n x n CSR), D_val[n], coefs[cheby_order]; P (n x n_coarse CSR) or NULLs on the coarsest level */
void orc_amg_set_level(orc_amg *a, int l, int n, const int *A_ptr, const int *A_col, const double *A_val, const double *D_val, const double *coefs, int n_coarse, const int *P_ptr, const int *P_col, const double *P_val)
{
    amg_level *L = &a->lev[l];
    csr_copy(&L->A, n, n, A_ptr, A_col, A_val);
    L->D_val = (double *)xcalloc((size_t)n, sizeof(double));
    memcpy(L->D_val, D_val, (size_t)n * sizeof(double));
    L->coefs = (double *)xcalloc((size_t)a->cheby_order, sizeof(double));
    memcpy(L->coefs, coefs, (size_t)a->cheby_order * sizeof(double));
    if (P_ptr)
    {
        csr_copy(&L->P, n, n_coarse, P_ptr, P_col, P_val);
        orc_csr_transpose(&L->P, &L->R); /* R_fem[l] = transpose of the hierarchy's P (subdomain.tpp:3526-3545) */
    }
    L->f = (double *)xcalloc((size_t)n, sizeof(double));
    L->u = (double *)xcalloc((size_t)n, sizeof(double));
    L->r = (double *)xcalloc((size_t)n, sizeof(double));
    L->v = (double *)xcalloc((size_t)n, sizeof(double));
    L->w = (double *)xcalloc((size_t)n, sizeof(double));

    if (l == a->num_levels - 1)
    {
        a->coarse_dense = (double *)xcalloc((size_t)n * n, sizeof(double));
        for (int i = 0; i < n; i++)
            for (int j = A_ptr[i]; j < A_ptr[i + 1]; j++) a->coarse_dense[(size_t)i * n + A_col[j]] += A_val[j];
    }
}

void orc_amg_destroy(orc_amg *a)
{
    if (!a) return;
    for (int l = 0; l < a->num_levels; l++)
    {
        amg_level *L = &a->lev[l];
        orc_csr_free(&L->A);
        orc_csr_free(&L->P);
        orc_csr_free(&L->R);
        free(L->D_val);
        free(L->coefs);
        free(L->f);
        free(L->u);
        free(L->r);
        free(L->v);
        free(L->w);
    }
    free(a->lev);
    free(a->coarse_dense);
    free(a);
}

int orc_amg_level_size(const orc_amg *a, int l) { return a->lev[l].A.num_rows; }

/* Chebyshev smoother u += D p(DAD) D (f - A u): subdomain.tpp:19-83 host branches */
static void smooth(orc_amg *a, int l)
{
    amg_level *L = &a->lev[l];
    int c = a->cheby_order;
    orc_amg_scaled_residual_host(L->r, L->w, L->A.ptr, L->A.col, L->A.val, L->u, L->f, L->D_val, L->coefs[c - 1], L->A.num_rows);
    for (int p = c - 2; p >= 0; p--)
        orc_amg_polynomial_evaluation_host(L->w, L->v, L->A.ptr, L->A.col, L->A.val, L->r, L->D_val, L->coefs[p], L->A.num_rows);
    orc_amg_main_update_field(L->u, L->w, L->D_val, L->A.num_rows);
}

/* dense Gaussian elimination without pivoting on a copy (hypre_GaussElimSolve's role) */
static void gauss_solve(const double *A, const double *b, double *x, int n)
{
    double *M = (double *)xcalloc((size_t)n * n, sizeof(double));
    double *y = (double *)xcalloc((size_t)n, sizeof(double));
    memcpy(M, A, (size_t)n * n * sizeof(double));
    memcpy(y, b, (size_t)n * sizeof(double));
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
        for (int j = i + 1; j < n; j++) s -= M[(size_t)i * n + j] * x[j];
        x[i] = s / M[(size_t)i * n + i];
    }
    free(M);
    free(y);
}

/* subdomain.tpp:4011-4139: u_fem[0] = V-cycle(s) applied to f_fem[0], from u = 0 */
void orc_amg_vcycle(orc_amg *a, double *u0, const double *f0)
{
    int L = a->num_levels;
    memcpy(a->lev[0].f, f0, (size_t)a->lev[0].A.num_rows * sizeof(double));
    orc_amg_vector_set_to_value(a->lev[0].u, 0.0, a->lev[0].A.num_rows); /* :4012 */

    for (int iter = 0; iter < a->num_vcycles; iter++)
    {
        /* down leg (:4023-4073) */
        for (int l = 0; l < L - 1; l++)
        {
            amg_level *lv = &a->lev[l];
            if (l > 0) orc_amg_vector_set_to_value(lv->u, 0.0, lv->A.num_rows);
            smooth(a, l);
            memcpy(lv->v, lv->f, (size_t)lv->A.num_rows * sizeof(double));                                   /* v = f */
            orc_amg_matvec(lv->v, lv->A.ptr, lv->A.col, lv->A.val, lv->u, -1.0, 1.0, lv->A.num_rows);        /* v = -A u + v */
            orc_amg_matvec(a->lev[l + 1].f, lv->R.ptr, lv->R.col, lv->R.val, lv->v, 1.0, 0.0, lv->R.num_rows); /* f_{l+1} = R v */
        }

        /* coarse grid solve (:4077-4090) */
        gauss_solve(a->coarse_dense, a->lev[L - 1].f, a->lev[L - 1].u, a->lev[L - 1].A.num_rows);

        /* up leg (:4095-4135) */
        for (int l = L - 1; l > 0; l--)
        {
            amg_level *fine = &a->lev[l - 1];
            orc_amg_matvec(fine->u, fine->P.ptr, fine->P.col, fine->P.val, a->lev[l].u, 1.0, 1.0, fine->P.num_rows); /* u_{l-1} += P u_l */
            smooth(a, l - 1);
        }
    }

    memcpy(u0, a->lev[0].u, (size_t)a->lev[0].A.num_rows * sizeof(double));
}
