/*
 * fdd_oracle_composite.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Serial restatement of the Subdomain<double> constructor's composite
 * (subdomain.tpp:86-2747) and of the exchange half of tree_operator
 * (subdomain.tpp:4613-4645) for R MPI ranks simulated in one process:
 *   - global element graph from all elements' corner ids (:198-453);
 *   - computational regions: own elements, `subdomain_overlap` rings per
 *     polynomial level, the extended ring, the superdomain (:455-579);
 *   - region data pulled from the owners (the gs exchanges of :644-805 are
 *     plain copies here: every rank's meshes are in this process);
 *   - interface nodes (:810-843), region connectivity (:845-878);
 *   - global numbering with per-level offsets, hanging edges / faces zeroed,
 *     interface and extended nodes shifted to the end, ranking (:920-1176);
 *   - the non-conforming Q with J_cf rows (:1179-1585), per-point indirection
 *     arrays of the mixed-degree stiffness kernels (:1587-1630);
 *   - coarse dofs, Qt_coarse, the degree-1 operator of the whole domain
 *     (:1632-1848), the dof markers (:1860-1905);
 *   - superdomain A / Pt (:1907-2576).  The reference walks HYPRE BoomerAMG's
 *     hierarchy here; HYPRE is absent, so product and oracle both use the
 *     build's own graded smoothed aggregation (documented deviation):
 *     this file restates that algorithm independently in C (same passes, same
 *     arithmetic order, so both sides take the same aggregates);
 *   - interface operators Q_int / Qt_int / QQt_int and the weights (:2581-2747).
 *
 * The solve path on these operators is fdd_oracle_subdomain.c.
 * See fdd_oracle.h for who may use this file and the parity-pin statement.
 */
#include "fdd_oracle_priv.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define xcalloc orc_xcalloc

struct orc_fdd
{
    int num_ranks;
    int num_levels;
    int dim;
    int num_vertices;
    int *poly_degree;
    int num_total_elements;
    int *proc_count, *proc_offset;
    orc_subdomain **sub; /* one composite per rank */

    /* per rank: the region's elements (global id, level) and what the tree exchange needs */
    int *num_sub_elems, *num_sub_ext_elems;
    int **region_id, **region_level, **region_offset;
    int num_coarse_dofs;
    double **tree;       /* per rank: the degree tree of its own elements */
    double *coarse_all;  /* the gathered coarsest level of all ranks (subdomain.tpp:4620-4621) */
    double *coarse_dofs; /* Qt_coarse * coarse_all */
    int *sup_num_dofs;
    int **dof_sup;       /* per rank, per coarse dof (1-based superdomain dof, 0 none) */
    int **comp_levels;   /* per rank: composite dofs per coarsening level, -1 terminated */
    orc_csr *A_fem;      /* per rank: the low-order operator of the composite (subdomain.tpp:2749-3472) */
    int **point_dof;     /* per rank: 0-based subdomain dof of every region point that carries one directly, -1 otherwise */
};

/* ------------------------------------------------------------------ */
/* small helpers                                                        */
/* ------------------------------------------------------------------ */
static int ipow(int b, int e)
{
    int r = 1;
    for (int i = 0; i < e; i++) r *= b;
    return r;
}

static int cmp_ll(const void *a, const void *b)
{
    long long x = *(const long long *)a, y = *(const long long *)b;
    return (x < y) ? -1 : (x > y);
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x < y) ? -1 : (x > y);
}

/* sorted unique copy; returns count */
static int sort_unique_ll(long long *v, int n)
{
    if (n == 0) return 0;
    qsort(v, (size_t)n, sizeof(long long), cmp_ll);
    int m = 1;
    for (int i = 1; i < n; i++)
        if (v[i] != v[m - 1]) v[m++] = v[i];
    return m;
}

static int find_ll(const long long *v, int n, long long key)
{
    int lo = 0, hi = n - 1;
    while (lo <= hi)
    {
        int mid = (lo + hi) / 2;
        if (v[mid] == key) return mid;
        if (v[mid] < key)
            lo = mid + 1;
        else
            hi = mid - 1;
    }
    return -1;
}

/* local index of vertex v (v = i + 2j + 4k) in an element with n points per direction */
static int corner_index(int v, int n, int dim)
{
    int i = (v & 1) ? n - 1 : 0, j = (v & 2) ? n - 1 : 0, k = (dim == 3 && (v & 4)) ? n - 1 : 0;
    return i + j * n + k * n * n;
}

/* edge tables of the reference (subdomain.tpp:312-362) */
static const int edge_pairs_3d[12][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {4, 5}, {6, 7}, {4, 6}, {5, 7}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
static const int edge_pairs_2d[4][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}};
/* face tables (subdomain.tpp:393-401) */
static const int face_quads[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {0, 2, 4, 6}, {1, 3, 5, 7}};

/* points of edge eid in increasing local coordinate: the idx_i tables of matching_edge (subdomain.tpp:1197-1308) */
static void edge_idx(int eid, int n, int dim, int *idx)
{
    int nn = n * n;
    for (int k = 0; k < n; k++)
    {
        if (dim == 2)
        {
            switch (eid)
            {
            case 0: idx[k] = k + 0 * n; break;
            case 1: idx[k] = k + (n - 1) * n; break;
            case 2: idx[k] = 0 + k * n; break;
            default: idx[k] = (n - 1) + k * n; break;
            }
        }
        else
        {
            switch (eid)
            {
            case 0: idx[k] = k + 0 * n + 0 * nn; break;
            case 1: idx[k] = k + (n - 1) * n + 0 * nn; break;
            case 2: idx[k] = 0 + k * n + 0 * nn; break;
            case 3: idx[k] = (n - 1) + k * n + 0 * nn; break;
            case 4: idx[k] = k + 0 * n + (n - 1) * nn; break;
            case 5: idx[k] = k + (n - 1) * n + (n - 1) * nn; break;
            case 6: idx[k] = 0 + k * n + (n - 1) * nn; break;
            case 7: idx[k] = (n - 1) + k * n + (n - 1) * nn; break;
            case 8: idx[k] = 0 + 0 * n + k * nn; break;
            case 9: idx[k] = (n - 1) + 0 * n + k * nn; break;
            case 10: idx[k] = 0 + (n - 1) * n + k * nn; break;
            default: idx[k] = (n - 1) + (n - 1) * n + k * nn; break;
            }
        }
    }
}

/* points of face fid: the idx_i tables of matching_face (subdomain.tpp:1366-1431) */
static void face_idx(int fid, int n, int *idx)
{
    int nn = n * n;
    for (int b = 0; b < n; b++)
    {
        for (int a = 0; a < n; a++)
        {
            switch (fid)
            {
            case 0: idx[a + b * n] = a + b * n + 0 * nn; break;
            case 1: idx[a + b * n] = a + b * n + (n - 1) * nn; break;
            case 2: idx[a + b * n] = a + 0 * n + b * nn; break;
            case 3: idx[a + b * n] = a + (n - 1) * n + b * nn; break;
            case 4: idx[a + b * n] = 0 + a * n + b * nn; break;
            default: idx[a + b * n] = (n - 1) + a * n + b * nn; break;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* growing COO list -> CSR (duplicates summed in insertion order)        */
/* ------------------------------------------------------------------ */
typedef struct
{
    int *row, *col;
    double *val;
    long n, cap;
} coo;

static void coo_add(coo *c, int row, int col, double val)
{
    if (c->n == c->cap)
    {
        c->cap = c->cap ? 2 * c->cap : 1024;
        c->row = (int *)realloc(c->row, (size_t)c->cap * sizeof(int));
        c->col = (int *)realloc(c->col, (size_t)c->cap * sizeof(int));
        c->val = (double *)realloc(c->val, (size_t)c->cap * sizeof(double));
    }
    c->row[c->n] = row;
    c->col[c->n] = col;
    c->val[c->n] = val;
    c->n++;
}

static void coo_free(coo *c)
{
    free(c->row);
    free(c->col);
    free(c->val);
    memset(c, 0, sizeof(*c));
}

/* all entries kept (no drop tolerance): the HYPRE_IJMatrixAddToValues + Assemble of the reference */
typedef struct
{
    int row, col;
    double val;
    long seq;
} ij_entry;

static int ij_cmp(const void *a_, const void *b_)
{
    const ij_entry *a = (const ij_entry *)a_, *b = (const ij_entry *)b_;
    if (a->row != b->row) return (a->row < b->row) ? -1 : 1;
    if (a->col != b->col) return (a->col < b->col) ? -1 : 1;
    return (a->seq < b->seq) ? -1 : (a->seq > b->seq);
}

static void ij_assemble(orc_csr *A, int num_rows, int num_cols, const coo *c)
{
    ij_entry *e = (ij_entry *)xcalloc((size_t)c->n, sizeof(ij_entry));
    for (long i = 0; i < c->n; i++)
    {
        e[i].row = c->row[i];
        e[i].col = c->col[i];
        e[i].val = c->val[i];
        e[i].seq = i;
    }
    qsort(e, (size_t)c->n, sizeof(ij_entry), ij_cmp);
    A->num_rows = num_rows;
    A->num_cols = num_cols;
    A->ptr = (int *)xcalloc((size_t)num_rows + 1, sizeof(int));
    A->col = (int *)xcalloc((size_t)c->n, sizeof(int));
    A->val = (double *)xcalloc((size_t)c->n, sizeof(double));
    int nnz = 0;
    for (long i = 0; i < c->n; i++)
    {
        if (i > 0 && e[i].row == e[i - 1].row && e[i].col == e[i - 1].col)
        {
            A->val[nnz - 1] += e[i].val;
        }
        else
        {
            A->col[nnz] = e[i].col;
            A->val[nnz] = e[i].val;
            A->ptr[e[i].row + 1]++;
            nnz++;
        }
    }
    for (int r = 0; r < num_rows; r++) A->ptr[r + 1] += A->ptr[r];
    A->num_nnz = nnz;
    free(e);
}

/* ------------------------------------------------------------------ */
/* sparse products for the graded aggregation (row-wise, one dense       */
/* accumulator row; entries accumulate in the order of the left rows)    */
/* ------------------------------------------------------------------ */
static void csr_matmat(orc_csr *C, const orc_csr *A, const orc_csr *B)
{
    int n = A->num_rows, m = B->num_cols;
    long *marker = (long *)xcalloc((size_t)m, sizeof(long));
    double *acc = (double *)xcalloc((size_t)m, sizeof(double));
    int *cols = (int *)xcalloc((size_t)m, sizeof(int));
    coo out;
    memset(&out, 0, sizeof(out));
    for (int j = 0; j < m; j++) marker[j] = -1;

    for (int i = 0; i < n; i++)
    {
        int nc = 0;
        for (int p = A->ptr[i]; p < A->ptr[i + 1]; p++)
        {
            int k = A->col[p];
            double a = A->val[p];
            for (int q = B->ptr[k]; q < B->ptr[k + 1]; q++)
            {
                int j = B->col[q];
                if (marker[j] != i)
                {
                    marker[j] = i;
                    acc[j] = 0.0;
                    cols[nc++] = j;
                }
                acc[j] += a * B->val[q];
            }
        }
        qsort(cols, (size_t)nc, sizeof(int), cmp_int);
        for (int k = 0; k < nc; k++) coo_add(&out, i, cols[k], acc[cols[k]]);
    }
    ij_assemble(C, n, m, &out);
    coo_free(&out);
    free(marker);
    free(acc);
    free(cols);
}

static void csr_transpose_keep(orc_csr *T, const orc_csr *A)
{
    T->num_rows = A->num_cols;
    T->num_cols = A->num_rows;
    T->num_nnz = A->num_nnz;
    T->ptr = (int *)xcalloc((size_t)T->num_rows + 1, sizeof(int));
    T->col = (int *)xcalloc((size_t)A->num_nnz, sizeof(int));
    T->val = (double *)xcalloc((size_t)A->num_nnz, sizeof(double));
    for (int p = 0; p < A->num_nnz; p++) T->ptr[A->col[p] + 1]++;
    for (int i = 0; i < T->num_rows; i++) T->ptr[i + 1] += T->ptr[i];
    int *fill = (int *)xcalloc((size_t)T->num_rows, sizeof(int));
    for (int i = 0; i < T->num_rows; i++) fill[i] = T->ptr[i];
    for (int i = 0; i < A->num_rows; i++)
    {
        for (int p = A->ptr[i]; p < A->ptr[i + 1]; p++)
        {
            int q = fill[A->col[p]]++;
            T->col[q] = i;
            T->val[q] = A->val[p];
        }
    }
    free(fill);
}

static void csr_copy(orc_csr *B, const orc_csr *A)
{
    B->num_rows = A->num_rows;
    B->num_cols = A->num_cols;
    B->num_nnz = A->num_nnz;
    B->ptr = (int *)xcalloc((size_t)A->num_rows + 1, sizeof(int));
    B->col = (int *)xcalloc((size_t)A->num_nnz, sizeof(int));
    B->val = (double *)xcalloc((size_t)A->num_nnz, sizeof(double));
    memcpy(B->ptr, A->ptr, ((size_t)A->num_rows + 1) * sizeof(int));
    memcpy(B->col, A->col, (size_t)A->num_nnz * sizeof(int));
    memcpy(B->val, A->val, (size_t)A->num_nnz * sizeof(double));
}

/* ------------------------------------------------------------------ */
/* graded smoothed aggregation of the superdomain (the build's stand-in  */
/* for subdomain.tpp:1907-2400; see the header and DESIGN.md)            */
/* ------------------------------------------------------------------ */
#define GRADE_STRENGTH 0.08
#define GRADE_OMEGA (2.0 / 3.0)
#define GRADE_MAX_LEVELS 12
#define GRADE_KEEP_AT_MOST 8
#define GRADE_TIE 1.0e-10

static int is_strong(const orc_csr *A, const double *d, const char *active, double theta, int i, int p)
{
    int j = A->col[p];
    return j != i && active[j] && fabs(A->val[p]) >= theta * sqrt(fabs(d[i] * d[j])) * (1.0 - GRADE_TIE);
}

/* greedy aggregation of the active rows: seed with untouched strong neighbourhoods, join the strongest
 * aggregated neighbour, then the leftovers among themselves */
static int aggregate_active(const orc_csr *A, const char *active, double theta, int *agg)
{
    int n = A->num_rows;
    double *d = (double *)xcalloc((size_t)n, sizeof(double));
    orc_csr_diagonal(A, d);
    for (int i = 0; i < n; i++) agg[i] = -1;
    int count = 0;

    for (int i = 0; i < n; i++)
    {
        if (!active[i] || agg[i] != -1) continue;
        int free_nbhd = 1, has_strong = 0;
        for (int p = A->ptr[i]; p < A->ptr[i + 1] && free_nbhd; p++)
        {
            if (is_strong(A, d, active, theta, i, p))
            {
                has_strong = 1;
                if (agg[A->col[p]] != -1) free_nbhd = 0;
            }
        }
        if (!free_nbhd || !has_strong) continue;
        agg[i] = count;
        for (int p = A->ptr[i]; p < A->ptr[i + 1]; p++)
            if (is_strong(A, d, active, theta, i, p)) agg[A->col[p]] = count;
        count++;
    }

    int *joined = (int *)xcalloc((size_t)n, sizeof(int));
    for (int i = 0; i < n; i++) joined[i] = -1;
    for (int i = 0; i < n; i++)
    {
        if (!active[i] || agg[i] != -1) continue;
        double best = 0.0;
        for (int p = A->ptr[i]; p < A->ptr[i + 1]; p++)
        {
            if (is_strong(A, d, active, theta, i, p) && agg[A->col[p]] != -1 && fabs(A->val[p]) > best * (1.0 + GRADE_TIE))
            {
                best = fabs(A->val[p]);
                joined[i] = agg[A->col[p]];
            }
        }
    }
    for (int i = 0; i < n; i++)
        if (active[i] && agg[i] == -1 && joined[i] != -1) agg[i] = joined[i];

    for (int i = 0; i < n; i++)
    {
        if (!active[i] || agg[i] != -1) continue;
        agg[i] = count;
        for (int p = A->ptr[i]; p < A->ptr[i + 1]; p++)
            if (is_strong(A, d, active, theta, i, p) && agg[A->col[p]] == -1) agg[A->col[p]] = count;
        count++;
    }

    free(joined);
    free(d);
    return count;
}

/* P_comp (rows of A_c x composite dofs), A_comp = P^T A_c P, the composite dof of every coarse dof kept at
 * level 0, the sizes of the marker groups 1..4, the composite dofs added per level (-1 terminated) */
static void grade_superdomain(const orc_csr *A_c, const int *marker, int superdomain_overlap, orc_csr *P_comp, orc_csr *A_comp, int *comp_of_fine, int group_count[4], int *kept_per_level)
{
    int n0 = A_c->num_rows;
    orc_csr A;
    csr_copy(&A, A_c);
    int *D = (int *)xcalloc((size_t)n0, sizeof(int));
    for (int i = 0; i < n0; i++) D[i] = (marker[i] > 0) ? 1 : 0; /* subdomain.tpp:1929-1931 */
    for (int g = 0; g < 4; g++) group_count[g] = 0;
    for (int i = 0; i < n0; i++)
        if (marker[i] > 0) group_count[marker[i] - 1]++;
    for (int i = 0; i < n0; i++) comp_of_fine[i] = -1;
    int have_P = 0;
    int overlap = superdomain_overlap;
    int nlev = 0;

    for (int level = 0;; level++)
    {
        int n = A.num_rows;
        double *w = (double *)xcalloc((size_t)n, sizeof(double));
        double *w2 = (double *)xcalloc((size_t)n, sizeof(double));
        for (int i = 0; i < n; i++) w[i] = (D[i] > 0) ? 1.0 : 0.0;

        /* subdomain.tpp:1944-1957 */
        for (int nu = 0; nu < overlap; nu++)
        {
            for (int row = 0; row < n; row++)
            {
                double val = 0.0;
                for (int p = A.ptr[row]; p < A.ptr[row + 1]; p++) val += w[A.col[p]];
                w2[row] = val;
            }
            memcpy(w, w2, (size_t)n * sizeof(double));
        }
        if (overlap == 0) overlap = 1; /* :1959 */

        int remaining = 0;
        for (int i = 0; i < n; i++)
        {
            if (D[i] == 0 && w[i] > 0.0) D[i] = 2; /* :1965-1967 */
            if (D[i] == 0) remaining++;
        }

        char *active = (char *)xcalloc((size_t)n, sizeof(char));
        int *agg = (int *)xcalloc((size_t)n, sizeof(int));
        int num_agg = 0;
        if (remaining > 0 && remaining > GRADE_KEEP_AT_MOST && level < GRADE_MAX_LEVELS - 1)
        {
            for (int i = 0; i < n; i++) active[i] = (D[i] == 0);
            num_agg = aggregate_active(&A, active, GRADE_STRENGTH * pow(0.5, level), agg);
        }
        if (remaining > 0 && (num_agg == 0 || num_agg >= remaining))
        {
            for (int i = 0; i < n; i++)
                if (D[i] == 0) D[i] = 2; /* the last level keeps what is left, :1961-1963 */
            remaining = 0;
        }

        int *pos = (int *)xcalloc((size_t)n, sizeof(int));
        for (int i = 0; i < n; i++) pos[i] = -1;
        int nk = 0;
        if (level == 0)
        {
            /* subdomain.tpp:2043-2058: marker groups first, then the level-0 overlap */
            for (int m = 1; m <= 4; m++)
                for (int i = 0; i < n; i++)
                    if (marker[i] == m) pos[i] = nk++;
            for (int i = 0; i < n; i++)
                if (D[i] == 2) pos[i] = nk++;
            for (int i = 0; i < n; i++) comp_of_fine[i] = pos[i];
            kept_per_level[nlev++] = nk;
        }
        else
        {
            for (int i = 0; i < n; i++)
                if (D[i] == 1) pos[i] = nk++;
            int before = nk;
            for (int i = 0; i < n; i++)
                if (D[i] == 2) pos[i] = nk++; /* :2062-2065 */
            kept_per_level[nlev++] = nk - before;
        }

        /* interpolator of this level: identity on what is kept, smoothed aggregation on the rest */
        int pc = nk + ((remaining > 0) ? num_agg : 0);
        coo Pl;
        memset(&Pl, 0, sizeof(Pl));
        {
            double *d = (double *)xcalloc((size_t)n, sizeof(double));
            double *acc = (double *)xcalloc((size_t)pc, sizeof(double));
            int *stamp = (int *)xcalloc((size_t)pc, sizeof(int));
            int *cols = (int *)xcalloc((size_t)pc, sizeof(int));
            orc_csr_diagonal(&A, d);
            for (int c = 0; c < pc; c++) stamp[c] = -1;
            for (int i = 0; i < n; i++)
            {
                if (pos[i] >= 0)
                {
                    coo_add(&Pl, i, pos[i], 1.0);
                    continue;
                }
                int nc = 0;
                int c0 = nk + agg[i];
                stamp[c0] = i;
                acc[c0] = 0.0;
                cols[nc++] = c0;
                acc[c0] += 1.0;
                for (int p = A.ptr[i]; p < A.ptr[i + 1]; p++)
                {
                    int j = A.col[p];
                    int c = (pos[j] >= 0) ? pos[j] : nk + agg[j];
                    if (stamp[c] != i)
                    {
                        stamp[c] = i;
                        acc[c] = 0.0;
                        cols[nc++] = c;
                    }
                    acc[c] += -GRADE_OMEGA * A.val[p] / d[i];
                }
                qsort(cols, (size_t)nc, sizeof(int), cmp_int);
                for (int k = 0; k < nc; k++) coo_add(&Pl, i, cols[k], acc[cols[k]]);
            }
            free(d);
            free(acc);
            free(stamp);
            free(cols);
        }
        orc_csr P, R, AP, An;
        ij_assemble(&P, n, pc, &Pl);
        coo_free(&Pl);
        csr_transpose_keep(&R, &P);
        csr_matmat(&AP, &A, &P);
        csr_matmat(&An, &R, &AP);
        if (have_P)
        {
            orc_csr PP;
            csr_matmat(&PP, P_comp, &P);
            orc_csr_free(P_comp);
            *P_comp = PP;
        }
        else
        {
            csr_copy(P_comp, &P);
        }
        have_P = 1;
        orc_csr_free(&P);
        orc_csr_free(&R);
        orc_csr_free(&AP);
        orc_csr_free(&A);
        A = An;

        free(w);
        free(w2);
        free(active);
        free(agg);
        free(pos);
        free(D);
        D = (int *)xcalloc((size_t)A.num_rows, sizeof(int));
        for (int i = 0; i < nk; i++) D[i] = 1; /* :1987-1990: what is kept is "local" on the next level */
        if (remaining == 0) break;
    }
    kept_per_level[nlev] = -1;
    free(D);
    *A_comp = A;
}

/* ------------------------------------------------------------------ */
/* the composite of one rank                                            */
/* ------------------------------------------------------------------ */
typedef struct
{
    int id;     /* global element */
    int level;  /* index into poly_degree */
    int N, n, num_points, offset;
    int owner, owner_elem;
} relem;

#define MESH(f, r, l) (&(meshes)[(r) * (f)->num_levels + (l)])

static void build_rank(orc_fdd *F, int me, const orc_mesh *meshes, const double *const *D_hat, const double *const *J_cf_pairs, const long long *geometry_mesh, const int *vert_of, const int *v2e_ptr, const int *v2e,
                       const int *owner, const int *owner_elem, const long long *glo_num_coarse, const int *dof_num_coarse, const orc_csr *A_coarse, int subdomain_overlap, int superdomain_overlap)
{
    const int dim = F->dim, nv = F->num_vertices, num_levels = F->num_levels;
    const int *poly_degree = F->poly_degree;
    const int num_total_elements = F->num_total_elements;
    const int num_local_elements = F->proc_count[me];
    const int num_edges = (dim == 2) ? 4 : 12;
    const int num_faces = (dim == 2) ? 0 : 6;
    const int num_coarse_dofs = F->num_coarse_dofs;

    orc_subdomain *s = (orc_subdomain *)xcalloc(1, sizeof(orc_subdomain));
    F->sub[me] = s;
    s->dim = dim;
    s->num_levels = num_levels;
    s->poly_degree = (int *)xcalloc((size_t)num_levels, sizeof(int));
    s->levels = (orc_level *)xcalloc((size_t)num_levels, sizeof(orc_level));
    s->D_hat = (double **)xcalloc((size_t)num_levels, sizeof(double *));
    s->J_cf = (double **)xcalloc((size_t)num_levels, sizeof(double *));
    for (int l = 0; l < num_levels; l++)
    {
        int n = poly_degree[l] + 1;
        s->poly_degree[l] = poly_degree[l];
        s->levels[l].num_elements = num_local_elements;
        s->levels[l].num_points = num_local_elements * ipow(n, dim);
        s->levels[l].poly_degree = poly_degree[l];
        s->levels[l].offset = (l > 0) ? s->levels[l - 1].offset + s->levels[l - 1].num_points : 0; /* :112-120 */
        s->D_hat[l] = (double *)xcalloc((size_t)n * n, sizeof(double));
        memcpy(s->D_hat[l], D_hat[l], (size_t)n * n * sizeof(double));
        if (l + 1 < num_levels)
        {
            int n_c = poly_degree[l + 1] + 1;
            s->J_cf[l] = (double *)xcalloc((size_t)n * n_c, sizeof(double));
            memcpy(s->J_cf[l], J_cf_pairs[l * num_levels + (l + 1)], (size_t)n * n_c * sizeof(double));
        }
    }

    /* ---- computational regions (subdomain.tpp:455-579) ---- */
    int cap = num_total_elements + 1;
    relem *region = (relem *)xcalloc((size_t)2 * cap, sizeof(relem));
    int num_subdomain_elems = 0, num_subdomain_extended_elems = 0;
    double *work0 = (double *)xcalloc((size_t)num_total_elements, sizeof(double));
    double *work1 = (double *)xcalloc((size_t)num_total_elements, sizeof(double));
    double *mark = (double *)xcalloc((size_t)num_total_elements, sizeof(double)); /* work_hst[1] of the reference */
    int nreg = 0;

#define ADD_ELEM(e_, l_)                                                  \
    do                                                                    \
    {                                                                     \
        relem *r_ = &region[nreg++];                                      \
        r_->id = (e_);                                                    \
        r_->level = (l_);                                                 \
        r_->N = poly_degree[(l_)];                                        \
        r_->n = r_->N + 1;                                                \
        r_->num_points = ipow(r_->n, dim);                                \
        r_->owner = owner[(e_)];                                          \
        r_->owner_elem = owner_elem[(e_)];                                \
    } while (0)

    /* expander * in: an element reaches itself and every element it shares a vertex with (:432-453) */
#define EXPAND(out_, in_)                                                                   \
    do                                                                                      \
    {                                                                                       \
        for (int e_ = 0; e_ < num_total_elements; e_++)                                     \
        {                                                                                   \
            double v_ = 0.0;                                                                \
            for (int c_ = 0; c_ < nv; c_++)                                                 \
            {                                                                               \
                int vid_ = vert_of[e_ * nv + c_];                                           \
                for (int q_ = v2e_ptr[vid_]; q_ < v2e_ptr[vid_ + 1]; q_++) v_ += (in_)[v2e[q_]]; \
            }                                                                               \
            (out_)[e_] = v_;                                                                \
        }                                                                                   \
    } while (0)

    for (int e = 0; e < num_local_elements; e++)
    {
        ADD_ELEM(F->proc_offset[me] + e, 0);
        num_subdomain_elems++;
        num_subdomain_extended_elems++;
        work0[F->proc_offset[me] + e] = 1.0;
        mark[F->proc_offset[me] + e] = (double)(e + 1);
    }

    for (int l = 0; l < num_levels; l++)
    {
        for (int nu = 0; nu < subdomain_overlap; nu++)
        {
            EXPAND(work1, work0);
            memcpy(work0, work1, (size_t)num_total_elements * sizeof(double));
        }
        for (int e = 0; e < num_total_elements; e++)
        {
            if (work0[e] > 0.0 && mark[e] == 0.0)
            {
                mark[e] = (double)num_subdomain_elems;
                ADD_ELEM(e, l);
                num_subdomain_elems++;
                num_subdomain_extended_elems++;
            }
        }
        if (subdomain_overlap == 0) subdomain_overlap = 1; /* :509 */
    }

    EXPAND(work1, work0); /* :512-513 */
    int num_superdomain_elems = 0, num_superdomain_extended_elems = 0;
    for (int e = 0; e < num_total_elements; e++)
    {
        if (mark[e] == 0.0)
        {
            if (work1[e] > 0.0)
            {
                ADD_ELEM(e, num_levels - 1);
                num_subdomain_extended_elems++;
            }
            num_superdomain_elems++;
            num_superdomain_extended_elems++;
        }
    }
    for (int e = 0; e < num_total_elements; e++) work0[e] = (mark[e] == 0.0) ? 1.0 : 0.0; /* :533-539 */
    EXPAND(work1, work0);
    int *sup_ext = (int *)xcalloc((size_t)num_total_elements + 1, sizeof(int)); /* region indices of the superdomain-extended elements */
    int num_sup_ext = 0;
    int *region_index = (int *)xcalloc((size_t)num_total_elements, sizeof(int));
    for (int e = 0; e < num_total_elements; e++) region_index[e] = -1;
    for (int r = 0; r < nreg; r++) region_index[region[r].id] = r;
    for (int e = 0; e < num_total_elements; e++)
    {
        if (work1[e] > 0.0 && mark[e] > 0.0) /* :545-553 */
        {
            sup_ext[num_sup_ext++] = region_index[e];
            num_superdomain_extended_elems++;
        }
    }

    int num_subdomain_points = 0, num_subdomain_extended_points = 0;
    for (int r = 0; r < num_subdomain_extended_elems; r++)
    {
        region[r].offset = num_subdomain_extended_points;
        if (r < num_subdomain_elems) num_subdomain_points += region[r].num_points;
        num_subdomain_extended_points += region[r].num_points;
    }
    const int NP = num_subdomain_extended_points;
    (void)num_subdomain_points;

    F->num_sub_elems[me] = num_subdomain_elems;
    F->num_sub_ext_elems[me] = num_subdomain_extended_elems;
    F->region_id[me] = (int *)xcalloc((size_t)nreg + 1, sizeof(int));
    F->region_level[me] = (int *)xcalloc((size_t)nreg + 1, sizeof(int));
    F->region_offset[me] = (int *)xcalloc((size_t)nreg + 1, sizeof(int));
    for (int r = 0; r < nreg; r++)
    {
        F->region_id[me][r] = region[r].id;
        F->region_level[me][r] = region[r].level;
        F->region_offset[me][r] = region[r].offset;
    }

    /* ---- region data from the owners (the gs pulls of subdomain.tpp:644-805) ---- */
    double *mask = (double *)xcalloc((size_t)NP, sizeof(double));
    long long *glo = (long long *)xcalloc((size_t)NP, sizeof(long long));
    long long *dofn = (long long *)xcalloc((size_t)NP, sizeof(long long)); /* elem.dof_num */
    for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++) s->geom_fact[g] = (double *)xcalloc((size_t)NP, sizeof(double));
    for (int r = 0; r < nreg; r++)
    {
        const relem *el = &region[r];
        const orc_mesh *m = &meshes[el->owner * num_levels + el->level];
        size_t src = (size_t)el->owner_elem * el->num_points;
        for (int v = 0; v < el->num_points; v++)
        {
            mask[el->offset + v] = m->p_mask[src + v];
            glo[el->offset + v] = m->glo_num[src + v];
            for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++) s->geom_fact[g][el->offset + v] = m->g[g] ? m->g[g][src + v] : 0.0;
        }
    }

    /* ---- interface nodes (subdomain.tpp:810-843) ---- */
    long long *subdomain_glo_num = (long long *)xcalloc((size_t)NP + 1, sizeof(long long));
    int n_sub_glo = 0;
    for (int r = 0; r < num_subdomain_elems; r++)
    {
        const relem *el = &region[r];
        if (el->N == 1)
            for (int v = 0; v < el->num_points; v++)
                if (mask[el->offset + v] > 0.0) subdomain_glo_num[n_sub_glo++] = glo[el->offset + v];
    }
    n_sub_glo = sort_unique_ll(subdomain_glo_num, n_sub_glo);
    long long *interface_glo_num = (long long *)xcalloc((size_t)num_total_elements * nv + 1, sizeof(long long));
    int n_interface = 0;
    for (int e = 0; e < num_total_elements; e++)
    {
        if (mark[e] != 0.0) continue; /* superdomain elements */
        const orc_mesh *m = &meshes[owner[e] * num_levels + (num_levels - 1)];
        for (int v = 0; v < nv; v++)
        {
            long long g = m->glo_num[(size_t)owner_elem[e] * nv + v];
            if (find_ll(subdomain_glo_num, n_sub_glo, g) >= 0) interface_glo_num[n_interface++] = g;
        }
    }
    n_interface = sort_unique_ll(interface_glo_num, n_interface);
    free(subdomain_glo_num);

    for (int r = 0; r < nreg; r++)
    {
        const relem *el = &region[r];
        for (int v = 0; v < el->num_points; v++)
            if (find_ll(interface_glo_num, n_interface, glo[el->offset + v]) >= 0) dofn[el->offset + v] = glo[el->offset + v]; /* :835-838 */
    }

    /* ---- connectivity of the region: elements around every edge and across every face (subdomain.tpp:845-878) ---- */
    /* edge_conn[r][eid]: region elements sharing the edge's two corner ids, ascending, r excluded */
    int **edge_conn = (int **)xcalloc((size_t)nreg * (num_edges ? num_edges : 1), sizeof(int *));
    int *edge_conn_n = (int *)xcalloc((size_t)nreg * (num_edges ? num_edges : 1), sizeof(int));
    int **face_conn = (int **)xcalloc((size_t)nreg * (num_faces ? num_faces : 1), sizeof(int *));
    int *face_conn_n = (int *)xcalloc((size_t)nreg * (num_faces ? num_faces : 1), sizeof(int));
    {
        const int(*pairs)[2] = (dim == 2) ? edge_pairs_2d : edge_pairs_3d;
        int *cand = (int *)xcalloc((size_t)nreg + 1, sizeof(int));
        for (int r = 0; r < nreg; r++)
        {
            const int e = region[r].id;
            for (int eid = 0; eid < num_edges; eid++)
            {
                long long a = geometry_mesh[(size_t)e * nv + pairs[eid][0]], b = geometry_mesh[(size_t)e * nv + pairs[eid][1]];
                /* candidates: elements around vertex a */
                int nc = 0;
                int vid = vert_of[e * nv + pairs[eid][0]];
                for (int q = v2e_ptr[vid]; q < v2e_ptr[vid + 1]; q++)
                {
                    int ej = v2e[q];
                    if (ej == e || region_index[ej] < 0) continue;
                    int has = 0;
                    for (int k = 0; k < num_edges && !has; k++)
                    {
                        long long c0 = geometry_mesh[(size_t)ej * nv + pairs[k][0]], c1 = geometry_mesh[(size_t)ej * nv + pairs[k][1]];
                        if ((c0 == a && c1 == b) || (c0 == b && c1 == a)) has = 1;
                    }
                    if (has) cand[nc++] = region_index[ej];
                }
                qsort(cand, (size_t)nc, sizeof(int), cmp_int);
                edge_conn[r * num_edges + eid] = (int *)xcalloc((size_t)nc + 1, sizeof(int));
                memcpy(edge_conn[r * num_edges + eid], cand, (size_t)nc * sizeof(int));
                edge_conn_n[r * num_edges + eid] = nc;
            }
            for (int fid = 0; fid < num_faces; fid++)
            {
                long long fk[4], gk[4];
                for (int k = 0; k < 4; k++) fk[k] = geometry_mesh[(size_t)e * nv + face_quads[fid][k]];
                qsort(fk, 4, sizeof(long long), cmp_ll);
                int nc = 0;
                int vid = vert_of[e * nv + face_quads[fid][0]];
                for (int q = v2e_ptr[vid]; q < v2e_ptr[vid + 1]; q++)
                {
                    int ej = v2e[q];
                    if (ej == e || region_index[ej] < 0) continue;
                    int has = 0;
                    for (int f2 = 0; f2 < num_faces && !has; f2++)
                    {
                        for (int k = 0; k < 4; k++) gk[k] = geometry_mesh[(size_t)ej * nv + face_quads[f2][k]];
                        qsort(gk, 4, sizeof(long long), cmp_ll);
                        if (gk[0] == fk[0] && gk[1] == fk[1] && gk[2] == fk[2] && gk[3] == fk[3]) has = 1;
                    }
                    if (has) cand[nc++] = region_index[ej];
                }
                qsort(cand, (size_t)nc, sizeof(int), cmp_int);
                face_conn[r * num_faces + fid] = (int *)xcalloc((size_t)nc + 1, sizeof(int));
                memcpy(face_conn[r * num_faces + fid], cand, (size_t)nc * sizeof(int));
                face_conn_n[r * num_faces + fid] = nc;
            }
        }
        free(cand);
    }

    /* ---- global numbering (subdomain.tpp:920-1176) ---- */
    long long *global_offset = (long long *)xcalloc((size_t)num_levels, sizeof(long long));
    for (int l = 1; l < num_levels; l++) global_offset[l] = global_offset[l - 1] + (long long)num_total_elements * (long long)ipow(poly_degree[l - 1] + 1, dim);

    for (int r = 0; r < nreg; r++) /* :926-967 */
    {
        const relem *el = &region[r];
        long long corners[8];
        for (int v = 0; v < nv; v++) corners[v] = glo[el->offset + corner_index(v, el->n, dim)];
        for (int v = 0; v < el->num_points; v++) glo[el->offset + v] += global_offset[el->level];
        for (int v = 0; v < nv; v++) glo[el->offset + corner_index(v, el->n, dim)] = corners[v];
    }

    {
        int *idx = (int *)xcalloc((size_t)ipow(poly_degree[0] + 1, 2) + 4, sizeof(int));
        for (int r = 0; r < nreg; r++) /* :969-1098 */
        {
            const relem *ei = &region[r];
            for (int eid = 0; eid < num_edges; eid++)
            {
                for (int k = 0; k < edge_conn_n[r * num_edges + eid]; k++)
                {
                    const relem *ej = &region[edge_conn[r * num_edges + eid][k]];
                    if (ej->N < ei->N)
                    {
                        edge_idx(eid, ei->n, dim, idx);
                        for (int i = 1; i < ei->n - 1; i++) glo[ei->offset + idx[i]] = 0;
                    }
                }
            }
            for (int fid = 0; fid < num_faces; fid++)
            {
                for (int k = 0; k < face_conn_n[r * num_faces + fid]; k++)
                {
                    const relem *ej = &region[face_conn[r * num_faces + fid][k]];
                    if (ej->N < ei->N)
                    {
                        face_idx(fid, ei->n, idx);
                        for (int j = 1; j < ei->n - 1; j++)
                            for (int i = 1; i < ei->n - 1; i++) glo[ei->offset + idx[i + j * ei->n]] = 0;
                    }
                }
            }
        }
        free(idx);
    }

    /* interface nodes second to last, extended nodes last (:1100-1124) */
    {
        long long max_subdomain_num = 0;
        for (int p = 0; p < NP; p++)
            if (glo[p] > max_subdomain_num) max_subdomain_num = glo[p];
        for (int r = 0; r < nreg; r++)
        {
            const relem *el = &region[r];
            if (el->N == 1)
                for (int v = 0; v < el->num_points; v++)
                    if (dofn[el->offset + v] > 0) glo[el->offset + v] += max_subdomain_num;
        }
        for (int p = 0; p < NP; p++)
            if (glo[p] > max_subdomain_num) max_subdomain_num = glo[p];
        for (int r = num_subdomain_elems; r < nreg; r++)
        {
            const relem *el = &region[r];
            for (int v = 0; v < el->num_points; v++)
                if (mask[el->offset + v] > 0.0 && dofn[el->offset + v] == 0) glo[el->offset + v] += max_subdomain_num;
        }
    }

    /* ranking (:1151-1176) */
    {
        double *w = (double *)xcalloc((size_t)NP + 1, sizeof(double));
        for (int p = 0; p < NP; p++) w[p] = (double)glo[p];
        orc_ranking(w, NP);
        for (int p = 0; p < NP; p++) glo[p] = (long long)w[p];
        for (int p = 0; p < NP; p++) w[p] = (double)glo[p] * mask[p];
        orc_ranking(w, NP);
        for (int p = 0; p < NP; p++) dofn[p] = (long long)w[p];
        free(w);
    }

    /* ---- Q (subdomain.tpp:1496-1585) ---- */
    int num_ext_dofs = 0;
    for (int p = 0; p < NP; p++)
        if ((int)dofn[p] > num_ext_dofs) num_ext_dofs = (int)dofn[p];
    {
        coo Q;
        memset(&Q, 0, sizeof(Q));
        int nmax = poly_degree[0] + 1;
        int *idx_i = (int *)xcalloc((size_t)nmax * nmax + 4, sizeof(int));
        int *idx_j = (int *)xcalloc((size_t)nmax * nmax + 4, sizeof(int));
        const int(*pairs)[2] = (dim == 2) ? edge_pairs_2d : edge_pairs_3d;

        for (int r = 0; r < nreg; r++)
        {
            const relem *ei = &region[r];
            int N_i = ei->N, n_i = ei->n;

            for (int vid = 0; vid < ei->num_points; vid++) /* :1517-1520 */
                if (dofn[ei->offset + vid] > 0) coo_add(&Q, ei->offset + vid, (int)dofn[ei->offset + vid] - 1, 1.0);

            for (int eid = 0; eid < num_edges; eid++) /* :1522-1551 */
            {
                int e_j = -1, N_j = N_i, n_j = N_j + 1;
                for (int k = 0; k < edge_conn_n[r * num_edges + eid]; k++)
                {
                    int e = edge_conn[r * num_edges + eid][k];
                    if (region[e].N < N_j)
                    {
                        e_j = e;
                        N_j = region[e].N;
                        n_j = N_j + 1;
                    }
                }
                if (e_j < 0) continue;
                const relem *ej = &region[e_j];

                /* matching_edge (:1179-1348): the edge of elem_j with the same two corner ids, each side in its own direction */
                long long a = geometry_mesh[(size_t)ei->id * nv + pairs[eid][0]], b = geometry_mesh[(size_t)ei->id * nv + pairs[eid][1]];
                int eid_j = -1;
                for (int k = 0; k < num_edges && eid_j < 0; k++)
                {
                    long long c0 = geometry_mesh[(size_t)ej->id * nv + pairs[k][0]], c1 = geometry_mesh[(size_t)ej->id * nv + pairs[k][1]];
                    if ((c0 == a || c0 == b) && (c1 == a || c1 == b)) eid_j = k;
                }
                if (eid_j < 0) continue;
                edge_idx(eid, n_i, dim, idx_i);
                edge_idx(eid_j, n_j, dim, idx_j);

                /* J_cf[(N_j, N_i)]: n_i x n_j */
                int lf = ei->level, lc = ej->level;
                const double *J = J_cf_pairs[lf * num_levels + lc];
                for (int i = 1; i < n_i - 1; i++)
                    for (int j = 0; j < n_j; j++)
                        if (dofn[ej->offset + idx_j[j]] > 0) coo_add(&Q, ei->offset + idx_i[i], (int)dofn[ej->offset + idx_j[j]] - 1, J[i * n_j + j]);
            }

            for (int fid = 0; fid < num_faces; fid++) /* :1553-1578 */
            {
                for (int k = 0; k < face_conn_n[r * num_faces + fid]; k++)
                {
                    const relem *ej = &region[face_conn[r * num_faces + fid][k]];
                    int N_j = ej->N, n_j = N_j + 1;
                    if (!(N_i > N_j)) continue;

                    /* matching_face (:1350-1494) */
                    long long fk[4], gk[4];
                    for (int q = 0; q < 4; q++) fk[q] = geometry_mesh[(size_t)ei->id * nv + face_quads[fid][q]];
                    qsort(fk, 4, sizeof(long long), cmp_ll);
                    int fid_j = -1;
                    for (int f2 = 0; f2 < num_faces && fid_j < 0; f2++)
                    {
                        for (int q = 0; q < 4; q++) gk[q] = geometry_mesh[(size_t)ej->id * nv + face_quads[f2][q]];
                        qsort(gk, 4, sizeof(long long), cmp_ll);
                        if (gk[0] == fk[0] && gk[1] == fk[1] && gk[2] == fk[2] && gk[3] == fk[3]) fid_j = f2;
                    }
                    if (fid_j < 0) continue;
                    face_idx(fid, n_i, idx_i);
                    face_idx(fid_j, n_j, idx_j);

                    const double *J = J_cf_pairs[ei->level * num_levels + ej->level];
                    for (int j = 1; j < n_i - 1; j++)
                        for (int i = 1; i < n_i - 1; i++)
                            for (int q = 0; q < n_j; q++)
                                for (int p = 0; p < n_j; p++)
                                    if (dofn[ej->offset + idx_j[p + q * n_j]] > 0)
                                        coo_add(&Q, ei->offset + idx_i[i + j * n_i], (int)dofn[ej->offset + idx_j[p + q * n_j]] - 1, J[i * n_j + p] * J[j * n_j + q]);
                }
            }
        }
        orc_csr_assemble(&s->Q, NP, num_ext_dofs, Q.row, Q.col, Q.val, Q.n);
        orc_csr_transpose(&s->Q, &s->Qt);
        coo_free(&Q);
        free(idx_i);
        free(idx_j);
    }

    /* subdomain stiffness operator (:1587-1630) */
    s->num_dofs = 0;
    for (int r = 0; r < num_subdomain_elems; r++)
        for (int v = 0; v < region[r].num_points; v++)
            if ((int)dofn[region[r].offset + v] > s->num_dofs) s->num_dofs = (int)dofn[region[r].offset + v];
    s->num_points = NP;
    s->num_extended_dofs = num_ext_dofs;
    s->offset = (int *)xcalloc((size_t)NP + 1, sizeof(int));
    s->vertex = (int *)xcalloc((size_t)NP + 1, sizeof(int));
    s->level = (int *)xcalloc((size_t)NP + 1, sizeof(int));
    for (int r = 0; r < nreg; r++)
        for (int v = 0; v < region[r].num_points; v++)
        {
            s->offset[region[r].offset + v] = region[r].offset;
            s->vertex[region[r].offset + v] = v;
            s->level[region[r].offset + v] = region[r].level;
        }

    /* ---- superdomain (subdomain.tpp:1850-2576) ---- */
    int *dof_sup = (int *)xcalloc((size_t)num_coarse_dofs + 1, sizeof(int));
    int *dof_marker = (int *)xcalloc((size_t)num_coarse_dofs + 1, sizeof(int));
    F->dof_sup[me] = dof_sup;
    F->comp_levels[me] = (int *)xcalloc(GRADE_MAX_LEVELS + 2, sizeof(int));
    F->comp_levels[me][0] = -1;
    int sup_num_dofs = 0, sup_num_ext_dofs = 0;
    if (num_superdomain_elems > 0)
    {
        for (int r = 0; r < num_subdomain_elems; r++) /* :1862-1877 */
        {
            int eid = region[r].id;
            for (int v = 0; v < nv; v++)
            {
                int dof = dof_num_coarse[eid * nv + v];
                long long g = glo_num_coarse[eid * nv + v];
                if (dof > 0) dof_marker[dof - 1] = 1;
                if (dof > 0 && find_ll(interface_glo_num, n_interface, g) >= 0) dof_marker[dof - 1] = 2;
            }
        }
        for (int r = num_subdomain_elems; r < num_subdomain_extended_elems; r++) /* :1879-1891 */
        {
            int eid = region[r].id;
            for (int v = 0; v < nv; v++)
            {
                int dof = dof_num_coarse[eid * nv + v];
                if (dof > 0 && dof_marker[dof - 1] == 0) dof_marker[dof - 1] = 3;
            }
        }
        for (int k = 0; k < num_sup_ext; k++) /* :1893-1905 */
        {
            int eid = region[sup_ext[k]].id;
            for (int v = 0; v < nv; v++)
            {
                int dof = dof_num_coarse[eid * nv + v];
                if (dof > 0 && dof_marker[dof - 1] == 1) dof_marker[dof - 1] = 4;
            }
        }

        orc_csr P_comp, A_comp;
        memset(&P_comp, 0, sizeof(P_comp));
        memset(&A_comp, 0, sizeof(A_comp));
        int *comp_of_fine = (int *)xcalloc((size_t)num_coarse_dofs + 1, sizeof(int));
        int group[4];
        grade_superdomain(A_coarse, dof_marker, superdomain_overlap, &P_comp, &A_comp, comp_of_fine, group, F->comp_levels[me]);

        /* :2404-2424 */
        int ncomp = A_comp.num_rows;
        int *R_sup = (int *)xcalloc((size_t)ncomp + 1, sizeof(int));
        for (int i = 0; i < ncomp; i++) R_sup[i] = -1;
        int marker_offset[5];
        marker_offset[0] = 0;
        for (int m = 1; m < 5; m++) marker_offset[m] = marker_offset[m - 1] + group[m - 1];
        int dof = 0;
        for (int i = marker_offset[1]; i < marker_offset[3]; i++) R_sup[i] = dof++;
        for (int i = marker_offset[4]; i < ncomp; i++) R_sup[i] = dof++;
        for (int i = marker_offset[3]; i < marker_offset[4]; i++) R_sup[i] = dof++;
        sup_num_ext_dofs = dof;
        sup_num_dofs = dof - group[3]; /* :2545 */

        coo Ac, Pc;
        memset(&Ac, 0, sizeof(Ac));
        memset(&Pc, 0, sizeof(Pc));
        for (int i = 0; i < ncomp; i++) /* :2437-2449 */
            for (int p = A_comp.ptr[i]; p < A_comp.ptr[i + 1]; p++)
                if (R_sup[i] >= 0 && R_sup[A_comp.col[p]] >= 0) coo_add(&Ac, R_sup[i], R_sup[A_comp.col[p]], A_comp.val[p]);
        ij_assemble(&s->sup_A, dof, dof, &Ac);
        for (int row = 0; row < P_comp.num_rows; row++) /* :2465-2476, transposed as at :2549-2561 */
            for (int p = P_comp.ptr[row]; p < P_comp.ptr[row + 1]; p++)
                if (R_sup[P_comp.col[p]] >= 0) coo_add(&Pc, R_sup[P_comp.col[p]], row, P_comp.val[p]);
        ij_assemble(&s->sup_Pt, dof, num_coarse_dofs, &Pc);
        coo_free(&Ac);
        coo_free(&Pc);

        for (int i = 0; i < num_coarse_dofs; i++)
            if (comp_of_fine[i] >= 0 && R_sup[comp_of_fine[i]] >= 0) dof_sup[i] = R_sup[comp_of_fine[i]] + 1;

        free(R_sup);
        free(comp_of_fine);
        orc_csr_free(&P_comp);
        orc_csr_free(&A_comp);
    }
    else
    {
        s->sup_A.ptr = (int *)xcalloc(1, sizeof(int));
        s->sup_Pt.ptr = (int *)xcalloc(1, sizeof(int));
        s->sup_Pt.num_cols = num_coarse_dofs;
    }
    s->sup_num_dofs = sup_num_dofs;
    s->sup_num_extended_dofs = sup_num_ext_dofs;
    F->sup_num_dofs[me] = sup_num_dofs;

    /* Qt_coarse is the same on every rank (:1706-1713) */
    {
        coo Qc;
        memset(&Qc, 0, sizeof(Qc));
        for (int e = 0; e < num_total_elements; e++)
            for (int v = 0; v < nv; v++)
                if (dof_num_coarse[e * nv + v] > 0) coo_add(&Qc, dof_num_coarse[e * nv + v] - 1, e * nv + v, 1.0);
        orc_csr_assemble(&s->Qt_coarse, num_coarse_dofs, num_total_elements * nv, Qc.row, Qc.col, Qc.val, Qc.n);
        coo_free(&Qc);
    }

    /* ---- interface operator (subdomain.tpp:2581-2729) ---- */
    {
        const int num_interface_dofs = n_interface;
        const int nse = s->num_extended_dofs, nue = sup_num_ext_dofs, ns = s->num_dofs, nu = sup_num_dofs;
        const int num_dofs = ns + nu - num_interface_dofs;
        s->num_interface_dofs = num_interface_dofs;
        s->num_unique_dofs = num_dofs;

        long long *subdomain_dof_mapping = (long long *)xcalloc((size_t)nse + 2, sizeof(long long));   /* key: 1-based subdomain dof */
        long long *superdomain_dof_mapping = (long long *)xcalloc((size_t)nue + 2, sizeof(long long)); /* key: 1-based superdomain dof */
        for (int r = 0; r < num_subdomain_elems; r++) /* :2587-2594 */
            for (int v = 0; v < region[r].num_points; v++)
                if (dofn[region[r].offset + v] > 0) subdomain_dof_mapping[dofn[region[r].offset + v]] = dofn[region[r].offset + v];
        for (int r = num_subdomain_elems; r < num_subdomain_extended_elems; r++) /* :2596-2612 */
        {
            const relem *el = &region[r];
            for (int v = 0; v < el->num_points; v++)
            {
                int dof = dof_num_coarse[el->id * nv + v];
                if (dof > 0)
                {
                    dof -= 1;
                    if (dof_sup[dof] > 0) subdomain_dof_mapping[dofn[el->offset + v]] = dof_sup[dof] + (ns - num_interface_dofs);
                }
            }
        }
        /* :2616-2630 -- with aggregates in the composite not every superdomain dof has a level-0 node, so the
         * identity part of the map is written for every dof directly (same values where the reference writes them) */
        for (int i = 1; i <= nu; i++) superdomain_dof_mapping[i] = i + (ns - num_interface_dofs);
        for (int k = 0; k < num_sup_ext; k++) /* :2632-2651 */
        {
            const relem *el = &region[sup_ext[k]];
            for (int v = 0; v < el->num_points; v++)
            {
                int dof = dof_num_coarse[el->id * nv + v];
                if (dof > 0)
                {
                    dof -= 1;
                    if (dof_marker[dof] == 4) superdomain_dof_mapping[dof_sup[dof]] = dofn[el->offset + v];
                }
            }
        }

        coo M;
        memset(&M, 0, sizeof(M));
        for (int i = 0; i < nse; i++) coo_add(&M, i, (int)subdomain_dof_mapping[i + 1] - 1, 1.0); /* :2655-2656 */
        for (int i = 0; i < nue; i++) coo_add(&M, nse + i, (int)superdomain_dof_mapping[i + 1] - 1, 1.0);
        orc_csr_assemble(&s->Q_int, nse + nue, num_dofs, M.row, M.col, M.val, M.n);
        M.n = 0;
        for (int i = 0; i < ns; i++) coo_add(&M, i, i, 1.0); /* :2665-2669 */
        for (int i = 0; i < nu - num_interface_dofs; i++) coo_add(&M, ns + i, nse + num_interface_dofs + i, 1.0);
        orc_csr_assemble(&s->Qt_int, num_dofs, nse + nue, M.row, M.col, M.val, M.n);
        M.n = 0;
        double *seen = (double *)xcalloc((size_t)nse + nue + 1, sizeof(double));
        for (int i = 0; i < ns; i++) /* :2677-2681 */
        {
            coo_add(&M, i, i, 1.0);
            seen[i] = 1.0;
        }
        for (int r = num_subdomain_elems; r < num_subdomain_extended_elems; r++) /* :2683-2695 */
        {
            const relem *el = &region[r];
            for (int v = 0; v < el->num_points; v++)
            {
                long long d = dofn[el->offset + v];
                if (d > 0 && seen[d - 1] == 0)
                {
                    coo_add(&M, (int)d - 1, nse + dof_sup[dof_num_coarse[el->id * nv + v] - 1] - 1, 1.0);
                    seen[d - 1] = 1.0;
                }
            }
        }
        for (int i = 0; i < num_interface_dofs; i++) /* :2697-2701 */
        {
            coo_add(&M, nse + i, ns - num_interface_dofs + i, 1.0);
            seen[nse + i] = 1.0;
        }
        for (int i = num_interface_dofs; i < nu; i++) /* :2703-2707 */
        {
            coo_add(&M, nse + i, nse + i, 1.0);
            seen[nse + i] = 1.0;
        }
        for (int k = 0; k < num_sup_ext; k++) /* :2709-2727 */
        {
            const relem *el = &region[sup_ext[k]];
            for (int v = 0; v < el->num_points; v++)
            {
                if (dof_num_coarse[el->id * nv + v] > 0)
                {
                    int dof = dof_sup[dof_num_coarse[el->id * nv + v] - 1];
                    if (dof > 0 && seen[nse + dof - 1] == 0)
                    {
                        coo_add(&M, nse + dof - 1, (int)dofn[el->offset + v] - 1, 1.0);
                        seen[nse + dof - 1] = 1.0;
                    }
                }
            }
        }
        orc_csr_assemble(&s->QQt_int, nse + nue, nse + nue, M.row, M.col, M.val, M.n);
        coo_free(&M);
        free(seen);
        free(subdomain_dof_mapping);
        free(superdomain_dof_mapping);

        /* norm weighting (:2731-2737) */
        s->norm_weight = (double *)xcalloc((size_t)nse + nue + 1, sizeof(double));
        for (int i = 0; i < nse + nue; i++) s->norm_weight[i] = 1.0;
        for (int i = ns; i < nse; i++) s->norm_weight[i] = 0.0;
        for (int i = 0; i < num_interface_dofs; i++) s->norm_weight[nse + i] = 0.0;
        for (int i = nu; i < nue; i++) s->norm_weight[nse + i] = 0.0;

        /* inner product weight (:2739-2747) */
        s->num_values = NP + nue; /* :3858 */
        s->num_blocks = (s->num_values + ORC_BLOCK_SIZE - 1) / ORC_BLOCK_SIZE;
        s->inner_weight = (double *)xcalloc((size_t)s->num_values + 1, sizeof(double));
        orc_csr_multiply(s->inner_weight, s->Q.ptr, s->Q.col, s->Q.val, s->norm_weight, s->Q.num_rows);
        for (int i = 0; i < nue; i++) s->inner_weight[NP + i] = s->norm_weight[nse + i];
        for (int i = 0; i < s->num_values; i++)
            if (s->inner_weight[i] > 0.0) s->inner_weight[i] = 1.0;
    }

    /* ---- low-order operator of the composite (subdomain.tpp:2749-3472) ---- */
    F->point_dof[me] = (int *)xcalloc((size_t)NP + 1, sizeof(int));
    for (int p = 0; p < NP; p++) F->point_dof[me][p] = (glo[p] > 0 && dofn[p] > 0) ? (int)dofn[p] - 1 : -1;
    if (dim == 3)
    {
        const double epsilon = 1.0e-12;
        const int(*pairs)[2] = edge_pairs_3d;
        const int nse = s->num_extended_dofs;
        /* edges around every face, in the order the reference fills the face's boundary (:3205-3264) */
        static const int face_edges[6][4] = {{0, 1, 2, 3}, {4, 5, 6, 7}, {0, 4, 8, 9}, {1, 5, 10, 11}, {2, 6, 8, 10}, {3, 7, 9, 11}};
        static const int low_order_elems[6][4][3] = {{{0, 0, 0}, {0, 1, 0}, {1, 0, 0}, {1, 0, 1}}, {{1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 0, 1}}, {{0, 0, 0}, {0, 0, 1}, {0, 1, 0}, {1, 0, 1}},
                                                    {{1, 0, 1}, {1, 1, 0}, {1, 1, 1}, {0, 1, 0}}, {{0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 1, 0}}, {{1, 0, 1}, {1, 1, 1}, {0, 1, 1}, {0, 1, 0}}}; /* :2877-2882 */
        static const double D_fem[3][16] = {{1.0, 0.0, 0.0, -1.0, 1.0, 0.0, 0.0, -1.0, 1.0, 0.0, 0.0, -1.0, 1.0, 0.0, 0.0, -1.0},
                                            {0.0, 1.0, 0.0, -1.0, 0.0, 1.0, 0.0, -1.0, 0.0, 1.0, 0.0, -1.0, 0.0, 1.0, 0.0, -1.0},
                                            {0.0, 0.0, 1.0, -1.0, 0.0, 0.0, 1.0, -1.0, 0.0, 0.0, 1.0, -1.0, 0.0, 0.0, 1.0, -1.0}}; /* :2840-2842 */
        const double weight = 24.0;

        /* GLL nodes of every level from the meshes (reference element coordinates of an element's x-line) are not
         * available as such: J_cf_fem needs r_gll, recovered from the J_cf table is not possible either, so the nodes
         * are recomputed from the level's D_hat-independent definition: they are passed in through the meshes' first
         * element only when it is affine.  Instead the caller hands them over (orc_fdd_set_gll_nodes) before create
         * returns?  No: they come with the tables -- see r_gll below. */
        double **r_gll = (double **)xcalloc((size_t)num_levels, sizeof(double *));
        for (int l = 0; l < num_levels; l++)
        {
            /* nodes of level l on [-1, 1]: the interpolator from level l to itself is not in the tables, but the
             * interpolator J_cf[(N_c = 1, N_f = N_l)] holds h^c_1(xi^f_i) = (1 + xi_i) / 2 in its second column */
            int n = poly_degree[l] + 1;
            r_gll[l] = (double *)xcalloc((size_t)n, sizeof(double));
            if (poly_degree[l] == 1)
            {
                r_gll[l][0] = -1.0;
                r_gll[l][1] = 1.0;
            }
            else
            {
                const double *J = J_cf_pairs[l * num_levels + (num_levels - 1)]; /* n x 2 */
                for (int i = 0; i < n; i++) r_gll[l][i] = 2.0 * J[i * 2 + 1] - 1.0;
                r_gll[l][0] = -1.0;
                r_gll[l][n - 1] = 1.0;
            }
        }

        double *rx = (double *)xcalloc((size_t)NP + 1, sizeof(double));
        double *ry = (double *)xcalloc((size_t)NP + 1, sizeof(double));
        double *rz = (double *)xcalloc((size_t)NP + 1, sizeof(double));
        for (int r = 0; r < nreg; r++)
        {
            const relem *el = &region[r];
            const orc_mesh *m = &meshes[el->owner * num_levels + el->level];
            size_t src = (size_t)el->owner_elem * el->num_points;
            for (int v = 0; v < el->num_points; v++)
            {
                rx[el->offset + v] = m->x[src + v];
                ry[el->offset + v] = m->y[src + v];
                rz[el->offset + v] = m->z[src + v];
            }
        }

        /* the coarse differentiation matrices once more (:1731-1748) */
        double D[3][64], G[ORC_NUM_GEOM_FACTS][64], GD[3][64];
        memset(D, 0, sizeof(D));
        memset(G, 0, sizeof(G));
        {
            const double *Dh = D_hat[num_levels - 1];
            for (int p = 0; p < 2; p++)
                for (int q = 0; q < 2; q++)
                    for (int i = 0; i < 2; i++)
                        for (int j = 0; j < 2; j++)
                        {
                            D[0][(i + (p * 2 + q) * 2) * 8 + (j + (p * 2 + q) * 2)] = Dh[i * 2 + j];
                            D[1][(i * 8 + j) * 2 + ((p + p * 8) * (2 * 2) + (q + q * 8))] = Dh[i * 2 + j];
                            D[2][(i * 8 + j) * (2 * 2) + (p + q * 2) * (1 + 8)] = Dh[i * 2 + j];
                        }
        }

        coo Asub;
        memset(&Asub, 0, sizeof(Asub));
        int nmax = poly_degree[0] + 1, n3max = nmax * nmax * nmax;
        double *A_e = (double *)xcalloc((size_t)n3max * n3max, sizeof(double));
        int maxcols = n3max + 12 * nmax + 6 * nmax * nmax + 8;
        double *J_e = (double *)xcalloc((size_t)n3max * maxcols, sizeof(double));
        double *T_e = (double *)xcalloc((size_t)n3max * maxcols, sizeof(double));
        double *JtAJ = (double *)xcalloc((size_t)maxcols * maxcols, sizeof(double));
        int *col_dof = (int *)xcalloc((size_t)maxcols, sizeof(int));
        int *idx_i = (int *)xcalloc((size_t)nmax * nmax + 4, sizeof(int));
        int *idx_j = (int *)xcalloc((size_t)nmax * nmax + 4, sizeof(int));
        /* per element: local column (1-based, 0 none) and dof of vertices, hanging-edge masters, hanging-face masters */
        int *vert_first = (int *)xcalloc((size_t)n3max, sizeof(int));
        int *edge_first = (int *)xcalloc((size_t)12 * nmax, sizeof(int));
        int *edge_dof = (int *)xcalloc((size_t)12 * nmax, sizeof(int));
        int *edge_nj = (int *)xcalloc(12, sizeof(int));
        int *face_first = (int *)xcalloc((size_t)6 * nmax * nmax, sizeof(int));
        int *face_dof = (int *)xcalloc((size_t)6 * nmax * nmax, sizeof(int));
        int *face_nj = (int *)xcalloc(6, sizeof(int));
        double **Jf_cache = (double **)xcalloc((size_t)num_levels * num_levels, sizeof(double *));

        for (int r = 0; r < nreg; r++)
        {
            const relem *ei = &region[r];
            const int N_i = ei->N, n_i = ei->n, n3 = ei->num_points;
            memset(A_e, 0, (size_t)n3 * n3 * sizeof(double));

            if (N_i > 1) /* :2932-3039 */
            {
                for (int s_z = 0; s_z < N_i; s_z++)
                    for (int s_y = 0; s_y < N_i; s_y++)
                        for (int s_x = 0; s_x < N_i; s_x++)
                            for (int t = 0; t < 6; t++)
                            {
                                int loc_sub[4];
                                double x_sub[4], y_sub[4], z_sub[4], H_fem[9], inv_H_fem[9], G_fem[9], work0[16], work1[16];
                                for (int vid = 0; vid < 4; vid++)
                                {
                                    int i = low_order_elems[t][vid][0], j = low_order_elems[t][vid][1], k = low_order_elems[t][vid][2];
                                    loc_sub[vid] = (s_x + i) + (s_y + j) * n_i + (s_z + k) * (n_i * n_i);
                                    x_sub[vid] = rx[ei->offset + loc_sub[vid]];
                                    y_sub[vid] = ry[ei->offset + loc_sub[vid]];
                                    z_sub[vid] = rz[ei->offset + loc_sub[vid]];
                                }
                                H_fem[0] = x_sub[0] - x_sub[3];
                                H_fem[1] = x_sub[1] - x_sub[3];
                                H_fem[2] = x_sub[2] - x_sub[3];
                                H_fem[3] = y_sub[0] - y_sub[3];
                                H_fem[4] = y_sub[1] - y_sub[3];
                                H_fem[5] = y_sub[2] - y_sub[3];
                                H_fem[6] = z_sub[0] - z_sub[3];
                                H_fem[7] = z_sub[1] - z_sub[3];
                                H_fem[8] = z_sub[2] - z_sub[3];
                                {
                                    const double *A = H_fem;
                                    double det_A = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]); /* :2785-2791 */
                                    inv_H_fem[0] = (1.0 / det_A) * (A[4] * A[8] - A[7] * A[5]); /* :2806-2814 */
                                    inv_H_fem[1] = (1.0 / det_A) * (A[2] * A[7] - A[8] * A[1]);
                                    inv_H_fem[2] = (1.0 / det_A) * (A[1] * A[5] - A[4] * A[2]);
                                    inv_H_fem[3] = (1.0 / det_A) * (A[5] * A[6] - A[8] * A[3]);
                                    inv_H_fem[4] = (1.0 / det_A) * (A[0] * A[8] - A[6] * A[2]);
                                    inv_H_fem[5] = (1.0 / det_A) * (A[2] * A[3] - A[5] * A[0]);
                                    inv_H_fem[6] = (1.0 / det_A) * (A[3] * A[7] - A[6] * A[4]);
                                    inv_H_fem[7] = (1.0 / det_A) * (A[1] * A[6] - A[7] * A[0]);
                                    inv_H_fem[8] = (1.0 / det_A) * (A[0] * A[4] - A[3] * A[1]);
                                    double det_H_fem = det_A;
                                    for (int m = 0; m < 3; m++) /* :2985-2999: the same value at each of the 4 quadrature points */
                                        for (int n = 0; n < 3; n++)
                                        {
                                            double G_val = 0.0;
                                            for (int k = 0; k < 3; k++) G_val += (det_H_fem / weight) * inv_H_fem[m * 3 + k] * inv_H_fem[n * 3 + k];
                                            G_fem[n + m * 3] = G_val;
                                        }
                                }
                                for (int i = 0; i < 16; i++) work1[i] = 0.0;
                                for (int m = 0; m < 3; m++) /* :3003-3019 */
                                    for (int n = 0; n < 3; n++)
                                    {
                                        for (int i = 0; i < 16; i++) work0[i] = 0.0;
                                        for (int i = 0; i < 4; i++)
                                            for (int j = 0; j < 4; j++)
                                                for (int k = 0; k < 4; k++) work0[i * 4 + j] += ((i == k) ? G_fem[n + m * 3] : 0.0) * D_fem[n][k * 4 + j];
                                        for (int i = 0; i < 4; i++)
                                            for (int j = 0; j < 4; j++)
                                                for (int k = 0; k < 4; k++) work1[i * 4 + j] += D_fem[m][k * 4 + i] * work0[k * 4 + j];
                                    }
                                for (int i = 0; i < 4; i++) /* :3021-3034 */
                                    for (int j = 0; j < 4; j++)
                                        if (fabs(work1[i * 4 + j]) > epsilon) A_e[(size_t)loc_sub[i] * n3 + loc_sub[j]] += work1[i * 4 + j];
                            }
            }
            else /* :3082-3123 */
            {
                for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++)
                    for (int v = 0; v < 8; v++) G[g][v * 8 + v] = s->geom_fact[g][ei->offset + v];
                for (int i = 0; i < 8; i++)
                    for (int j = 0; j < 8; j++)
                    {
                        double GD_1 = 0.0, GD_2 = 0.0, GD_3 = 0.0;
                        for (int k = 0; k < 8; k++)
                        {
                            GD_1 += G[0][i * 8 + k] * D[0][k * 8 + j] + G[3][i * 8 + k] * D[1][k * 8 + j] + G[4][i * 8 + k] * D[2][k * 8 + j];
                            GD_2 += G[3][i * 8 + k] * D[0][k * 8 + j] + G[1][i * 8 + k] * D[1][k * 8 + j] + G[5][i * 8 + k] * D[2][k * 8 + j];
                            GD_3 += G[4][i * 8 + k] * D[0][k * 8 + j] + G[5][i * 8 + k] * D[1][k * 8 + j] + G[2][i * 8 + k] * D[2][k * 8 + j];
                        }
                        GD[0][i * 8 + j] = GD_1;
                        GD[1][i * 8 + j] = GD_2;
                        GD[2][i * 8 + j] = GD_3;
                    }
                for (int i = 0; i < 8; i++)
                    for (int j = 0; j < 8; j++)
                    {
                        double val = 0.0;
                        for (int k = 0; k < 8; k++) val += D[0][k * 8 + i] * GD[0][k * 8 + j] + D[1][k * 8 + i] * GD[1][k * 8 + j] + D[2][k * 8 + i] * GD[2][k * 8 + j];
                        if (fabs(val) > epsilon) A_e[(size_t)i * n3 + j] += val;
                    }
            }

            /* local columns (:3130-3276) */
            int rank = 1;
            for (int vid = 0; vid < n3; vid++) vert_first[vid] = (glo[ei->offset + vid] > 0) ? rank++ : 0;
            for (int eid = 0; eid < 12; eid++) edge_nj[eid] = 0;
            for (int fid = 0; fid < 6; fid++) face_nj[fid] = 0;
            for (int eid = 0; eid < num_edges; eid++)
            {
                int e_j = -1, N_j = N_i, n_j = N_j + 1;
                for (int k = 0; k < edge_conn_n[r * num_edges + eid]; k++)
                {
                    int e = edge_conn[r * num_edges + eid][k];
                    if (region[e].N < N_j)
                    {
                        e_j = e;
                        N_j = region[e].N;
                        n_j = N_j + 1;
                    }
                }
                if (e_j < 0) continue;
                const relem *ej = &region[e_j];
                long long a = geometry_mesh[(size_t)ei->id * nv + pairs[eid][0]], b = geometry_mesh[(size_t)ei->id * nv + pairs[eid][1]];
                int eid_j = -1;
                for (int k = 0; k < num_edges && eid_j < 0; k++)
                {
                    long long c0 = geometry_mesh[(size_t)ej->id * nv + pairs[k][0]], c1 = geometry_mesh[(size_t)ej->id * nv + pairs[k][1]];
                    if ((c0 == a || c0 == b) && (c1 == a || c1 == b)) eid_j = k;
                }
                if (eid_j < 0) continue;
                edge_idx(eid, n_i, dim, idx_i);
                edge_idx(eid_j, n_j, dim, idx_j);
                edge_nj[eid] = n_j;
                edge_first[eid * nmax + 0] = vert_first[idx_i[0]];
                edge_dof[eid * nmax + 0] = (int)dofn[ei->offset + idx_i[0]];
                edge_first[eid * nmax + (n_j - 1)] = vert_first[idx_i[n_i - 1]];
                edge_dof[eid * nmax + (n_j - 1)] = (int)dofn[ei->offset + idx_i[n_i - 1]];
                for (int k = 1; k < n_j - 1; k++)
                {
                    edge_first[eid * nmax + k] = rank++;
                    edge_dof[eid * nmax + k] = (int)dofn[ej->offset + idx_j[k]];
                }
            }
            for (int fid = 0; fid < num_faces; fid++)
            {
                for (int k = 0; k < face_conn_n[r * num_faces + fid]; k++)
                {
                    const relem *ej = &region[face_conn[r * num_faces + fid][k]];
                    int N_j = ej->N, n_j = N_j + 1;
                    if (!(N_i > N_j)) continue;
                    long long fk[4], gk[4];
                    for (int q = 0; q < 4; q++) fk[q] = geometry_mesh[(size_t)ei->id * nv + face_quads[fid][q]];
                    qsort(fk, 4, sizeof(long long), cmp_ll);
                    int fid_j = -1;
                    for (int f2 = 0; f2 < num_faces && fid_j < 0; f2++)
                    {
                        for (int q = 0; q < 4; q++) gk[q] = geometry_mesh[(size_t)ej->id * nv + face_quads[f2][q]];
                        qsort(gk, 4, sizeof(long long), cmp_ll);
                        if (gk[0] == fk[0] && gk[1] == fk[1] && gk[2] == fk[2] && gk[3] == fk[3]) fid_j = f2;
                    }
                    if (fid_j < 0) continue;
                    face_idx(fid, n_i, idx_i);
                    face_idx(fid_j, n_j, idx_j);
                    face_nj[fid] = n_j;
                    int *ff = face_first + (size_t)fid * nmax * nmax, *fd = face_dof + (size_t)fid * nmax * nmax;
                    /* corners (:3200-3203) */
                    ff[0 + 0 * n_j] = vert_first[idx_i[0 + 0 * n_i]];
                    fd[0 + 0 * n_j] = (int)dofn[ei->offset + idx_i[0 + 0 * n_i]];
                    ff[(n_j - 1) + 0 * n_j] = vert_first[idx_i[(n_i - 1) + 0 * n_i]];
                    fd[(n_j - 1) + 0 * n_j] = (int)dofn[ei->offset + idx_i[(n_i - 1) + 0 * n_i]];
                    ff[0 + (n_j - 1) * n_j] = vert_first[idx_i[0 + (n_i - 1) * n_i]];
                    fd[0 + (n_j - 1) * n_j] = (int)dofn[ei->offset + idx_i[0 + (n_i - 1) * n_i]];
                    ff[(n_j - 1) + (n_j - 1) * n_j] = vert_first[idx_i[(n_i - 1) + (n_i - 1) * n_i]];
                    fd[(n_j - 1) + (n_j - 1) * n_j] = (int)dofn[ei->offset + idx_i[(n_i - 1) + (n_i - 1) * n_i]];
                    /* boundary from the four edges (:3205-3264) */
                    for (int q = 1; q < n_j - 1; q++)
                    {
                        const int *fe = face_edges[fid];
                        ff[q + 0 * n_j] = edge_first[fe[0] * nmax + q];
                        fd[q + 0 * n_j] = edge_dof[fe[0] * nmax + q];
                        ff[q + (n_j - 1) * n_j] = edge_first[fe[1] * nmax + q];
                        fd[q + (n_j - 1) * n_j] = edge_dof[fe[1] * nmax + q];
                        ff[0 + q * n_j] = edge_first[fe[2] * nmax + q];
                        fd[0 + q * n_j] = edge_dof[fe[2] * nmax + q];
                        ff[(n_j - 1) + q * n_j] = edge_first[fe[3] * nmax + q];
                        fd[(n_j - 1) + q * n_j] = edge_dof[fe[3] * nmax + q];
                    }
                    for (int jj = 1; jj < n_j - 1; jj++) /* :3266-3273 */
                        for (int ii = 1; ii < n_j - 1; ii++)
                        {
                            ff[ii + jj * n_j] = rank++;
                            fd[ii + jj * n_j] = (int)dofn[ej->offset + idx_j[ii + jj * n_j]];
                        }
                }
            }
            int num_cols = rank - 1;

            /* J_e (:3287-3355) */
            memset(J_e, 0, (size_t)n3 * num_cols * sizeof(double));
            for (int vid = 0; vid < n3; vid++)
                if (vert_first[vid] > 0) J_e[(size_t)vid * num_cols + vert_first[vid] - 1] += 1.0;
            for (int eid = 0; eid < num_edges; eid++)
            {
                if (edge_nj[eid] == 0) continue;
                int n_j = edge_nj[eid], N_j = n_j - 1;
                int lc = -1;
                for (int l = 0; l < num_levels; l++)
                    if (poly_degree[l] == N_j) lc = l;
                double **slot = &Jf_cache[ei->level * num_levels + lc];
                if (!*slot)
                {
                    /* J_cf_fem (:2754-2783) */
                    int n_f = n_i, n_c = n_j, N_f = N_i, N_c = N_j;
                    double *Jf = (double *)xcalloc((size_t)n_f * n_c, sizeof(double));
                    const double *rf = r_gll[ei->level], *rc = r_gll[lc];
                    Jf[0 * n_c + 0] = 1.0;
                    for (int i = 1; i < N_f; i++)
                        for (int j = 0; j < N_c; j++)
                            if ((rc[j] <= rf[i]) && (rf[i] <= rc[j + 1]))
                            {
                                Jf[i * n_c + (j + 0)] = (rc[j + 1] - rf[i]) / (rc[j + 1] - rc[j]);
                                Jf[i * n_c + (j + 1)] = (rf[i + 0] - rc[j]) / (rc[j + 1] - rc[j]);
                            }
                    Jf[(n_f - 1) * n_c + (n_c - 1)] = 1.0;
                    *slot = Jf;
                }
                const double *Jf = *slot;
                edge_idx(eid, n_i, dim, idx_i);
                for (int i = 1; i < n_i - 1; i++)
                    for (int j = 0; j < n_j; j++)
                    {
                        int col = edge_first[eid * nmax + j] - 1;
                        double val = Jf[i * n_j + j];
                        if (fabs(val) > epsilon && col >= 0) J_e[(size_t)idx_i[i] * num_cols + col] += val;
                    }
            }
            for (int fid = 0; fid < num_faces; fid++)
            {
                if (face_nj[fid] == 0) continue;
                int n_j = face_nj[fid], N_j = n_j - 1;
                int lc = -1;
                for (int l = 0; l < num_levels; l++)
                    if (poly_degree[l] == N_j) lc = l;
                const double *Jf = Jf_cache[ei->level * num_levels + lc]; /* filled by the face's edges above */
                const int *ff = face_first + (size_t)fid * nmax * nmax;
                face_idx(fid, n_i, idx_i);
                for (int j = 1; j < n_i - 1; j++)
                    for (int i = 1; i < n_i - 1; i++)
                        for (int q = 0; q < n_j; q++)
                            for (int pp = 0; pp < n_j; pp++)
                            {
                                int col = ff[pp + q * n_j] - 1;
                                double val = Jf[i * n_j + pp] * Jf[j * n_j + q];
                                if (fabs(val) > epsilon && col >= 0) J_e[(size_t)idx_i[i + j * n_i] * num_cols + col] += val;
                            }
            }

            /* dofs of the local columns (:3363-3375) */
            for (int k = 0; k < num_cols; k++) col_dof[k] = 0;
            for (int vid = 0; vid < n3; vid++)
                if (vert_first[vid] > 0) col_dof[vert_first[vid] - 1] = (int)dofn[ei->offset + vid];
            for (int eid = 0; eid < num_edges; eid++)
                for (int k = 0; k < edge_nj[eid]; k++)
                    if (edge_first[eid * nmax + k] > 0) col_dof[edge_first[eid * nmax + k] - 1] = edge_dof[eid * nmax + k];
            for (int fid = 0; fid < num_faces; fid++)
                for (int k = 0; k < face_nj[fid] * face_nj[fid]; k++)
                    if (face_first[(size_t)fid * nmax * nmax + k] > 0) col_dof[face_first[(size_t)fid * nmax * nmax + k] - 1] = face_dof[(size_t)fid * nmax * nmax + k];

            /* J^T A J (:3360-3361) */
            memset(T_e, 0, (size_t)n3 * num_cols * sizeof(double));
            for (int i = 0; i < n3; i++)
                for (int k = 0; k < n3; k++)
                {
                    double a = A_e[(size_t)i * n3 + k];
                    if (a == 0.0) continue;
                    for (int j = 0; j < num_cols; j++) T_e[(size_t)i * num_cols + j] += a * J_e[(size_t)k * num_cols + j];
                }
            memset(JtAJ, 0, (size_t)num_cols * num_cols * sizeof(double));
            for (int k = 0; k < n3; k++)
                for (int i = 0; i < num_cols; i++)
                {
                    double a = J_e[(size_t)k * num_cols + i];
                    if (a == 0.0) continue;
                    for (int j = 0; j < num_cols; j++) JtAJ[(size_t)i * num_cols + j] += a * T_e[(size_t)k * num_cols + j];
                }
            for (int i = 0; i < num_cols; i++) /* :3385-3403 */
                for (int j = 0; j < num_cols; j++)
                {
                    int row = col_dof[i], col = col_dof[j];
                    double val = JtAJ[(size_t)i * num_cols + j];
                    if (row > 0 && col > 0 && fabs(val) > epsilon) coo_add(&Asub, row - 1, col - 1, val);
                }
        }
        orc_csr A_sub_fem;
        ij_assemble(&A_sub_fem, nse, nse, &Asub);
        coo_free(&Asub);

        /* the combined operator on the unique dofs (:3414-3472) */
        {
            const int nue = s->sup_num_extended_dofs, ns = s->num_dofs, nu = s->sup_num_dofs, nI = s->num_interface_dofs;
            int *unique_of = (int *)xcalloc((size_t)nse + nue + 1, sizeof(int)); /* Q_int * (0, 1, 2, ...) */
            for (int i = 0; i < nse + nue; i++) unique_of[i] = (s->Q_int.ptr[i + 1] > s->Q_int.ptr[i]) ? s->Q_int.col[s->Q_int.ptr[i]] : -1;
            coo Af;
            memset(&Af, 0, sizeof(Af));
            for (int i = 0; i < ns; i++)
                for (int p = A_sub_fem.ptr[i]; p < A_sub_fem.ptr[i + 1]; p++) coo_add(&Af, unique_of[i], unique_of[A_sub_fem.col[p]], A_sub_fem.val[p]);
            if (nu > 0)
                for (int i = nI; i < nu; i++)
                    for (int p = s->sup_A.ptr[i]; p < s->sup_A.ptr[i + 1]; p++) coo_add(&Af, unique_of[nse + i], unique_of[nse + s->sup_A.col[p]], s->sup_A.val[p]);
            ij_assemble(&F->A_fem[me], s->num_unique_dofs, s->num_unique_dofs, &Af);
            coo_free(&Af);
            free(unique_of);
        }
        orc_csr_free(&A_sub_fem);

        for (int i = 0; i < num_levels * num_levels; i++) free(Jf_cache[i]);
        free(Jf_cache);
        for (int l = 0; l < num_levels; l++) free(r_gll[l]);
        free(r_gll);
        free(rx);
        free(ry);
        free(rz);
        free(A_e);
        free(J_e);
        free(T_e);
        free(JtAJ);
        free(col_dof);
        free(idx_i);
        free(idx_j);
        free(vert_first);
        free(edge_first);
        free(edge_dof);
        free(edge_nj);
        free(face_first);
        free(face_dof);
        free(face_nj);
    }
    else
    {
        F->A_fem[me].ptr = (int *)xcalloc(1, sizeof(int));
    }

    /* work arrays (:588-595) */
    {
        size_t tree = (size_t)s->levels[num_levels - 1].offset + (size_t)s->levels[num_levels - 1].num_points;
        size_t need = tree + (size_t)NP + 16;
        size_t alt = (size_t)s->num_extended_dofs + (size_t)s->sup_num_extended_dofs + 16;
        if (alt > need) need = alt;
        if ((size_t)num_total_elements * nv + 16 > need) need = (size_t)num_total_elements * nv + 16;
        if ((size_t)num_coarse_dofs + 16 > need) need = (size_t)num_coarse_dofs + 16;
        s->own_points = s->levels[0].num_points;
        orc_subdomain_alloc_solver(s, need);
    }
    s->tree_done = 1; /* the right-hand side comes from orc_fdd_tree_operator */

    for (int i = 0; i < nreg * num_edges; i++) free(edge_conn[i]);
    for (int i = 0; i < nreg * num_faces; i++) free(face_conn[i]);
    free(edge_conn);
    free(edge_conn_n);
    free(face_conn);
    free(face_conn_n);
    free(global_offset);
    free(interface_glo_num);
    free(dof_marker);
    free(mask);
    free(glo);
    free(dofn);
    free(region);
    free(region_index);
    free(sup_ext);
    free(work0);
    free(work1);
    free(mark);
#undef ADD_ELEM
#undef EXPAND
}

/* meshes[rank * num_levels + level]; J_cf_pairs[l_f * num_levels + l_c] for l_f < l_c (n_f x n_c row-major, subdomain.tpp:142-164) */
orc_fdd *orc_fdd_create(int num_ranks, int num_levels, const int *poly_degree, const double *const *D_hat, const double *const *J_cf_pairs, const orc_mesh *meshes, int subdomain_overlap, int superdomain_overlap)
{
    orc_fdd *F = (orc_fdd *)xcalloc(1, sizeof(orc_fdd));
    const int dim = meshes[0].dim;
    const int nv = (dim == 2) ? 4 : 8;
    F->num_ranks = num_ranks;
    F->num_levels = num_levels;
    F->dim = dim;
    F->num_vertices = nv;
    F->poly_degree = (int *)xcalloc((size_t)num_levels, sizeof(int));
    memcpy(F->poly_degree, poly_degree, (size_t)num_levels * sizeof(int));
    F->proc_count = (int *)xcalloc((size_t)num_ranks, sizeof(int));
    F->proc_offset = (int *)xcalloc((size_t)num_ranks, sizeof(int));
    for (int p = 0; p < num_ranks; p++) F->proc_count[p] = meshes[p * num_levels].num_local_elements;
    for (int p = 1; p < num_ranks; p++) F->proc_offset[p] = F->proc_offset[p - 1] + F->proc_count[p - 1];
    const int E = F->proc_offset[num_ranks - 1] + F->proc_count[num_ranks - 1];
    F->num_total_elements = E;

    F->sub = (orc_subdomain **)xcalloc((size_t)num_ranks, sizeof(orc_subdomain *));
    F->num_sub_elems = (int *)xcalloc((size_t)num_ranks, sizeof(int));
    F->num_sub_ext_elems = (int *)xcalloc((size_t)num_ranks, sizeof(int));
    F->region_id = (int **)xcalloc((size_t)num_ranks, sizeof(int *));
    F->region_level = (int **)xcalloc((size_t)num_ranks, sizeof(int *));
    F->region_offset = (int **)xcalloc((size_t)num_ranks, sizeof(int *));
    F->tree = (double **)xcalloc((size_t)num_ranks, sizeof(double *));
    F->sup_num_dofs = (int *)xcalloc((size_t)num_ranks, sizeof(int));
    F->dof_sup = (int **)xcalloc((size_t)num_ranks, sizeof(int *));
    F->comp_levels = (int **)xcalloc((size_t)num_ranks, sizeof(int *));
    F->A_fem = (orc_csr *)xcalloc((size_t)num_ranks, sizeof(orc_csr));
    F->point_dof = (int **)xcalloc((size_t)num_ranks, sizeof(int *));

    /* corner ids of every element (subdomain.tpp:225-262), element -> (rank, local id) (:270-280) */
    long long *geometry_mesh = (long long *)xcalloc((size_t)E * nv, sizeof(long long));
    int *owner = (int *)xcalloc((size_t)E, sizeof(int));
    int *owner_elem = (int *)xcalloc((size_t)E, sizeof(int));
    {
        const int n0 = poly_degree[0] + 1, np0 = ipow(n0, dim);
        for (int p = 0; p < num_ranks; p++)
        {
            const orc_mesh *m = &meshes[p * num_levels];
            for (int e = 0; e < F->proc_count[p]; e++)
            {
                int g = F->proc_offset[p] + e;
                owner[g] = p;
                owner_elem[g] = e;
                for (int v = 0; v < nv; v++) geometry_mesh[(size_t)g * nv + v] = m->glo_num[(size_t)e * np0 + corner_index(v, n0, dim)];
            }
        }
    }

    /* vertex -> elements: the vertex part of the connectivity maps (:289-306); edge and face neighbours of an
     * element are among the elements around its vertices */
    long long *ids = (long long *)xcalloc((size_t)E * nv, sizeof(long long));
    memcpy(ids, geometry_mesh, (size_t)E * nv * sizeof(long long));
    int nvert = sort_unique_ll(ids, E * nv);
    int *vert_of = (int *)xcalloc((size_t)E * nv, sizeof(int));
    for (int i = 0; i < E * nv; i++) vert_of[i] = find_ll(ids, nvert, geometry_mesh[i]);
    int *v2e_ptr = (int *)xcalloc((size_t)nvert + 1, sizeof(int));
    for (int i = 0; i < E * nv; i++) v2e_ptr[vert_of[i] + 1]++;
    for (int v = 0; v < nvert; v++) v2e_ptr[v + 1] += v2e_ptr[v];
    int *v2e = (int *)xcalloc((size_t)E * nv, sizeof(int));
    {
        int *fill = (int *)xcalloc((size_t)nvert, sizeof(int));
        for (int v = 0; v < nvert; v++) fill[v] = v2e_ptr[v];
        for (int e = 0; e < E; e++)
            for (int v = 0; v < nv; v++) v2e[fill[vert_of[e * nv + v]]++] = e;
        free(fill);
    }
    free(ids);

    /* coarse level: masked ids and dense dof numbers of every element's vertices (:1653-1704) */
    long long *glo_num_coarse = (long long *)xcalloc((size_t)E * nv, sizeof(long long));
    int *dof_num_coarse = (int *)xcalloc((size_t)E * nv, sizeof(int));
    {
        double *w = (double *)xcalloc((size_t)E * nv, sizeof(double));
        for (int e = 0; e < E; e++)
        {
            const orc_mesh *m = &meshes[owner[e] * num_levels + (num_levels - 1)];
            for (int v = 0; v < nv; v++)
            {
                size_t i = (size_t)owner_elem[e] * nv + v;
                glo_num_coarse[(size_t)e * nv + v] = (m->p_mask[i] > 0.0) ? m->glo_num[i] : 0;
                w[(size_t)e * nv + v] = (double)glo_num_coarse[(size_t)e * nv + v];
            }
        }
        orc_ranking(w, E * nv);
        F->num_coarse_dofs = 0;
        for (int i = 0; i < E * nv; i++)
        {
            dof_num_coarse[i] = (int)w[i];
            if (dof_num_coarse[i] > F->num_coarse_dofs) F->num_coarse_dofs = dof_num_coarse[i];
        }
        free(w);
    }

    /* the degree-1 operator of the whole domain (:1715-1848) */
    orc_csr A_coarse;
    {
        const double *Dh = D_hat[num_levels - 1];
        double D[3][64], G[ORC_NUM_GEOM_FACTS][64], GD[3][64], A_e[64];
        memset(D, 0, sizeof(D));
        memset(G, 0, sizeof(G));
        if (dim == 2)
        {
            for (int k = 0; k < 2; k++)
                for (int i = 0; i < 2; i++)
                    for (int j = 0; j < 2; j++) D[0][(i + k * 2) * 4 + (j + k * 2)] = Dh[i * 2 + j];
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++)
                    for (int k = 0; k < 2; k++) D[1][(i * 2 + k) * 4 + (j * 2 + k)] = Dh[i * 2 + j];
        }
        else
        {
            for (int p = 0; p < 2; p++)
                for (int q = 0; q < 2; q++)
                    for (int i = 0; i < 2; i++)
                        for (int j = 0; j < 2; j++)
                        {
                            D[0][(i + (p * 2 + q) * 2) * 8 + (j + (p * 2 + q) * 2)] = Dh[i * 2 + j];
                            D[1][(i * 8 + j) * 2 + ((p + p * 8) * (2 * 2) + (q + q * 8))] = Dh[i * 2 + j];
                            D[2][(i * 8 + j) * (2 * 2) + (p + q * 2) * (1 + 8)] = Dh[i * 2 + j];
                        }
        }
        coo Ac;
        memset(&Ac, 0, sizeof(Ac));
        const double epsilon = 1.0e-12; /* subdomain.hpp:233 */
        for (int e = 0; e < E; e++)
        {
            const orc_mesh *m = &meshes[owner[e] * num_levels + (num_levels - 1)];
            for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++)
                for (int v = 0; v < nv; v++) G[g][v * nv + v] = m->g[g] ? m->g[g][(size_t)owner_elem[e] * nv + v] : 0.0;
            for (int i = 0; i < nv; i++)
            {
                for (int j = 0; j < nv; j++)
                {
                    double GD_1 = 0.0, GD_2 = 0.0, GD_3 = 0.0;
                    for (int k = 0; k < nv; k++)
                    {
                        if (dim == 2)
                        {
                            GD_1 += G[0][i * 4 + k] * D[0][k * 4 + j] + G[2][i * 4 + k] * D[1][k * 4 + j];
                            GD_2 += G[2][i * 4 + k] * D[0][k * 4 + j] + G[1][i * 4 + k] * D[1][k * 4 + j];
                        }
                        else
                        {
                            GD_1 += G[0][i * 8 + k] * D[0][k * 8 + j] + G[3][i * 8 + k] * D[1][k * 8 + j] + G[4][i * 8 + k] * D[2][k * 8 + j];
                            GD_2 += G[3][i * 8 + k] * D[0][k * 8 + j] + G[1][i * 8 + k] * D[1][k * 8 + j] + G[5][i * 8 + k] * D[2][k * 8 + j];
                            GD_3 += G[4][i * 8 + k] * D[0][k * 8 + j] + G[5][i * 8 + k] * D[1][k * 8 + j] + G[2][i * 8 + k] * D[2][k * 8 + j];
                        }
                    }
                    GD[0][i * nv + j] = GD_1;
                    GD[1][i * nv + j] = GD_2;
                    GD[2][i * nv + j] = GD_3;
                }
            }
            for (int i = 0; i < nv * nv; i++) A_e[i] = 0.0;
            for (int i = 0; i < nv; i++)
                for (int j = 0; j < nv; j++)
                    for (int k = 0; k < nv; k++)
                    {
                        if (dim == 2)
                            A_e[i * 4 + j] += D[0][k * 4 + i] * GD[0][k * 4 + j] + D[1][k * 4 + i] * GD[1][k * 4 + j];
                        else
                            A_e[i * 8 + j] += D[0][k * 8 + i] * GD[0][k * 8 + j] + D[1][k * 8 + i] * GD[1][k * 8 + j] + D[2][k * 8 + i] * GD[2][k * 8 + j];
                    }
            for (int i = 0; i < nv; i++)
                for (int j = 0; j < nv; j++)
                {
                    int row = dof_num_coarse[e * nv + i] - 1, col = dof_num_coarse[e * nv + j] - 1;
                    double val = A_e[i * nv + j];
                    if (row >= 0 && col >= 0 && fabs(val) > epsilon) coo_add(&Ac, row, col, val);
                }
        }
        ij_assemble(&A_coarse, F->num_coarse_dofs, F->num_coarse_dofs, &Ac);
        coo_free(&Ac);
    }

    for (int me = 0; me < num_ranks; me++)
        build_rank(F, me, meshes, D_hat, J_cf_pairs, geometry_mesh, vert_of, v2e_ptr, v2e, owner, owner_elem, glo_num_coarse, dof_num_coarse, &A_coarse, subdomain_overlap, superdomain_overlap);

    for (int p = 0; p < num_ranks; p++)
    {
        const orc_subdomain *s = F->sub[p];
        size_t tree = (size_t)s->levels[num_levels - 1].offset + (size_t)s->levels[num_levels - 1].num_points;
        F->tree[p] = (double *)xcalloc(tree + 1, sizeof(double));
    }
    F->coarse_all = (double *)xcalloc((size_t)E * nv + 1, sizeof(double));
    F->coarse_dofs = (double *)xcalloc((size_t)F->num_coarse_dofs + 1, sizeof(double));

    orc_csr_free(&A_coarse);
    free(geometry_mesh);
    free(owner);
    free(owner_elem);
    free(vert_of);
    free(v2e_ptr);
    free(v2e);
    free(glo_num_coarse);
    free(dof_num_coarse);
    return F;
}

void orc_fdd_destroy(orc_fdd *F)
{
    if (!F) return;
    for (int p = 0; p < F->num_ranks; p++)
    {
        orc_subdomain_destroy(F->sub[p]);
        free(F->region_id[p]);
        free(F->region_level[p]);
        free(F->region_offset[p]);
        free(F->tree[p]);
        free(F->dof_sup[p]);
        free(F->comp_levels[p]);
        orc_csr_free(&F->A_fem[p]);
        free(F->point_dof[p]);
    }
    free(F->sub);
    free(F->num_sub_elems);
    free(F->num_sub_ext_elems);
    free(F->region_id);
    free(F->region_level);
    free(F->region_offset);
    free(F->tree);
    free(F->sup_num_dofs);
    free(F->dof_sup);
    free(F->comp_levels);
    free(F->A_fem);
    free(F->point_dof);
    free(F->coarse_all);
    free(F->coarse_dofs);
    free(F->poly_degree);
    free(F->proc_count);
    free(F->proc_offset);
    free(F);
}

orc_subdomain *orc_fdd_subdomain(orc_fdd *F, int rank) { return F->sub[rank]; }

/* info[]: 0 sub elems, 1 sub extended elems, 2 points, 3 sub dofs, 4 sub extended dofs, 5 interface dofs,
 * 6 sup dofs, 7 sup extended dofs, 8 unique dofs, 9 coarse dofs, 10 num_values, 11 own points */
void orc_fdd_info(const orc_fdd *F, int rank, int *info)
{
    const orc_subdomain *s = F->sub[rank];
    info[0] = F->num_sub_elems[rank];
    info[1] = F->num_sub_ext_elems[rank];
    info[2] = s->num_points;
    info[3] = s->num_dofs;
    info[4] = s->num_extended_dofs;
    info[5] = s->num_interface_dofs;
    info[6] = s->sup_num_dofs;
    info[7] = s->sup_num_extended_dofs;
    info[8] = s->num_unique_dofs;
    info[9] = F->num_coarse_dofs;
    info[10] = s->num_values;
    info[11] = s->own_points;
}

void orc_fdd_region(const orc_fdd *F, int rank, int *id, int *level)
{
    for (int r = 0; r < F->num_sub_ext_elems[rank]; r++)
    {
        id[r] = F->region_id[rank][r];
        level[r] = F->region_level[rank][r];
    }
}

const int *orc_fdd_composite_levels(const orc_fdd *F, int rank) { return F->comp_levels[rank]; }

/* which: 0 Q, 1 Qt, 2 Q_int, 3 Qt_int, 4 QQt_int, 5 superdomain A, 6 superdomain Pt, 7 Qt_coarse */
const orc_csr *orc_fdd_matrix(const orc_fdd *F, int rank, int which)
{
    const orc_subdomain *s = F->sub[rank];
    switch (which)
    {
    case 0: return &s->Q;
    case 1: return &s->Qt;
    case 2: return &s->Q_int;
    case 3: return &s->Qt_int;
    case 4: return &s->QQt_int;
    case 5: return &s->sup_A;
    case 6: return &s->sup_Pt;
    case 7: return &s->Qt_coarse;
    default: return NULL;
    }
}

const orc_csr *orc_fdd_low_order_matrix(const orc_fdd *F, int rank) { return &F->A_fem[rank]; }
const int *orc_fdd_point_dofs(const orc_fdd *F, int rank) { return F->point_dof[rank]; }
const double *orc_fdd_norm_weight(const orc_fdd *F, int rank) { return F->sub[rank]->norm_weight; }
const double *orc_fdd_inner_weight(const orc_fdd *F, int rank) { return F->sub[rank]->inner_weight; }

/* tree_operator for all ranks at once (subdomain.tpp:4566-4646): every rank restricts its own residual down
 * the degree tree, the coarsest level is gathered (:4620-4621), the region copies pull the owners' data (the gs
 * with +/- ids, :4626-4630), and the assembled coarse level goes through Pt into the tail (:4635-4644).
 * Tu[rank] has num_values entries; it is also left in the subdomain's f, where the solvers pick it up. */
void orc_fdd_tree_operator(orc_fdd *F, double *const *Tu, const double *const *u)
{
    const int dim = F->dim, nv = F->num_vertices, L = F->num_levels;

    for (int p = 0; p < F->num_ranks; p++)
    {
        orc_subdomain *s = F->sub[p];
        double *tree = F->tree[p];
        orc_sub_copy_f64_f64(tree, u[p], s->levels[0].num_points); /* :4571 */
        for (int l = 0; l < L - 1; l++) /* :4576-4609 */
        {
            int n_f = s->levels[l].poly_degree + 1, n_c = s->levels[l + 1].poly_degree + 1;
            const double *J = s->J_cf[l];
            double *u_f = tree + s->levels[l].offset, *u_c = tree + s->levels[l + 1].offset;
            int num_points;
            if (dim == 2)
            {
                num_points = s->levels[l].num_elements * (n_f * n_c);
                orc_sub_restriction_1(s->work[1], J, u_f, num_points, n_f, n_c, dim);
                num_points = s->levels[l].num_elements * (n_c * n_c);
                orc_sub_restriction_2(u_c, J, s->work[1], num_points, n_f, n_c, dim);
            }
            else
            {
                num_points = s->levels[l].num_elements * (n_f * n_f * n_c);
                orc_sub_restriction_1(s->work[1], J, u_f, num_points, n_f, n_c, dim);
                num_points = s->levels[l].num_elements * (n_f * n_c * n_c);
                orc_sub_restriction_2(s->work[2], J, s->work[1], num_points, n_f, n_c, dim);
                num_points = s->levels[l].num_elements * (n_c * n_c * n_c);
                orc_sub_restriction_3(u_c, J, s->work[2], num_points, n_f, n_c);
            }
        }
        /* coarse level of rank p into the gathered array (:4620-4621) */
        memcpy(F->coarse_all + (size_t)F->proc_offset[p] * nv, tree + s->levels[L - 1].offset, (size_t)s->levels[L - 1].num_points * sizeof(double));
    }

    for (int p = 0; p < F->num_ranks; p++)
    {
        orc_subdomain *s = F->sub[p];
        double *out = s->f;
        /* region copies take the owners' tree data (:4626-4630) */
        for (int r = 0; r < F->num_sub_ext_elems[p]; r++)
        {
            int e = F->region_id[p][r], l = F->region_level[p][r];
            int q = 0;
            while (q + 1 < F->num_ranks && F->proc_offset[q + 1] <= e) q++;
            int le = e - F->proc_offset[q];
            int np = ipow(F->poly_degree[l] + 1, dim);
            const orc_subdomain *so = F->sub[q];
            memcpy(out + F->region_offset[p][r], F->tree[q] + so->levels[l].offset + (size_t)le * np, (size_t)np * sizeof(double));
        }
        /* superdomain data (:4635-4644) */
        if (s->sup_num_extended_dofs > 0)
        {
            orc_csr_multiply(F->coarse_dofs, s->Qt_coarse.ptr, s->Qt_coarse.col, s->Qt_coarse.val, F->coarse_all, s->Qt_coarse.num_rows);
            if (s->sup_Pt.num_nnz > 0)
                orc_csr_multiply(out + s->num_points, s->sup_Pt.ptr, s->sup_Pt.col, s->sup_Pt.val, F->coarse_dofs, s->sup_Pt.num_rows);
            else
                memset(out + s->num_points, 0, (size_t)s->sup_num_extended_dofs * sizeof(double));
        }
        if (Tu && Tu[p] && Tu[p] != out) memcpy(Tu[p], out, (size_t)s->num_values * sizeof(double));
    }
}

/* z = M^-1 r on all ranks: tree_operator, then every rank's inner solve (no communication inside, the FDD
 * property).  method 0: flexible CG (subdomain.tpp:4161-4268), 1: GMRES (:4309-4489).  history: num_ranks rows of
 * history_cap entries; num_hist[rank] entries are valid. */
void orc_fdd_precondition(orc_fdd *F, double *const *z, const double *const *r, int method, const orc_subdomain_opts *opts, double *history, int history_cap, int *num_hist)
{
    orc_fdd_tree_operator(F, NULL, r);
    for (int p = 0; p < F->num_ranks; p++)
    {
        int nh = 0;
        double *h = history ? history + (size_t)p * history_cap : NULL;
        if (method == 0)
            orc_subdomain_fcg(F->sub[p], z[p], r[p], opts, h, history_cap, &nh);
        else
            orc_subdomain_gmres(F->sub[p], z[p], r[p], opts, h, history_cap, &nh);
        if (num_hist) num_hist[p] = nh;
    }
}
