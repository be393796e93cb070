/*
 * fdd_oracle_host.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Serial restatement of the host orchestration on the hot path:
 * CSR_Matrix (csr_matrix.tpp), Domain<double> setup of Q/Qt/weights
 * (domain.tpp:233-302), direct_stiffness_summation (domain.tpp:582-600),
 * stiffness_matrix (domain.tpp:602-609), the flexible CG and flexible
 * GMRES(m) drivers (domain.tpp:611-914) and their reductions
 * (domain.tpp:916-1002).
 *
 * MPI ranks are simulated in ONE process: an orc_world holds R domains and
 * every MPI_Allreduce(SUM) becomes an in-rank-order sum; gslib's gs_add on
 * the boundary-node prefix (domain.tpp:284, 592) becomes scatter-add into a
 * dense interface-slot vector followed by a gather (slots = sorted unique
 * global ids that appear in any rank's boundary prefix).
 *
 * See fdd_oracle.h for who may use this file and the parity-pin statement.
 */
#include "fdd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ====================================================================== */
/* small utilities                                                         */
/* ====================================================================== */
static void *xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    if (!p)
    {
        fprintf(stderr, "fdd_oracle: out of memory (%zu x %zu)\n", n, sz);
        abort();
    }
    return p;
}

/* open-addressing int64 -> int map (keys are 1-based node ids, never 0) */
typedef struct
{
    long long *keys;
    int *vals;
    size_t cap; /* power of two */
    size_t size;
} i64map;

static size_t i64hash(long long k)
{
    uint64_t x = (uint64_t)k;
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (size_t)x;
}

static void i64map_init(i64map *m, size_t expected)
{
    size_t cap = 16;
    while (cap < 2 * expected + 8) cap <<= 1;
    m->cap = cap;
    m->size = 0;
    m->keys = (long long *)xcalloc(cap, sizeof(long long));
    m->vals = (int *)xcalloc(cap, sizeof(int));
}

static void i64map_free(i64map *m)
{
    free(m->keys);
    free(m->vals);
    m->keys = NULL;
    m->vals = NULL;
}

/* returns pointer to value slot; *found tells whether the key existed */
static int *i64map_slot(i64map *m, long long key, int *found)
{
    size_t mask = m->cap - 1;
    size_t h = i64hash(key) & mask;

    while (m->keys[h] != 0 && m->keys[h] != key) h = (h + 1) & mask;

    if (m->keys[h] == key)
    {
        *found = 1;
        return &m->vals[h];
    }

    *found = 0;
    m->keys[h] = key;
    m->size++;
    return &m->vals[h];
}

static int i64map_get(const i64map *m, long long key, int *val)
{
    size_t mask = m->cap - 1;
    size_t h = i64hash(key) & mask;

    while (m->keys[h] != 0 && m->keys[h] != key) h = (h + 1) & mask;

    if (m->keys[h] == key)
    {
        *val = m->vals[h];
        return 1;
    }
    return 0;
}

/* ====================================================================== */
/* CSR_Matrix host class                                                   */
/* ====================================================================== */
typedef struct
{
    int row, col;
    double val;
    long seq; /* insertion order: makes the duplicate-sum order well defined
                 (std::sort in csr_matrix.tpp:101 leaves it unspecified) */
} coo_entry;

static int coo_cmp(const void *a_, const void *b_)
{
    const coo_entry *a = (const coo_entry *)a_;
    const coo_entry *b = (const coo_entry *)b_;
    if (a->row != b->row) return (a->row < b->row) ? -1 : 1;
    if (a->col != b->col) return (a->col < b->col) ? -1 : 1;
    if (a->seq != b->seq) return (a->seq < b->seq) ? -1 : 1;
    return 0;
}

/* csr_matrix.tpp:70-81 (add_entry) + :94-180 (assemble) */
int orc_csr_assemble(orc_csr *A, int num_rows, int num_cols, const int *rows, const int *cols, const double *vals, long n_entries)
{
    const double sparse_tolerance = 1.0e-12; /* csr_matrix.tpp:61-64 */

    A->num_rows = num_rows;
    A->num_cols = num_cols;
    A->num_nnz = 0;
    A->ptr = NULL;
    A->col = NULL;
    A->val = NULL;

    coo_entry *entries = (coo_entry *)xcalloc((size_t)n_entries, sizeof(coo_entry));
    long count_in = 0;

    for (long e = 0; e < n_entries; e++)
    {
        int row = rows[e], col = cols[e];

        if ((row < 0) || (row >= num_rows) || (col < 0) || (col >= num_cols))
        {
            free(entries);
            return -1; /* reference: printf + exit(EXIT_FAILURE), csr_matrix.tpp:72-76 */
        }

        if (fabs(vals[e]) > sparse_tolerance)
        {
            entries[count_in].row = row;
            entries[count_in].col = col;
            entries[count_in].val = vals[e];
            entries[count_in].seq = count_in;
            count_in++;
        }
    }

    /* csr_matrix.tpp:96: silent no-op for empty matrices */
    if ((num_rows == 0) || (num_cols == 0) || (count_in == 0))
    {
        free(entries);
        A->ptr = (int *)xcalloc((size_t)num_rows + 1, sizeof(int));
        return 0;
    }

    qsort(entries, (size_t)count_in, sizeof(coo_entry), coo_cmp);

    int num_nnz = 1;
    for (long i = 1; i < count_in; i++)
        if ((entries[i].row != entries[i - 1].row) || (entries[i].col != entries[i - 1].col)) num_nnz++;

    A->num_nnz = num_nnz;
    A->ptr = (int *)xcalloc((size_t)num_rows + 1, sizeof(int));
    A->col = (int *)xcalloc((size_t)num_nnz, sizeof(int));
    A->val = (double *)xcalloc((size_t)num_nnz, sizeof(double));

    int count = 0;
    A->ptr[entries[0].row + 1]++;
    A->col[0] = entries[0].col;
    A->val[0] = entries[0].val;

    for (long i = 1; i < count_in; i++)
    {
        if ((entries[i].row != entries[i - 1].row) || (entries[i].col != entries[i - 1].col))
        {
            count++;
            A->ptr[entries[i].row + 1]++;
            A->col[count] = entries[i].col;
            A->val[count] = entries[i].val;
        }
        else
        {
            A->val[count] += entries[i].val;
        }
    }

    for (int i = 1; i <= num_rows; i++) A->ptr[i] += A->ptr[i - 1];

    free(entries);
    return 0;
}

/* csr_matrix.tpp:228-258 */
void orc_csr_transpose(const orc_csr *A, orc_csr *At)
{
    long n = A->num_nnz;
    int *rows = (int *)xcalloc((size_t)n, sizeof(int));
    int *cols = (int *)xcalloc((size_t)n, sizeof(int));
    double *vals = (double *)xcalloc((size_t)n, sizeof(double));
    long c = 0;

    for (int i = 0; i < A->num_rows; i++)
    {
        for (int j = A->ptr[i]; j < A->ptr[i + 1]; j++)
        {
            rows[c] = A->col[j];
            cols[c] = i;
            vals[c] = A->val[j];
            c++;
        }
    }

    orc_csr_assemble(At, A->num_cols, A->num_rows, rows, cols, vals, c);

    free(rows);
    free(cols);
    free(vals);
}

/* csr_matrix.tpp:261-299 */
void orc_csr_diagonal(const orc_csr *A, double *D)
{
    for (int i = 0; i < A->num_rows; i++)
    {
        D[i] = 0.0;

        for (int j = A->ptr[i]; j < A->ptr[i + 1]; j++)
        {
            if (i == A->col[j])
            {
                D[i] = A->val[j];
                break;
            }
        }
    }
}

void orc_csr_free(orc_csr *A)
{
    free(A->ptr);
    free(A->col);
    free(A->val);
    A->ptr = NULL;
    A->col = NULL;
    A->val = NULL;
    A->num_nnz = 0;
}

/* ====================================================================== */
/* Domain                                                                  */
/* ====================================================================== */
struct orc_domain
{
    int dim;
    int poly_degree;
    int num_local_elements;
    int num_elem_points;
    int num_local_points;
    int num_local_nodes;
    int num_bdary_nodes;
    int num_blocks;

    orc_csr Q, Qt;
    double *assembled_weight; /* num_local_nodes */
    double *dirichlet_mask;   /* num_local_points */
    double *geom_fact[ORC_NUM_GEOM_FACTS];
    double *D_hat;

    long long *boundary_nodes; /* global ids of the prefix */
    int *bdary_slot;           /* world interface slot per boundary node */

    /* work_dev[0..dim) and solver vectors */
    double *work[3];
    double *r_k, *r_kp1, *q_k, *z_k, *p_k;
};

struct orc_world
{
    int num_ranks;
    orc_domain *dom;
    int num_slots;
    double *slots;
    /* scratch arrays-of-pointers */
    double **ptrs_a, **ptrs_b;
};

static void domain_setup(orc_domain *d, const orc_mesh *mesh, const double *D_hat)
{
    int n = mesh->poly_degree + 1;

    d->dim = mesh->dim;
    d->poly_degree = mesh->poly_degree;
    d->num_local_elements = mesh->num_local_elements;
    d->num_elem_points = (mesh->dim == 2) ? n * n : n * n * n;         /* domain.tpp:52 */
    d->num_local_points = d->num_local_elements * d->num_elem_points;  /* domain.tpp:53 */
    d->num_blocks = (d->num_local_points + ORC_BLOCK_SIZE - 1) / ORC_BLOCK_SIZE; /* domain.tpp:335 */

    int P = d->num_local_points;

    d->dirichlet_mask = (double *)xcalloc((size_t)P, sizeof(double));
    memcpy(d->dirichlet_mask, mesh->p_mask, (size_t)P * sizeof(double));

    for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++)
    {
        d->geom_fact[g] = (double *)xcalloc((size_t)P, sizeof(double));
        if (mesh->g[g]) memcpy(d->geom_fact[g], mesh->g[g], (size_t)P * sizeof(double));
    }

    d->D_hat = (double *)xcalloc((size_t)n * n, sizeof(double));
    memcpy(d->D_hat, D_hat, (size_t)n * n * sizeof(double));

    for (int w = 0; w < 3; w++) d->work[w] = (double *)xcalloc((size_t)P, sizeof(double));
    d->r_k = (double *)xcalloc((size_t)P, sizeof(double));
    d->r_kp1 = (double *)xcalloc((size_t)P, sizeof(double));
    d->q_k = (double *)xcalloc((size_t)P, sizeof(double));
    d->z_k = (double *)xcalloc((size_t)P, sizeof(double));
    d->p_k = (double *)xcalloc((size_t)P, sizeof(double));

    /* domain.tpp:236-247: local multiplicity of every global node */
    i64map local_node_degree;
    i64map_init(&local_node_degree, (size_t)P);

    for (int p = 0; p < P; p++)
    {
        int found;
        int *slot = i64map_slot(&local_node_degree, mesh->glo_num[p], &found);
        if (!found)
            *slot = 1;
        else
            (*slot)++;
    }

    /* domain.tpp:249-267: boundary nodes first, in first-encounter order */
    i64map local_node_idx;
    i64map_init(&local_node_idx, (size_t)P);
    d->boundary_nodes = (long long *)xcalloc((size_t)P, sizeof(long long));
    int count = 0;

    for (int p = 0; p < P; p++)
    {
        int deg = 0;
        i64map_get(&local_node_degree, mesh->glo_num[p], &deg);

        if (deg != mesh->node_degree[p])
        {
            int found;
            int *slot = i64map_slot(&local_node_idx, mesh->glo_num[p], &found);
            if (!found)
            {
                d->boundary_nodes[count] = mesh->glo_num[p];
                *slot = count;
                count++;
            }
        }
    }

    d->num_bdary_nodes = count; /* domain.tpp:269 */

    /* domain.tpp:271-281: the rest, in first-encounter order */
    for (int p = 0; p < P; p++)
    {
        int found;
        int *slot = i64map_slot(&local_node_idx, mesh->glo_num[p], &found);
        if (!found)
        {
            *slot = count;
            count++;
        }
    }

    d->num_local_nodes = (int)local_node_degree.size; /* domain.tpp:286 */

    /* domain.tpp:287-294: boolean Q (points x nodes), Qt = Q^T */
    int *rows = (int *)xcalloc((size_t)P, sizeof(int));
    int *cols = (int *)xcalloc((size_t)P, sizeof(int));
    double *vals = (double *)xcalloc((size_t)P, sizeof(double));

    for (int p = 0; p < P; p++)
    {
        int idx = 0;
        i64map_get(&local_node_idx, mesh->glo_num[p], &idx);
        rows[p] = p;
        cols[p] = idx;
        vals[p] = 1.0;
    }

    orc_csr_assemble(&d->Q, P, d->num_local_nodes, rows, cols, vals, P);
    orc_csr_transpose(&d->Q, &d->Qt);

    free(rows);
    free(cols);
    free(vals);
    i64map_free(&local_node_degree);
    i64map_free(&local_node_idx);

    d->assembled_weight = (double *)xcalloc((size_t)d->num_local_nodes, sizeof(double));
}

static int cmp_ll(const void *a, const void *b)
{
    long long x = *(const long long *)a, y = *(const long long *)b;
    return (x < y) ? -1 : (x > y);
}

/* gslib_gs(..., gs_add, ...) on each rank's boundary prefix
 * (domain.tpp:592): every id's values are summed over all ranks that list it
 * and the sum is written back to all of them. */
static void world_gs_add(orc_world *w, double *const *prefix)
{
    for (int s = 0; s < w->num_slots; s++) w->slots[s] = 0.0;

    for (int r = 0; r < w->num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];
        for (int b = 0; b < d->num_bdary_nodes; b++) w->slots[d->bdary_slot[b]] += prefix[r][b];
    }

    for (int r = 0; r < w->num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];
        for (int b = 0; b < d->num_bdary_nodes; b++) prefix[r][b] = w->slots[d->bdary_slot[b]];
    }
}

orc_world *orc_world_create(int num_ranks, const orc_mesh *meshes, const double *D_hat)
{
    orc_world *w = (orc_world *)xcalloc(1, sizeof(orc_world));
    w->num_ranks = num_ranks;
    w->dom = (orc_domain *)xcalloc((size_t)num_ranks, sizeof(orc_domain));
    w->ptrs_a = (double **)xcalloc((size_t)num_ranks, sizeof(double *));
    w->ptrs_b = (double **)xcalloc((size_t)num_ranks, sizeof(double *));

    long total_bdary = 0;

    for (int r = 0; r < num_ranks; r++)
    {
        domain_setup(&w->dom[r], &meshes[r], D_hat);
        total_bdary += w->dom[r].num_bdary_nodes;
    }

    /* gs_setup (domain.tpp:283-284): slots = sorted unique boundary ids */
    long long *all = (long long *)xcalloc((size_t)total_bdary, sizeof(long long));
    long c = 0;
    for (int r = 0; r < num_ranks; r++)
        for (int b = 0; b < w->dom[r].num_bdary_nodes; b++) all[c++] = w->dom[r].boundary_nodes[b];

    qsort(all, (size_t)total_bdary, sizeof(long long), cmp_ll);

    long uniq = 0;
    for (long i = 0; i < total_bdary; i++)
        if (i == 0 || all[i] != all[i - 1]) all[uniq++] = all[i];

    w->num_slots = (int)uniq;
    w->slots = (double *)xcalloc((size_t)uniq, sizeof(double));

    for (int r = 0; r < num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];
        d->bdary_slot = (int *)xcalloc((size_t)d->num_bdary_nodes, sizeof(int));

        for (int b = 0; b < d->num_bdary_nodes; b++)
        {
            long long key = d->boundary_nodes[b];
            long lo = 0, hi = uniq - 1;
            while (lo < hi)
            {
                long mid = (lo + hi) / 2;
                if (all[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            d->bdary_slot[b] = (int)lo;
        }
    }

    free(all);

    /* domain.tpp:296-302: assembled_weight = 1 / gs_add(Qt * 1) */
    for (int r = 0; r < num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];
        orc_set_to_value(d->work[0], 1.0, d->num_local_points, 0);
        orc_csr_multiply(d->assembled_weight, d->Qt.ptr, d->Qt.col, d->Qt.val, d->work[0], d->Qt.num_rows);
        w->ptrs_a[r] = d->assembled_weight;
    }

    world_gs_add(w, w->ptrs_a);

    for (int r = 0; r < num_ranks; r++)
        orc_invert_vector_elements(w->dom[r].assembled_weight, w->dom[r].num_local_nodes);

    return w;
}

void orc_world_destroy(orc_world *w)
{
    if (!w) return;

    for (int r = 0; r < w->num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];
        orc_csr_free(&d->Q);
        orc_csr_free(&d->Qt);
        free(d->assembled_weight);
        free(d->dirichlet_mask);
        for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++) free(d->geom_fact[g]);
        free(d->D_hat);
        free(d->boundary_nodes);
        free(d->bdary_slot);
        for (int k = 0; k < 3; k++) free(d->work[k]);
        free(d->r_k);
        free(d->r_kp1);
        free(d->q_k);
        free(d->z_k);
        free(d->p_k);
    }

    free(w->dom);
    free(w->slots);
    free(w->ptrs_a);
    free(w->ptrs_b);
    free(w);
}

int orc_world_num_ranks(const orc_world *w) { return w->num_ranks; }
int orc_world_num_local_points(const orc_world *w, int rank) { return w->dom[rank].num_local_points; }
int orc_world_num_local_nodes(const orc_world *w, int rank) { return w->dom[rank].num_local_nodes; }
int orc_world_num_bdary_nodes(const orc_world *w, int rank) { return w->dom[rank].num_bdary_nodes; }
int orc_world_num_interface_slots(const orc_world *w) { return w->num_slots; }
const orc_csr *orc_world_Q(const orc_world *w, int rank) { return &w->dom[rank].Q; }
const orc_csr *orc_world_Qt(const orc_world *w, int rank) { return &w->dom[rank].Qt; }
const double *orc_world_assembled_weight(const orc_world *w, int rank) { return w->dom[rank].assembled_weight; }

/* domain.tpp:582-600.  QQtu[r] may alias u[r] (domain.tpp:645). */
void orc_world_dssum(orc_world *w, double *const *QQtu, const double *const *u, int apply_mask, int apply_weight)
{
    for (int r = 0; r < w->num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];

        if (apply_weight)
            orc_csr_multiply_weight(d->work[0], d->Qt.ptr, d->Qt.col, d->Qt.val, u[r], d->assembled_weight, d->Qt.num_rows);
        else
            orc_csr_multiply(d->work[0], d->Qt.ptr, d->Qt.col, d->Qt.val, u[r], d->Qt.num_rows);

        w->ptrs_a[r] = d->work[0];
    }

    world_gs_add(w, w->ptrs_a);

    for (int r = 0; r < w->num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];

        if (apply_mask)
            orc_csr_multiply_weight(QQtu[r], d->Q.ptr, d->Q.col, d->Q.val, d->work[0], d->dirichlet_mask, d->Q.num_rows);
        else
            orc_csr_multiply(QQtu[r], d->Q.ptr, d->Q.col, d->Q.val, d->work[0], d->Q.num_rows);
    }
}

/* domain.tpp:602-609 */
void orc_world_stiffness(orc_world *w, double *const *Au, const double *const *u, int apply_dssum)
{
    for (int r = 0; r < w->num_ranks; r++)
    {
        orc_domain *d = &w->dom[r];
        double *GDu[3] = {d->work[0], d->work[1], d->work[2]};
        const double *G[6] = {d->geom_fact[0], d->geom_fact[1], d->geom_fact[2], d->geom_fact[3], d->geom_fact[4], d->geom_fact[5]};

        orc_dom_stiffness_matrix_1(GDu, u[r], d->D_hat, G, d->num_local_points, d->poly_degree, d->dim);
        orc_dom_stiffness_matrix_2(Au[r], (const double *const *)GDu, d->D_hat, d->num_local_points, d->poly_degree, d->dim);
    }

    if (apply_dssum) orc_world_dssum(w, Au, (const double *const *)Au, 1, 0);
}

/* domain.tpp:916-931 */
double orc_world_residual_norm(orc_world *w, const double *const *r)
{
    double r_norm = 0.0;

    for (int k = 0; k < w->num_ranks; k++) w->ptrs_b[k] = w->dom[k].work[1];
    orc_world_dssum(w, w->ptrs_b, r, 1, 0);

    for (int k = 0; k < w->num_ranks; k++)
    {
        orc_domain *d = &w->dom[k];
        orc_dom_residual_norm(d->work[0], r[k], d->work[1], d->dirichlet_mask, d->num_local_points, d->num_blocks);
        r_norm += orc_block_sum(d->work[0], d->num_blocks); /* + MPI_Allreduce(SUM) */
    }

    return sqrt(r_norm);
}

/* domain.tpp:933-947 */
double orc_world_assembled_inner_product(orc_world *w, const double *const *u, const double *const *v)
{
    double uv = 0.0;

    for (int k = 0; k < w->num_ranks; k++) w->ptrs_b[k] = w->dom[k].work[1];
    orc_world_dssum(w, w->ptrs_b, v, 1, 0);

    for (int k = 0; k < w->num_ranks; k++)
    {
        orc_domain *d = &w->dom[k];
        orc_dom_inner_product(d->work[0], u[k], d->work[1], d->dirichlet_mask, d->num_local_points, d->num_blocks);
        uv += orc_block_sum(d->work[0], d->num_blocks);
    }

    return uv;
}

static void push_hist(double *history, int cap, int *n, double v)
{
    if (history && *n < cap) history[*n] = v;
    (*n)++;
}

/* apply M^{-1} then the "stitching" dssum (domain.tpp:637-651, 697-711) */
static void apply_preconditioner(orc_world *w, const orc_solver_opts *opts, double *const *z, const double *const *r)
{
    if (opts->precond)
    {
        opts->precond(opts->precond_ctx, z, r);
        orc_world_dssum(w, z, (const double *const *)z, 1, 1);
    }
    else
    {
        orc_world_dssum(w, z, r, 1, 0);
    }
}

/* domain.tpp:611-725 */
int orc_world_fcg(orc_world *w, double *const *u, const double *const *f, const orc_solver_opts *opts, double *history, int history_cap, int *num_hist)
{
    int R = w->num_ranks;
    int nh = 0;

    double **r_k = (double **)xcalloc((size_t)R, sizeof(double *));
    double **r_kp1 = (double **)xcalloc((size_t)R, sizeof(double *));
    double **q_k = (double **)xcalloc((size_t)R, sizeof(double *));
    double **z_k = (double **)xcalloc((size_t)R, sizeof(double *));
    double **p_k = (double **)xcalloc((size_t)R, sizeof(double *));

    for (int k = 0; k < R; k++)
    {
        orc_domain *d = &w->dom[k];
        r_k[k] = d->r_k;
        r_kp1[k] = d->r_kp1;
        q_k[k] = d->q_k;
        z_k[k] = d->z_k;
        p_k[k] = d->p_k;
        orc_dom_initialize_arrays(u[k], r_k[k], f[k], d->num_local_points);
    }

    double r_norm;
    double r_0_norm = orc_world_residual_norm(w, (const double *const *)r_k);
    push_hist(history, history_cap, &nh, r_0_norm);

    double alpha_k, beta_k, gamma_k, theta_k;

    apply_preconditioner(w, opts, z_k, (const double *const *)r_k);

    for (int k = 0; k < R; k++) memcpy(p_k[k], z_k[k], (size_t)w->dom[k].num_local_points * sizeof(double));

    int num_iterations = 0;

    for (int iter = 0; iter < opts->max_iterations; iter++)
    {
        orc_world_stiffness(w, q_k, (const double *const *)p_k, 0);

        /* projection_inner_products (domain.tpp:949-973) */
        gamma_k = 0.0;
        theta_k = 0.0;
        for (int k = 0; k < R; k++)
        {
            orc_domain *d = &w->dom[k];
            orc_dom_projection_inner_products(d->work[0], z_k[k], r_k[k], p_k[k], q_k[k], d->num_local_points, d->num_blocks);

            double g = 0.0, t = 0.0;
            for (int b = 0; b < d->num_blocks; b++)
            {
                g += d->work[0][b];
                t += d->work[0][b + d->num_blocks];
            }
            gamma_k += g;
            theta_k += t;
        }

        alpha_k = gamma_k / theta_k;

        for (int k = 0; k < R; k++)
            orc_dom_solution_and_residual_update(u[k], r_kp1[k], r_k[k], p_k[k], q_k[k], alpha_k, w->dom[k].num_local_points);

        r_norm = orc_world_residual_norm(w, (const double *const *)r_kp1);
        push_hist(history, history_cap, &nh, r_norm);

        if (opts->use_relative)
        {
            if (r_norm / r_0_norm < opts->tolerance) break;
        }
        else
        {
            if (r_norm < opts->tolerance) break;
        }

        if (isnan(r_norm)) break;

        apply_preconditioner(w, opts, z_k, (const double *const *)r_kp1);

        /* inner_product_flexible (domain.tpp:981-996) */
        theta_k = 0.0;
        for (int k = 0; k < R; k++)
        {
            orc_domain *d = &w->dom[k];
            orc_dom_inner_product_flexible(d->work[0], r_k[k], r_kp1[k], z_k[k], d->num_local_points, d->num_blocks);
            theta_k += orc_block_sum(d->work[0], d->num_blocks);
        }

        beta_k = theta_k / gamma_k;

        for (int k = 0; k < R; k++)
            orc_dom_residual_and_search_update(p_k[k], r_k[k], z_k[k], r_kp1[k], beta_k, w->dom[k].num_local_points);

        num_iterations++;
    }

    free(r_k);
    free(r_kp1);
    free(q_k);
    free(z_k);
    free(p_k);

    if (num_hist) *num_hist = nh;
    return num_iterations;
}

/* domain.tpp:727-914 */
int orc_world_gmres(orc_world *w, double *const *u, const double *const *f, const orc_solver_opts *opts, double *history, int history_cap, int *num_hist)
{
    int R = w->num_ranks;
    int m = opts->num_vectors;
    int nh = 0;

    /* V[m+1], Z[m] per rank (domain.tpp:325-330) */
    double ***V = (double ***)xcalloc((size_t)m + 1, sizeof(double **));
    double ***Z = (double ***)xcalloc((size_t)m, sizeof(double **));
    for (int i = 0; i < m + 1; i++)
    {
        V[i] = (double **)xcalloc((size_t)R, sizeof(double *));
        for (int k = 0; k < R; k++) V[i][k] = (double *)xcalloc((size_t)w->dom[k].num_local_points, sizeof(double));
    }
    for (int i = 0; i < m; i++)
    {
        Z[i] = (double **)xcalloc((size_t)R, sizeof(double *));
        for (int k = 0; k < R; k++) Z[i][k] = (double *)xcalloc((size_t)w->dom[k].num_local_points, sizeof(double));
    }

    double *H = (double *)xcalloc((size_t)m * m, sizeof(double)); /* H[i][j] = H[i*m+j] */
    double *c_gmres = (double *)xcalloc((size_t)m, sizeof(double));
    double *s_gmres = (double *)xcalloc((size_t)m, sizeof(double));
    double *gamma = (double *)xcalloc((size_t)m + 1, sizeof(double));

    double **r_k = (double **)xcalloc((size_t)R, sizeof(double *));
    double **q_k = (double **)xcalloc((size_t)R, sizeof(double *));

    for (int k = 0; k < R; k++)
    {
        r_k[k] = w->dom[k].r_k;
        q_k[k] = w->dom[k].q_k;
        orc_dom_initialize_arrays(u[k], r_k[k], f[k], w->dom[k].num_local_points);
    }

    double r_norm;
    double r_0_norm = orc_world_residual_norm(w, (const double *const *)r_k);
    push_hist(history, history_cap, &nh, r_0_norm);

    int converged = 0;
    int iter = 0;
    int j;

    double alpha_j, beta_j, gamma_j, gamma_k;

    while (iter < opts->max_iterations)
    {
        if (iter > 0)
        {
            orc_world_stiffness(w, r_k, (const double *const *)u, 0);

            for (int k = 0; k < R; k++)
                orc_vector_vector_addition(r_k[k], 1.0, f[k], -1.0, r_k[k], w->dom[k].num_local_points);

            r_norm = orc_world_residual_norm(w, (const double *const *)r_k);
            gamma[0] = r_norm;
        }
        else
        {
            gamma[0] = r_0_norm;
        }

        for (int k = 0; k < R; k++)
            orc_vector_scaling(V[0][k], 1.0 / gamma[0], r_k[k], w->dom[k].num_local_points);

        for (j = 0; j < m; j++)
        {
            apply_preconditioner(w, opts, Z[j], (const double *const *)V[j]);

            orc_world_stiffness(w, q_k, (const double *const *)Z[j], 0);

            /* classical Gram-Schmidt: all H[i][j] first (domain.tpp:810-815) */
            for (int i = 0; i < j + 1; i++)
                H[i * m + j] = orc_world_assembled_inner_product(w, (const double *const *)q_k, (const double *const *)V[i]);

            for (int i = 0; i < j + 1; i++)
                for (int k = 0; k < R; k++)
                    orc_vector_vector_addition(q_k[k], 1.0, q_k[k], -H[i * m + j], V[i][k], w->dom[k].num_local_points);

            /* Givens rotations (domain.tpp:825-830) */
            for (int i = 0; i < j; i++)
            {
                double h_ij = H[i * m + j];
                H[i * m + j] = c_gmres[i] * h_ij + s_gmres[i] * H[(i + 1) * m + j];
                H[(i + 1) * m + j] = -s_gmres[i] * h_ij + c_gmres[i] * H[(i + 1) * m + j];
            }

            alpha_j = orc_world_residual_norm(w, (const double *const *)q_k);

            if (fabs(alpha_j) == 0.0)
            {
                converged = 1;
                break;
            }

            beta_j = sqrt(H[j * m + j] * H[j * m + j] + alpha_j * alpha_j);
            gamma_j = 1.0 / beta_j;
            c_gmres[j] = H[j * m + j] * gamma_j;
            s_gmres[j] = alpha_j * gamma_j;
            H[j * m + j] = beta_j;
            gamma[j + 1] = -s_gmres[j] * gamma[j];
            gamma[j] = c_gmres[j] * gamma[j];

            r_norm = fabs(gamma[j + 1]);
            push_hist(history, history_cap, &nh, r_norm);

            if (opts->use_relative)
            {
                if (r_norm / r_0_norm < opts->tolerance)
                {
                    converged = 1;
                    break;
                }
            }
            else
            {
                if (r_norm < opts->tolerance)
                {
                    converged = 1;
                    break;
                }
            }

            if (iter >= opts->max_iterations)
            {
                converged = 1;
                break;
            }

            if (isnan(r_norm))
            {
                converged = 1;
                break;
            }

            for (int k = 0; k < R; k++)
                orc_vector_scaling(V[j + 1][k], 1.0 / alpha_j, q_k[k], w->dom[k].num_local_points);

            iter++;
        }

        if (j == m) j--;

        /* back substitution into c_gmres (domain.tpp:891-899) */
        for (int k = j; k >= 0; k--)
        {
            gamma_k = gamma[k];

            for (int i = j; i > k; i--) gamma_k -= H[k * m + i] * c_gmres[i];

            c_gmres[k] = gamma_k / H[k * m + k];
        }

        for (int i = 0; i < j + 1; i++)
            for (int k = 0; k < R; k++)
                orc_vector_vector_addition(u[k], 1.0, u[k], c_gmres[i], Z[i][k], w->dom[k].num_local_points);

        if (converged) break;
    }

    for (int i = 0; i < m + 1; i++)
    {
        for (int k = 0; k < R; k++) free(V[i][k]);
        free(V[i]);
    }
    for (int i = 0; i < m; i++)
    {
        for (int k = 0; k < R; k++) free(Z[i][k]);
        free(Z[i]);
    }
    free(V);
    free(Z);
    free(H);
    free(c_gmres);
    free(s_gmres);
    free(gamma);
    free(r_k);
    free(q_k);

    if (num_hist) *num_hist = nh;
    return iter;
}
