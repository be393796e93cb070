/*
 * fdd_oracle_subdomain.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Serial restatement of the Subdomain<double> solve path
 * (subdomain.tpp:3942-4646) on the operators the reference's constructor
 * (subdomain.tpp:86-2747) produces when the subdomain region holds only
 * conforming elements of the rank itself: a single-rank run (configs C1-C3),
 * or the "block-local" region of a multi-rank run (no neighbour rings, no
 * superdomain; SURVEY.md section 8(e) caveat).  In that case:
 *
 *   - subdomain_region = own elements at degree N (subdomain.tpp:468-474);
 *     rings and superdomain are empty (subdomain.tpp:487-553 add nothing);
 *   - dof_num = 1-based rank of glo_num*mask with 0 for Dirichlet points
 *     (ranking lambda, subdomain.tpp:881-918, 1151-1176);
 *   - Q has one 1.0 per non-Dirichlet point (subdomain.tpp:1517-1520), Qt=Q^T;
 *   - Q_int = Qt_int = QQt_int = identity on the dofs (subdomain.tpp:2653-2729);
 *   - norm_weight = 1 on every dof, inner_weight = (Q*norm_weight > 0)
 *     (subdomain.tpp:2731-2747);
 *   - superdomain_operator.A / .Pt are empty, so their multiplies are the
 *     silent no-ops of csr_matrix.tpp:304, 334.
 *
 * The degree tree (restriction_1/2/3 down the levels, subdomain.tpp:4576-4609)
 * and the coarse assembly Qt_coarse (subdomain.tpp:1706-1713, 4639) are still
 * computed, as the reference does on every preconditioner application.
 *
 * See fdd_oracle.h for who may use this file and the parity-pin statement.
 */
#include "fdd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fdd_oracle_priv.h"

void *orc_xcalloc(size_t n, size_t sz);
static void *xcalloc(size_t n, size_t sz) { return orc_xcalloc(n, sz); }

void *orc_xcalloc(size_t n, size_t sz)
{
    void *p = calloc(n ? n : 1, sz);
    if (!p)
    {
        fprintf(stderr, "fdd_oracle: out of memory\n");
        abort();
    }
    return p;
}

typedef struct
{
    unsigned int idx;
    double val;
} rank_entry;

static int rank_cmp_val(const void *a_, const void *b_)
{
    const rank_entry *a = (const rank_entry *)a_, *b = (const rank_entry *)b_;
    if (a->val < b->val) return -1;
    if (a->val > b->val) return 1;
    /* tie-break on index: std::sort is unstable but ties get equal ranks */
    return (a->idx < b->idx) ? -1 : (a->idx > b->idx);
}

static void ranking(double *data, int size) { orc_ranking(data, size); }

/* ranking lambda, subdomain.tpp:881-918: dense ranks, 0 stays 0 */
void orc_ranking(double *data, int size)
{
    if (size == 0) return;

    rank_entry *entries = (rank_entry *)xcalloc((size_t)size, sizeof(rank_entry));

    for (int i = 0; i < size; i++)
    {
        entries[i].idx = (unsigned int)i;
        entries[i].val = data[i];
    }

    qsort(entries, (size_t)size, sizeof(rank_entry), rank_cmp_val);

    double value = entries[0].val;
    double rank = (value == 0.0) ? 0.0 : 1.0;

    entries[0].val = rank;

    for (int i = 1; i < size; i++)
    {
        if (entries[i].val == value)
        {
            entries[i].val = rank;
        }
        else
        {
            rank += 1.0;
            value = entries[i].val;
            entries[i].val = rank;
        }
    }

    for (int i = 0; i < size; i++) data[entries[i].idx] = entries[i].val;

    free(entries);
}

static void csr_identity(orc_csr *A, int n)
{
    int *rows = (int *)xcalloc((size_t)n, sizeof(int));
    double *vals = (double *)xcalloc((size_t)n, sizeof(double));
    for (int i = 0; i < n; i++)
    {
        rows[i] = i;
        vals[i] = 1.0;
    }
    orc_csr_assemble(A, n, n, rows, rows, vals, n);
    free(rows);
    free(vals);
}

orc_subdomain *orc_subdomain_create(int num_levels, const int *poly_degree, const double *const *D_hat, const double *const *J_cf_tables, const orc_mesh *level_meshes)
{
    orc_subdomain *s = (orc_subdomain *)xcalloc(1, sizeof(orc_subdomain));
    const orc_mesh *fine = &level_meshes[0];
    int dim = fine->dim;

    s->dim = dim;
    s->num_levels = num_levels;
    s->poly_degree = (int *)xcalloc((size_t)num_levels, sizeof(int));
    s->levels = (orc_level *)xcalloc((size_t)num_levels, sizeof(orc_level));
    s->D_hat = (double **)xcalloc((size_t)num_levels, sizeof(double *));
    s->J_cf = (double **)xcalloc((size_t)num_levels, sizeof(double *));

    /* subdomain.tpp:112-120 */
    for (int l = 0; l < num_levels; l++)
    {
        int n = poly_degree[l] + 1;
        int nep = (dim == 2) ? n * n : n * n * n;

        s->poly_degree[l] = poly_degree[l];
        s->levels[l].num_elements = level_meshes[l].num_local_elements;
        s->levels[l].num_points = level_meshes[l].num_local_elements * nep;
        s->levels[l].poly_degree = poly_degree[l];
        s->levels[l].offset = (l > 0) ? s->levels[l - 1].offset + s->levels[l - 1].num_points : 0;

        s->D_hat[l] = (double *)xcalloc((size_t)n * n, sizeof(double));
        memcpy(s->D_hat[l], D_hat[l], (size_t)n * n * sizeof(double));

        if (l + 1 < num_levels)
        {
            int n_c = poly_degree[l + 1] + 1;
            s->J_cf[l] = (double *)xcalloc((size_t)n * n_c, sizeof(double));
            memcpy(s->J_cf[l], J_cf_tables[l], (size_t)n * n_c * sizeof(double));
        }
    }

    int n0 = poly_degree[0] + 1;
    int nep0 = (dim == 2) ? n0 * n0 : n0 * n0 * n0;
    int P = fine->num_local_elements * nep0;
    s->num_points = P;

    /* geometry of the region = the rank's own fine-level data (subdomain.tpp:667-699) */
    for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++)
    {
        s->geom_fact[g] = (double *)xcalloc((size_t)P, sizeof(double));
        if (fine->g[g]) memcpy(s->geom_fact[g], fine->g[g], (size_t)P * sizeof(double));
    }

    /* dof numbering: subdomain.tpp:1151-1176 with a conforming single-degree region */
    double *tmp = (double *)xcalloc((size_t)P, sizeof(double));
    for (int p = 0; p < P; p++) tmp[p] = (double)(fine->glo_num[p]);
    ranking(tmp, P);
    for (int p = 0; p < P; p++) tmp[p] = tmp[p] * fine->p_mask[p];
    ranking(tmp, P);

    int num_dofs = 0;
    for (int p = 0; p < P; p++)
        if ((int)tmp[p] > num_dofs) num_dofs = (int)tmp[p];

    /* Q: subdomain.tpp:1510-1520 */
    int *rows = (int *)xcalloc((size_t)P, sizeof(int));
    int *cols = (int *)xcalloc((size_t)P, sizeof(int));
    double *vals = (double *)xcalloc((size_t)P, sizeof(double));
    long ne = 0;
    for (int p = 0; p < P; p++)
    {
        if (tmp[p] > 0.0)
        {
            rows[ne] = p;
            cols[ne] = (int)tmp[p] - 1;
            vals[ne] = 1.0;
            ne++;
        }
    }
    orc_csr_assemble(&s->Q, P, num_dofs, rows, cols, vals, ne);
    orc_csr_transpose(&s->Q, &s->Qt); /* subdomain.tpp:1584 */
    free(rows);
    free(cols);
    free(vals);
    free(tmp);

    s->num_dofs = num_dofs;          /* subdomain.tpp:1591-1598 */
    s->num_extended_dofs = num_dofs; /* subdomain.tpp:1601 (= Q.num_cols) */
    s->sup_num_extended_dofs = 0;

    /* per-point indirection arrays: subdomain.tpp:1603-1630 */
    s->offset = (int *)xcalloc((size_t)P, sizeof(int));
    s->vertex = (int *)xcalloc((size_t)P, sizeof(int));
    s->level = (int *)xcalloc((size_t)P, sizeof(int));
    for (int e = 0; e < fine->num_local_elements; e++)
    {
        for (int v = 0; v < nep0; v++)
        {
            s->offset[e * nep0 + v] = e * nep0;
            s->vertex[e * nep0 + v] = v;
            s->level[e * nep0 + v] = 0;
        }
    }

    /* Qt_coarse: subdomain.tpp:1653-1713 (coarsest level, degree 1 => 2^dim vertices) */
    {
        const orc_mesh *coarse = &level_meshes[num_levels - 1];
        int nc = poly_degree[num_levels - 1] + 1;
        int nv = (dim == 2) ? nc * nc : nc * nc * nc;
        int size = coarse->num_local_elements * nv;
        double *dof = (double *)xcalloc((size_t)size, sizeof(double));

        for (int i = 0; i < size; i++)
            dof[i] = (coarse->p_mask[i] > 0.0) ? (double)(coarse->glo_num[i]) : 0.0;

        ranking(dof, size);

        int num_coarse_dofs = 0;
        for (int i = 0; i < size; i++)
            if ((int)dof[i] > num_coarse_dofs) num_coarse_dofs = (int)dof[i];

        int *r2 = (int *)xcalloc((size_t)size, sizeof(int));
        int *c2 = (int *)xcalloc((size_t)size, sizeof(int));
        double *v2 = (double *)xcalloc((size_t)size, sizeof(double));
        long n2 = 0;
        for (int i = 0; i < size; i++)
        {
            if (dof[i] > 0.0)
            {
                r2[n2] = (int)dof[i] - 1;
                c2[n2] = i;
                v2[n2] = 1.0;
                n2++;
            }
        }
        orc_csr_assemble(&s->Qt_coarse, num_coarse_dofs, size, r2, c2, v2, n2);
        free(r2);
        free(c2);
        free(v2);
        free(dof);
    }

    /* interface maps: subdomain.tpp:2653-2729 reduce to identities */
    csr_identity(&s->Q_int, num_dofs);
    csr_identity(&s->Qt_int, num_dofs);
    csr_identity(&s->QQt_int, num_dofs);

    /* weights: subdomain.tpp:2731-2747 */
    s->norm_weight = (double *)xcalloc((size_t)num_dofs, sizeof(double));
    for (int i = 0; i < num_dofs; i++) s->norm_weight[i] = 1.0;

    s->num_values = s->num_points + s->sup_num_extended_dofs; /* subdomain.tpp:3858 */
    s->num_blocks = (s->num_values + ORC_BLOCK_SIZE - 1) / ORC_BLOCK_SIZE;

    s->inner_weight = (double *)xcalloc((size_t)s->num_values, sizeof(double));
    orc_csr_multiply(s->inner_weight, s->Q.ptr, s->Q.col, s->Q.val, s->norm_weight, s->Q.num_rows);
    for (int i = 0; i < s->num_values; i++)
        if (s->inner_weight[i] > 0.0) s->inner_weight[i] = 1.0;

    /* work arrays hold the whole tree (subdomain.tpp:588-595) */
    size_t tree = (size_t)s->levels[num_levels - 1].offset + (size_t)s->levels[num_levels - 1].num_points;
    s->own_points = s->levels[0].num_points;
    s->num_unique_dofs = num_dofs;
    orc_subdomain_alloc_solver(s, tree + (size_t)P + 16);

    return s;
}

/* solver vectors (subdomain.tpp:3860-3873) and work arrays */
void orc_subdomain_alloc_solver(orc_subdomain *s, size_t wsize)
{
    for (int w = 0; w < 3; w++) s->work[w] = (double *)xcalloc(wsize, sizeof(double));

    size_t nv = (size_t)s->num_values;
    s->f = (double *)xcalloc(nv, sizeof(double));
    s->u_k = (double *)xcalloc(nv, sizeof(double));
    s->r_k = (double *)xcalloc(nv, sizeof(double));
    s->r_kp1 = (double *)xcalloc(nv, sizeof(double));
    s->q_k = (double *)xcalloc(nv, sizeof(double));
    s->z_k = (double *)xcalloc(nv, sizeof(double));
    s->p_k = (double *)xcalloc(nv, sizeof(double));
    s->cap_vectors = 0;
    s->V = NULL;
    s->Z = NULL;
}

void orc_subdomain_destroy(orc_subdomain *s)
{
    if (!s) return;
    for (int l = 0; l < s->num_levels; l++)
    {
        free(s->D_hat[l]);
        free(s->J_cf[l]);
    }
    free(s->D_hat);
    free(s->J_cf);
    free(s->poly_degree);
    free(s->levels);
    orc_csr_free(&s->Q);
    orc_csr_free(&s->Qt);
    orc_csr_free(&s->Qt_coarse);
    orc_csr_free(&s->Q_int);
    orc_csr_free(&s->Qt_int);
    orc_csr_free(&s->QQt_int);
    orc_csr_free(&s->sup_A);
    orc_csr_free(&s->sup_Pt);
    for (int g = 0; g < ORC_NUM_GEOM_FACTS; g++) free(s->geom_fact[g]);
    free(s->offset);
    free(s->vertex);
    free(s->level);
    free(s->norm_weight);
    free(s->inner_weight);
    free(s->jacobi_dinv);
    for (int w = 0; w < 3; w++) free(s->work[w]);
    free(s->f);
    free(s->u_k);
    free(s->r_k);
    free(s->r_kp1);
    free(s->q_k);
    free(s->z_k);
    free(s->p_k);
    if (s->V)
        for (int i = 0; i < s->cap_vectors + 1; i++) free(s->V[i]);
    if (s->Z)
        for (int i = 0; i < s->cap_vectors; i++) free(s->Z[i]);
    free(s->V);
    free(s->Z);
    free(s);
}

int orc_subdomain_num_values(const orc_subdomain *s) { return s->num_values; }
int orc_subdomain_num_dofs(const orc_subdomain *s) { return s->num_dofs; }

/* subdomain.tpp:4566-4646 */
void orc_subdomain_tree_operator(orc_subdomain *s, double *Tu, const double *u)
{
    int dim = s->dim;

    if (s->tree_done)
    {
        /* composite region: the tree was built and exchanged for all ranks at once (orc_fdd_tree_operator) */
        if (Tu != s->f) memcpy(Tu, s->f, (size_t)s->num_values * sizeof(double));
        return;
    }

    /* fill up tree: cast copy of the outer vector (subdomain.tpp:4571) */
    orc_sub_copy_f64_f64(s->work[0], u, s->levels[0].num_points);

    for (int l = 0; l < s->num_levels - 1; l++)
    {
        int n_f = s->levels[l].poly_degree + 1;
        int n_c = s->levels[l + 1].poly_degree + 1;
        const double *J = s->J_cf[l];
        double *u_f = s->work[0] + s->levels[l].offset;
        double *u_c = s->work[0] + s->levels[l + 1].offset;
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

    /* tree exchange: the gs pull of the region's own elements is a copy of
     * the level-0 slice (subdomain.tpp:4626-4630 with ids -k matching +k on
     * the same rank) */
    memcpy(Tu, s->work[0], (size_t)s->num_points * sizeof(double));

    /* coarse level assembled with Qt_coarse (subdomain.tpp:4635-4639); Pt is
     * empty, so nothing reaches the (empty) tail (subdomain.tpp:4643-4644) */
    {
        const double *coarse = s->work[0] + s->levels[s->num_levels - 1].offset;
        memcpy(s->work[2], coarse, (size_t)s->Qt_coarse.num_cols * sizeof(double));
        orc_csr_multiply(s->work[1], s->Qt_coarse.ptr, s->Qt_coarse.col, s->Qt_coarse.val, s->work[2], s->Qt_coarse.num_rows);
    }
}

/* subdomain.tpp:3942-3967 */
void orc_subdomain_stiffness(orc_subdomain *s, double *Au, const double *u)
{
    double *GDu[3] = {s->work[0], s->work[1], s->work[2]};
    const double *G[6] = {s->geom_fact[0], s->geom_fact[1], s->geom_fact[2], s->geom_fact[3], s->geom_fact[4], s->geom_fact[5]};

    /* superdomain_operator.A.multiply on the tail (subdomain.tpp:3951); an empty matrix is a no-op */
    if (s->sup_num_extended_dofs > 0 && s->sup_A.num_nnz > 0)
        orc_csr_multiply(Au + s->num_points, s->sup_A.ptr, s->sup_A.col, s->sup_A.val, u + s->num_points, s->sup_A.num_rows);

    orc_sub_stiffness_matrix_1(GDu, u, (const double *const *)s->D_hat, s->offset, s->vertex, s->level, s->poly_degree, G, s->num_points, s->dim);
    orc_sub_stiffness_matrix_2(Au, (const double *const *)GDu, (const double *const *)s->D_hat, s->offset, s->vertex, s->level, s->poly_degree, s->num_points, s->dim);
}

/* subdomain.tpp:3969-3985 */
void orc_subdomain_dssum(orc_subdomain *s, double *QQtu, const double *u)
{
    orc_csr_multiply(s->work[0], s->Qt.ptr, s->Qt.col, s->Qt.val, u, s->Qt.num_rows);
    memcpy(s->work[0] + s->num_extended_dofs, u + s->num_points, (size_t)s->sup_num_extended_dofs * sizeof(double)); /* :3977 */
    orc_csr_multiply(s->work[1], s->QQt_int.ptr, s->QQt_int.col, s->QQt_int.val, s->work[0], s->QQt_int.num_rows);
    orc_csr_multiply(QQtu, s->Q.ptr, s->Q.col, s->Q.val, s->work[1], s->Q.num_rows);
    memcpy(QQtu + s->num_points, s->work[1] + s->num_extended_dofs, (size_t)s->sup_num_extended_dofs * sizeof(double)); /* :3984 */
}

/* subdomain.tpp:4491-4515 */
double orc_subdomain_residual_norm(orc_subdomain *s, const double *r)
{
    orc_csr_multiply_weight(s->work[1], s->Qt.ptr, s->Qt.col, s->Qt.val, r, s->norm_weight, s->Qt.num_rows);
    memcpy(s->work[1] + s->num_extended_dofs, r + s->num_points, (size_t)s->sup_num_extended_dofs * sizeof(double)); /* :4501 */

    int num_values = s->num_extended_dofs + s->sup_num_extended_dofs;
    int num_blocks = (num_values + ORC_BLOCK_SIZE - 1) / ORC_BLOCK_SIZE;

    orc_sub_weighted_inner_product(s->work[0], s->work[1], s->work[1], s->norm_weight, num_values, num_blocks);

    return sqrt(orc_block_sum(s->work[0], num_blocks));
}

/* subdomain.tpp:4277-4307 (scribbles on p_k as the reference does, :4301) */
static double assembled_inner_product(orc_subdomain *s, const double *u, const double *v)
{
    orc_csr_multiply_weight(s->work[0], s->Qt.ptr, s->Qt.col, s->Qt.val, u, s->norm_weight, s->Qt.num_rows);
    memcpy(s->work[0] + s->num_extended_dofs, u + s->num_points, (size_t)s->sup_num_extended_dofs * sizeof(double)); /* :4286 */
    orc_csr_multiply_weight(s->work[1], s->Qt.ptr, s->Qt.col, s->Qt.val, v, s->norm_weight, s->Qt.num_rows);
    memcpy(s->work[1] + s->num_extended_dofs, v + s->num_points, (size_t)s->sup_num_extended_dofs * sizeof(double)); /* :4294 */

    int num_values = s->num_extended_dofs + s->sup_num_extended_dofs;
    int num_blocks = (num_values + ORC_BLOCK_SIZE - 1) / ORC_BLOCK_SIZE;

    orc_sub_weighted_inner_product(s->p_k, s->work[0], s->work[1], s->norm_weight, num_values, num_blocks);

    return orc_block_sum(s->p_k, num_blocks);
}

void orc_subdomain_attach_amg(orc_subdomain *s, orc_amg *amg) { s->amg = amg; }

/* dof of every level-0 point, -1 where the point has none (the rows of the boolean Q, subdomain.tpp:1588-1601) */
void orc_subdomain_point_dofs(const orc_subdomain *s, int *dof)
{
    for (int p = 0; p < s->num_points; p++) dof[p] = (s->Q.ptr[p + 1] > s->Q.ptr[p]) ? s->Q.col[s->Q.ptr[p]] : -1;
}

/* subdomain.tpp:3987-4159 on the conforming composite: Qt_int = Q_int = I, empty tail */
void orc_subdomain_low_order_preconditioner(orc_subdomain *s, double *z, const double *r)
{
    if (!s->amg || orc_amg_level_size(s->amg, 0) != s->num_unique_dofs)
    {
        fprintf(stderr, "fdd_oracle: low_order_preconditioner needs an attached AMG hierarchy over the %d dofs\n", s->num_unique_dofs);
        abort();
    }
    orc_csr_multiply(s->work[0], s->Qt.ptr, s->Qt.col, s->Qt.val, r, s->Qt.num_rows);                        /* :3996 */
    memcpy(s->work[0] + s->num_extended_dofs, r + s->num_points, (size_t)s->sup_num_extended_dofs * sizeof(double)); /* :4000 */
    orc_csr_multiply(s->work[1], s->Qt_int.ptr, s->Qt_int.col, s->Qt_int.val, s->work[0], s->Qt_int.num_rows); /* :4004 */
    orc_amg_vcycle(s->amg, s->work[2], s->work[1]);                                                            /* :4008-4142 */
    orc_csr_multiply(s->work[0], s->Q_int.ptr, s->Q_int.col, s->Q_int.val, s->work[2], s->Q_int.num_rows);     /* :4146 */
    orc_csr_multiply(z, s->Q.ptr, s->Q.col, s->Q.val, s->work[0], s->Q.num_rows);                              /* :4153 */
    memcpy(z + s->num_points, s->work[0] + s->num_extended_dofs, (size_t)s->sup_num_extended_dofs * sizeof(double)); /* :4157 */
}

/* ------------------------------------------------------------------------------------------------------------------
 * Point-Jacobi in the preconditioner slot (a labelled option of the build under test, not in the reference):
 * z = Q Q_int D^-1 (Qt_int Qt r), D = diag(Qt_int Qt' A Q' Q_int) over the unique dofs, with Q' = [Q 0; 0 I] and
 * A = [A_L 0; 0 A_sup] the reference's operators on composite vectors (subdomain.tpp:3942-3985).  The diagonal is
 * formed from those operators as they stand: entry u = (row of the dof's representative, Qt_int) . (column of the
 * dof's copies, Q_int); the element part element by element with the restated kernels of subdomain.okl:4-101.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Au = A_e u on the np points of one element starting at point `first` (element-local vectors) */
static void element_apply(const orc_subdomain *s, int first, int np, double *Au, const double *u, double *const GDu[3], const int *zero_offset)
{
    const double *G[6];
    for (int g = 0; g < 6; g++) G[g] = s->geom_fact[g] + first;
    orc_sub_stiffness_matrix_1(GDu, u, (const double *const *)s->D_hat, zero_offset, s->vertex + first, s->level + first, s->poly_degree, G, np, s->dim);
    orc_sub_stiffness_matrix_2(Au, (const double *const *)GDu, (const double *const *)s->D_hat, zero_offset, s->vertex + first, s->level + first, s->poly_degree, np, s->dim);
}

/* coefficient of u(v) in (A_e u)(v), read off subdomain.okl:4-101 */
static double element_diagonal_entry(const orc_subdomain *s, int first, int v, int n)
{
    const double *D = s->D_hat[s->level[first]];
    const double *const *G = (const double *const *)s->geom_fact;
    if (s->dim == 2)
    {
        const int i = v % n, j = v / n;
        double a = 0.0;
        for (int p = 0; p < n; p++) a += D[i + p * n] * D[i + p * n] * G[0][first + p + j * n] + D[j + p * n] * D[j + p * n] * G[1][first + i + p * n];
        return a + 2.0 * D[i + i * n] * D[j + j * n] * G[2][first + v];
    }
    const int nn = n * n, i = v % n, j = (v / n) % n, k = v / nn;
    double a = 0.0;
    for (int p = 0; p < n; p++)
        a += D[i + p * n] * D[i + p * n] * G[0][first + p + j * n + k * nn] + D[j + p * n] * D[j + p * n] * G[1][first + i + p * n + k * nn] + D[k + p * n] * D[k + p * n] * G[2][first + i + j * n + p * nn];
    const double di = D[i + i * n], dj = D[j + j * n], dk = D[k + k * n];
    return a + 2.0 * (di * dj * G[3][first + v] + di * dk * G[4][first + v] + dj * dk * G[5][first + v]);
}

/* test hook: the closed form above against the kernels applied to a unit vector */
double orc_subdomain_element_diagonal_check(const orc_subdomain *s)
{
    double worst = 0.0;
    int first = 0;
    while (first < s->num_points)
    {
        const int n = s->poly_degree[s->level[first]] + 1;
        int np = 1;
        for (int d = 0; d < s->dim; d++) np *= n;
        double *u = (double *)xcalloc((size_t)np, sizeof(double)), *Au = (double *)xcalloc((size_t)np, sizeof(double));
        double *GDu[3] = {(double *)xcalloc((size_t)np, sizeof(double)), (double *)xcalloc((size_t)np, sizeof(double)), (double *)xcalloc((size_t)np, sizeof(double))};
        int *zero = (int *)xcalloc((size_t)np, sizeof(int));
        for (int v = 0; v < np; v += (np > 64 ? 7 : 1))
        {
            u[v] = 1.0;
            element_apply(s, first, np, Au, u, GDu, zero);
            u[v] = 0.0;
            const double a = element_diagonal_entry(s, first, v, n);
            const double err = fabs(a - Au[v]) / fabs(Au[v]);
            if (err > worst) worst = err;
        }
        free(u);
        free(Au);
        for (int g = 0; g < 3; g++) free(GDu[g]);
        free(zero);
        first += np;
    }
    return worst;
}

typedef struct
{
    int dof, point;
    double weight;
} jac_entry;

static int jac_cmp(const void *a_, const void *b_)
{
    const jac_entry *a = (const jac_entry *)a_, *b = (const jac_entry *)b_;
    if (a->dof != b->dof) return a->dof < b->dof ? -1 : 1;
    return a->point - b->point;
}

/* diag[u], u < num_unique_dofs */
void orc_subdomain_jacobi_diagonal(const orc_subdomain *s, double *diag)
{
    const int nse = s->num_extended_dofs, nu = s->num_unique_dofs;
    /* representative (row of Qt_int) of every unique dof, and the unique dof every extended index is a copy of (Q_int) */
    int *rep = (int *)xcalloc((size_t)nu, sizeof(int));
    int *uniq_of = (int *)xcalloc((size_t)(nse + s->sup_num_extended_dofs) + 1, sizeof(int));
    for (int u = 0; u < nu; u++) rep[u] = s->Qt_int.col[s->Qt_int.ptr[u]];
    for (int i = 0; i < s->Q_int.num_rows; i++) uniq_of[i] = (s->Q_int.ptr[i + 1] > s->Q_int.ptr[i]) ? s->Q_int.col[s->Q_int.ptr[i]] : -1;
    for (int u = 0; u < nu; u++) diag[u] = 0.0;

    /* element part: rows of representatives below nse.  left[e] = Q e_rep, right[e] = Q (copies of the same unique dof) */
    int first = 0;
    while (first < s->num_points)
    {
        const int n = s->poly_degree[s->level[first]] + 1;
        int np = 1;
        for (int d = 0; d < s->dim; d++) np *= n;
        int cap = 0;
        for (int v = 0; v < np; v++) cap += s->Q.ptr[first + v + 1] - s->Q.ptr[first + v];
        jac_entry *ent = (jac_entry *)xcalloc((size_t)cap + 1, sizeof(jac_entry));
        int ne = 0, simple = 1;
        for (int v = 0; v < np; v++)
            for (int t = s->Q.ptr[first + v]; t < s->Q.ptr[first + v + 1]; t++)
            {
                const int u = uniq_of[s->Q.col[t]];
                if (u < 0) continue;
                ent[ne].dof = u;
                ent[ne].point = v;
                ent[ne].weight = s->Q.val[t];
                if (s->Q.col[t] != rep[u] || s->Q.val[t] != 1.0) simple = 0;
                ne++;
            }
        qsort(ent, (size_t)ne, sizeof(jac_entry), jac_cmp);
        for (int b = 1; b < ne && simple; b++)
            if (ent[b].dof == ent[b - 1].dof) simple = 0;
        if (simple)
        {
            for (int b = 0; b < ne; b++) diag[ent[b].dof] += element_diagonal_entry(s, first, ent[b].point, n);
        }
        else
        {
            double *wl = (double *)xcalloc((size_t)np, sizeof(double)), *wr = (double *)xcalloc((size_t)np, sizeof(double)), *Aw = (double *)xcalloc((size_t)np, sizeof(double));
            double *GDu[3] = {(double *)xcalloc((size_t)np, sizeof(double)), (double *)xcalloc((size_t)np, sizeof(double)), (double *)xcalloc((size_t)np, sizeof(double))};
            int *zero = (int *)xcalloc((size_t)np, sizeof(int));
            for (int b = 0; b < ne;)
            {
                int e2 = b + 1;
                while (e2 < ne && ent[e2].dof == ent[b].dof) e2++;
                const int u = ent[b].dof;
                if (rep[u] < nse)
                {
                    /* right: all copies; left: the representative's entries only -- recover them from Q */
                    for (int v = 0; v < np; v++) wl[v] = wr[v] = 0.0;
                    for (int t = b; t < e2; t++) wr[ent[t].point] += ent[t].weight;
                    int any_left = 0;
                    for (int v = 0; v < np; v++)
                        for (int t = s->Q.ptr[first + v]; t < s->Q.ptr[first + v + 1]; t++)
                            if (s->Q.col[t] == rep[u])
                            {
                                wl[v] += s->Q.val[t];
                                any_left = 1;
                            }
                    if (any_left)
                    {
                        element_apply(s, first, np, Aw, wr, GDu, zero);
                        double acc = 0.0;
                        for (int v = 0; v < np; v++) acc += wl[v] * Aw[v];
                        diag[u] += acc;
                    }
                }
                b = e2;
            }
            free(wl);
            free(wr);
            free(Aw);
            for (int g = 0; g < 3; g++) free(GDu[g]);
            free(zero);
        }
        free(ent);
        first += np;
    }

    /* superdomain part: the dofs whose representative is a superdomain entry take that row of A against their copies there */
    for (int u = 0; u < nu; u++)
    {
        if (rep[u] < nse) continue;
        const int row = rep[u] - nse;
        double acc = 0.0;
        for (int t = s->sup_A.ptr[row]; t < s->sup_A.ptr[row + 1]; t++)
            if (uniq_of[nse + s->sup_A.col[t]] == u) acc += s->sup_A.val[t];
        diag[u] = acc; /* the row of the representative alone (Qt_int picks one row); element sums of its subdomain copies are not its row */
    }
    free(rep);
    free(uniq_of);
}

static void jacobi_preconditioner(orc_subdomain *s, double *z, const double *r)
{
    if (!s->jacobi_dinv)
    {
        s->jacobi_dinv = (double *)xcalloc((size_t)s->num_unique_dofs + 1, sizeof(double));
        orc_subdomain_jacobi_diagonal(s, s->jacobi_dinv);
        for (int u = 0; u < s->num_unique_dofs; u++) s->jacobi_dinv[u] = 1.0 / s->jacobi_dinv[u];
    }
    orc_csr_multiply(s->work[0], s->Qt.ptr, s->Qt.col, s->Qt.val, r, s->Qt.num_rows);
    memcpy(s->work[0] + s->num_extended_dofs, r + s->num_points, (size_t)s->sup_num_extended_dofs * sizeof(double));
    orc_csr_multiply(s->work[1], s->Qt_int.ptr, s->Qt_int.col, s->Qt_int.val, s->work[0], s->Qt_int.num_rows);
    orc_amg_vector_multiplication(s->work[2], s->jacobi_dinv, s->work[1], s->num_unique_dofs);
    orc_csr_multiply(s->work[0], s->Q_int.ptr, s->Q_int.col, s->Q_int.val, s->work[2], s->Q_int.num_rows);
    orc_csr_multiply(z, s->Q.ptr, s->Q.col, s->Q.val, s->work[0], s->Q.num_rows);
    memcpy(z + s->num_points, s->work[0] + s->num_extended_dofs, (size_t)s->sup_num_extended_dofs * sizeof(double));
}

static void apply_inner_preconditioner(orc_subdomain *s, const orc_subdomain_opts *opts, double *z, const double *r)
{
    if (opts->use_preconditioner == 2)
        jacobi_preconditioner(s, z, r);
    else if (opts->use_preconditioner)
        orc_subdomain_low_order_preconditioner(s, z, r);
    else
        orc_subdomain_dssum(s, z, r);
}

static void push_hist(double *history, int cap, int *n, double v)
{
    if (history && *n < cap) history[*n] = v;
    (*n)++;
}

static void ensure_vectors(orc_subdomain *s, int m)
{
    if (s->cap_vectors >= m) return;

    if (s->V)
        for (int i = 0; i < s->cap_vectors + 1; i++) free(s->V[i]);
    if (s->Z)
        for (int i = 0; i < s->cap_vectors; i++) free(s->Z[i]);
    free(s->V);
    free(s->Z);

    s->V = (double **)xcalloc((size_t)m + 1, sizeof(double *));
    s->Z = (double **)xcalloc((size_t)m, sizeof(double *));
    for (int i = 0; i < m + 1; i++) s->V[i] = (double *)xcalloc((size_t)s->num_values, sizeof(double));
    for (int i = 0; i < m; i++) s->Z[i] = (double *)xcalloc((size_t)s->num_values, sizeof(double));
    s->cap_vectors = m;
}

/* subdomain.tpp:4309-4489.  use_relative = false as at the call sites
 * (domain.tpp:642, 702, 792 pass no flags; defaults subdomain.hpp:248). */
int orc_subdomain_gmres(orc_subdomain *s, double *u_l, const double *f_l, const orc_subdomain_opts *opts, double *history, int history_cap, int *num_hist)
{
    int m = opts->num_vectors;
    int nv = s->num_values;
    int nh = 0;

    ensure_vectors(s, m);

    double *H = (double *)xcalloc((size_t)m * m, sizeof(double));
    double *c_gmres = (double *)xcalloc((size_t)m, sizeof(double));
    double *s_gmres = (double *)xcalloc((size_t)m, sizeof(double));
    double *gamma = (double *)xcalloc((size_t)m + 1, sizeof(double));

    orc_subdomain_tree_operator(s, s->f, f_l);

    orc_sub_initialize_arrays(s->u_k, s->r_k, s->f, nv);

    double r_norm;
    double r_0_norm = orc_subdomain_residual_norm(s, s->r_k);
    push_hist(history, history_cap, &nh, r_0_norm);

    int converged = 0;
    int iter = 0;
    int j;
    double alpha_j, beta_j, gamma_j, gamma_k;

    while (iter < opts->max_iterations)
    {
        if (iter > 0)
        {
            orc_subdomain_stiffness(s, s->r_k, s->u_k);
            orc_vector_vector_addition(s->r_k, 1.0, s->f, -1.0, s->r_k, nv);
            r_norm = orc_subdomain_residual_norm(s, s->r_k);
            gamma[0] = r_norm;
        }
        else
        {
            gamma[0] = r_0_norm;
        }

        orc_vector_scaling(s->V[0], 1.0 / gamma[0], s->r_k, nv);

        for (j = 0; j < m; j++)
        {
            iter++;

            /* subdomain.tpp:4373-4382 */
            apply_inner_preconditioner(s, opts, s->Z[j], s->V[j]);

            orc_subdomain_stiffness(s, s->q_k, s->Z[j]);

            for (int i = 0; i < j + 1; i++) H[i * m + j] = assembled_inner_product(s, s->q_k, s->V[i]);

            for (int i = 0; i < j + 1; i++) orc_vector_vector_addition(s->q_k, 1.0, s->q_k, -H[i * m + j], s->V[i], nv);

            for (int i = 0; i < j; i++)
            {
                double h_ij = H[i * m + j];
                H[i * m + j] = c_gmres[i] * h_ij + s_gmres[i] * H[(i + 1) * m + j];
                H[(i + 1) * m + j] = -s_gmres[i] * h_ij + c_gmres[i] * H[(i + 1) * m + j];
            }

            alpha_j = orc_subdomain_residual_norm(s, s->q_k);

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

            if (r_norm < opts->tolerance)
            {
                converged = 1;
                break;
            }

            if (iter >= opts->max_iterations)
            {
                converged = 1;
                break;
            }

            orc_vector_scaling(s->V[j + 1], 1.0 / alpha_j, s->q_k, nv);
        }

        if (j == m) j--;

        for (int k = j; k >= 0; k--)
        {
            gamma_k = gamma[k];
            for (int i = j; i > k; i--) gamma_k -= H[k * m + i] * c_gmres[i];
            c_gmres[k] = gamma_k / H[k * m + k];
        }

        for (int i = 0; i < j + 1; i++) orc_vector_vector_addition(s->u_k, 1.0, s->u_k, c_gmres[i], s->Z[i], nv);

        if (converged) break;
    }

    orc_sub_copy_f64_f64(u_l, s->u_k, s->levels[0].num_points); /* subdomain.tpp:4485 */

    free(H);
    free(c_gmres);
    free(s_gmres);
    free(gamma);

    if (num_hist) *num_hist = nh;
    return iter;
}

/* subdomain.tpp:4161-4268 with use_preconditioner == false */
int orc_subdomain_fcg(orc_subdomain *s, double *u_l, const double *f_l, const orc_subdomain_opts *opts, double *history, int history_cap, int *num_hist)
{
    int nv = s->num_values;
    int nb = s->num_blocks;
    int nh = 0;

    orc_subdomain_tree_operator(s, s->r_k, f_l);

    orc_set_to_value(s->u_k, 0.0, nv, 0);

    double r_norm;
    double r_0_norm = orc_subdomain_residual_norm(s, s->r_k);
    push_hist(history, history_cap, &nh, r_0_norm);

    double alpha_k, beta_k, gamma_k, theta_k;

    apply_inner_preconditioner(s, opts, s->z_k, s->r_k); /* subdomain.tpp:4190-4193 */
    memcpy(s->p_k, s->z_k, (size_t)nv * sizeof(double));

    int iter = 0;

    while (iter < opts->max_iterations)
    {
        orc_subdomain_stiffness(s, s->q_k, s->p_k);

        /* projection_inner_products (subdomain.tpp:4517-4535) */
        orc_sub_projection_inner_products(s->work[0], s->z_k, s->r_k, s->p_k, s->q_k, s->inner_weight, nv, nb);
        gamma_k = 0.0;
        theta_k = 0.0;
        for (int b = 0; b < nb; b++)
        {
            gamma_k += s->work[0][b];
            theta_k += s->work[0][b + nb];
        }

        alpha_k = gamma_k / theta_k;

        orc_sub_solution_and_residual_update(s->u_k, s->r_kp1, s->r_k, s->p_k, s->q_k, alpha_k, nv);

        r_norm = orc_subdomain_residual_norm(s, s->r_kp1);

        iter++;
        push_hist(history, history_cap, &nh, r_norm);

        if (r_norm < opts->tolerance) break;
        if (iter == opts->max_iterations) break;

        apply_inner_preconditioner(s, opts, s->z_k, s->r_kp1); /* subdomain.tpp:4245-4248 */

        /* search_update_inner_product (subdomain.tpp:4544-4557) */
        orc_sub_search_update_inner_product(s->work[0], s->r_k, s->r_kp1, s->z_k, s->inner_weight, nv, nb);
        theta_k = orc_block_sum(s->work[0], nb);

        beta_k = theta_k / gamma_k;

        orc_sub_residual_and_search_update(s->p_k, s->r_k, s->z_k, s->r_kp1, beta_k, nv);
    }

    orc_sub_copy_f64_f64(u_l, s->u_k, s->levels[0].num_points); /* subdomain.tpp:4266 */

    if (num_hist) *num_hist = nh;
    return iter;
}
