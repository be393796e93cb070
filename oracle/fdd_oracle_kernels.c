/*
 * fdd_oracle_kernels.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Serial restatement of every device kernel on the hot path, keeping the
 * reference's per-thread arithmetic order.  See fdd_oracle.h for the rules
 * about who may use this file and for the parity-pin statement.
 */
#include "fdd_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ====================================================================== */
/* csr_matrix.okl                                                          */
/* ====================================================================== */

/* csr_matrix.okl:5-18 */
void orc_csr_multiply(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int n)
{
    for (int i = 0; i < n; i++)
    {
        double Au_i = 0.0;

        for (int j = A_ptr[i]; j < A_ptr[i + 1]; j++)
            Au_i += A_val[j] * u[A_col[j]];

        Au[i] = Au_i;
    }
}

/* csr_matrix.okl:20-33 (row_end is INCLUSIVE) */
void orc_csr_multiply_range(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int row_start, int row_end)
{
    for (int i = row_start; i <= row_end; i++)
    {
        double Au_i = 0.0;

        for (int j = A_ptr[i]; j < A_ptr[i + 1]; j++)
            Au_i += A_val[j] * u[A_col[j]];

        Au[i] = Au_i;
    }
}

/* csr_matrix.okl:35-48 */
void orc_csr_multiply_weight(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *weight, int n)
{
    for (int i = 0; i < n; i++)
    {
        double Au_i = 0.0;

        for (int j = A_ptr[i]; j < A_ptr[i + 1]; j++)
            Au_i += A_val[j] * u[A_col[j]];

        Au[i] = Au_i * weight[i];
    }
}

/* ====================================================================== */
/* math.okl                                                                */
/* ====================================================================== */

/* math.okl:5-11 */
void orc_set_to_value(double *u, double alpha, int n, int offset)
{
    for (int i = 0; i < n; i++) u[i + offset] = alpha;
}

/* math.okl:13-19 */
void orc_invert_vector_elements(double *u, int n)
{
    for (int i = 0; i < n; i++) u[i] = 1.0 / u[i];
}

/* math.okl:21-27 (uv may alias u or v: domain.tpp:767) */
void orc_vector_vector_addition(double *uv, double alpha, const double *u, double beta, const double *v, int n)
{
    for (int i = 0; i < n; i++) uv[i] = alpha * u[i] + beta * v[i];
}

/* math.okl:29-35 */
void orc_vector_scaling(double *au, double alpha, const double *u, int n)
{
    for (int i = 0; i < n; i++) au[i] = alpha * u[i];
}

/* ====================================================================== */
/* Block reduction helper: the 128-wide @shared tree of domain.okl:125-131 */
/* ====================================================================== */
static double tree_reduce(double *s)
{
    for (int alive = ((ORC_BLOCK_SIZE + 1) / 2); 0 < alive; alive /= 2)
        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
            if (item < alive) s[item] += s[item + alive];

    return s[0];
}

/* domain.tpp:926, 943, 960-964, 991-992; subdomain.tpp:4306, 4512, 4530-4534, 4556 */
double orc_block_sum(const double *block, int num_blocks)
{
    double s = 0.0;
    for (int b = 0; b < num_blocks; b++) s += block[b];
    return s;
}

/* ====================================================================== */
/* domain.okl                                                              */
/* ====================================================================== */

/* domain.okl:5-52 */
void orc_dom_stiffness_matrix_1(double *const GDu[3], const double *u, const double *D_hat, const double *const G[6], int num_points, int poly_degree, int dim)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        int n_x = poly_degree + 1;
        int n_xy = n_x * n_x;
        int num_elem_points = (dim == 2) ? n_x * n_x : n_x * n_x * n_x;

        int e = idx / num_elem_points;
        int v = idx % num_elem_points;

        if (dim == 2)
        {
            int i = v % n_x;
            int j = v / n_x;

            double Du_1 = 0.0;
            double Du_2 = 0.0;

            for (int k = 0; k < n_x; k++)
            {
                Du_1 += D_hat[k + i * n_x] * u[e * num_elem_points + (k + j * n_x)];
                Du_2 += D_hat[k + j * n_x] * u[e * num_elem_points + (i + k * n_x)];
            }

            GDu[0][idx] = G[0][idx] * Du_1 + G[2][idx] * Du_2;
            GDu[1][idx] = G[2][idx] * Du_1 + G[1][idx] * Du_2;
        }
        else
        {
            int i = v % n_x;
            int j = (v / n_x) % n_x;
            int k = v / n_xy;

            double Du_1 = 0.0;
            double Du_2 = 0.0;
            double Du_3 = 0.0;

            for (int p = 0; p < n_x; p++)
            {
                Du_1 += D_hat[p + i * n_x] * u[e * num_elem_points + (p + j * n_x + k * n_xy)];
                Du_2 += D_hat[p + j * n_x] * u[e * num_elem_points + (i + p * n_x + k * n_xy)];
                Du_3 += D_hat[p + k * n_x] * u[e * num_elem_points + (i + j * n_x + p * n_xy)];
            }

            GDu[0][idx] = G[0][idx] * Du_1 + G[3][idx] * Du_2 + G[4][idx] * Du_3;
            GDu[1][idx] = G[3][idx] * Du_1 + G[1][idx] * Du_2 + G[5][idx] * Du_3;
            GDu[2][idx] = G[4][idx] * Du_1 + G[5][idx] * Du_2 + G[2][idx] * Du_3;
        }
    }
}

/* domain.okl:54-98 */
void orc_dom_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *D_hat, int num_points, int poly_degree, int dim)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        int n_x = poly_degree + 1;
        int n_xy = n_x * n_x;
        int num_elem_points = (dim == 2) ? n_x * n_x : n_x * n_x * n_x;

        int e = idx / num_elem_points;
        int v = idx % num_elem_points;

        if (dim == 2)
        {
            int i = v % n_x;
            int j = v / n_x;

            double Au_1 = 0.0;
            double Au_2 = 0.0;

            for (int k = 0; k < n_x; k++)
            {
                Au_1 += D_hat[i + k * n_x] * GDu[0][e * num_elem_points + (k + j * n_x)];
                Au_2 += D_hat[j + k * n_x] * GDu[1][e * num_elem_points + (i + k * n_x)];
            }

            Au[idx] = Au_1 + Au_2;
        }
        else
        {
            int i = v % n_x;
            int j = (v / n_x) % n_x;
            int k = v / n_xy;

            double Au_1 = 0.0;
            double Au_2 = 0.0;
            double Au_3 = 0.0;

            for (int p = 0; p < n_x; p++)
            {
                Au_1 += D_hat[i + p * n_x] * GDu[0][e * num_elem_points + (p + j * n_x + k * n_xy)];
                Au_2 += D_hat[j + p * n_x] * GDu[1][e * num_elem_points + (i + p * n_x + k * n_xy)];
                Au_3 += D_hat[k + p * n_x] * GDu[2][e * num_elem_points + (i + j * n_x + p * n_xy)];
            }

            Au[idx] = Au_1 + Au_2 + Au_3;
        }
    }
}

/* domain.okl:100-107 */
void orc_dom_initialize_arrays(double *u_k, double *r_k, const double *f, int num_points)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        u_k[idx] = 0.0;
        r_k[idx] = f[idx];
    }
}

/* domain.okl:109-138 */
void orc_dom_residual_norm(double *block, const double *r_k, const double *QQt_r_k, const double *dirichlet_mask, int num_points, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double r_norm[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_points)
                r_norm[item] = r_k[idx] * QQt_r_k[idx] * dirichlet_mask[idx];
            else
                r_norm[item] = 0.0;
        }

        block[group] = tree_reduce(r_norm);
    }
}

/* domain.okl:140-184 */
void orc_dom_projection_inner_products(double *block, const double *z_k, const double *r_k, const double *p_k, const double *q_k, int num_points, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double gamma_sum[ORC_BLOCK_SIZE];
        double theta_sum[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_points)
            {
                gamma_sum[item] = z_k[idx] * r_k[idx];
                theta_sum[item] = p_k[idx] * q_k[idx];
            }
            else
            {
                gamma_sum[item] = 0.0;
                theta_sum[item] = 0.0;
            }
        }

        block[group] = tree_reduce(gamma_sum);
        block[group + num_blocks] = tree_reduce(theta_sum);
    }
}

/* domain.okl:186-193 */
void orc_dom_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_points)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        u_k[idx] += alpha_k * p_k[idx];
        r_kp1[idx] = r_k[idx] - alpha_k * q_k[idx];
    }
}

/* domain.okl:195-224 */
void orc_dom_inner_product_flexible(double *block, const double *r_k, const double *r_kp1, const double *z_k, int num_points, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double theta_sum[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_points)
                theta_sum[item] = (r_kp1[idx] - r_k[idx]) * z_k[idx];
            else
                theta_sum[item] = 0.0;
        }

        block[group] = tree_reduce(theta_sum);
    }
}

/* domain.okl:226-233 */
void orc_dom_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_points)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        p_k[idx] = z_k[idx] + beta_k * p_k[idx];
        r_k[idx] = r_kp1[idx];
    }
}

/* domain.okl:235-264 */
void orc_dom_inner_product(double *block, const double *u_k, const double *v_k, const double *dirichlet_mask, int num_points, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double sum[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_points)
                sum[item] = u_k[idx] * v_k[idx] * dirichlet_mask[idx];
            else
                sum[item] = 0.0;
        }

        block[group] = tree_reduce(sum);
    }
}

/* ====================================================================== */
/* subdomain.okl                                                           */
/* ====================================================================== */

/* subdomain.okl:4-53.  `poly_degree` is the JIT-injected POLY_DEGREE table
 * (subdomain.tpp:3886-3890). */
void orc_sub_stiffness_matrix_1(double *const GDu[3], const double *u, const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, const double *const G[6], int num_points, int dim)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        int o = offset[idx];
        int v = vert[idx];
        int l = level[idx];
        int n_x = poly_degree[l] + 1;
        int n_xy = n_x * n_x;
        const double *D_hat = D_hat_ptr[l];

        if (dim == 2)
        {
            int i = v % n_x;
            int j = v / n_x;

            double Du_1 = 0.0;
            double Du_2 = 0.0;

            for (int k = 0; k < n_x; k++)
            {
                Du_1 += D_hat[k + i * n_x] * u[o + (k + j * n_x)];
                Du_2 += D_hat[k + j * n_x] * u[o + (i + k * n_x)];
            }

            GDu[0][idx] = G[0][idx] * Du_1 + G[2][idx] * Du_2;
            GDu[1][idx] = G[2][idx] * Du_1 + G[1][idx] * Du_2;
        }
        else
        {
            int i = v % n_x;
            int j = (v / n_x) % n_x;
            int k = v / n_xy;

            double Du_1 = 0.0;
            double Du_2 = 0.0;
            double Du_3 = 0.0;

            for (int p = 0; p < n_x; p++)
            {
                Du_1 += D_hat[p + i * n_x] * u[o + (p + j * n_x + k * n_xy)];
                Du_2 += D_hat[p + j * n_x] * u[o + (i + p * n_x + k * n_xy)];
                Du_3 += D_hat[p + k * n_x] * u[o + (i + j * n_x + p * n_xy)];
            }

            GDu[0][idx] = G[0][idx] * Du_1 + G[3][idx] * Du_2 + G[4][idx] * Du_3;
            GDu[1][idx] = G[3][idx] * Du_1 + G[1][idx] * Du_2 + G[5][idx] * Du_3;
            GDu[2][idx] = G[4][idx] * Du_1 + G[5][idx] * Du_2 + G[2][idx] * Du_3;
        }
    }
}

/* subdomain.okl:55-101 */
void orc_sub_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, int num_points, int dim)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        int o = offset[idx];
        int v = vert[idx];
        int l = level[idx];
        int n_x = poly_degree[l] + 1;
        int n_xy = n_x * n_x;
        const double *D_hat = D_hat_ptr[l];

        if (dim == 2)
        {
            int i = v % n_x;
            int j = v / n_x;

            double Au_1 = 0.0;
            double Au_2 = 0.0;

            for (int k = 0; k < n_x; k++)
            {
                Au_1 += D_hat[i + k * n_x] * GDu[0][o + (k + j * n_x)];
                Au_2 += D_hat[j + k * n_x] * GDu[1][o + (i + k * n_x)];
            }

            Au[idx] = Au_1 + Au_2;
        }
        else
        {
            int i = v % n_x;
            int j = (v / n_x) % n_x;
            int k = v / n_xy;

            double Au_1 = 0.0;
            double Au_2 = 0.0;
            double Au_3 = 0.0;

            for (int p = 0; p < n_x; p++)
            {
                Au_1 += D_hat[i + p * n_x] * GDu[0][o + (p + j * n_x + k * n_xy)];
                Au_2 += D_hat[j + p * n_x] * GDu[1][o + (i + p * n_x + k * n_xy)];
                Au_3 += D_hat[k + p * n_x] * GDu[2][o + (i + j * n_x + p * n_xy)];
            }

            Au[idx] = Au_1 + Au_2 + Au_3;
        }
    }
}

/* subdomain.okl:103-132 */
void orc_sub_inner_product(double *block, const double *u, const double *v, int num_values, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double uv[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_values)
                uv[item] = u[idx] * v[idx];
            else
                uv[item] = 0.0;
        }

        block[group] = tree_reduce(uv);
    }
}

/* subdomain.okl:134-163 */
void orc_sub_weighted_inner_product(double *block, const double *u, const double *v, const double *w, int num_values, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double uv[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_values)
                uv[item] = u[idx] * v[idx] * w[idx];
            else
                uv[item] = 0.0;
        }

        block[group] = tree_reduce(uv);
    }
}

/* subdomain.okl:165-209 */
void orc_sub_projection_inner_products(double *block, const double *z_k, const double *r_k, const double *p_k, const double *q_k, const double *weight, int num_values, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double gamma_sum[ORC_BLOCK_SIZE];
        double theta_sum[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_values)
            {
                gamma_sum[item] = z_k[idx] * r_k[idx] * weight[idx];
                theta_sum[item] = p_k[idx] * q_k[idx] * weight[idx];
            }
            else
            {
                gamma_sum[item] = 0.0;
                theta_sum[item] = 0.0;
            }
        }

        block[group] = tree_reduce(gamma_sum);
        block[group + num_blocks] = tree_reduce(theta_sum);
    }
}

/* subdomain.okl:211-218 */
void orc_sub_initialize_arrays(double *u_k, double *r_k, const double *f, int num_values)
{
    for (int idx = 0; idx < num_values; idx++)
    {
        u_k[idx] = 0.0;
        r_k[idx] = f[idx];
    }
}

/* subdomain.okl:220-227 */
void orc_sub_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_values)
{
    for (int idx = 0; idx < num_values; idx++)
    {
        u_k[idx] += alpha_k * p_k[idx];
        r_kp1[idx] = r_k[idx] - alpha_k * q_k[idx];
    }
}

/* subdomain.okl:229-258 */
void orc_sub_search_update_inner_product(double *block, const double *r_k, const double *r_kp1, const double *z_k, const double *weight, int num_points, int num_blocks)
{
    for (int group = 0; group < num_blocks; ++group)
    {
        double theta_sum[ORC_BLOCK_SIZE];

        for (int item = 0; item < ORC_BLOCK_SIZE; ++item)
        {
            int idx = group * ORC_BLOCK_SIZE + item;

            if (idx < num_points)
                theta_sum[item] = (r_kp1[idx] - r_k[idx]) * z_k[idx] * weight[idx];
            else
                theta_sum[item] = 0.0;
        }

        block[group] = tree_reduce(theta_sum);
    }
}

/* subdomain.okl:259-266 */
void orc_sub_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_values)
{
    for (int idx = 0; idx < num_values; idx++)
    {
        p_k[idx] = z_k[idx] + beta_k * p_k[idx];
        r_k[idx] = r_kp1[idx];
    }
}

/* subdomain.okl:268-282 with DType == EType == double */
void orc_sub_copy_f64_f64(double *u, const double *v, int num_points)
{
    for (int idx = 0; idx < num_points; idx++) u[idx] = (double)(v[idx]);
}

/* subdomain.okl:268-274 with DType = float (preconditioner), EType = double */
void orc_sub_copy_f32_f64(float *u, const double *v, int num_points)
{
    for (int idx = 0; idx < num_points; idx++) u[idx] = (float)(v[idx]);
}

/* subdomain.okl:276-282 with EType = double, DType = float */
void orc_sub_copy_f64_f32(double *u, const float *v, int num_points)
{
    for (int idx = 0; idx < num_points; idx++) u[idx] = (double)(v[idx]);
}

/* subdomain.okl:284-313 */
void orc_sub_restriction_1(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        int num_elem_points_fine = (dim == 2) ? n_f * n_f : n_f * n_f * n_f;
        int num_elem_points_coarse = (dim == 2) ? n_f * n_c : n_f * n_f * n_c;

        int e = idx / num_elem_points_coarse;
        int v = idx % num_elem_points_coarse;

        double Ju_ij = 0.0;

        if (dim == 2)
        {
            int i = v % n_f;
            int j = v / n_f;

            for (int k = 0; k < n_f; k++) Ju_ij += J_cf[j + k * n_c] * u[(i + k * n_f) + e * num_elem_points_fine];

            Ju[(i + j * n_f) + e * num_elem_points_coarse] = Ju_ij;
        }
        else
        {
            int i = v % n_c;
            int j = (v / n_c) % n_f;
            int k = v / (n_c * n_f);

            for (int l = 0; l < n_f; l++) Ju_ij += J_cf[i + l * n_c] * u[(l + j * n_f + k * (n_f * n_f)) + e * num_elem_points_fine];

            Ju[(i + j * n_c + k * (n_c * n_f)) + e * num_elem_points_coarse] = Ju_ij;
        }
    }
}

/* subdomain.okl:315-344 */
void orc_sub_restriction_2(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        int num_elem_points_fine = (dim == 2) ? n_f * n_c : n_f * n_f * n_c;
        int num_elem_points_coarse = (dim == 2) ? n_c * n_c : n_f * n_c * n_c;

        int e = idx / num_elem_points_coarse;
        int v = idx % num_elem_points_coarse;

        double Ju_ij = 0.0;

        if (dim == 2)
        {
            int i = v % n_c;
            int j = v / n_c;

            for (int k = 0; k < n_f; k++) Ju_ij += u[(j * n_f + k) + e * num_elem_points_fine] * J_cf[k * n_c + i];

            Ju[(i + j * n_c) + e * num_elem_points_coarse] = Ju_ij;
        }
        else
        {
            int i = v % n_c;
            int j = (v / n_c) % n_c;
            int k = v / (n_c * n_c);

            for (int l = 0; l < n_f; l++) Ju_ij += J_cf[j + l * n_c] * u[(i + l * n_c + k * (n_c * n_f)) + e * num_elem_points_fine];

            Ju[(i + j * n_c + k * (n_c * n_c)) + e * num_elem_points_coarse] = Ju_ij;
        }
    }
}

/* subdomain.okl:346-366 */
void orc_sub_restriction_3(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c)
{
    for (int idx = 0; idx < num_points; idx++)
    {
        int num_elem_points_fine = n_f * n_c * n_c;
        int num_elem_points_coarse = n_c * n_c * n_c;

        int e = idx / num_elem_points_coarse;
        int v = idx % num_elem_points_coarse;

        double Ju_ij = 0.0;

        int i = v % n_c;
        int j = (v / n_c) % n_c;
        int k = v / (n_c * n_c);

        for (int l = 0; l < n_f; l++) Ju_ij += J_cf[k + l * n_c] * u[(i + j * n_c + l * (n_c * n_c)) + e * num_elem_points_fine];

        Ju[(i + j * n_c + k * (n_c * n_c)) + e * num_elem_points_coarse] = Ju_ij;
    }
}

/* ====================================================================== */
/* AMG/kernels.cu, AMG/csr_matrix.cpp, subdomain.tpp:19-83                 */
/* ====================================================================== */

/* AMG/kernels.cu:11-23 */
void orc_amg_vector_set_to_value(double *data, double value, int size)
{
    for (int idx = 0; idx < size; idx++) data[idx] = value;
}

/* AMG/kernels.cu:25-41 */
void orc_amg_main_scaled_residual(double *Sr, double *w, const double *f_m_Au, const double *S, double alpha, int size)
{
    for (int idx = 0; idx < size; idx++)
    {
        Sr[idx] = S[idx] * f_m_Au[idx];
        w[idx] = alpha * Sr[idx];
    }
}

/* AMG/kernels.cu:43-59 */
void orc_amg_main_polynomial_evaluation(double *w, double *v, const double *r, const double *D_val, double alpha, int size)
{
    for (int idx = 0; idx < size; idx++)
    {
        v[idx] *= D_val[idx];
        w[idx] = alpha * r[idx] + v[idx];
    }
}

/* AMG/kernels.cu:61-76 */
void orc_amg_main_update_field(double *u, const double *w, const double *D_val, int size)
{
    for (int idx = 0; idx < size; idx++) u[idx] += D_val[idx] * w[idx];
}

/* AMG/kernels.cu:79-94 */
void orc_amg_vector_multiplication(double *uv, const double *u, const double *v, int size)
{
    for (int idx = 0; idx < size; idx++) uv[idx] = u[idx] * v[idx];
}

/* AMG/csr_matrix.cpp:112-134, host branch: y = alpha * A x + beta * y */
void orc_amg_matvec(double *y, const int *ptr, const int *col, const double *val, const double *x, double alpha, double beta, int num_rows)
{
    for (int row = 0; row < num_rows; row++)
    {
        double Ax = 0.0;

        for (int idx = ptr[row]; idx < ptr[row + 1]; idx++)
            Ax += val[idx] * x[col[idx]];

        /* cusparseSpMV does not read y when beta == 0 (AMG/csr_matrix.cpp:129-131) */
        y[row] = (beta == 0.0) ? alpha * Ax : alpha * Ax + beta * y[row];
    }
}

/* subdomain.tpp:21-33 ("host" branch of scaled_residual) */
void orc_amg_scaled_residual_host(double *Sr, double *w, const int *ptr, const int *col, const double *val, const double *u, const double *f, const double *S, double alpha, int num_rows)
{
    for (int row = 0; row < num_rows; row++)
    {
        double Ax = 0.0;

        for (int idx = ptr[row]; idx < ptr[row + 1]; idx++)
            Ax += val[idx] * u[col[idx]];

        Sr[row] = S[row] * (f[row] - Ax);
        w[row] = alpha * Sr[row];
    }
}

/* subdomain.tpp:47-61 ("host" branch of polynomial_evaluation) */
void orc_amg_polynomial_evaluation_host(double *w, double *v, const int *ptr, const int *col, const double *val, const double *r, const double *D_val, double alpha, int num_rows)
{
    for (int row = 0; row < num_rows; row++)
    {
        double tmp = 0.0;

        for (int idx = ptr[row]; idx < ptr[row + 1]; idx++)
            tmp += val[idx] * D_val[col[idx]] * w[col[idx]];

        v[row] = D_val[row] * tmp;
    }

    for (int row = 0; row < num_rows; row++)
        w[row] = alpha * r[row] + v[row];
}
