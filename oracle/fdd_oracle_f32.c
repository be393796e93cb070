/*
 * fdd_oracle_f32.c -- TEST INFRASTRUCTURE (CPU oracle; never linked, loaded or called by the product).
 *
 * IEEE-single restatement of the inner solve's kernels as the reference compiles them with
 * PTYPE = Float = float (config.hpp:19-20, poisson.cpp:206: Subdomain<PTYPE>; run.py:157 sweeps Float = float):
 * every OKL kernel body with DType = float, OCCA-Serial semantics (loops run sequentially), compiled
 * -O2 -ffp-contract=off so that every float operation is one rounded IEEE-single operation.
 *
 * What is pinned to what:
 *   orc_f32_sub_stiffness        subdomain.okl:4-101 with DType = float (both launches, uniform degree, element-major
 *                                points): every product, sum and the (Au_1 + Au_2) + Au_3 order in float
 *   orc_f32_vector_vector_addition / orc_f32_vector_scaling    math.okl:21-35 with DType = float
 *   orc_f32_csr_gather           csr_matrix.okl:5-18 with DType = float on a boolean matrix (values 1.0f not read)
 *   orc_f32_gather_indexed*      the index copies of the dof-space form (a copy has no arithmetic)
 * The reductions of the float inner solve keep DOUBLE accumulators in this build (stated deviation from the
 * reference's float block sums, subdomain.okl:134-180 with DType = float: a 10^7-term float dot loses half its digits);
 * their twins restate exactly that arithmetic, term order 0..n-1:
 *   orc_f32_multi_inner_product_scaled, orc_f32_multi_axpy_norm2_scaled, orc_f32_multi_lincomb
 */
#include <stddef.h>

/* subdomain.okl:4-101, DType = float, 3-D, num_elements elements of (N+1)^3 points each.  `index` non-NULL: the
 * element reads u[e,i,j,k] = scale * v[index[point]] (0 where index < 0): the boolean scatter Q and the
 * vector_scaling of math.okl:29-35 applied on load, the product in float as the reference's float kernels form it. */
void orc_f32_sub_stiffness(float *Au, const float *v, const int *index, const float *scale, const float *D_hat, const float *const G[6], int num_elements, int poly_degree)
{
    const int n = poly_degree + 1, nn = n * n, n3 = nn * n;
    float u[16 * 16 * 16], g1[16 * 16 * 16], g2[16 * 16 * 16], g3[16 * 16 * 16];
    for (int e = 0; e < num_elements; e++)
    {
        const size_t o = (size_t)e * n3;
        for (int p = 0; p < n3; p++)
        {
            if (index)
            {
                const int d = index[o + p];
                float x = d < 0 ? 0.0f : v[d];
                if (scale && d >= 0) x = (*scale) * x;
                u[p] = x;
            }
            else
                u[p] = v[o + p];
        }
        /* subdomain.okl:4-53 */
        for (int k = 0; k < n; k++)
            for (int j = 0; j < n; j++)
                for (int i = 0; i < n; i++)
                {
                    float Du_1 = 0.0f, Du_2 = 0.0f, Du_3 = 0.0f;
                    for (int p = 0; p < n; p++)
                    {
                        Du_1 += D_hat[p + i * n] * u[p + j * n + k * nn];
                        Du_2 += D_hat[p + j * n] * u[i + p * n + k * nn];
                        Du_3 += D_hat[p + k * n] * u[i + j * n + p * nn];
                    }
                    const size_t idx = o + (size_t)(i + j * n + k * nn);
                    g1[i + j * n + k * nn] = G[0][idx] * Du_1 + G[3][idx] * Du_2 + G[4][idx] * Du_3;
                    g2[i + j * n + k * nn] = G[3][idx] * Du_1 + G[1][idx] * Du_2 + G[5][idx] * Du_3;
                    g3[i + j * n + k * nn] = G[4][idx] * Du_1 + G[5][idx] * Du_2 + G[2][idx] * Du_3;
                }
        /* subdomain.okl:55-101 */
        for (int k = 0; k < n; k++)
            for (int j = 0; j < n; j++)
                for (int i = 0; i < n; i++)
                {
                    float Au_1 = 0.0f, Au_2 = 0.0f, Au_3 = 0.0f;
                    for (int p = 0; p < n; p++)
                    {
                        Au_1 += D_hat[i + p * n] * g1[p + j * n + k * nn];
                        Au_2 += D_hat[j + p * n] * g2[i + p * n + k * nn];
                        Au_3 += D_hat[k + p * n] * g3[i + j * n + p * nn];
                    }
                    Au[o + (size_t)(i + j * n + k * nn)] = Au_1 + Au_2 + Au_3;
                }
    }
}

/* math.okl:21-27, DType = float */
void orc_f32_vector_vector_addition(float *uv, float alpha, const float *u, float beta, const float *v, int n)
{
    for (int i = 0; i < n; i++) uv[i] = alpha * u[i] + beta * v[i];
}

/* math.okl:29-35, DType = float; the scale is a device-resident double (1/norm of the Krylov vector) rounded to float
 * once, as a float build of the reference holds it */
void orc_f32_vector_scaling(float *au, double alpha, const float *u, int n)
{
    const float a = (float)alpha;
    for (int i = 0; i < n; i++) au[i] = a * u[i];
}

/* csr_matrix.okl:5-18, DType = float, boolean matrix (every value 1.0f: the products are exact and left out) */
void orc_f32_csr_gather(float *t, const int *ptr, const int *col, const float *u, int row_lo, int row_hi)
{
    for (int r = row_lo; r < row_hi; r++)
    {
        float s = 0.0f;
        for (int j = ptr[r]; j < ptr[r + 1]; j++) s += u[col[j]];
        t[r] = s;
    }
}

void orc_f32_gather_indexed(float *out, const float *in, const int *index, int n)
{
    for (int i = 0; i < n; i++) out[i] = index[i] < 0 ? 0.0f : in[index[i]];
}

void orc_f32_gather_indexed_f64(double *out, const float *in, const int *index, int n)
{
    for (int i = 0; i < n; i++) out[i] = index[i] < 0 ? 0.0 : (double)in[index[i]];
}

/* out[k] = sum_i a_i * (s_k * b_k,i), float data, double products and sums, i ascending */
void orc_f32_multi_inner_product_scaled(double *out, const float *a, const float *const *b, const double *b_scale, int m, int n)
{
    for (int k = 0; k < m; k++)
    {
        const double sk = b_scale ? b_scale[k] : 1.0;
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc += (double)a[i] * (sk * (double)b[k][i]);
        out[k] = acc;
    }
}

/* dst = float(y + sum_k (sign * c_k * s_k) * x_k) accumulated in double, k ascending; returns |dst|^2 summed in double */
double orc_f32_multi_axpy_norm2_scaled(float *dst, const float *y, const double *c, double sign, const float *const *x, const double *x_scale, int m, int n)
{
    double norm2 = 0.0;
    for (int i = 0; i < n; i++)
    {
        double v = (double)y[i];
        for (int k = 0; k < m; k++)
        {
            const double ck = sign * c[k] * (x_scale ? x_scale[k] : 1.0);
            v += ck * (double)x[k][i];
        }
        const float r = (float)v;
        dst[i] = r;
        norm2 += (double)r * (double)r;
    }
    return norm2;
}

/* q (+)= sum_{k < use} (c_k * s_k) * v_k accumulated in double, rounded to float once; use = min(m, last + 1) */
void orc_f32_multi_lincomb(float *q, int q_is_zero, const double *c, const float *const *v, const double *v_scale, int last, int m, int n)
{
    int use = last >= 0 ? last + 1 : m;
    if (use > m) use = m;
    for (int i = 0; i < n; i++)
    {
        double acc = q_is_zero ? 0.0 : (double)q[i];
        for (int k = 0; k < use; k++) acc += (c[k] * (v_scale ? v_scale[k] : 1.0)) * (double)v[k][i];
        q[i] = (float)acc;
    }
}
