// The scalar bookkeeping of the inner flexible GMRES(m) (Hessenberg column,
// Givens rotations, residual recurrence, stopping tests, back-substitution:
// subdomain.tpp:4396-4477) as one-thread kernels, so that a whole restart
// cycle is enqueued without a host round trip per step.  The arithmetic is the
// host code's, statement for statement (IEEE sqrt and division, no contraction).
//
// Stopping is recorded, not acted on: once `stopped` is set the later steps of
// the cycle still run on the device (their vectors are never used) and the
// solution update takes the columns 0..j_last the reference would have taken.
#include "fdd_common.h"

namespace
{
constexpr int kMax = FDD_MULTI_MAX;

struct GmresState
{
    double H[kMax][kMax];
    double c[kMax], s[kMax], gamma[kMax + 1];
    double y[kMax];        // solution coefficients of the cycle (0 beyond j_last)
    double hist[kMax + 1]; // residual norms recorded in this cycle (hist[0]: its starting norm)
    double r0;             // starting norm of the first cycle (relative test)
    double num_hist;       // entries of hist
    double stopped;        // 0 / 1
    double j_last;         // last column that enters the update
    double steps;          // steps counted until the stop (the reference's `iter` increment of this cycle)
    double converged;      // the reference's `converged` flag at the end of the cycle
    double inv[kMax + 1];  // 1/gamma_0 and 1/||q_j||: the normalisation of basis vector j, applied by its readers
};

__global__ void gmres_begin_kernel(GmresState *st, const double *norm2, int first_cycle)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double g0 = sqrt(*norm2);
    st->gamma[0] = g0;
    st->inv[0] = 1.0 / g0; // subdomain.tpp:4358
    if (first_cycle) st->r0 = g0;
    st->hist[0] = g0;
    st->num_hist = 1.0;
    st->stopped = 0.0;
    st->j_last = -1.0;
    st->steps = 0.0;
    st->converged = 0.0;
    for (int k = 0; k < kMax; k++) st->y[k] = 0.0;
}

// column j: dots[0..j] = <q, v_i>, dots[j+1] = ||q - sum h v||^2
__global__ void gmres_step_kernel(GmresState *st, const double *dots, int j, int iterations_before, int max_iterations, double tolerance, int use_relative)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (st->stopped != 0.0) return;
    st->steps += 1.0;
    const int iter = iterations_before + (int)st->steps;

    for (int i = 0; i < j + 1; i++) st->H[i][j] = dots[i];
    for (int i = 0; i < j; i++)
    {
        const double h_ij = st->H[i][j];
        st->H[i][j] = st->c[i] * h_ij + st->s[i] * st->H[i + 1][j];
        st->H[i + 1][j] = -st->s[i] * h_ij + st->c[i] * st->H[i + 1][j];
    }
    const double alpha_j = sqrt(dots[j + 1]);
    st->j_last = (double)j;
    if (fabs(alpha_j) == 0.0) // subdomain.tpp:4418-4422
    {
        st->stopped = 1.0;
        st->converged = 1.0;
        return;
    }
    st->inv[j + 1] = 1.0 / alpha_j; // subdomain.tpp:4457
    const double beta_j = sqrt(st->H[j][j] * st->H[j][j] + alpha_j * alpha_j);
    const double gamma_j = 1.0 / beta_j;
    st->c[j] = st->H[j][j] * gamma_j;
    st->s[j] = alpha_j * gamma_j;
    st->H[j][j] = beta_j;
    st->gamma[j + 1] = -st->s[j] * st->gamma[j];
    st->gamma[j] = st->c[j] * st->gamma[j];

    const double r_norm = fabs(st->gamma[j + 1]);
    st->hist[(int)st->num_hist] = r_norm;
    st->num_hist += 1.0;
    const bool small = use_relative ? (r_norm / st->r0 < tolerance) : (r_norm < tolerance);
    if (small || iter >= max_iterations) // subdomain.tpp:4436-4455
    {
        st->stopped = 1.0;
        st->converged = 1.0;
    }
}

// back-substitution (subdomain.tpp:4462-4468): y[0..j_last]
__global__ void gmres_finish_kernel(GmresState *st, int m)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int j = (int)st->j_last;
    if (st->stopped == 0.0) j = m - 1; // the loop ran out: `if (j == num_vectors) j--`
    st->j_last = (double)j;
    for (int k = j; k >= 0; k--)
    {
        double gamma_k = st->gamma[k];
        for (int i = j; i > k; i--) gamma_k -= st->H[k][i] * st->c[i];
        st->c[k] = gamma_k / st->H[k][k];
    }
    for (int k = 0; k < kMax; k++) st->y[k] = (k <= j) ? st->c[k] : 0.0;
}
} // namespace

extern "C" {

size_t fdd_gmres_state_bytes(void) { return sizeof(GmresState); }

int fdd_gmres_begin_dev(void *state, const double *norm2_dev, int first_cycle, void *stream)
{
    FDD_REQUIRE(state != nullptr && norm2_dev != nullptr);
    hipLaunchKernelGGL(gmres_begin_kernel, dim3(1), dim3(64), 0, fdd_stream(stream), static_cast<GmresState *>(state), norm2_dev, first_cycle);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_gmres_step_dev(void *state, const double *dots_dev, int j, int iterations_before, int max_iterations, double tolerance, int use_relative, void *stream)
{
    FDD_REQUIRE(state != nullptr && dots_dev != nullptr && j >= 0 && j < kMax);
    hipLaunchKernelGGL(gmres_step_kernel, dim3(1), dim3(64), 0, fdd_stream(stream), static_cast<GmresState *>(state), dots_dev, j, iterations_before, max_iterations, tolerance, use_relative);
    FDD_LAUNCH_CHECK();
    return 0;
}

int fdd_gmres_finish_dev(void *state, int m, void *stream)
{
    FDD_REQUIRE(state != nullptr && m >= 1 && m <= kMax);
    hipLaunchKernelGGL(gmres_finish_kernel, dim3(1), dim3(64), 0, fdd_stream(stream), static_cast<GmresState *>(state), m);
    FDD_LAUNCH_CHECK();
    return 0;
}

// one blocking copy of what the host needs after a cycle
int fdd_gmres_fetch(void *state, double *y, double *hist, int *num_hist, int *j_last, int *steps, int *converged, void *stream)
{
    FDD_REQUIRE(state != nullptr);
    GmresState h;
    static_assert(sizeof(GmresState) <= 4096, "fetched through the pinned staging buffer");
    int rc = fdd_fetch_scalars(&h, state, sizeof(GmresState), stream);
    if (rc) return rc;
    if (y)
        for (int k = 0; k < kMax; k++) y[k] = h.y[k];
    const int nh = (int)h.num_hist;
    if (hist)
        for (int k = 0; k < nh; k++) hist[k] = h.hist[k];
    if (num_hist) *num_hist = nh;
    if (j_last) *j_last = (int)h.j_last;
    if (steps) *steps = (int)h.steps;
    if (converged) *converged = (int)h.converged;
    return 0;
}

int fdd_gmres_scales(void *state, const double **inv_dev)
{
    FDD_REQUIRE(state != nullptr && inv_dev != nullptr);
    *inv_dev = static_cast<GmresState *>(state)->inv;
    return 0;
}

int fdd_gmres_last_column(void *state, const double **j_last_dev)
{
    FDD_REQUIRE(state != nullptr && j_last_dev != nullptr);
    *j_last_dev = &static_cast<GmresState *>(state)->j_last;
    return 0;
}

int fdd_gmres_coefficients(void *state, const double **y_dev)
{
    FDD_REQUIRE(state != nullptr && y_dev != nullptr);
    *y_dev = static_cast<GmresState *>(state)->y;
    return 0;
}

} // extern "C"
