/*
 * math.hpp -- Math<DType>: the BLAS-1 helper class of the reference
 * (math.hpp:12-35, math.tpp:45-92) on the gfx950 kernels (math.okl ->
 * fdd_set_to_value / fdd_invert_vector_elements / fdd_vector_vector_addition /
 * fdd_vector_scaling).  No JIT build step: kernels are AOT-compiled, so the
 * reference's rank-0-first buildKernel + MPI_Barrier dance (math.tpp:18-36)
 * has no counterpart.
 */
#ifndef FDD_MATH_HPP
#define FDD_MATH_HPP

#include <type_traits>

#include "config.hpp"

template <typename DType>
class Math
{
    static_assert(std::is_same<DType, double>::value, "the gfx950 kernels are fp64 (SURVEY.md 2b: scope is DType = double)");

  public:
    Math() {}
    ~Math() {}

    void set_to_value(fdd::memory &u, DType alpha, int n, int offset = 0) { FDD_CALL(fdd_set_to_value(u.as<double>(), alpha, n, offset, fdd::dev().stream)); }

    void invert_vector_elements(fdd::memory &u, int n) { FDD_CALL(fdd_invert_vector_elements(u.as<double>(), n, fdd::dev().stream)); }

    void vector_vector_addition(fdd::memory &uv, const DType alpha, const fdd::memory &u, const DType beta, const fdd::memory &v, const int n)
    {
        FDD_CALL(fdd_vector_vector_addition(uv.as<double>(), alpha, u.as<double>(), beta, v.as<double>(), n, fdd::dev().stream));
    }

    void vector_scaling(fdd::memory &au, const DType alpha, const fdd::memory &u, const int n) { FDD_CALL(fdd_vector_scaling(au.as<double>(), alpha, u.as<double>(), n, fdd::dev().stream)); }

    // Host helper of the reference (math.tpp:70-92); unused there too.  The
    // reference indexes B with stride p (B[k*p + j], math.tpp:81), which is
    // only right for square B; this one uses the row length m.
    void matrix_matrix_multiply(DType *C, const DType *A, const DType *B, int n, int p, int m, bool A_t = false, bool B_t = false)
    {
        if (A_t || B_t)
        {
            pstdout("Not implemented, yet");
            fdd::quit();
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < m; j++)
            {
                DType C_ij = 0.0;
                for (int k = 0; k < p; k++) C_ij += A[i * p + k] * B[k * m + j];
                C[i * m + j] = C_ij;
            }
    }
};

#endif
