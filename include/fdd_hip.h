/*
 * fdd_hip.h -- C-ABI of libfdd_hip.so: the MI355X (gfx950) kernels of the
 * preconditioned-CG hot path of the SEM Poisson solve, replacing the OCCA
 * launch layer (occa::device / occa::memory / occa::kernel::operator()) of the
 * reference.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every entry returns int: 0 = ok, >0 = hipError_t, <0 = FDD_ERR_*;
 *     fdd_last_error() gives the text of the last failure on this thread;
 *   - all pointers named like reference kernel arguments are DEVICE pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     kernel entry is asynchronous on it.  Only fdd_memcpy_h2d / _d2h,
 *     fdd_stream_sync and fdd_device_sync block (occa::memory::copyFrom/copyTo
 *     and occa::device::finish semantics);
 *   - the caller owns all buffers, kernels never allocate;
 *   - argument order follows the OKL kernel it replaces, then workspaces,
 *     then `stream`;
 *   - arithmetic is fp64 with int32 indices (csr_matrix.tpp:132-134), compiled
 *     with -ffp-contract=off: element-wise kernels, thread-per-row and
 *     LDS-staged SpMV and the tensor-product kernels keep the reference's
 *     per-output operation order and are bit-identical to the OCCA-Serial
 *     arithmetic; reductions use a different summation tree (tolerance stated
 *     in tests/), and the fp64-MFMA kernels fuse multiply-add.
 *
 * Each entry cites the reference interface it replaces (file:line relative to
 * the reference repository root).
 */
#ifndef FDD_HIP_H
#define FDD_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDD_ERR_INVALID_ARGUMENT (-1)
#define FDD_ERR_UNSUPPORTED (-2)
#define FDD_ERR_NOT_INITIALIZED (-3)

#define FDD_NUM_GEOM_FACTS 6      /* element.hpp:10-12 */
#define FDD_REDUCE_MAX_BLOCKS 2048 /* partial sums any reduction entry may write into its workspace */

/* ------------------------------------------------------------------ */
/* runtime: replaces occa::device / occa::memory (config.hpp:52;        */
/* usage inventory SURVEY.md section 8(b))                              */
/* ------------------------------------------------------------------ */
const char *fdd_version(void);
const char *fdd_last_error(void);
int fdd_device_count(int *count);
int fdd_set_device(int device);                 /* occa::device::setup, poisson.cpp:137 */
int fdd_get_device(int *device);
int fdd_device_name(char *buf, size_t buf_len);
int fdd_malloc(void **ptr, size_t bytes);       /* occa::device::malloc<T>(n) */
int fdd_free(void *ptr);                        /* occa::memory::free() */
int fdd_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream); /* occa::memory::copyFrom(host): blocking */
int fdd_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream); /* occa::memory::copyTo(host): blocking */
int fdd_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream); /* occa::memory::copyFrom/To(mem): async */
int fdd_fetch_scalars(void *dst, const void *src, size_t bytes, void *stream); /* blocking D2H of <= 4 KiB of reduction results via a pinned staging buffer */
int fdd_memset(void *dst, int value, size_t bytes, void *stream);
int fdd_stream_create(void **stream);           /* cudaStreamCreate, subdomain.tpp:3475 */
int fdd_stream_destroy(void *stream);
int fdd_stream_sync(void *stream);
int fdd_device_sync(void);                      /* occa::device::finish(), timer.tpp:50,59 */
/* hipGraph capture / replay of a launch sequence (cudaGraph of the V-cycle legs,
 * subdomain.tpp:3644-3704, 4021, 4113).  Capture needs a non-default stream. */
int fdd_graph_begin_capture(void *stream);
int fdd_graph_end_capture(void *stream, void **graph_exec);
int fdd_graph_launch(void *graph_exec, void *stream);
int fdd_graph_destroy(void *graph_exec);
int fdd_event_create(void **event);
int fdd_event_destroy(void *event);
int fdd_event_record(void *event, void *stream);
int fdd_event_elapsed_ms(float *ms, void *start, void *stop); /* synchronises on `stop` */

/* ------------------------------------------------------------------ */
/* csr_matrix.okl -- CSR y = A x                                        */
/* ------------------------------------------------------------------ */
/* csr_matrix.okl:5-18  multiply(Au, A_ptr, A_col, A_val, u, n); launched from csr_matrix.tpp:310 */
int fdd_csr_multiply(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int n, void *stream);
/* csr_matrix.okl:20-33 multiply_range(..., row_start, row_end) -- row_end INCLUSIVE; csr_matrix.tpp:328 */
int fdd_csr_multiply_range(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int row_start, int row_end, void *stream);
/* csr_matrix.okl:35-48 multiply_weight(..., weight, n); csr_matrix.tpp:340 */
int fdd_csr_multiply_weight(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *weight, int n, void *stream);

/* Planned SpMV: the row partition CSR_Matrix::assemble (csr_matrix.tpp:94-180)
 * can compute once while it still holds the host `ptr` array.  Rows are grouped
 * into blocks of <= FDD_CSR_BLOCK_NNZ non-zeros that one workgroup streams
 * coalesced through LDS; each row is then summed in column order by one lane,
 * which keeps the reference's summation order.  Rows longer than a block are
 * reduced by a whole workgroup (order differs; tolerance in tests/). */
#define FDD_CSR_BLOCK_NNZ 2048
typedef struct fdd_csr_plan fdd_csr_plan;
int fdd_csr_plan_create(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz);
int fdd_csr_plan_destroy(fdd_csr_plan *plan);
int fdd_csr_plan_num_blocks(const fdd_csr_plan *plan, int *num_blocks);
int fdd_csr_plan_kind(const fdd_csr_plan *plan, int *kind); /* 0 = thread-per-row, 1 = LDS-staged row blocks */
int fdd_csr_plan_pipelined(const fdd_csr_plan *plan, int *pipelined); /* 1: a short-row plan whose SpMV / gather entries run on the persistent, software-pipelined kernel (profile labels) */
/* Tell the plan that every stored value is exactly 1.0 (the boolean gather / scatter matrices Q, Qt,
 * Q_int, ...; CSR_Matrix::assemble checks its host values): the kernels then skip the val array --
 * 1.0*x is x, so results are unchanged and 8 of the 12 bytes per non-zero are not moved. */
int fdd_csr_plan_set_unit_values(fdd_csr_plan *plan, int unit_values);
/* Give the plan a sliced-ELL copy of the matrix (slices of 64 rows, column-major, padded to the slice's longest row)
 * when that costs at most max_padding times the stored entries and no row exceeds 64 entries: the form for the
 * short, even rows of the AMG levels.  *attached = 1: fdd_csr_plan_multiply / _matvec[_to] / fdd_amg_smooth_* of this
 * plan run on it from now on, same row sums in the same order.  A_ptr_host: host copy of the row pointers; A_ptr,
 * A_col, A_val: the device arrays (A_val double, or float on a plan of fdd_csr_plan_create_f32). */
int fdd_csr_plan_attach_sell(fdd_csr_plan *plan, const int *A_ptr_host, const int *A_ptr, const int *A_col, const void *A_val, double max_padding, int *attached, void *stream);
/* What the attach made: slices of 64 rows (0: none) and how many of them store their columns in the compact form -- entry k of
 * a slice's 64 rows as one base (the smallest of the 64 columns) + a 16-bit offset per row, possible wherever the 64 columns
 * of every slot lie within 65535 of each other (on the lattice-numbered AMG levels they lie within 63): 10 instead of 12
 * bytes per entry, 6 instead of 8 in single precision; the same columns, values and order. */
int fdd_csr_plan_sell_info(const fdd_csr_plan *plan, int *slices, int *compact_slices);
/* weight may be NULL (multiply) or a device vector of num_rows (multiply_weight) */
int fdd_csr_plan_multiply(const fdd_csr_plan *plan, double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *weight, void *stream);

/* ------------------------------------------------------------------ */
/* math.okl -- BLAS-1                                                    */
/* ------------------------------------------------------------------ */
int fdd_set_to_value(double *u, double alpha, int n, int offset, void *stream);                                              /* math.okl:5-11,  math.tpp:48 */
int fdd_invert_vector_elements(double *u, int n, void *stream);                                                              /* math.okl:13-19, math.tpp:54 */
int fdd_vector_vector_addition(double *uv, double alpha, const double *u, double beta, const double *v, int n, void *stream); /* math.okl:21-27, math.tpp:60; uv may alias u or v */
int fdd_vector_scaling(double *au, double alpha, const double *u, int n, void *stream);                                      /* math.okl:29-35, math.tpp:66 */

/* ------------------------------------------------------------------ */
/* domain.okl -- outer solver kernels                                   */
/* ------------------------------------------------------------------ */
/* Reference two-kernel form, one thread per GLL point, global scratch GDu.
 * G / GDu are HOST arrays of device pointers (the reference keeps device
 * pointer tables, domain.tpp:65-67, 221-224).  dim = 2 or 3. */
int fdd_dom_stiffness_matrix_1(double *const GDu[3], const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], int num_points, int poly_degree, int dim, void *stream); /* domain.okl:5-52,  domain.tpp:605 */
int fdd_dom_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *D_hat, int num_points, int poly_degree, int dim, void *stream);                                           /* domain.okl:54-98, domain.tpp:606 */
/* Fused Au = D^T G D u (both passes in one launch, slabs staged in LDS, no
 * global scratch): 64 B/point.  3-D, poly_degree 1..15.  Replaces the pair of
 * launches in Domain::stiffness_matrix (domain.tpp:602-607). */
int fdd_dom_stiffness_matrix(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], int num_elements, int poly_degree, void *stream);

/* The same fused operator on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) for poly_degree 8..15:
 * the six contractions of an element are 16x16x16 products on LDS-staged views, one persistent
 * 1024-lane workgroup per CU, next element prefetched.  MFMA fuses multiply-add, so this entry is
 * NOT bit-identical to the reference arithmetic (agrees to ~1e-15 * max|Au|).  elem_offset as in
 * fdd_sub_stiffness_matrix (NULL => contiguous elements).  Au must not alias u. */
int fdd_stiffness_matrix_mfma(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream);
int fdd_stiffness_matrix_mfma_gather(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream); /* u[p] = (*v_scale_dev) * v[point_dof[p]] on load (scale may be NULL) */

int fdd_dom_initialize_arrays(double *u_k, double *r_k, const double *f, int num_points, void *stream); /* domain.okl:100-107, domain.tpp:618,734 */

/* Reductions.  The reference kernels write one partial per 128-thread block
 * and the host sums them after a D2H copy (domain.tpp:916-996).  Here `out`
 * (device) receives the FINAL scalar(s); `ws` is a device workspace of
 * fdd_reduce_workspace_doubles() doubles. */
size_t fdd_reduce_workspace_doubles(void);
int fdd_dom_residual_norm(double *out, double *ws, const double *r_k, const double *QQt_r_k, const double *dirichlet_mask, int num_points, void *stream);                  /* domain.okl:109-138; out[0] = sum r*QQt_r*mask (no sqrt) */
int fdd_dom_projection_inner_products(double *out2, double *ws, const double *z_k, const double *r_k, const double *p_k, const double *q_k, int num_points, void *stream); /* domain.okl:140-184; out2 = {gamma, theta} */
int fdd_dom_inner_product_flexible(double *out, double *ws, const double *r_k, const double *r_kp1, const double *z_k, int num_points, void *stream);                      /* domain.okl:195-224 */
/* The same sum and, from the same three vectors, the NEXT iteration's gamma = <z_k, r_kp1> (the first sum of
 * domain.okl:140-184 one iteration early; fdd_sub_inner_product(p, q) is what is left of that kernel):
 * out2 = {gamma_next, theta}.  Same bits as the two separate entries. */
int fdd_dom_inner_product_flexible_gamma(double *out2, double *ws, const double *r_k, const double *r_kp1, const double *z_k, int num_points, void *stream);
int fdd_dom_inner_product(double *out, double *ws, const double *u_k, const double *v_k, const double *dirichlet_mask, int num_points, void *stream);                      /* domain.okl:235-264 */

int fdd_dom_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_points, void *stream); /* domain.okl:186-193, domain.tpp:978 */
int fdd_dom_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_points, void *stream);                      /* domain.okl:226-233, domain.tpp:720 */
/* Same updates with the step length taken from device scalars (no host
 * round trip between the dot products and the update):
 *   alpha = num[0] / den[0];   beta = num[0] / den[0]. */
int fdd_dom_solution_and_residual_update_dev(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, const double *alpha_num, const double *alpha_den, int num_points, void *stream);
int fdd_dom_residual_and_search_update_dev(double *p_k, double *r_k, const double *z_k, const double *r_kp1, const double *beta_num, const double *beta_den, int num_points, void *stream);

/* ------------------------------------------------------------------ */
/* subdomain.okl -- FDD local-solve kernels                             */
/* ------------------------------------------------------------------ */
/* Reference form with per-point indirection (offset / vert / level int arrays,
 * subdomain.tpp:1603-1630).  D_hat_ptr is a HOST array of num_levels device
 * pointers; poly_degree is the HOST table injected as the JIT macro
 * POLY_DEGREE (subdomain.tpp:3886-3890), num_levels <= 16. */
int fdd_sub_stiffness_matrix_1(double *const GDu[3], const double *u, const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, int num_levels, const double *const G[FDD_NUM_GEOM_FACTS], int num_points, int dim, void *stream); /* subdomain.okl:4-53,   subdomain.tpp:3953 */
int fdd_sub_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, int num_levels, int num_points, int dim, void *stream);                                           /* subdomain.okl:55-101, subdomain.tpp:3961 */
/* Fused, level-sorted form: one launch per polynomial level over the list of
 * that level's elements; elem_offset[e] (device) is the first point of element
 * e in u / Au / G (NULL => e * (N+1)^3). */
int fdd_sub_stiffness_matrix(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream);
/* 2-D form of the fused kernel (the DIM == 2 branches of domain.okl:20-33 and :69-80 in one launch, 40 B/point):
 * elements of (poly_degree+1)^2 points, G[0] = G11, G[1] = G22, G[2] = G12 (the other three are not read);
 * elem_offset as in fdd_sub_stiffness_matrix (nullptr: element e starts at e*(poly_degree+1)^2).  Au may alias u.
 * Bit-identical to fdd_dom_stiffness_matrix_1 + _2 with dim = 2. */
int fdd_stiffness_matrix_2d(double *Au, const double *u, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream);
/* The same with the boolean scatter Q of the subdomain fused into the load: u[p] = v[point_dof[p]]
 * (0 where point_dof[p] < 0), v a vector over the subdomain's dofs (Q v then A, subdomain.tpp:3977-3981 + 3942-3967). */
int fdd_sub_stiffness_matrix_gather(double *Au, const double *v, const int *point_dof, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream);

int fdd_sub_inner_product(double *out, double *ws, const double *u, const double *v, int num_values, void *stream);                                                                                  /* subdomain.okl:103-132 */
int fdd_sub_weighted_inner_product(double *out, double *ws, const double *u, const double *v, const double *w, int num_values, void *stream);                                                       /* subdomain.okl:134-163, subdomain.tpp:4302,4508 */
int fdd_sub_projection_inner_products(double *out2, double *ws, const double *z_k, const double *r_k, const double *p_k, const double *q_k, const double *weight, int num_values, void *stream);    /* subdomain.okl:165-209, subdomain.tpp:4526 */
int fdd_sub_search_update_inner_product(double *out, double *ws, const double *r_k, const double *r_kp1, const double *z_k, const double *weight, int num_points, void *stream);                    /* subdomain.okl:229-258, subdomain.tpp:4552 */
int fdd_sub_initialize_arrays(double *u_k, double *r_k, const double *f, int num_values, void *stream);                                                                                             /* subdomain.okl:211-218, subdomain.tpp:4274 */
int fdd_sub_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_values, void *stream);                        /* subdomain.okl:220-227, subdomain.tpp:4541 */
int fdd_sub_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_values, void *stream);                                              /* subdomain.okl:259-266, subdomain.tpp:4563 */
/* precision-cast copies between outer (EType) and preconditioner (DType) data */
int fdd_sub_copy_f64_f64(double *u, const double *v, int num_points, void *stream); /* subdomain.okl:268-282 with DType=EType=double */
int fdd_sub_copy_f32_f64(float *u, const double *v, int num_points, void *stream);  /* subdomain.okl:268-274, DType=float */
int fdd_sub_copy_f64_f32(double *u, const float *v, int num_points, void *stream);  /* subdomain.okl:276-282, DType=float */
/* degree-tree restriction, reference three-launch form with global intermediates */
int fdd_sub_restriction_1(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim, void *stream); /* subdomain.okl:284-313, subdomain.tpp:4593,4601 */
int fdd_sub_restriction_2(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim, void *stream); /* subdomain.okl:315-344, subdomain.tpp:4596,4604 */
int fdd_sub_restriction_3(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, void *stream);          /* subdomain.okl:346-366, subdomain.tpp:4607 */
/* Fused J^T (x) J^T (x) J^T per element through LDS, 3-D: 8*(n_f^3+n_c^3) B/element */
int fdd_sub_restriction(double *u_c, const double *J_cf, const double *u_f, int num_elements, int n_f, int n_c, void *stream);
/* the `dim == 2` branches of restriction_1 and _2 (subdomain.okl:284-344) in one launch, bit-identical to the pair */
int fdd_sub_restriction_2d(double *u_c, const double *J_cf, const double *u_f, int num_elements, int n_f, int n_c, void *stream);

/* ------------------------------------------------------------------ */
/* AMG/kernels.cu + AMG/csr_matrix.cpp -- Chebyshev-smoothed V-cycle     */
/* ------------------------------------------------------------------ */
int fdd_amg_vector_set_to_value(double *data, double value, int size, void *stream);                                                   /* AMG/kernels.cu:11-23 */
int fdd_amg_main_scaled_residual(double *Sr, double *w, const double *f_m_Au, const double *S, double alpha, int size, void *stream);   /* AMG/kernels.cu:25-41 */
int fdd_amg_main_polynomial_evaluation(double *w, double *v, const double *r, const double *D_val, double alpha, int size, void *stream); /* AMG/kernels.cu:43-59 */
int fdd_amg_main_update_field(double *u, const double *w, const double *D_val, int size, void *stream);                                /* AMG/kernels.cu:61-76 */
int fdd_amg_vector_multiplication(double *uv, const double *u, const double *v, int size, void *stream);                               /* AMG/kernels.cu:79-94 */
/* y = alpha*A*x + beta*y (cusparseSpMV CSR_ALG1 at AMG/csr_matrix.cpp:129-131); y must not alias x */
int fdd_amg_matvec(double *y, const int *ptr, const int *col, const double *val, const double *x, double alpha, double beta, int num_rows, void *stream);
/* the same on a plan (LDS row staging for the ~27 non-zeros per row of the AMG levels) */
int fdd_csr_plan_matvec(const fdd_csr_plan *plan, double *y, const int *A_ptr, const int *A_col, const double *A_val, const double *x, double alpha, double beta, void *stream);
int fdd_csr_plan_matvec_to(const fdd_csr_plan *plan, double *y, const double *y_in, const int *A_ptr, const int *A_col, const double *A_val, const double *x, double alpha, double beta, void *stream); /* y = alpha*A*x + beta*y_in (y_in NULL: y itself): f - A u without first copying f */
/* The Chebyshev smoother (subdomain.tpp:19-83) with its element-wise kernels (AMG/kernels.cu:25-94) fused into the
 * SpMV in front of them, statement for statement the same arithmetic:
 *   residual:    Sr = D*(f - A u);  work = D*(coef*Sr)          [matvec(-1,1) + scaled_residual + vector_multiplication]
 *   polynomial:  work_out = D*(coef*Sr + D*(A work_in))         [matvec(1,0) + polynomial_evaluation + vector_multiplication]
 *   update:      u += D*(coef*Sr + D*(A work_in))               [matvec(1,0) + polynomial_evaluation + update_field]
 *   start:       Sr = D*f;  work = D*(coef*Sr)                  [scaled_residual from u = 0 + vector_multiplication] */
int fdd_amg_smooth_residual_matvec(const fdd_csr_plan *plan, double *work, double *Sr, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *f, const double *D_val, double coef, void *stream);
int fdd_amg_smooth_polynomial_matvec(const fdd_csr_plan *plan, double *work_out, const int *A_ptr, const int *A_col, const double *A_val, const double *work_in, const double *Sr, const double *D_val, double coef, void *stream);
int fdd_amg_smooth_update_matvec(const fdd_csr_plan *plan, double *u, const int *A_ptr, const int *A_col, const double *A_val, const double *work_in, const double *Sr, const double *D_val, double coef, void *stream);
int fdd_amg_smooth_start(double *work, double *Sr, const double *f, const double *D_val, double coef, int size, void *stream);

/* ---- matrix-free grid transfer of a geometric AMG level (csrc/fdd_transfer.hip) ------------------------------------
 * Replaces the two SpMVs with the interpolator of a level whose coarse grid is a coarsened GLL lattice (u += P e and
 * f_c = P^T v: the role of P_fem / R_fem, subdomain.tpp:3526-3545, 4064-4068, 4097-4101) where that interpolator is
 * multi-linear in the elements' reference coordinates: the same n x m table of 1-D weights in every element and direction
 * (host/low_order.hpp: geometric_level).  n lattice nodes per direction and element, m kept ones; fine node i sits between
 * kept nodes lo[i] <= hi[i] with weights wl[i] and 1 - wl[i] (lo == hi: a kept node).  lo, hi, wl: HOST arrays of n.
 *   owner_dof[num_elements * n^3]   the fine dof of a lattice point where the point is the first of its dof, -1 elsewhere
 *   coarse_dof[num_elements * m^3]  the coarse dof of an element's kept node (x fastest), -1 on a Dirichlet node
 * prolong: u[dof] += interpolated coarse value, every fine dof once.  restrict: partial[e * m^3 + t] = the element's own
 * weighted sum for its kept node t; the caller adds the partial sums of a coarse dof (a boolean gather over coarse_dof's
 * transpose: fdd_csr_plan_multiply).  The CSR interpolator's operator with its sums in another order: equal to rounding.
 * 3-D; n = 8 and n = 16 are built (fdd_lattice_supported), others are refused. */
int fdd_lattice_supported(int n, int m, int *supported);
int fdd_lattice_prolong(double *u, const double *coarse, const int *owner_dof, const int *coarse_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream);
int fdd_lattice_prolong_f32(float *u, const float *coarse, const int *owner_dof, const int *coarse_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream);
int fdd_lattice_restrict(double *partial, const double *fine, const int *owner_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream);
int fdd_lattice_restrict_f32(float *partial, const float *fine, const int *owner_dof, int n, int m, const int *lo, const int *hi, const double *wl, long long num_elements, void *stream);
/* Float = float (AMG/config.hpp:4, run.py:157): the V-cycle on f32 values and vectors.  Only the fused sequence is
 * provided (SpMV + fused smoother + set/start); casts at the V-cycle's ends: fdd_sub_copy_f32_f64 / _f64_f32.
 * Plans for these entries come from fdd_csr_plan_create_f32 (row blocks whatever the row lengths); the fp64
 * entries refuse such a plan and these refuse an fp64 one. */
int fdd_csr_plan_create_f32(fdd_csr_plan **plan, const int *A_ptr_host, int num_rows, int num_cols, int num_nnz);
int fdd_csr_plan_matvec_to_f32(const fdd_csr_plan *plan, float *y, const float *y_in, const int *A_ptr, const int *A_col, const float *A_val, const float *x, float alpha, float beta, void *stream);
int fdd_amg_smooth_residual_matvec_f32(const fdd_csr_plan *plan, float *work, float *Sr, const int *A_ptr, const int *A_col, const float *A_val, const float *u, const float *f, const float *D_val, float coef, void *stream);
int fdd_amg_smooth_polynomial_matvec_f32(const fdd_csr_plan *plan, float *work_out, const int *A_ptr, const int *A_col, const float *A_val, const float *work_in, const float *Sr, const float *D_val, float coef, void *stream);
/* the update from u = 0 (pre-smoothing): u = 0 + D*(coef*Sr + D*(A work_in)), u not read -- the bits of the update on a zeroed u */
int fdd_amg_smooth_update_matvec_from_zero(const fdd_csr_plan *plan, double *u, const int *A_ptr, const int *A_col, const double *A_val, const double *work_in, const double *Sr, const double *D_val, double coef, void *stream);
int fdd_amg_smooth_update_matvec_from_zero_f32(const fdd_csr_plan *plan, float *u, const int *A_ptr, const int *A_col, const float *A_val, const float *work_in, const float *Sr, const float *D_val, float coef, void *stream);
int fdd_amg_smooth_update_matvec_f32(const fdd_csr_plan *plan, float *u, const int *A_ptr, const int *A_col, const float *A_val, const float *work_in, const float *Sr, const float *D_val, float coef, void *stream);
int fdd_amg_smooth_start_f32(float *work, float *Sr, const float *f, const float *D_val, float coef, int size, void *stream);
int fdd_amg_vector_set_to_value_f32(float *data, float value, int size, void *stream);
/* cublasDdot replacement (AMG/vector.cpp:100,129): out[0] = sum x*y */
int fdd_amg_dot(double *out, double *ws, const double *x, const double *y, int size, void *stream);

/* ------------------------------------------------------------------ */
/* multi-vector forms of the Gram-Schmidt sweep of the Krylov solvers   */
/* (same per-element arithmetic as the launch-per-vector reference       */
/* sequence; vectors are read once)                                      */
/* ------------------------------------------------------------------ */
#define FDD_MULTI_MAX 8
/* doubles per result slot of one Arnoldi step on the device: (j + 1) projections and the norm behind them, j < FDD_MULTI_MAX
 * (the inner solve runs up to FDD_MULTI_MAX steps per cycle: run.py:151-152 sweeps 1, 2, 4, 8) */
#define FDD_GMRES_SLOT (FDD_MULTI_MAX + 2)
/* out[i] = sum a*b[i]*w, i < m <= 8: the (j+1) weighted_inner_product launches of one
 * Arnoldi step (subdomain.tpp:4389-4394).  b is a HOST array of m device pointers. */
int fdd_multi_weighted_inner_product(double *out, double *ws, const double *a, const double *const *b, int m, const double *w, int n, void *stream);
/* q = 1.0*q + coeffs[0]*v[0]; q = 1.0*q + coeffs[1]*v[1]; ... (m <= 8) in one pass: the successive
 * vector_vector_addition launches of domain.tpp:817-822,902-907 / subdomain.tpp:4396-4401,4473-4478.
 * coeffs is a HOST array, v a HOST array of device pointers (none may alias q). */
int fdd_multi_axpy(double *q, const double *coeffs, const double *const *v, int m, int n, void *stream);
/* y += sign * sum_k coeffs_dev[k] * x_k and out[0] = sum y*y*w of the result, one pass, coefficients read
 * from device memory (the output of fdd_multi_weighted_inner_product): Gram-Schmidt update + norm with no
 * host round trip in between (subdomain.tpp:4396-4416). */
int fdd_multi_axpy_norm2_dev(double *out, double *ws, double *y, const double *coeffs_dev, double sign, const double *const *x, int m, const double *w, int n, void *stream);
/* au = (1 / sqrt(*norm2_dev)) * u (subdomain.tpp:4457 with the norm still on the device) */
int fdd_vector_scaling_rsqrt_dev(double *au, const double *norm2_dev, const double *u, int n, void *stream);
int fdd_multi_axpy_dev(double *q, const double *coeffs_dev, const double *const *v, int m, int n, void *stream); /* fdd_multi_axpy, coefficients in device memory */

/* Scalar bookkeeping of one restart cycle of the inner flexible GMRES(m) (subdomain.tpp:4396-4477) on the device:
 * Hessenberg column + Givens rotations + residual recurrence + stopping tests per step, back-substitution at the
 * end, so that a cycle is enqueued without a host round trip per step.  `state` = fdd_gmres_state_bytes() bytes of
 * device memory.  A stop is recorded, not acted on: later steps of the cycle run on vectors nobody uses, and
 * fdd_gmres_fetch reports the columns 0..j_last the reference would have used.
 *   begin : gamma[0] = sqrt(*norm2_dev) (also the relative-test norm when first_cycle)
 *   step j: dots_dev[0..j] = <q, v_i>, dots_dev[j+1] = ||q - sum_i h_i v_i||^2
 *   finish: y = H^-1 gamma on columns 0..j_last; fdd_gmres_coefficients gives the device pointer to y[FDD_MULTI_MAX] */
size_t fdd_gmres_state_bytes(void);
int fdd_gmres_begin_dev(void *state, const double *norm2_dev, int first_cycle, void *stream);
int fdd_gmres_step_dev(void *state, const double *dots_dev, int j, int iterations_before, int max_iterations, double tolerance, int use_relative, void *stream);
int fdd_gmres_finish_dev(void *state, int m, void *stream);
int fdd_gmres_fetch(void *state, double *y, double *hist, int *num_hist, int *j_last, int *steps, int *converged, void *stream);
int fdd_gmres_coefficients(void *state, const double **y_dev);
/* the basis may stay unnormalised: inv[0] = 1/gamma_0, inv[j+1] = 1/||q_j|| are kept in the state (the factors
 * vector_scaling would have applied, subdomain.tpp:4358, 4457) for the *_scaled entries below to apply on load */
int fdd_gmres_scales(void *state, const double **inv_dev);
int fdd_gmres_last_column(void *state, const double **j_last_dev); /* device address of j_last (a double), for fdd_multi_lincomb_limited_dev */
int fdd_sub_stiffness_matrix_gather_scaled(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream);
/* in the multi-vector reductions below w == NULL means unit weights: nothing is read, and x * 1.0 is x bit for bit */
int fdd_multi_weighted_inner_product_scaled(double *out, double *ws, const double *a, const double *const *b, const double *b_scale_dev, int m, const double *w, int n, void *stream);
int fdd_multi_axpy_norm2_scaled_dev(double *out, double *ws, double *dst, const double *y, const double *coeffs_dev, double sign, const double *const *x, const double *x_scale_dev, int m, const double *w, int n, void *stream); /* dst == NULL (also in the _f32 form): the updated vector is not stored, only its norm is formed -- the last Arnoldi step of a cycle, whose basis vector nobody reads */
int fdd_multi_axpy_scaled_dev(double *q, const double *coeffs_dev, const double *const *v, const double *v_scale_dev, int m, int n, void *stream);
int fdd_vector_scaling_dev(double *au, const double *scale_dev, const double *u, int n, void *stream); /* au = (*scale_dev) * u */
int fdd_multi_lincomb_scaled_dev(double *q, int q_is_zero, const double *coeffs_dev, const double *const *v, const double *v_scale_dev, int m, int n, void *stream); /* fdd_multi_axpy_scaled_dev; q_is_zero: q is taken to be 0 and is not read (it need not have been cleared) */
int fdd_multi_lincomb_limited_dev(double *q, int q_is_zero, const double *coeffs_dev, const double *const *v, const double *v_scale_dev, const double *last_dev, int m, int n, void *stream); /* only vectors 0..(int)*last_dev enter (NULL: all m) */
int fdd_sqrt_sum_dev(double *out, const double *parts_dev, int nparts, void *stream); /* out[0] = sqrt(sum parts): a residual norm appended to a device-side history */
/* ---- single-precision preconditioner (the reference's PTYPE = Float = float, config.hpp:19-20, poisson.cpp:206): the
 * kernels of the dof-space inner solve on float vectors.  Device-resident scalars and reduction accumulators stay double. ---- */
/* The same operator on elements that are AFFINE images of the reference cube (every element of a box mesh), an option of
 * this build: the six factors of a point are formed from six numbers per element and the GLL weights,
 *   G_f(e; i, j, k) = elem_factors[6 e + f] * (w_i w_j) w_k,     gll_weights[0 .. poly_degree],
 * instead of being streamed (48 of the 64 bytes per point of domain.okl:5-98 are not read).  point_dof == NULL: u = v
 * point by point (Domain); otherwise gathered and scaled as above (v_scale_dev may be NULL).  The host layer uses it only
 * where the mesh's own factor arrays have that form to rounding (Stiffness_Operator::affine); results agree with the
 * streamed form to a few ulp of the factors, not bit for bit. */
int fdd_stiffness_matrix_affine(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *elem_factors, const double *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream);
/* the same on the matrix-core kernel (poly_degree 8..15; Au != v) */
int fdd_stiffness_matrix_mfma_affine(double *Au, const double *v, const double *v_scale_dev, const int *point_dof, const double *D_hat, const double *elem_factors, const double *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream);
/* Do the factor arrays have that form?  Per element e of the list: elem_factors[6 e + f] = G_f / W at the element's middle
 * point, deviation[e] = max over points and factors of |G_f(p) - elem_factors[6 e + f] W(p)| / (max_f |elem_factors| W(p)),
 * W(p) = (w_i w_j) w_k.  3-D elements. */
int fdd_stiffness_affine_detect(double *elem_factors, double *deviation, const double *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, const double *gll_weights, int num_elements, int poly_degree, void *stream);
int fdd_stiffness_matrix_affine_f32(float *Au, const float *v, const double *v_scale_dev, const int *point_dof, const float *D_hat, const float *elem_factors, const float *gll_weights, const int *elem_offset, int num_elements, int poly_degree, void *stream);
int fdd_sub_stiffness_matrix_gather_scaled_f32(float *Au, const float *v, const double *v_scale_dev, const int *point_dof, const float *D_hat, const float *const G[FDD_NUM_GEOM_FACTS], const int *elem_offset, int num_elements, int poly_degree, void *stream);
int fdd_multi_inner_product_scaled_f32(double *out, double *ws, const float *a, const float *const *b, const double *b_scale_dev, int m, int n, void *stream); /* out[k] = sum a * (s_k b_k), k < m <= 8 */
int fdd_multi_axpy_norm2_scaled_dev_f32(double *out, double *ws, float *dst, const float *y, const double *coeffs_dev, double sign, const float *const *x, const double *x_scale_dev, int m, int n, void *stream); /* dst = y + sign sum c_k (s_k x_k); out = |dst|^2 */
int fdd_vector_scaling_dev_f32(float *au, const double *scale_dev, const float *u, int n, void *stream);
/* z = d .* ((*scale_dev) * u), scale_dev NULL: z = d .* u.  The point-Jacobi preconditioner of the inner solve (a labelled
 * option of this build, host/subdomain.hpp) on a Krylov vector kept unnormalised: math.okl:29-35, then
 * AMG/kernels.cu:64-73 (vector_multiplication), in that order */
int fdd_vector_diagonal_scaling_dev(double *z, const double *d, const double *scale_dev, const double *u, int n, void *stream);
int fdd_vector_diagonal_scaling_dev_f32(float *z, const float *d, const double *scale_dev, const float *u, int n, void *stream);
int fdd_vector_vector_addition_f32(float *uv, float alpha, const float *u, float beta, const float *v, int n, void *stream);
int fdd_multi_lincomb_limited_dev_f32(float *q, int q_is_zero, const double *coeffs_dev, const float *const *v, const double *v_scale_dev, const double *last_dev, int m, int n, void *stream); /* q (+)= sum_{k <= *last} c_k (s_k v_k); last_dev NULL: all m */
int fdd_gather_rows_f32(float *t, const int *ptr, const int *col, const float *u, int row_lo, int row_hi, void *stream); /* t[row] = sum of u over the row's entries (boolean gather) */
/* fdd_gather_rows_f32 on the row blocks of a plan (LDS-staged, coalesced index stream): same sums in the same order */
int fdd_csr_plan_gather_f32(const fdd_csr_plan *plan, float *t, const int *ptr, const int *col, const float *u, int row_lo, int row_hi, void *stream);
int fdd_gather_indexed_f32(float *out, const float *in, const int *index, int n, void *stream);         /* out[i] = in[index[i]], 0 where index[i] < 0 */
int fdd_gather_indexed_f32_f64(double *out, const float *in, const int *index, int n, void *stream);    /* the same, cast up (subdomain.okl:276-282 at the solve's exit) */
int fdd_xmay_ratio_dev(double *out, const double *x, const double *num_dev, const double *den_dev, const double *y, int n, void *stream); /* out = x - (*num / *den) * y (domain.okl:191: r+ = r - alpha q with alpha on the device; out may be x) */
int fdd_xpby_ratio_dev(double *out, const double *x, const double *num_dev, const double *den_dev, const double *y, int n, void *stream); /* out = x + (*num / *den) * y (domain.okl:226: p = z + beta p; out may be y) */
/* out[0] = sum_nodes s*s*w with s = (Qt u)[node]*w[node]: Subdomain::residual_norm
 * (subdomain.tpp:4491-4515: multiply_weight + weighted_inner_product) without the dof vector */
int fdd_gather_weighted_norm2(double *out, double *ws, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, int num_nodes, void *stream);

/* ------------------------------------------------------------------ */
/* fused direct-stiffness summation (gather-scatter)                    */
/* ------------------------------------------------------------------ */
/* direct_stiffness_summation (domain.tpp:582-600, subdomain.tpp:3969-3985) is
 *   t = (Qt u) .* node_weight ; [gs_add on the boundary prefix] ; out = (Q t) .* point_mask
 * with boolean Qt (all values 1.0) and Q = Qt^T.  One lane per assembled node
 * gathers its points and scatters the sum back: ~36 B/point instead of two
 * SpMVs (66-74 B/point), same arithmetic, bit-identical.  node_weight /
 * point_mask / t may be NULL.  QQtu may alias u.  Nodes [node_start, node_end). */
int fdd_dssum_fused(double *QQtu, double *t, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, const double *point_mask, int node_start, int node_end, void *stream);
int fdd_dssum_gather(double *t, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, int node_start, int node_end, void *stream);      /* t = (Qt u) .* w on a node range */
int fdd_dssum_scatter(double *QQtu, const double *t, const int *Qt_ptr, const int *Qt_col, const double *point_mask, int node_start, int node_end, void *stream);  /* out = (Q t) .* mask on a node range */
int fdd_fill_indexed(double *out, const int *idx, double value, int n, void *stream); /* out[idx[i]] = value (points without a dof) */
int fdd_gather_indexed(double *out, const double *in, const int *index, const double *scale, int n, void *stream); /* out[i] = in[index[i]] * scale[i] (0 where index[i] < 0; scale may be NULL) */
/* The same (no scale) from a vector whose head [0, split) lives in `lo` and whose tail lives in `hi` (indexed by the
 * same, unshifted index): out[i] = (index[i] < split ? lo : hi)[index[i]].  Packs the ring data of tree_operator's
 * pull (subdomain.tpp:4626) from the caller's level-0 vector and the restricted levels without copying the former. */
int fdd_gather_indexed_split(double *out, const double *lo, const double *hi, int split, const int *index, int n, void *stream);
/* y[index[i]] += t[i], distinct indices: the result of a transposed SpMV whose non-empty rows are few, added into the
 * full vector (the hanging-point rows S^T of the composite region, subdomain.tpp:1522-1578 transposed). */
int fdd_scatter_add_indexed(double *y, const int *index, const double *t, int n, void *stream);
int fdd_scatter_add_indexed_f32(float *y, const int *index, const float *t, int n, void *stream);
/* The same three operations on the row blocks of Qt's SpMV plan (unit-value plans only): entries are
 * staged through LDS so that no global access depends on a row length.  mode 0 = gather + scatter,
 * 1 = gather only (t out), 2 = scatter only (t in); nodes [row_lo, row_hi).  Same bits as above. */
int fdd_csr_plan_dssum(const fdd_csr_plan *plan, double *QQtu, double *t, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, const double *point_mask, int row_lo, int row_hi, int mode, void *stream);
int fdd_csr_plan_gather_weighted_norm2(const fdd_csr_plan *plan, double *out, double *ws, const int *Qt_ptr, const int *Qt_col, const double *u, const double *node_weight, void *stream);

/* ------------------------------------------------------------------ */
/* interface exchange helpers: gslib gs(gs_add) on the boundary-node    */
/* prefix (domain.tpp:590-594) becomes pack -> all-reduce -> unpack on a */
/* dense interface-slot vector                                           */
/* ------------------------------------------------------------------ */
int fdd_interface_pack(double *slots, const int *slot_of, const double *prefix, int n, void *stream);   /* slots[slot_of[i]] = prefix[i]; slot_of is injective */
int fdd_interface_unpack(double *prefix, const double *slots, const int *slot_of, int n, void *stream); /* prefix[i] = slots[slot_of[i]] */
/* neighbour form of gs(gs_add) (domain.tpp:590-594) for point-to-point links: buf[i*nc + c] = {a, b}[c][index[i]] (nc = b ? 2 : 1) fills the
 * own copies and the parts sent to the peers; after the grouped send / receive, {a, b}[c][r] = sum_k buf[col[k]*nc + c] over ptr[r] <= k < ptr[r+1],
 * in the order of col (ascending rank of the contributor) */
int fdd_interface_gather(double *buf, const int *index, int n, const double *a, const double *b, void *stream);
int fdd_interface_sum(double *a, double *b, const int *ptr, const int *col, int rows, const double *buf, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* FDD_HIP_H */
