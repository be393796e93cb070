/*
 * fdd_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement, with OCCA-Serial semantics, of the reference's
 * preconditioned-CG hot path.  Every function cites the reference file:line
 * it follows (paths relative to /root/reference).  "OCCA-Serial semantics"
 * means: @outer/@inner loops run sequentially, @shared arrays are per-@outer
 * scratch, so block reductions are the exact 128-wide pairwise tree
 * (alive = 64, 32, ..., 1) followed by an in-order sum of the block partials
 * on the host.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library, and only as the checker.  The product
 * (libfdd_hip.so / libfdd_host.so) never links it.
 *
 * PARITY PIN: the reference ships no tests, golden vectors or fixtures for
 * this path, and cannot be built here (OCCA, HYPRE,
 * gslib, MPI, CUDA absent).  What IS pinned by the reference itself: the GLL
 * tables (nodes, weights, D_hat, J_cf) against special_functions.f compiled
 * with flang (oracle/_ref, tests/golden/gll_tables.json).  Kernel and solver
 * arithmetic is "parity unpinned" upstream: this restatement is the contract.
 *
 * Compile with -O2 -ffp-contract=off (x86-64 baseline has no FMA, matching
 * the reference Makefile's plain `-O2` g++ build of the OCCA-Serial path).
 */
#ifndef FDD_ORACLE_H
#define FDD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BLOCK_SIZE 128 /* AMG/config.hpp:5 (wins over config.hpp:38-40) */
#define ORC_NUM_GEOM_FACTS 6 /* element.hpp:10-12 */

/* ------------------------------------------------------------------ */
/* csr_matrix.okl                                                       */
/* ------------------------------------------------------------------ */
void orc_csr_multiply(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int n);
void orc_csr_multiply_range(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, int row_start, int row_end);
void orc_csr_multiply_weight(double *Au, const int *A_ptr, const int *A_col, const double *A_val, const double *u, const double *weight, int n);

/* ------------------------------------------------------------------ */
/* math.okl                                                             */
/* ------------------------------------------------------------------ */
void orc_set_to_value(double *u, double alpha, int n, int offset);
void orc_invert_vector_elements(double *u, int n);
void orc_vector_vector_addition(double *uv, double alpha, const double *u, double beta, const double *v, int n);
void orc_vector_scaling(double *au, double alpha, const double *u, int n);

/* ------------------------------------------------------------------ */
/* domain.okl                                                           */
/* ------------------------------------------------------------------ */
void orc_dom_stiffness_matrix_1(double *const GDu[3], const double *u, const double *D_hat, const double *const G[6], int num_points, int poly_degree, int dim);
void orc_dom_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *D_hat, int num_points, int poly_degree, int dim);
void orc_dom_initialize_arrays(double *u_k, double *r_k, const double *f, int num_points);
void orc_dom_residual_norm(double *block, const double *r_k, const double *QQt_r_k, const double *dirichlet_mask, int num_points, int num_blocks);
void orc_dom_projection_inner_products(double *block, const double *z_k, const double *r_k, const double *p_k, const double *q_k, int num_points, int num_blocks);
void orc_dom_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_points);
void orc_dom_inner_product_flexible(double *block, const double *r_k, const double *r_kp1, const double *z_k, int num_points, int num_blocks);
void orc_dom_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_points);
void orc_dom_inner_product(double *block, const double *u_k, const double *v_k, const double *dirichlet_mask, int num_points, int num_blocks);

/* ------------------------------------------------------------------ */
/* subdomain.okl                                                        */
/* ------------------------------------------------------------------ */
void orc_sub_stiffness_matrix_1(double *const GDu[3], const double *u, const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, const double *const G[6], int num_points, int dim);
void orc_sub_stiffness_matrix_2(double *Au, const double *const GDu[3], const double *const *D_hat_ptr, const int *offset, const int *vert, const int *level, const int *poly_degree, int num_points, int dim);
void orc_sub_inner_product(double *block, const double *u, const double *v, int num_values, int num_blocks);
void orc_sub_weighted_inner_product(double *block, const double *u, const double *v, const double *w, int num_values, int num_blocks);
void orc_sub_projection_inner_products(double *block, const double *z_k, const double *r_k, const double *p_k, const double *q_k, const double *weight, int num_values, int num_blocks);
void orc_sub_initialize_arrays(double *u_k, double *r_k, const double *f, int num_values);
void orc_sub_solution_and_residual_update(double *u_k, double *r_kp1, const double *r_k, const double *p_k, const double *q_k, double alpha_k, int num_values);
void orc_sub_search_update_inner_product(double *block, const double *r_k, const double *r_kp1, const double *z_k, const double *weight, int num_points, int num_blocks);
void orc_sub_residual_and_search_update(double *p_k, double *r_k, const double *z_k, const double *r_kp1, double beta_k, int num_values);
void orc_sub_copy_f64_f64(double *u, const double *v, int num_points);
void orc_sub_copy_f32_f64(float *u, const double *v, int num_points);
void orc_sub_copy_f64_f32(double *u, const float *v, int num_points);
void orc_sub_restriction_1(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim);
void orc_sub_restriction_2(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c, int dim);
void orc_sub_restriction_3(double *Ju, const double *J_cf, const double *u, int num_points, int n_f, int n_c);

/* ------------------------------------------------------------------ */
/* AMG/kernels.cu + AMG/csr_matrix.cpp + subdomain.tpp:19-83            */
/* ------------------------------------------------------------------ */
void orc_amg_vector_set_to_value(double *data, double value, int size);
void orc_amg_main_scaled_residual(double *Sr, double *w, const double *f_m_Au, const double *S, double alpha, int size);
void orc_amg_main_polynomial_evaluation(double *w, double *v, const double *r, const double *D_val, double alpha, int size);
void orc_amg_main_update_field(double *u, const double *w, const double *D_val, int size);
void orc_amg_vector_multiplication(double *uv, const double *u, const double *v, int size);
void orc_amg_matvec(double *y, const int *ptr, const int *col, const double *val, const double *x, double alpha, double beta, int num_rows);
/* host-branch Chebyshev smoother pieces (subdomain.tpp:19-83, "host" branches) */
void orc_amg_scaled_residual_host(double *Sr, double *w, const int *ptr, const int *col, const double *val, const double *u, const double *f, const double *S, double alpha, int num_rows);
void orc_amg_polynomial_evaluation_host(double *w, double *v, const int *ptr, const int *col, const double *val, const double *r, const double *D_val, double alpha, int num_rows);

/* host-side in-order block sum (domain.tpp:926, 943, 960-964, 991-992) */
double orc_block_sum(const double *block, int num_blocks);

/* ------------------------------------------------------------------ */
/* CSR_Matrix host class (csr_matrix.tpp)                               */
/* ------------------------------------------------------------------ */
typedef struct orc_csr
{
    int num_rows, num_cols, num_nnz;
    int *ptr;
    int *col;
    double *val;
} orc_csr;

/* add_entry + assemble (csr_matrix.tpp:70-81, 94-180).  `n_entries` COO
 * triples in insertion order.  Returns 0, or -1 for an out-of-range entry
 * (the reference prints and exits).  Entries with |v| <= 1e-12 are dropped. */
int orc_csr_assemble(orc_csr *A, int num_rows, int num_cols, const int *rows, const int *cols, const double *vals, long n_entries);
void orc_csr_transpose(const orc_csr *A, orc_csr *At);   /* csr_matrix.tpp:228-258 */
void orc_csr_diagonal(const orc_csr *A, double *D);       /* csr_matrix.tpp:261-299 */
void orc_csr_free(orc_csr *A);

/* ------------------------------------------------------------------ */
/* Domain<double> (domain.tpp), R ranks simulated in one process        */
/* ------------------------------------------------------------------ */
typedef struct orc_domain orc_domain;
typedef struct orc_world orc_world;

/* Mesh arrays exactly as Domain::initialize reads them (domain.tpp:45-224),
 * for ONE rank: element-major, (N+1)^dim values per element. */
typedef struct orc_mesh
{
    int dim;
    int poly_degree;
    int num_local_elements;
    const double *x, *y, *z;         /* may be NULL (only initial_function uses them) */
    const long long *glo_num;        /* 1-based global node ids */
    const int *node_degree;          /* global multiplicity per local point */
    const double *p_mask;            /* 0 on Dirichlet boundary else 1 */
    const double *g[ORC_NUM_GEOM_FACTS];
} orc_mesh;

/* D_hat: (N+1)^2 row-major, D_hat[k + i*n] = dl_k/dxi(xi_i) (domain.tpp:311-316). */
orc_world *orc_world_create(int num_ranks, const orc_mesh *meshes, const double *D_hat);
void orc_world_destroy(orc_world *w);

int orc_world_num_ranks(const orc_world *w);
int orc_world_num_local_points(const orc_world *w, int rank);
int orc_world_num_local_nodes(const orc_world *w, int rank);
int orc_world_num_bdary_nodes(const orc_world *w, int rank);
int orc_world_num_interface_slots(const orc_world *w);
const orc_csr *orc_world_Q(const orc_world *w, int rank);
const orc_csr *orc_world_Qt(const orc_world *w, int rank);
const double *orc_world_assembled_weight(const orc_world *w, int rank);

/* The vectors are arrays-of-pointers, one per rank.
 * direct_stiffness_summation: domain.tpp:582-600. */
void orc_world_dssum(orc_world *w, double *const *QQtu, const double *const *u, int apply_mask, int apply_weight);
/* stiffness_matrix: domain.tpp:602-609 */
void orc_world_stiffness(orc_world *w, double *const *Au, const double *const *u, int apply_dssum);
/* residual_norm / assembled_inner_product: domain.tpp:916-947 */
double orc_world_residual_norm(orc_world *w, const double *const *r);
double orc_world_assembled_inner_product(orc_world *w, const double *const *u, const double *const *v);

typedef void (*orc_precond_fn)(void *ctx, double *const *z, const double *const *r);

typedef struct orc_solver_opts
{
    int max_iterations;      /* domain.hpp:115 (500) */
    int num_vectors;         /* domain.hpp:114 (20)  */
    double tolerance;        /* domain.hpp:118 (1e-7) */
    int use_relative;        /* default true */
    orc_precond_fn precond;  /* NULL => use_preconditioner=false path (domain.tpp:648-651) */
    void *precond_ctx;
} orc_solver_opts;

/* flexible_conjugate_gradient (domain.tpp:611-725).  history[0..*num_hist)
 * receives the residual norms printed by the reference (iteration 0 first);
 * returns num_iterations. */
int orc_world_fcg(orc_world *w, double *const *u, const double *const *f, const orc_solver_opts *opts, double *history, int history_cap, int *num_hist);
/* generalized_minimum_residual (domain.tpp:727-914) */
int orc_world_gmres(orc_world *w, double *const *u, const double *const *f, const orc_solver_opts *opts, double *history, int history_cap, int *num_hist);

/* ------------------------------------------------------------------ */
/* Subdomain<double> solve path, conforming composite                    */
/* (subdomain.tpp:3942-4646 on operators built as subdomain.tpp:86-2747  */
/*  builds them when every region element is conforming: single rank, or */
/*  block-local = own elements only)                                      */
/* ------------------------------------------------------------------ */
typedef struct orc_subdomain orc_subdomain;

typedef struct orc_subdomain_opts
{
    int num_vectors;        /* subdomain.hpp:229 (4) */
    int max_iterations;     /* subdomain.hpp:230 (4) */
    double tolerance;       /* subdomain.hpp:232 (1e-12) */
    int use_preconditioner; /* 0: direct_stiffness_summation identity path, 1: low-order AMG V-cycle, 2: point-Jacobi (labelled option of the build, not in the reference) */
} orc_subdomain_opts;

/* levels: poly degrees N, N-r, ..., 1 (subdomain.tpp:98-110); D_hat[l] and
 * J_cf[(l_c, l_f)] tables are inputs (from GLL tables).  J_cf_tables[l] is the
 * interpolator from level l+1 (coarse) to level l (fine), n_f x n_c row-major
 * (subdomain.tpp:153-159).  level_meshes[l] is this rank's mesh at degree
 * poly_degree[l] (the reference builds one Domain per level, poisson.cpp:176-199). */
orc_subdomain *orc_subdomain_create(int num_levels, const int *poly_degree, const double *const *D_hat, const double *const *J_cf_tables, const orc_mesh *level_meshes);
void orc_subdomain_destroy(orc_subdomain *s);
int orc_subdomain_num_values(const orc_subdomain *s);
int orc_subdomain_num_dofs(const orc_subdomain *s);
/* tree_operator (subdomain.tpp:4566-4646), single-region form */
void orc_subdomain_tree_operator(orc_subdomain *s, double *Tu, const double *u);
void orc_subdomain_stiffness(orc_subdomain *s, double *Au, const double *u);   /* :3942-3967 */
void orc_subdomain_dssum(orc_subdomain *s, double *QQtu, const double *u);     /* :3969-3985 */
double orc_subdomain_residual_norm(orc_subdomain *s, const double *r);         /* :4491-4515 */
/* generalized_minimum_residual (subdomain.tpp:4309-4489); returns iterations */
int orc_subdomain_gmres(orc_subdomain *s, double *u_l, const double *f_l, const orc_subdomain_opts *opts, double *history, int history_cap, int *num_hist);
/* flexible_conjugate_gradient (subdomain.tpp:4161-4268) */
int orc_subdomain_fcg(orc_subdomain *s, double *u_l, const double *f_l, const orc_subdomain_opts *opts, double *history, int history_cap, int *num_hist);

/* ------------------------------------------------------------------ */
/* The full-domain-decomposition composite (subdomain.tpp:86-2747) of    */
/* every rank of an R-rank run, and the exchange half of tree_operator   */
/* (subdomain.tpp:4613-4645): fdd_oracle_composite.c                     */
/* ------------------------------------------------------------------ */
typedef struct orc_fdd orc_fdd;
/* meshes[rank * num_levels + level]: rank's mesh at degree poly_degree[level];
 * J_cf_pairs[l_f * num_levels + l_c] (l_f < l_c): interpolator from level l_c to
 * level l_f, n_f x n_c row-major (subdomain.tpp:142-164) */
orc_fdd *orc_fdd_create(int num_ranks, int num_levels, const int *poly_degree, const double *const *D_hat, const double *const *J_cf_pairs, const orc_mesh *meshes, int subdomain_overlap, int superdomain_overlap);
void orc_fdd_destroy(orc_fdd *F);
orc_subdomain *orc_fdd_subdomain(orc_fdd *F, int rank); /* the rank's composite: orc_subdomain_stiffness / _dssum / _residual_norm apply */
/* info[12]: sub elems, sub extended elems, points, sub dofs, sub extended dofs, interface dofs, sup dofs,
 * sup extended dofs, unique dofs, coarse dofs, num_values, own points */
void orc_fdd_info(const orc_fdd *F, int rank, int *info);
void orc_fdd_region(const orc_fdd *F, int rank, int *id, int *level); /* global element and level of every region element */
const int *orc_fdd_composite_levels(const orc_fdd *F, int rank);       /* composite dofs per coarsening level, -1 terminated */
/* which: 0 Q, 1 Qt, 2 Q_int, 3 Qt_int, 4 QQt_int, 5 superdomain A, 6 superdomain Pt, 7 Qt_coarse */
const orc_csr *orc_fdd_matrix(const orc_fdd *F, int rank, int which);
/* the low-order operator of the composite over its unique dofs (subdomain.tpp:2749-3472; 3-D) */
const orc_csr *orc_fdd_low_order_matrix(const orc_fdd *F, int rank);
/* 0-based subdomain dof of every region point that carries one directly (-1: Dirichlet or hanging) */
const int *orc_fdd_point_dofs(const orc_fdd *F, int rank);
const double *orc_fdd_norm_weight(const orc_fdd *F, int rank);  /* sub extended + sup extended dofs */
const double *orc_fdd_inner_weight(const orc_fdd *F, int rank); /* num_values */
void orc_fdd_tree_operator(orc_fdd *F, double *const *Tu, const double *const *u);
/* z = M^-1 r on every rank; method 0 flexible CG, 1 GMRES; history: num_ranks rows of history_cap */
void orc_fdd_precondition(orc_fdd *F, double *const *z, const double *const *r, int method, const orc_subdomain_opts *opts, double *history, int history_cap, int *num_hist);

/* ------------------------------------------------------------------ */
/* Low-order AMG V-cycle given a hierarchy (subdomain.tpp:19-83,        */
/* 3987-4159); the hierarchy itself comes from HYPRE in the reference   */
/* and is an input here (fdd_oracle_amg.c)                              */
/* ------------------------------------------------------------------ */
typedef struct orc_amg orc_amg;
orc_amg *orc_amg_create(int num_levels, int cheby_order, int num_vcycles);
void orc_amg_set_level(orc_amg *a, int l, int n, const int *A_ptr, const int *A_col, const double *A_val, const double *D_val, const double *coefs, int n_coarse, const int *P_ptr, const int *P_col, const double *P_val);
void orc_amg_destroy(orc_amg *a);
int orc_amg_level_size(const orc_amg *a, int l);
void orc_amg_vcycle(orc_amg *a, double *u0, const double *f0);
void orc_subdomain_attach_amg(orc_subdomain *s, orc_amg *amg);
void orc_subdomain_point_dofs(const orc_subdomain *s, int *dof);
/* point-Jacobi option: diagonal of Qt_int Qt' A Q' Q_int over the unique dofs; and the closed-form element diagonal against the kernels (max relative error) */
void orc_subdomain_jacobi_diagonal(const orc_subdomain *s, double *diag);
double orc_subdomain_element_diagonal_check(const orc_subdomain *s);
void orc_subdomain_low_order_preconditioner(orc_subdomain *s, double *z, const double *r);

/* the same V-cycle with Float = float (AMG/config.hpp:4; fdd_oracle_amg_f32.c): arrays given in double are rounded */
typedef struct orc_amg32 orc_amg32;
orc_amg32 *orc_amg32_create(int num_levels, int cheby_order, int num_vcycles);
void orc_amg32_set_level(orc_amg32 *a, int l, int n, const int *A_ptr, const int *A_col, const double *A_val, const double *D_val, const double *coefs, int n_coarse, const int *P_ptr, const int *P_col, const double *P_val);
void orc_amg32_destroy(orc_amg32 *a);
void orc_amg32_vcycle(orc_amg32 *a, double *u0, const double *f0);

/* the inner solve's kernels with PTYPE = Float = float (config.hpp:19-20; fdd_oracle_f32.c): IEEE single throughout,
 * except the reductions, which restate this build's double accumulators over float data */
void orc_f32_sub_stiffness(float *Au, const float *v, const int *index, const float *scale, const float *D_hat, const float *const G[6], int num_elements, int poly_degree);
void orc_f32_vector_vector_addition(float *uv, float alpha, const float *u, float beta, const float *v, int n);
void orc_f32_vector_scaling(float *au, double alpha, const float *u, int n);
void orc_f32_csr_gather(float *t, const int *ptr, const int *col, const float *u, int row_lo, int row_hi);
void orc_f32_gather_indexed(float *out, const float *in, const int *index, int n);
void orc_f32_gather_indexed_f64(double *out, const float *in, const int *index, int n);
void orc_f32_multi_inner_product_scaled(double *out, const float *a, const float *const *b, const double *b_scale, int m, int n);
double orc_f32_multi_axpy_norm2_scaled(float *dst, const float *y, const double *c, double sign, const float *const *x, const double *x_scale, int m, int n);
void orc_f32_multi_lincomb(float *q, int q_is_zero, const double *c, const float *const *v, const double *v_scale, int last, int m, int n);

#ifdef __cplusplus
}
#endif

#endif
