/*
 * fdd_host.h -- C-ABI of libfdd_host.so: the C++ host classes
 * (CSR_Matrix / Math / Domain / Subdomain, mirrors of the reference's
 * csr_matrix.hpp / math.hpp / domain.hpp / subdomain.hpp) and what the
 * reference's driver does with them (poisson.cpp:150-251: one Domain per
 * polynomial level, a Subdomain preconditioner, manufactured right-hand side,
 * outer flexible CG / GMRES), exposed with plain pointers so that tests,
 * bench.py and foreign-language hosts can drive it.
 *
 * Vectors cross this boundary as HOST arrays of num_local_points doubles
 * (element-major, the layout of the mesh files); the solver works on device
 * copies.  Every entry returns 0 on success; fddh_last_error() has the text.
 */
#ifndef FDD_HOST_H
#define FDD_HOST_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *fddh_last_error(void);

/* Device and stream of this rank (OCCA_Initialize, poisson.cpp:127-148).
 * own_stream != 0 creates a private non-blocking stream (stream is ignored);
 * otherwise every kernel runs on `stream` exactly as given -- NULL is the
 * default stream.  Pass torch's current stream to share ordering with
 * torch.distributed collectives. */
int fddh_init(int device, void *stream, int own_stream);
int fddh_set_print(int on); /* rank-0 residual history lines (rstdout) */
int fddh_set_timer(int on); /* named timing regions (timer.hpp); off by default */
int fddh_timer_total(const char *key, double *seconds);
/* the same aggregated over the ranks, "max" (the reference's table, poisson.cpp:256, timer.tpp:67) or "sum": collective */
int fddh_timer_total_over_ranks(const char *key, const char *aggregation, double *seconds);

/* Communicator (MPI_Initialize, poisson.cpp:84-89).  Exactly one of: */
int fddh_comm_single(void);
int fddh_comm_rccl_unique_id(char *out128);                        /* rank 0; ship the 128 bytes to all ranks */
int fddh_comm_rccl_init(const char *id128, int rank, int size);    /* RCCL over xGMI, called directly */
typedef int (*fddh_allreduce_fn)(void *ctx, void *buf, long long n);
typedef int (*fddh_allgather_fn)(void *ctx, const void *send, void *recv, long long bytes);
typedef int (*fddh_barrier_fn)(void *ctx);
int fddh_comm_callbacks(int rank, int size, void *ctx, fddh_allreduce_fn allreduce_sum_f64, fddh_allreduce_fn allreduce_max_f64, fddh_allgather_fn allgather_bytes, fddh_barrier_fn barrier);
/* The same with the point-to-point exchange the composite region needs (the reference's gslib pull of neighbour
 * elements, subdomain.tpp:601-642, 4626): one grouped call, n peers; send_bytes[i] from send[i] go to peers[i] and
 * recv_bytes[i] from peers[i] arrive in recv[i] (device buffers of this rank; either side may be 0). */
typedef int (*fddh_exchange_fn)(void *ctx, int n, const int *peers, const void *const *send, const long long *send_bytes, void *const *recv, const long long *recv_bytes);
int fddh_comm_callbacks_ex(int rank, int size, void *ctx, fddh_allreduce_fn allreduce_sum_f64, fddh_allreduce_fn allreduce_max_f64, fddh_allgather_fn allgather_bytes, fddh_barrier_fn barrier, fddh_exchange_fn exchange_bytes);
/* Several ranks inside ONE process, each on its own host thread (every fddh_* entry acts on the calling thread's rank:
 * device stream, communicator, timer and print switches are per thread), sharing one GPU: collectives are device-to-device
 * copies and rank-ordered sums between the ranks' buffers.  For rehearsing the N-rank path where RCCL cannot be used (it
 * admits one rank per device): create the world once, then every rank thread calls fddh_init (own stream) and
 * fddh_comm_local with its rank. */
int fddh_local_world_create(void **world, int size);
int fddh_local_world_destroy(void *world);
/* a rank of the world failed outside a collective (its thread raised): wake the peers waiting for it with an error now */
int fddh_local_world_fail(void *world);
/* end of a rank thread: release what fddh_init(own_stream) / fddh_comm_* gave the calling thread (stream, communicator) */
int fddh_rank_finalize(void);
int fddh_comm_local(void *world, int rank);
int fddh_comm_info(int *rank, int *size, char *name, size_t name_len);
/* run every collective of the active communicator once on n doubles and verify the results */
int fddh_comm_selftest(int n);

/* A "problem" = what run_simulation builds (poisson.cpp:176-206): Domains for
 * the levels N, N-r, ..., 1 of this rank and, optionally, the Subdomain
 * preconditioner over them. */
typedef struct fddh_problem fddh_problem;

/* synthetic box mesh: E global elements, P rank blocks per direction */
int fddh_problem_create_box(fddh_problem **out, const int E[3], const int P[3], int poly_degree, int poly_reduction, int with_subdomain);
/* Nek5000-export directory, the reference's input (poisson.cpp:61-68, domain.tpp:45-224) */
int fddh_problem_create_dir(fddh_problem **out, const char *directory, int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int with_subdomain);
/* The same with the reference's overlap arguments (poisson.cpp:61-68) and construction flags:
 *   FDDH_WITH_SUBDOMAIN    build the Subdomain preconditioner
 *   FDDH_BLOCK_LOCAL       more than one rank: keep every rank's own elements only (no neighbour rings, no
 *                          superdomain): block-Jacobi, the comparison point of the full-domain-decomposition method
 *   FDDH_FORCE_COMPOSITE   build the region through the composite setup even on one rank (test hook) */
enum
{
    FDDH_WITH_SUBDOMAIN = 1,
    FDDH_BLOCK_LOCAL = 2,
    FDDH_FORCE_COMPOSITE = 4
};
int fddh_problem_create_box_ex(fddh_problem **out, const int E[3], const int P[3], int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int flags);
int fddh_problem_create_dir_ex(fddh_problem **out, const char *directory, int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int flags);
int fddh_problem_destroy(fddh_problem *p);
/* write the box mesh of this rank as the reference's file set under `directory` */
int fddh_write_box_mesh_files(const char *directory, const int E[3], const int P[3], int poly_degree, int rank);
/* The box mesh under the generalized Kershaw map (eps_y, eps_z in (0, 1]; 1 = the uniform box): the geometry of every
 * experiment of the reference (run.py:25-47, run.sh:30: Nek5000 "Kershaw" exports with eps = 0.3, which are not in the
 * repository).  GLL points moved by the map, the six geometric factors isoparametric at each level's own degree, all six
 * non-zero (host/box_mesh.hpp). */
int fddh_problem_create_kershaw_ex(fddh_problem **out, const int E[3], const int P[3], int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int flags, double eps_y, double eps_z);
int fddh_write_kershaw_mesh_files(const char *directory, const int E[3], const int P[3], int poly_degree, int rank, double eps_y, double eps_z);

enum
{
    FDDH_INFO_NUM_LOCAL_POINTS = 0,
    FDDH_INFO_NUM_LOCAL_NODES,
    FDDH_INFO_NUM_BDARY_NODES,
    FDDH_INFO_NUM_INTERFACE_SLOTS,
    FDDH_INFO_NUM_TOTAL_NODES,
    FDDH_INFO_NUM_TOTAL_ELEMENTS,
    FDDH_INFO_NUM_LOCAL_ELEMENTS,
    FDDH_INFO_NUM_LEVELS,
    FDDH_INFO_SUB_NUM_VALUES,
    FDDH_INFO_SUB_NUM_DOFS,
    FDDH_INFO_NUM_ITERATIONS,
    FDDH_INFO_DIM, /* 2 or 3: set by the mesh (domain.tpp:47) */
    FDDH_INFO_COUNT
};
int fddh_problem_info(const fddh_problem *p, long long *info, int n);
int fddh_problem_level_degree(const fddh_problem *p, int level, int *poly_degree);

/* The composite region of this rank (subdomain.tpp:455-579, 2581-2583) */
enum
{
    FDDH_SUB_IS_COMPOSITE = 0,   /* 1: rings + superdomain (more than one rank), 0: the rank's own conforming elements */
    FDDH_SUB_NUM_ELEMS,          /* own + ring elements */
    FDDH_SUB_NUM_EXT_ELEMS,      /* + the extended ring */
    FDDH_SUB_NUM_POINTS,         /* points of all of them = head of a composite vector */
    FDDH_SUB_NUM_SUB_DOFS,       /* subdomain dofs: regular + interface */
    FDDH_SUB_NUM_SUB_EXT_DOFS,   /* + extended */
    FDDH_SUB_NUM_INTERFACE_DOFS,
    FDDH_SUB_NUM_SUP_DOFS,       /* superdomain dofs: interface + regular */
    FDDH_SUB_NUM_SUP_EXT_DOFS,   /* + extended = tail of a composite vector */
    FDDH_SUB_NUM_UNIQUE_DOFS,    /* Subdomain::num_dofs */
    FDDH_SUB_NUM_COARSE_DOFS,    /* degree-1 dofs of the whole domain */
    FDDH_SUB_NUM_VALUES,
    FDDH_SUB_OWN_POINTS,
    FDDH_SUB_NUM_PEERS,          /* ranks this one exchanges ring data with */
    FDDH_SUB_INFO_COUNT
};
int fddh_problem_sub_info(const fddh_problem *p, long long *info, int n);
/* global element id and polynomial level of every region element (FDDH_SUB_NUM_EXT_ELEMS entries) */
int fddh_problem_sub_region(const fddh_problem *p, int *element, int *level, int n);
/* composite dofs kept per coarsening level of the superdomain (at most n values; *num_levels receives the count) */
int fddh_problem_sub_composite_levels(const fddh_problem *p, int *kept, int n, int *num_levels);

/* mesh arrays of a level as Domain::initialize holds them: name in
 * {"x","y","z","glo_num"(int64),"node_degree"(int32),"p_mask","g_1".."g_6"} */
int fddh_problem_mesh_array(const fddh_problem *p, int level, const char *name, void *out, size_t bytes);
/* CSR of the fine level's gather matrix Qt / scatter matrix Q (host copies) */
int fddh_problem_csr(const fddh_problem *p, int which /*0 = Q, 1 = Qt*/, int *num_rows, int *num_cols, int *num_nnz, int *ptr, int *col, double *val);
int fddh_problem_assembled_weight(const fddh_problem *p, double *out, int n);

/* Replace the GLL tables (D_hat of a level, J_cf between two levels) -- tests
 * feed the reference's own tables so parity does not hinge on the last bit of
 * the GLL nodes. */
int fddh_problem_set_D_hat(fddh_problem *p, int level, const double *D_hat, int n);
int fddh_problem_get_D_hat(const fddh_problem *p, int level, double *D_hat, int n);

/* Solver options: domain.hpp:112-118 and subdomain.hpp:228-238.  Negative
 * values (and NaN tolerance) leave a field unchanged. */
int fddh_problem_set_options(fddh_problem *p, int max_iterations, double tolerance, int num_vectors, int use_preconditioner, int preconditioner_type, int sub_num_vectors, int sub_max_iterations, int sub_build_tree);

/* Implementation switches (results are identical either way; the reference-shaped
 * launch sequences stay available for comparison):
 *   "fused_dssum"              1: one gather-scatter kernel per dssum (default), 0: the Qt / Q SpMV pair
 *   "restructured_inner_solve" 1: inner GMRES with cached assembled vectors, multi-dot / multi-axpy (default),
 *                              0: the reference's launch-by-launch sequence (subdomain.tpp:4309-4489)
 *   "assembled_outer_solve"    1: flexible CG on node vectors (one value per assembled node): Q fused into the stiffness
 *                              load, no dssum pass, the inner solve entered and left in dof numbering (default when the
 *                              inner solve is the assembled GMRES or there is no preconditioner); 0: point vectors
 *   "lazy_steps"               1: fddh_problem_pcg_steps(K) runs its K iterations with ONE host synchronisation, at the end
 *                              (norms kept in a device-side history, inner solves without their end-of-cycle read; default)
 *   "device_bookkeeping"       1: Givens rotations / stopping tests of the inner GMRES in one-thread kernels and alpha, beta
 *                              of the node-space PCG read from device memory: two host synchronisations per PCG step
 *                              (default); 0: the host computes them between launches, as the reference does
 *   "assembled_inner_solve"    1: inner GMRES on vectors over the dofs, Q fused into the stiffness load, one host
 *                              synchronisation per step (default); 0: the point-space forms above
 *   "mfma_stiffness"           1: degrees 11..15 apply the stiffness on the fp64 matrix cores (default;
 *                              agrees with the bit-exact kernel to ~1e-15, not bit for bit), 0: scalar fused kernel
 *   "sub_use_preconditioner"   1: the inner solver preconditions with the low-order AMG V-cycle
 *                              (Subdomain::use_preconditioner, subdomain.hpp:231; the reference's and this build's default),
 *                              0: the identity on assembled data (dssum, subdomain.tpp:4379-4382),
 *                              2: point-Jacobi, the exact diagonal of the inner iteration's operator -- a labelled option of
 *                              this build, not in the reference (host/subdomain.hpp)
 *   "amg_graph"                1: the V-cycle is replayed as one hipGraph when the stream allows capture (default)
 *   "amg_fused_smoother"       1: the smoother's element-wise kernels run as SpMV epilogues, bit-identical (default); 0: the reference's launch sequence 
 *   "amg_precision"            64 (default) or 32: the reference's `Float` (AMG/config.hpp:4): the V-cycle in double or in float */
/* Flag "affine_geometry" (an option of this build, off by default; the reference always streams the six factor
 * arrays): after fddh_problem_set_flag(p, "affine_geometry", 1), which of the operators run on the kernel that forms the
 * factors from six numbers per element (fdd_hip.h: fdd_stiffness_matrix_affine) -- the fine Domain's node-space operator,
 * and how many of the Subdomain's level lists -- and the largest relative deviation of the mesh's own factor arrays
 * from that form (-1 before the flag was ever set).  Any argument may be NULL. */
int fddh_problem_affine_info(fddh_problem *p, int *fine_domain_affine, int *sub_lists_affine, int *sub_lists, double *max_deviation);
int fddh_problem_set_flag(fddh_problem *p, const char *name, int value);

/* Low-order AMG preconditioner of the inner solve (Subdomain::low_order_preconditioner,
 * subdomain.tpp:3987-4159).  The reference builds the hierarchy with HYPRE BoomerAMG
 * (subdomain.tpp:3383-3549); here the caller hands it in, finest level first:
 * A (CSR), the Chebyshev diagonal scaling D_val and coefficients (hypre ds / coefs), and
 * the prolongation P to the next level (null + n_coarse 0 on the coarsest level).  Level 0
 * is numbered by the subdomain's dofs: fddh_problem_sub_point_dofs gives the dof of every
 * level-0 point (-1 on Dirichlet points). */
int fddh_problem_sub_point_dofs(const fddh_problem *p, int *dof, int n);
int fddh_problem_amg_add_level(fddh_problem *p, int n, const int *A_ptr, const int *A_col, const double *A_val, const double *D_val, const double *coefs, int num_coefs, int n_coarse, const int *P_ptr, const int *P_col, const double *P_val);
int fddh_problem_amg_finalize(fddh_problem *p);
int fddh_problem_amg_apply(fddh_problem *p, const double *r, double *z);
/* Or let the host layer build both the low-order FEM matrix (P1 on the 6 tetrahedra of every GLL sub-cell,
 * subdomain.tpp:2823-3060) and this build's own smoothed-aggregation hierarchy for it (HYPRE BoomerAMG's stand-in,
 * subdomain.tpp:3383-3549), and attach it.  coarsest_size / strength <= 0 take the defaults (400 rows, 0.08). */
int fddh_problem_amg_build(fddh_problem *p, int coarsest_size, double strength, int smooth_prolongator, int verbose, int *num_levels);
int fddh_problem_amg_level_info(const fddh_problem *p, int level, int *n, int *nnz_A, int *n_coarse, int *nnz_P);
/* 1 where the interpolator of `level` (to level + 1) is applied matrix-free: a geometric level of fddh_problem_amg_build's own
 * hierarchy on a conforming 3-D region with 8 or 16 lattice nodes per direction (fdd_lattice_prolong / _restrict,
 * include/fdd_hip.h), and the flag "amg_matrix_free_transfer" (default 1) on; 0: two SpMVs with the CSR interpolator */
int fddh_problem_amg_level_transfer(const fddh_problem *p, int level, int *matrix_free);
int fddh_problem_amg_level_arrays(const fddh_problem *p, int level, int *A_ptr, int *A_col, double *A_val, double *D_val, double *coefs, int *P_ptr, int *P_col, double *P_val);

/* Domain operations on host vectors of num_local_points */
int fddh_problem_dssum(fddh_problem *p, double *out, const double *in, int apply_mask, int apply_weight);          /* Domain::direct_stiffness_summation */
int fddh_problem_stiffness(fddh_problem *p, double *out, const double *in, int apply_dssum);                       /* Domain::stiffness_matrix */
int fddh_problem_residual_norm(fddh_problem *p, const double *r, double *norm);                                    /* Domain::residual_norm */
int fddh_problem_make_rhs(fddh_problem *p, int function_id, unsigned long long seed, double *u_star, double *f);   /* poisson.cpp:211-219 */
/* u* given on the host (e.g. a seeded vector): u* <- weighted dssum(u*), f = A_L u* */
int fddh_problem_make_rhs_from(fddh_problem *p, double *u_star_inout, double *f);

/* Outer solve: solver_id 0 = flexible_conjugate_gradient, 1 = generalized_minimum_residual
 * (poisson.cpp:224-231).  history receives the residual norms the reference
 * prints (iteration 0 first). */
int fddh_problem_solve(fddh_problem *p, int solver_id, const double *f, double *u, double *history, int history_cap, int *num_history, int *num_iterations);

/* The same solve with the right-hand side already uploaded and the clock around the device work only (stream
 * synchronised before and after): what bench.py reports as time to tolerance.  u may be NULL. */
int fddh_problem_solve_timed(fddh_problem *p, int solver_id, const double *f, double *u, double *history, int history_cap, int *num_history, int *num_iterations, double *seconds);

/* Subdomain (preconditioner) operations; type 0 = flexible_conjugate_gradient,
 * 1 = generalized_minimum_residual */
int fddh_problem_precond_apply(fddh_problem *p, int type, const double *r, double *z, double *history, int history_cap, int *num_history);
/* op 0 = tree_operator (in: an outer vector of num_local_points; collective on a composite region),
 * 1 = stiffness_matrix, 2 = direct_stiffness_summation; in/out of sub_num_values = [region points | superdomain dofs] */
int fddh_problem_sub_op(fddh_problem *p, int op, const double *in, double *out);
int fddh_problem_sub_residual_norm(fddh_problem *p, const double *r, double *norm);
/* The inner iteration in its dof-space form (host/subdomain.hpp: GMRES on Qt A_L Q over the unique dofs
 * [subdomain regular | interface | superdomain regular], n = FDDH_SUB_NUM_UNIQUE_DOFS):
 * op 0: out = operator(in), both n dofs; op 1: out (n dofs) = the right-hand side the inner solve sees for the outer
 * vector `in` (num_local_points; runs tree_operator: collective on a composite region).  Analysis and tests. */
int fddh_problem_sub_dof_op(fddh_problem *p, int op, const double *in, double *out, int n);
/* the diagonal of that operator over the n unique dofs, as the point-Jacobi option ("sub_use_preconditioner" = 2) uses it */
int fddh_problem_sub_jacobi_diagonal(fddh_problem *p, double *out, int n);

/* Average launch time (HIP events on the stream) and algorithmic bytes (12 B per non-zero,
 * 12 B per row, 8 B per column) of the assembly SpMVs of csr_matrix.okl on the problem's matrices:
 * which = 0: Q x (scatter), 1: Qt x (gather). */
int fddh_problem_spmv_time(fddh_problem *p, int which, int iterations, double *avg_us, double *algorithmic_bytes);
/* The solve path's exchanges alone, with the sizes and buffers the solve uses: avg_us[FDDH_COMM_TIME_COUNT],
 * bytes[FDDH_COMM_TIME_COUNT] = { all-reduce of 3 scalars, interface exchange of two prefixes as a dense all-reduce, coarse-level
 * all-gather (total bytes gathered), ring pull (bytes this rank sends), the same interface exchange as grouped sends / receives
 * between the ranks that share nodes (bytes this rank sends: the default path, flag "neighbour_interface_exchange"), ring pull
 * and coarse blocks in one group (bytes this rank sends: the default path, flag "fold_coarse_exchange") }; host wall clock around
 * device synchronisation, after a barrier.  Collective: every rank calls it.  Zeros on one rank / without a composite. */
#define FDDH_COMM_TIME_COUNT 6
int fddh_problem_comm_time(fddh_problem *p, int iterations, double *avg_us, double *bytes);
/* The general CSR case of the same metric: the 27-point trilinear stencil on an m^3 node grid
 * ((3m - 2)^3 non-zeros; m = 225 is the C2 node grid), values and x seeded, y = A x through CSR_Matrix::multiply. */
int fddh_spmv_stencil_time(int m, int iterations, double *avg_us, double *algorithmic_bytes, long long *num_nnz);

/* Stepwise PCG with vectors resident in HBM (what bench.py times): begin sets
 * u = 0, r = f, z = M^-1 r, p = z; each step is one full outer iteration with
 * no stopping test.  last_residual receives ||r|| of the last step. */
int fddh_problem_pcg_begin(fddh_problem *p, const double *f);
int fddh_problem_pcg_steps(fddh_problem *p, int steps, double *last_residual);
int fddh_problem_pcg_solution(fddh_problem *p, double *u);
/* Per-kernel timing with HIP events on the rank's stream (bench.py's roofline
 * object).  collect() synchronises and writes a JSON object
 * {"<kernel family>": {"count": n, "ms": total, "bytes": algorithmic total}}. */
int fddh_profile_enable(int on);
int fddh_profile_only(const char *kernel_key); /* after enable: time only this kernel family (NULL / "": all); two event records per launch cost a few microseconds */
int fddh_profile_collect(char *json, size_t json_len);

int fddh_sync(void);   /* stream synchronise */
int fddh_barrier(void); /* communicator barrier (stream-ordered, then synchronised) */

#ifdef __cplusplus
}
#endif

#endif /* FDD_HOST_H */
