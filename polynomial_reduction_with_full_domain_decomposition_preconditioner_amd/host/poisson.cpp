/*
 * poisson.cpp -- the driver, same command line as the reference
 * (poisson.cpp:61-68):
 *
 *     poisson <directory> <polynomial degree> <polynomial reduction>
 *             <subdomain overlap> <superdomain overlap>
 *
 * <directory> holds the Nek5000-export file sets lx1_<N+1>/ for every level
 * degree N, N-r, ..., 1 (domain.tpp:45-224).  Extensions (all optional, after
 * the five positional arguments):
 *     --box Ex Ey Ez     generate the synthetic unit-cube mesh instead of
 *                        reading <directory> (which is then ignored)
 *     --kershaw EPS      with --box: the box under the Kershaw map with
 *                        eps_y = eps_z = EPS (the geometry of the reference's
 *                        experiments: run.py:25-47 and run.sh:30 read Nek5000
 *                        exports of it with EPS = 0.3)
 *     --solver fcg|gmres outer solver (the reference hard-codes solver_id = 1,
 *                        GMRES, poisson.cpp:224; PCG is solver_id = 0)
 *     --function ID      manufactured solution id (poisson.cpp:211: 4)
 *     --no-precond       outer solve without the FDD preconditioner
 *     --no-amg           inner solves without the low-order AMG V-cycle
 *                        (Subdomain::use_preconditioner = false; the
 *                        reference's default is true, subdomain.hpp:231,
 *                        and so is this driver's); --amg says it explicitly
 *     --inner K          inner Krylov steps per preconditioner application
 *                        (subdomain.hpp:229-230 num_vectors = max_iterations;
 *                        run.py:151-152 sweeps 1, 2, 4, 8; default 4)
 *     --inner-solver fcg|gmres   domain.hpp:116 preconditioner_type (run.py:150)
 *     --vcycles V        subdomain.hpp:236 num_vcycles (run.py:153)
 *     --cheby C          subdomain.hpp:237 cheby_order, 1..4 (run.py:154)
 *     --float            the preconditioner in single precision
 *                        (config.hpp:19-20 PTYPE / AMG/config.hpp:4 Float; run.py:156)
 *     --block-local      more than one rank: every rank keeps its own
 *                        elements only (no neighbour rings, no superdomain:
 *                        block-Jacobi).  The overlap arguments then have no
 *                        effect and the driver says so.
 *     --write-mesh DIR   write the box mesh as a reference-format file set
 *
 * One process per GPU.  With WORLD_SIZE > 1 in the environment (RANK,
 * LOCAL_RANK, WORLD_SIZE as set by any launcher) the ranks join an RCCL
 * communicator whose unique id travels through the file named by
 * FDD_RCCL_ID_FILE (rank 0 writes it); MPI is not needed.
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "fdd_host.h"

static void die(const char *what)
{
    fprintf(stderr, "ERROR: %s: %s\n", what, fddh_last_error());
    exit(EXIT_FAILURE);
}

static int env_int(const char *name, int def)
{
    const char *v = getenv(name);
    return v ? atoi(v) : def;
}

static void library_banner(int rank)
{
    if (rank != 0) return;
    printf("----------------------------------------------------------------------------------\n");
    printf("|  Full domain decomposition with polynomial reduction -- MI355X (gfx950) build   |\n");
    printf("----------------------------------------------------------------------------------\n\n");
}

int main(int argc, char *argv[])
{
    const int rank = env_int("RANK", 0);
    const int size = env_int("WORLD_SIZE", 1);
    const int local_rank = env_int("LOCAL_RANK", 0);

    if (fddh_init(local_rank, nullptr, 1)) die("fddh_init");
    library_banner(rank);

    if (size > 1)
    {
        const char *id_env = getenv("FDD_RCCL_ID_FILE");
        if (!id_env)
        {
            fprintf(stderr, "ERROR: WORLD_SIZE > 1 needs FDD_RCCL_ID_FILE\n");
            return EXIT_FAILURE;
        }
        // keyed by a launcher-supplied job id (or the master port every launcher sets) so that a stale file of an
        // earlier run under the same path is never read
        const char *job = getenv("FDD_JOB_ID") ? getenv("FDD_JOB_ID") : getenv("MASTER_PORT");
        const std::string id_path = std::string(id_env) + (job ? std::string(".") + job : std::string());
        const char *id_file = id_path.c_str();
        if (rank == 0) unlink(id_file);
        char id[128];
        if (rank == 0)
        {
            // a file left by an earlier launch must not be picked up by this launch's other ranks: its name carries
            // the launcher's job id when one is given (FDD_JOB_ID), and it is removed again once every rank has joined
            if (fddh_comm_rccl_unique_id(id)) die("fddh_comm_rccl_unique_id");
            std::string tmp = std::string(id_file) + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, 128, f) != 128)
            {
                fprintf(stderr, "ERROR: cannot write %s\n", tmp.c_str());
                return EXIT_FAILURE;
            }
            fclose(f);
            rename(tmp.c_str(), id_file);
        }
        else
        {
            FILE *f = nullptr;
            for (int tries = 0; tries < 6000 && !(f = fopen(id_file, "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            if (!f || fread(id, 1, 128, f) != 128)
            {
                fprintf(stderr, "ERROR: cannot read %s\n", id_file);
                return EXIT_FAILURE;
            }
            fclose(f);
        }
        if (fddh_comm_rccl_init(id, rank, size)) die("fddh_comm_rccl_init");
        if (fddh_barrier()) die("fddh_barrier"); // every rank has read the id
        if (rank == 0) unlink(id_file);
    }
    else
    {
        fddh_comm_single();
    }

    if (argc < 6)
    {
        if (rank == 0) printf("ERROR: Use as 'poisson <directory> <polynomial degree> <polynomial reduction> <subdomain overlap> <superdomain overlap>'\n");
        return EXIT_SUCCESS; // the reference's quit() exits with SUCCESS (config.hpp:57-62)
    }

    const char *directory = argv[1];
    const int poly_degree = atoi(argv[2]);
    const int poly_reduction = atoi(argv[3]);
    const int subdomain_overlap = atoi(argv[4]);
    const int superdomain_overlap = atoi(argv[5]);

    int box[3] = {0, 0, 0};
    int solver_id = 1;
    int function_id = 4;
    int with_subdomain = 1;
    int use_amg = 1;     // Subdomain::use_preconditioner, subdomain.hpp:231
    int block_local = 0;
    const char *write_dir = nullptr;
    double kershaw_eps = 1.0;
    int inner_steps = 0, inner_solver = -1, vcycles = 0, cheby = 0, single = 0;
    for (int a = 6; a < argc; a++)
    {
        if (!strcmp(argv[a], "--box") && a + 3 < argc)
        {
            box[0] = atoi(argv[a + 1]);
            box[1] = atoi(argv[a + 2]);
            box[2] = atoi(argv[a + 3]);
            a += 3;
        }
        else if (!strcmp(argv[a], "--inner") && a + 1 < argc)
            inner_steps = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--inner-solver") && a + 1 < argc)
            inner_solver = !strcmp(argv[++a], "fcg") ? 0 : 1;
        else if (!strcmp(argv[a], "--vcycles") && a + 1 < argc)
            vcycles = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--cheby") && a + 1 < argc)
            cheby = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--float"))
            single = 1;
        else if (!strcmp(argv[a], "--kershaw") && a + 1 < argc)
            kershaw_eps = atof(argv[++a]);
        else if (!strcmp(argv[a], "--solver") && a + 1 < argc)
            solver_id = !strcmp(argv[++a], "fcg") ? 0 : 1;
        else if (!strcmp(argv[a], "--function") && a + 1 < argc)
            function_id = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--no-precond"))
            with_subdomain = 0;
        else if (!strcmp(argv[a], "--no-amg"))
            use_amg = 0;
        else if (!strcmp(argv[a], "--amg"))
            use_amg = 1;
        else if (!strcmp(argv[a], "--block-local"))
            block_local = 1;
        else if (!strcmp(argv[a], "--write-mesh") && a + 1 < argc)
            write_dir = argv[++a];
    }

    if (rank == 0)
    {
        printf("Running simulation with:\n");
        printf("- Directory: \"%s\"\n", directory);
        printf("- Polynomial degree: \"%d\"\n", poly_degree);
        printf("- Polynomial reduction: \"%d\"\n", poly_reduction);
        printf("- Subdomain overlap: \"%d\"\n", subdomain_overlap);
        printf("- Superdomain overlap: \"%d\"\n\n", superdomain_overlap);
        if (block_local && size > 1) printf("NOTE: --block-local: every rank preconditions with its own elements only; the overlap arguments are not used\n\n");
    }

    // rank blocks: powers of two go round-robin over x, y, z
    int P[3] = {1, 1, 1};
    for (int s = size, d = 0; s > 1 && s % 2 == 0; s /= 2, d = (d + 1) % 3) P[d] *= 2;

    fddh_problem *problem = nullptr;
    const int flags = (with_subdomain ? FDDH_WITH_SUBDOMAIN : 0) | (block_local ? FDDH_BLOCK_LOCAL : 0);
    if (box[0] > 0)
    {
        if (write_dir)
        {
            int deg = poly_degree;
            for (;;)
            {
                if (fddh_write_kershaw_mesh_files(write_dir, box, P, deg, rank, kershaw_eps, kershaw_eps)) die("fddh_write_kershaw_mesh_files");
                if (deg == 1 || !with_subdomain) break;
                deg = (deg - poly_reduction >= 1) ? deg - poly_reduction : 1;
            }
        }
        if (fddh_problem_create_kershaw_ex(&problem, box, P, poly_degree, poly_reduction, subdomain_overlap, superdomain_overlap, flags, kershaw_eps, kershaw_eps)) die("fddh_problem_create_kershaw_ex");
    }
    else
    {
        FILE *fp = fopen(directory, "r");
        if (fp == NULL)
        {
            if (rank == 0) printf("Directory '%s' does not exist. Make sure the directory has all the 'lx1' subdirectories", directory);
            return EXIT_SUCCESS;
        }
        fclose(fp);
        if (fddh_problem_create_dir_ex(&problem, directory, poly_degree, poly_reduction, subdomain_overlap, superdomain_overlap, flags)) die("fddh_problem_create_dir_ex");
    }
    if (with_subdomain && fddh_problem_set_flag(problem, "sub_use_preconditioner", use_amg)) die("fddh_problem_set_flag");
    // the parameters run.py:150-156 sweeps by rewriting header lines and rebuilding: run-time switches here
    if (with_subdomain)
    {
        if (cheby > 0 && fddh_problem_set_flag(problem, "amg_cheby_order", cheby)) die("amg_cheby_order");
        if (vcycles > 0 && fddh_problem_set_flag(problem, "amg_num_vcycles", vcycles)) die("amg_num_vcycles");
        if (single && fddh_problem_set_flag(problem, "preconditioner_precision", 32)) die("preconditioner_precision");
        if ((inner_steps > 0 || inner_solver >= 0) && fddh_problem_set_options(problem, -1, NAN, -1, -1, inner_solver, inner_steps > 0 ? inner_steps : -1, inner_steps > 0 ? inner_steps : -1, -1)) die("fddh_problem_set_options");
    }

    long long info[FDDH_INFO_COUNT];
    fddh_problem_info(problem, info, FDDH_INFO_COUNT);
    const size_t npts = (size_t)info[FDDH_INFO_NUM_LOCAL_POINTS];

    if (rank == 0) printf("\nSetting up exact function...\nSetting up right-hand-side...\n");
    double *u_star = (double *)malloc(npts * sizeof(double));
    double *f = (double *)malloc(npts * sizeof(double));
    double *u = (double *)malloc(npts * sizeof(double));
    if (fddh_problem_make_rhs(problem, function_id, 1234ULL + (unsigned long long)rank, u_star, f)) die("fddh_problem_make_rhs");

    if (rank == 0) printf("Solving Poisson problem...\n");
    fddh_set_timer(1);
    int nhist = 0, its = 0;
    fddh_barrier();
    auto t0 = std::chrono::high_resolution_clock::now();
    if (fddh_problem_solve(problem, solver_id, f, u, nullptr, 0, &nhist, &its)) die("fddh_problem_solve");
    fddh_barrier();
    auto t1 = std::chrono::high_resolution_clock::now();
    const double seconds = std::chrono::duration<double>(t1 - t0).count();

    double err = 0.0;
    for (size_t i = 0; i < npts; i++) err = std::max(err, std::fabs(u[i] - u_star[i]));

    if (rank == 0)
    {
        printf("\nRun info:\n");
        printf("-------------------------------------------------------------------------\n");
        printf("Number of dimensions: %lld\n", info[FDDH_INFO_DIM]);
        printf("Total number of elements: %lld\n", info[FDDH_INFO_NUM_TOTAL_ELEMENTS]);
        printf("Total number of unique nodes: %lld\n", info[FDDH_INFO_NUM_TOTAL_NODES]);
        printf("Polynomial degree: %d\n", poly_degree);
        printf("Function ID: %d\n", function_id);
        printf("Solver data precision: double\n");
        printf("Solver type: \"%s\"\n", (solver_id == 0) ? "FCG" : "GMRES");
        long long sub[FDDH_SUB_INFO_COUNT] = {0};
        if (with_subdomain) fddh_problem_sub_info(problem, sub, FDDH_SUB_INFO_COUNT);
        if (!with_subdomain)
            printf("Preconditioner: none\n");
        else if (sub[FDDH_SUB_IS_COMPOSITE])
            printf("Preconditioner: full domain decomposition (%lld + %lld region elements, %lld superdomain dofs of %lld coarse), inner GMRES(4)%s\n", sub[FDDH_SUB_NUM_ELEMS], sub[FDDH_SUB_NUM_EXT_ELEMS] - sub[FDDH_SUB_NUM_ELEMS],
                   sub[FDDH_SUB_NUM_SUP_DOFS], sub[FDDH_SUB_NUM_COARSE_DOFS], use_amg ? " + low-order AMG V-cycle" : "");
        else
            printf("Preconditioner: %s, inner GMRES(4)%s\n", size > 1 ? "block-local subdomain solve (own elements only)" : "subdomain solve (single rank: own elements = whole domain)", use_amg ? " + low-order AMG V-cycle" : "");
        printf("Iterations: %d\n", its);
        printf("Solve wall time: %.6f s\n", seconds);
        printf("max |u - u*| on rank 0: %.3e\n", err);
        if (its > 0) printf("DOF-updates/s: %.4e\n", (double)info[FDDH_INFO_NUM_TOTAL_NODES] * its / seconds);

    }

    // timing table (poisson.cpp:253-401): the reference's eight rows, each the MAXIMUM over the ranks (aggregation_type
    // "max", poisson.cpp:256; timer.tpp:67), tree construction / exchange summed over their regions (:322-357).
    // Collective: every rank takes part in the reductions, rank 0 prints.
    {
        struct Row
        {
            const char *label;
            std::vector<const char *> keys;
        };
        const std::vector<Row> rows = {
            {"Inner products", {"domain.inner_products"}},
            {"Residual norm", {"domain.residual_norm"}},
            {"Vector operations", {"domain.vector_operations"}},
            {"Operator application", {"domain.operator_application"}},
            {"Tree construction", {"subdomain.tree_construction.gpu_to_gpu", "subdomain.tree_construction.subdomain", "subdomain.tree_construction.gpu_to_cpu", "subdomain.tree_construction.assemble_coarse", "subdomain.tree_construction.superdomain"}},
            {"Tree exchange", {"subdomain.tree_exchange.superdomain", "subdomain.tree_exchange.subdomain", "subdomain.tree_exchange.cpu_to_gpu"}},
            {"Subdomain stitching", {"subdomain.stitching"}},
            {"Subdomain solver", {"subdomain.solver"}},
        };
        std::vector<double> t(rows.size(), 0.0);
        double total = 0.0;
        for (size_t r = 0; r < rows.size(); r++)
        {
            for (const char *k : rows[r].keys)
            {
                double s = 0.0;
                if (fddh_timer_total_over_ranks(k, "max", &s)) die("fddh_timer_total_over_ranks");
                t[r] += s;
            }
            total += t[r];
        }
        if (rank == 0)
        {
            printf("\nTimings (maximum over the ranks):\n");
            printf("-------------------------------------------------------------------------\n");
            printf("%-21s = %12.08f s ( %6.02f )\n", "Total", total, 100.0);
            for (size_t r = 0; r < rows.size(); r++) printf("%-21s = %12.08f s ( %6.02f )\n", rows[r].label, t[r], total > 0.0 ? 100.0 * t[r] / total : 0.0);
        }
    }

    free(u_star);
    free(f);
    free(u);
    fddh_problem_destroy(problem);
    return EXIT_SUCCESS;
}
