// C-ABI over the C++ host classes (include/fdd_host.h).  Everything here is
// plumbing: it instantiates Domain<double> / Subdomain<double> the way the
// reference's run_simulation does (poisson.cpp:150-251) and moves host vectors
// in and out of device memory.
#include "fdd_host.h"

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <limits>
#include <exception>
#include <memory>
#include <string>
#include <sys/stat.h>
#include <unordered_map>
#include <functional>
#include <vector>

#include "box_mesh.hpp"
#include "domain.hpp"
#include "subdomain.hpp"

typedef double SType; // config.hpp:19 STYPE
typedef double PType; // config.hpp:20 PTYPE (AMG/config.hpp:4 Float)

static thread_local char g_err[512] = "";

static int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

struct fddh_problem
{
    int poly_degree = 1;
    int poly_reduction = 1;
    std::vector<int> degrees; // N, N-r, ..., 1
    std::unordered_map<int, Domain<SType>> domains;
    std::unique_ptr<Subdomain<PType>> subdomain;
    NoPreconditioner none;

    // The rank (= host thread) that built the problem: its device stream, communicator and globals are thread_local, so a
    // call from any other thread would silently run on the legacy default stream with a one-rank communicator (every
    // collective skipped: wrong sums, or peers hanging in RCCL).  Every fddh_problem_* entry checks it (rank_check).
    const void *owner = nullptr;

    // device staging vectors
    fdd::memory a, b, c;
    fdd::memory sa, sb; // subdomain-sized

    Domain<SType> &fine() { return domains[poly_degree]; }
    const Domain<SType> &fine() const { return domains.at(poly_degree); }
};

// identity of the calling rank's per-thread state
static const void *this_rank() { return &fdd::dev(); }

// 0 when the calling thread is an initialised rank (fddh_init) and, for a problem, the one that built it
static int rank_check(const fddh_problem *p = nullptr)
{
    if (!fdd::dev().initialised) return fail("fddh_init has not been called on this thread: the host layer's device stream and communicator are per rank = per host thread");
    if (p && p->owner != this_rank()) return fail("this problem was built by another rank (host thread); its stream and communicator are not the calling thread's");
    return 0;
}

static std::vector<int> level_degrees(int N, int reduction)
{
    std::vector<int> d;
    d.push_back(N);
    while (d.back() > 1)
    {
        int r = d.back() - reduction;
        d.push_back(r >= 1 ? r : 1);
    }
    return d;
}

// FDD_SETUP_TIMING=1: rank 0 prints the host time of the setup's parts
static double setup_clock() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void setup_lap(const char *what, int degree, double &mark)
{
    static const bool on = getenv("FDD_SETUP_TIMING") != nullptr;
    const double t = setup_clock();
    if (on and fdd::comm().rank == 0) printf("setup: %-28s N = %-2d %8.3f s\n", what, degree, t - mark);
    mark = t;
}

static void finish_problem(fddh_problem *p, int flags, int sub_overlap, int sup_overlap)
{
    double mark = setup_clock();
    p->owner = this_rank();
    const int with_subdomain = flags & FDDH_WITH_SUBDOMAIN;
    Domain<SType> &dom = p->fine();
    p->a = fdd::dev().malloc<double>(dom.num_local_points);
    p->b = fdd::dev().malloc<double>(dom.num_local_points);
    p->c = fdd::dev().malloc<double>(dom.num_local_points);
    if (with_subdomain)
    {
        rstdout("Setting up subdomain object...\n");
        p->subdomain.reset(new Subdomain<PType>());
        p->subdomain->block_local = (flags & FDDH_BLOCK_LOCAL) != 0;
        p->subdomain->force_composite = (flags & FDDH_FORCE_COMPOSITE) != 0;
        p->subdomain->initialize(p->domains, p->poly_degree, p->poly_reduction, sub_overlap, sup_overlap);
        setup_lap("Subdomain::initialize", p->poly_degree, mark);
        p->sa = fdd::dev().malloc<double>(p->subdomain->num_values);
        p->sb = fdd::dev().malloc<double>(p->subdomain->num_values);
        dom.use_preconditioner = true;
    }
    else
    {
        dom.use_preconditioner = false;
    }
}

extern "C" {

const char *fddh_last_error(void) { return g_err; }

int fddh_init(int device, void *stream, int own_stream)
{
    try
    {
        if (fdd_set_device(device) != 0) return fail("fdd_set_device(%d): %s", device, fdd_last_error());
        if (own_stream)
        {
            void *s = nullptr;
            if (fdd_stream_create(&s) != 0) return fail("fdd_stream_create: %s", fdd_last_error());
            stream = s;
        }
        fdd::dev().owns_stream = own_stream != 0;
        fdd::dev().stream = stream;
        fdd::dev().initialised = true;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_set_print(int on)
{
    try
    {
        fdd::globals().print = on != 0;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_set_timer(int on)
{
    try
    {
        fdd_timer().enabled = on != 0;
        if (on) fdd_timer().initialize();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_timer_total(const char *key, double *seconds)
{
    try
    {
        if (!key || !seconds) return fail("null argument");
        *seconds = fdd_timer().total(key);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_timer_total_over_ranks(const char *key, const char *aggregation, double *seconds)
{
    try
    {
        if (!key || !aggregation || !seconds) return fail("null argument");
        *seconds = fdd_timer().total(key, aggregation); // collective: all-reduce (max or sum) over the ranks, timer.tpp:67
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_comm_single(void)
{
    try
    {
        if (int rc = rank_check()) return rc;
        fdd::set_comm(new fdd::SingleComm());
        fdd::globals().proc_id = 0;
        fdd::globals().num_procs = 1;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_local_world_create(void **world, int size)
{
    try
    {
        if (!world || size < 1) return fail("bad argument");
        *world = new std::shared_ptr<fdd::LocalWorld>(new fdd::LocalWorld(size));
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_local_world_destroy(void *world)
{
    delete static_cast<std::shared_ptr<fdd::LocalWorld> *>(world);
    return 0;
}

int fddh_local_world_fail(void *world)
{
    try
    {
        if (!world) return fail("null argument");
        (*static_cast<std::shared_ptr<fdd::LocalWorld> *>(world))->fail();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_rank_finalize(void)
{
    try
    {
        fdd::device_t &d = fdd::dev();
        if (!d.initialised) return 0;
        if (d.stream) fdd_stream_sync(d.stream);
        fdd::set_comm(new fdd::SingleComm()); // deletes the thread's communicator (and its device scratch)
        if (d.owns_stream && d.stream) fdd_stream_destroy(d.stream);
        d.stream = nullptr;
        d.owns_stream = false;
        d.initialised = false;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_comm_local(void *world, int rank)
{
    try
    {
        if (int rc = rank_check()) return rc;
        if (!world) return fail("null argument");
        std::shared_ptr<fdd::LocalWorld> w = *static_cast<std::shared_ptr<fdd::LocalWorld> *>(world);
        if (rank < 0 || rank >= w->size) return fail("bad rank");
        fdd::set_comm(new fdd::LocalComm(w, rank));
        fdd::globals().proc_id = rank;
        fdd::globals().num_procs = w->size;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_comm_rccl_unique_id(char *out128)
{
    try
    {
        if (!out128) return fail("null argument");
        fdd::RcclComm tmp;
        tmp.unique_id(out128);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_comm_rccl_init(const char *id128, int rank, int size)
{
    try
    {
        if (int rc = rank_check()) return rc;
        if (!id128 || rank < 0 || size < 1 || rank >= size) return fail("bad rank/size");
        fdd::RcclComm *c = new fdd::RcclComm();
        c->init(id128, rank, size);
        fdd::set_comm(c);
        fdd::globals().proc_id = rank;
        fdd::globals().num_procs = size;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_comm_callbacks(int rank, int size, void *ctx, fddh_allreduce_fn allreduce_sum_f64, fddh_allreduce_fn allreduce_max_f64, fddh_allgather_fn allgather_bytes, fddh_barrier_fn barrier)
{
    try
    {
        if (int rc = rank_check()) return rc;
        if (rank < 0 || size < 1 || rank >= size || !allreduce_sum_f64 || !allreduce_max_f64 || !allgather_bytes || !barrier) return fail("bad callback set");
        fdd::CommCallbacks cb;
        cb.ctx = ctx;
        cb.allreduce_sum_f64 = allreduce_sum_f64;
        cb.allreduce_max_f64 = allreduce_max_f64;
        cb.allgather_bytes = allgather_bytes;
        cb.barrier = barrier;
        cb.exchange_bytes = nullptr; // fddh_comm_callbacks_ex supplies it
        fdd::set_comm(new fdd::CallbackComm(rank, size, cb));
        fdd::globals().proc_id = rank;
        fdd::globals().num_procs = size;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_comm_callbacks_ex(int rank, int size, void *ctx, fddh_allreduce_fn allreduce_sum_f64, fddh_allreduce_fn allreduce_max_f64, fddh_allgather_fn allgather_bytes, fddh_barrier_fn barrier, fddh_exchange_fn exchange_bytes)
{
    try
    {
        if (int rc = rank_check()) return rc;
        if (rank < 0 || size < 1 || rank >= size || !allreduce_sum_f64 || !allreduce_max_f64 || !allgather_bytes || !barrier || !exchange_bytes) return fail("bad callback set");
        fdd::CommCallbacks cb;
        cb.ctx = ctx;
        cb.allreduce_sum_f64 = allreduce_sum_f64;
        cb.allreduce_max_f64 = allreduce_max_f64;
        cb.allgather_bytes = allgather_bytes;
        cb.barrier = barrier;
        cb.exchange_bytes = exchange_bytes;
        fdd::set_comm(new fdd::CallbackComm(rank, size, cb));
        fdd::globals().proc_id = rank;
        fdd::globals().num_procs = size;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

// Drives every collective of the active communicator once, bypassing the
// "size == 1" shortcuts of the solver, and checks the results.
int fddh_comm_selftest(int n)
{
    try
    {
        if (int rc = rank_check()) return rc;
        if (n < 1) return fail("n must be positive");
        fdd::Comm &c = fdd::comm();
        const int R = c.size, me = c.rank;
        std::vector<double> h(n);
        for (int i = 0; i < n; i++) h[i] = (double)(me + 1) * (i + 1);
        fdd::memory d = fdd::dev().malloc<double>(n);
        fdd::memory g = fdd::dev().malloc<double>((size_t)n * R);

        d.copyFrom(h.data(), n * sizeof(double));
        c.allreduce_sum(d.as<double>(), n);
        std::vector<double> out(n);
        d.copyTo(out.data(), n * sizeof(double));
        const double tri = 0.5 * R * (R + 1);
        for (int i = 0; i < n; i++)
            if (out[i] != tri * (i + 1)) return fail("allreduce_sum: element %d is %g, expected %g", i, out[i], tri * (i + 1));

        d.copyFrom(h.data(), n * sizeof(double));
        c.allreduce_max(d.as<double>(), n);
        d.copyTo(out.data(), n * sizeof(double));
        for (int i = 0; i < n; i++)
            if (out[i] != (double)R * (i + 1)) return fail("allreduce_max: element %d is %g", i, out[i]);

        d.copyFrom(h.data(), n * sizeof(double));
        c.allgather(d.ptr(), g.ptr(), n * sizeof(double));
        std::vector<double> all((size_t)n * R);
        g.copyTo(all.data(), all.size() * sizeof(double));
        for (int p = 0; p < R; p++)
            for (int i = 0; i < n; i++)
                if (all[(size_t)p * n + i] != (double)(p + 1) * (i + 1)) return fail("allgather: rank %d element %d is %g", p, i, all[(size_t)p * n + i]);

        std::vector<long long> mine(me + 2, 100 + me);
        std::vector<int> counts;
        std::vector<long long> cat = c.allgatherv_host(mine, counts);
        size_t expect = 0;
        for (int p = 0; p < R; p++) expect += p + 2;
        if (cat.size() != expect) return fail("allgatherv_host: %zu entries, expected %zu", cat.size(), expect);

        // point-to-point: every rank sends (me + 1) * (i + 1) + 1000 * peer to its two ring neighbours (one message per
        // peer and direction, as the composite's ring pull does), on device buffers and through the host helper
        if (R > 1)
        {
            const int left = (me + R - 1) % R, right = (me + 1) % R;
            std::vector<int> peers;
            peers.push_back(right);
            if (left != right) peers.push_back(left);
            std::vector<fdd::memory> sb(peers.size()), rb(peers.size());
            std::vector<fdd::ExchangeOp> ops(peers.size());
            for (size_t k = 0; k < peers.size(); k++)
            {
                const int ns = n + peers[k], nr = n + me; // sizes differ per direction
                for (int i = 0; i < n; i++) h[i] = (double)(me + 1) * (i + 1) + 1000.0 * peers[k];
                std::vector<double> msg(ns, -1.0);
                std::copy(h.begin(), h.end(), msg.begin());
                sb[k] = fdd::dev().malloc<double>(ns);
                rb[k] = fdd::dev().malloc<double>(nr);
                sb[k].copyFrom(msg.data(), ns * sizeof(double));
                ops[k].peer = peers[k];
                ops[k].send = sb[k].ptr();
                ops[k].send_bytes = ns * sizeof(double);
                ops[k].recv = rb[k].ptr();
                ops[k].recv_bytes = nr * sizeof(double);
            }
            c.exchange(ops.data(), (int)ops.size());
            for (size_t k = 0; k < peers.size(); k++)
            {
                std::vector<double> got(n + me);
                rb[k].copyTo(got.data(), got.size() * sizeof(double));
                for (int i = 0; i < n; i++)
                    if (got[i] != (double)(peers[k] + 1) * (i + 1) + 1000.0 * me) return fail("exchange: element %d from rank %d is %g", i, peers[k], got[i]);
                sb[k].free();
                rb[k].free();
            }
            std::vector<std::vector<char>> out(R);
            for (int p = 0; p < R; p++) out[p].assign((size_t)(3 + me + 2 * p), (char)(17 * me + p));
            std::vector<std::vector<char>> in = c.exchange_host(out);
            for (int p = 0; p < R; p++)
            {
                if (in[p].size() != (size_t)(3 + p + 2 * me)) return fail("exchange_host: %zu bytes from rank %d", in[p].size(), p);
                for (char ch : in[p])
                    if (ch != (char)(17 * p + me)) return fail("exchange_host: wrong payload from rank %d", p);
            }
        }

        c.barrier();
        d.free();
        g.free();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_comm_info(int *rank, int *size, char *name, size_t name_len)
{
    try
    {
        if (int rc = rank_check()) return rc;
        if (rank) *rank = fdd::comm().rank;
        if (size) *size = fdd::comm().size;
        if (name && name_len) snprintf(name, name_len, "%s", fdd::comm().name());
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_create_box(fddh_problem **out, const int E[3], const int P[3], int poly_degree, int poly_reduction, int with_subdomain)
{
    try
    {
        if (int rc = rank_check()) return rc;
        return fddh_problem_create_box_ex(out, E, P, poly_degree, poly_reduction, 1, 1, with_subdomain ? FDDH_WITH_SUBDOMAIN : 0);
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_create_box_ex(fddh_problem **out, const int E[3], const int P[3], int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int flags)
{
    return fddh_problem_create_kershaw_ex(out, E, P, poly_degree, poly_reduction, subdomain_overlap, superdomain_overlap, flags, 1.0, 1.0);
}

int fddh_problem_create_kershaw_ex(fddh_problem **out, const int E[3], const int P[3], int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int flags, double eps_y, double eps_z)
{
    try
    {
        if (int rc = rank_check()) return rc;
        if (!(eps_y > 0.0 && eps_y <= 1.0 && eps_z > 0.0 && eps_z <= 1.0)) return fail("Kershaw eps must lie in (0, 1]");
        const int with_subdomain = flags & FDDH_WITH_SUBDOMAIN;
        if (!out || !E || !P || poly_degree < 1 || poly_reduction < 1) return fail("bad argument");
        if (P[0] * P[1] * P[2] != fdd::comm().size) return fail("rank grid %dx%dx%d does not match communicator size %d", P[0], P[1], P[2], fdd::comm().size);
        for (int d = 0; d < 3; d++)
            if (E[d] < 1 || P[d] < 1 || E[d] % P[d] != 0) return fail("elements per direction must be a multiple of the rank blocks");

        fddh_problem *p = new fddh_problem();
        p->poly_degree = poly_degree;
        p->poly_reduction = poly_reduction;
        p->degrees = with_subdomain ? level_degrees(poly_degree, poly_reduction) : std::vector<int>(1, poly_degree);

        fdd::BoxSpec spec;
        for (int d = 0; d < 3; d++)
        {
            spec.E[d] = E[d];
            spec.P[d] = P[d];
        }
        spec.kershaw_eps[0] = eps_y;
        spec.kershaw_eps[1] = eps_z;

        for (int deg : p->degrees)
        {
            rstdout("Setting up domain \"N = %d\" object...\n", deg);
            double mark = setup_clock();
            MeshData<SType> mesh = fdd::make_box_mesh<SType>(spec, deg, fdd::comm().rank);
            setup_lap("box mesh", deg, mark);
            p->domains[deg].initialize(std::move(mesh));
            setup_lap("Domain::initialize", deg, mark);
        }
        // `dim` follows the last mesh read in the reference; keep the fine one current
        fdd::globals().dim = p->fine().mesh.dim;

        finish_problem(p, flags, subdomain_overlap, superdomain_overlap);
        *out = p;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_create_dir(fddh_problem **out, const char *directory, int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int with_subdomain)
{
    try
    {
        if (int rc = rank_check()) return rc;
        return fddh_problem_create_dir_ex(out, directory, poly_degree, poly_reduction, subdomain_overlap, superdomain_overlap, with_subdomain ? FDDH_WITH_SUBDOMAIN : 0);
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_create_dir_ex(fddh_problem **out, const char *directory, int poly_degree, int poly_reduction, int subdomain_overlap, int superdomain_overlap, int flags)
{
    try
    {
        if (int rc = rank_check()) return rc;
        const int with_subdomain = flags & FDDH_WITH_SUBDOMAIN;
        if (!out || !directory || poly_degree < 1 || poly_reduction < 1) return fail("bad argument");
        fddh_problem *p = new fddh_problem();
        p->poly_degree = poly_degree;
        p->poly_reduction = poly_reduction;
        p->degrees = with_subdomain ? level_degrees(poly_degree, poly_reduction) : std::vector<int>(1, poly_degree);
        for (int deg : p->degrees)
        {
            MeshData<SType> m;
            if (!Domain<SType>::read_mesh_files(directory, deg, fdd::comm().rank, m))
            {
                delete p;
                return fail("cannot read mesh files of degree %d for rank %d under '%s'", deg, fdd::comm().rank, directory);
            }
            rstdout("Setting up domain \"N = %d\" object...\n", deg);
            p->domains[deg].initialize(std::move(m));
        }
        fdd::globals().dim = p->fine().mesh.dim;
        finish_problem(p, flags, subdomain_overlap, superdomain_overlap);
        *out = p;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_destroy(fddh_problem *p)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        // Like the reference (empty destructors, domain.tpp:24-28), device memory of
        // the host classes is released at process end; the staging vectors are freed.
        if (!p) return 0;
        p->a.free();
        p->b.free();
        p->c.free();
        p->sa.free();
        p->sb.free();
        delete p;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_write_box_mesh_files(const char *directory, const int E[3], const int P[3], int poly_degree, int rank)
{
    return fddh_write_kershaw_mesh_files(directory, E, P, poly_degree, rank, 1.0, 1.0);
}

int fddh_write_kershaw_mesh_files(const char *directory, const int E[3], const int P[3], int poly_degree, int rank, double eps_y, double eps_z)
{
    try
    {
        if (!directory || !E || !P) return fail("null argument");
        if (!(eps_y > 0.0 && eps_y <= 1.0 && eps_z > 0.0 && eps_z <= 1.0)) return fail("Kershaw eps must lie in (0, 1]");
        fdd::BoxSpec spec;
        for (int d = 0; d < 3; d++)
        {
            spec.E[d] = E[d];
            spec.P[d] = P[d];
        }
        spec.kershaw_eps[0] = eps_y;
        spec.kershaw_eps[1] = eps_z;
        MeshData<SType> m = fdd::make_box_mesh<SType>(spec, poly_degree, rank);
        char sub[4096];
        mkdir(directory, 0777);
        snprintf(sub, sizeof(sub), "%s/lx1_%d", directory, poly_degree + 1);
        mkdir(sub, 0777);
        if (!Domain<SType>::write_mesh_files(directory, rank, m)) return fail("cannot write mesh files under '%s'", directory);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_info(const fddh_problem *p, long long *info, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !info) return fail("null argument");
        const Domain<SType> &d = p->fine();
        long long v[FDDH_INFO_COUNT];
        v[FDDH_INFO_NUM_LOCAL_POINTS] = d.num_local_points;
        v[FDDH_INFO_NUM_LOCAL_NODES] = d.num_local_nodes;
        v[FDDH_INFO_NUM_BDARY_NODES] = d.boundary_nodes_count();
        v[FDDH_INFO_NUM_INTERFACE_SLOTS] = d.interface_slots_count();
        v[FDDH_INFO_NUM_TOTAL_NODES] = d.num_total_nodes;
        v[FDDH_INFO_NUM_TOTAL_ELEMENTS] = d.num_total_elements;
        v[FDDH_INFO_NUM_LOCAL_ELEMENTS] = d.num_local_elements;
        v[FDDH_INFO_NUM_LEVELS] = (long long)p->degrees.size();
        v[FDDH_INFO_SUB_NUM_VALUES] = p->subdomain ? p->subdomain->num_values : 0;
        v[FDDH_INFO_SUB_NUM_DOFS] = p->subdomain ? p->subdomain->dofs() : 0;
        v[FDDH_INFO_NUM_ITERATIONS] = d.num_iterations;
        v[FDDH_INFO_DIM] = d.mesh.dim;
        for (int i = 0; i < n && i < FDDH_INFO_COUNT; i++) info[i] = v[i];
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_info(const fddh_problem *p, long long *info, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !info) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        const Subdomain<PType> &s = *p->subdomain;
        const fdd::composite::Composite &c = s.composite_description();
        long long v[FDDH_SUB_INFO_COUNT];
        const int own_elems = p->fine().num_local_elements, own_pts = p->fine().num_local_points;
        v[FDDH_SUB_IS_COMPOSITE] = s.composite() ? 1 : 0;
        v[FDDH_SUB_NUM_ELEMS] = s.composite() ? c.num_sub_elems : own_elems;
        v[FDDH_SUB_NUM_EXT_ELEMS] = s.composite() ? c.num_sub_ext_elems : own_elems;
        v[FDDH_SUB_NUM_POINTS] = s.composite() ? c.num_sub_ext_points : own_pts;
        v[FDDH_SUB_NUM_SUB_DOFS] = s.composite() ? c.sub_num_dofs : s.dofs();
        v[FDDH_SUB_NUM_SUB_EXT_DOFS] = s.composite() ? c.sub_num_ext_dofs : s.dofs();
        v[FDDH_SUB_NUM_INTERFACE_DOFS] = s.composite() ? c.num_interface_dofs : 0;
        v[FDDH_SUB_NUM_SUP_DOFS] = s.composite() ? c.sup_num_dofs : 0;
        v[FDDH_SUB_NUM_SUP_EXT_DOFS] = s.composite() ? c.sup_num_ext_dofs : 0;
        v[FDDH_SUB_NUM_UNIQUE_DOFS] = s.dofs();
        v[FDDH_SUB_NUM_COARSE_DOFS] = s.composite() ? c.num_coarse_dofs : 0;
        v[FDDH_SUB_NUM_VALUES] = s.num_values;
        v[FDDH_SUB_OWN_POINTS] = own_pts;
        v[FDDH_SUB_NUM_PEERS] = s.composite() ? (long long)c.peers.size() : 0;
        for (int i = 0; i < n && i < FDDH_SUB_INFO_COUNT; i++) info[i] = v[i];
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_region(const fddh_problem *p, int *element, int *level, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !element || !level) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        const Subdomain<PType> &s = *p->subdomain;
        if (!s.composite()) return fail("the region is the rank's own elements (no composite)");
        const fdd::composite::Composite &c = s.composite_description();
        if (n != c.num_sub_ext_elems) return fail("the region has %d elements", c.num_sub_ext_elems);
        for (int r = 0; r < n; r++)
        {
            element[r] = c.sub[r].id;
            level[r] = c.sub[r].level;
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_composite_levels(const fddh_problem *p, int *kept, int n, int *num_levels)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !num_levels) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        const fdd::composite::Composite &c = p->subdomain->composite_description();
        *num_levels = (int)c.comp_levels.size();
        for (int i = 0; i < n && i < *num_levels && kept; i++) kept[i] = c.comp_levels[i];
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_level_degree(const fddh_problem *p, int level, int *poly_degree)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !poly_degree || level < 0 || level >= (int)p->degrees.size()) return fail("bad level");
        *poly_degree = p->degrees[level];
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_mesh_array(const fddh_problem *p, int level, const char *name, void *out, size_t bytes)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !name || !out || level < 0 || level >= (int)p->degrees.size()) return fail("bad argument");
        const MeshData<SType> &m = p->domains.at(p->degrees[level]).mesh;
        const void *src = nullptr;
        size_t have = 0;
        std::string s(name);
        if (s == "x") { src = m.x.data(); have = m.x.size() * sizeof(SType); }
        else if (s == "y") { src = m.y.data(); have = m.y.size() * sizeof(SType); }
        else if (s == "z") { src = m.z.data(); have = m.z.size() * sizeof(SType); }
        else if (s == "glo_num") { src = m.glo_num.data(); have = m.glo_num.size() * sizeof(long long); }
        else if (s == "node_degree") { src = m.node_degree.data(); have = m.node_degree.size() * sizeof(int); }
        else if (s == "p_mask") { src = m.p_mask.data(); have = m.p_mask.size() * sizeof(SType); }
        else if (s.size() == 3 && s[0] == 'g' && s[1] == '_' && s[2] >= '1' && s[2] <= '6') { const int g = s[2] - '1'; src = m.g[g].data(); have = m.g[g].size() * sizeof(SType); }
        else return fail("unknown mesh array '%s'", name);
        if (bytes != have) return fail("mesh array '%s' has %zu bytes, caller gave %zu", name, have, bytes);
        memcpy(out, src, have);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_csr(const fddh_problem *cp, int which, int *num_rows, int *num_cols, int *num_nnz, int *ptr, int *col, double *val)
{
    try
    {
        if (!cp) return fail("null argument");
        fddh_problem *p = const_cast<fddh_problem *>(cp);
        CSR_Matrix<SType> &A = (which == 0) ? p->fine().scatter_matrix() : p->fine().gather_matrix();
        if (num_rows) *num_rows = A.num_rows;
        if (num_cols) *num_cols = A.num_cols;
        if (num_nnz) *num_nnz = A.num_nnz;
        if (ptr) memcpy(ptr, A.ptr_hst.data(), A.ptr_hst.size() * sizeof(int));
        if (col) memcpy(col, A.col_hst.data(), A.col_hst.size() * sizeof(int));
        if (val) memcpy(val, A.val_hst.data(), A.val_hst.size() * sizeof(double));
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_assembled_weight(const fddh_problem *cp, double *out, int n)
{
    try
    {
        if (!cp || !out) return fail("null argument");
        fddh_problem *p = const_cast<fddh_problem *>(cp);
        if (n != p->fine().num_local_nodes) return fail("assembled_weight has %d entries", p->fine().num_local_nodes);
        p->fine().assembled_weight_memory().copyTo(out, (size_t)n * sizeof(double));
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_set_D_hat(fddh_problem *p, int level, const double *D_hat, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !D_hat || level < 0 || level >= (int)p->degrees.size()) return fail("bad argument");
        if (n != p->degrees[level] + 1) return fail("D_hat of level %d is %d x %d", level, p->degrees[level] + 1, p->degrees[level] + 1);
        p->domains[p->degrees[level]].set_D_hat(D_hat, n);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_get_D_hat(const fddh_problem *p, int level, double *D_hat, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !D_hat || level < 0 || level >= (int)p->degrees.size()) return fail("bad argument");
        if (n != p->degrees[level] + 1) return fail("wrong size");
        const std::vector<double> &D = p->domains.at(p->degrees[level]).D_hat_hst;
        memcpy(D_hat, D.data(), D.size() * sizeof(double));
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_set_options(fddh_problem *p, int max_iterations, double tolerance, int num_vectors, int use_preconditioner, int preconditioner_type, int sub_num_vectors, int sub_max_iterations, int sub_build_tree)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p) return fail("null argument");
        Domain<SType> &d = p->fine();
        if (max_iterations >= 0) d.max_iterations = max_iterations;
        if (!std::isnan(tolerance) && tolerance >= 0.0) d.tolerance = tolerance;
        if (num_vectors > 0) d.num_vectors = num_vectors;
        if (use_preconditioner >= 0)
        {
            if (use_preconditioner && !p->subdomain) return fail("problem was created without a Subdomain");
            d.use_preconditioner = use_preconditioner != 0;
        }
        if (preconditioner_type >= 0) d.preconditioner_type = preconditioner_type;
        if (p->subdomain)
        {
            if (sub_num_vectors > 0) p->subdomain->num_vectors = sub_num_vectors;
            if (sub_max_iterations >= 0) p->subdomain->max_iterations = sub_max_iterations;
            if (sub_build_tree >= 0) p->subdomain->build_tree = sub_build_tree != 0;
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_affine_info(fddh_problem *p, int *fine_domain_affine, int *sub_lists_affine, int *sub_lists, double *max_deviation)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p) return fail("null argument");
        Domain<SType> &dom = p->fine();
        double worst = dom.affine_deviation;
        int lists = 0, affine = 0;
        if (p->subdomain)
            for (auto &ll : p->subdomain->operator_lists())
            {
                lists++;
                if (ll.affine) affine++;
                worst = std::max(worst, ll.affine_deviation);
            }
        if (fine_domain_affine) *fine_domain_affine = dom.affine_geometry ? 1 : 0;
        if (sub_lists_affine) *sub_lists_affine = affine;
        if (sub_lists) *sub_lists = lists;
        if (max_deviation) *max_deviation = worst;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_set_flag(fddh_problem *p, const char *name, int value)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !name) return fail("null argument");
        const std::string s(name);
        if (s == "fused_dssum")
        {
            for (auto &kv : p->domains) kv.second.fused_dssum = value != 0;
            if (p->subdomain) p->subdomain->fused_dssum = value != 0;
        }
        else if (s == "restructured_inner_solve")
        {
            if (p->subdomain) p->subdomain->restructured = value != 0;
            for (auto &kv : p->domains) kv.second.restructured_outer = value != 0; // the same restructure of the outer GMRES
        }
        else if (s == "neighbour_interface_exchange")
        {
            // gs_add on the boundary prefix as grouped sends / receives between the ranks that share nodes (default) or as
            // the dense interface-slot all-reduce; must be set alike on every rank
            for (auto &kv : p->domains) kv.second.neighbour_interface_exchange = value != 0;
        }
        else if (s == "fold_coarse_exchange")
        {
            // the coarse level's blocks inside the ring pull's group (default) or as an all-gather of their own; alike on every rank
            if (p->subdomain) p->subdomain->fold_coarse_exchange = value != 0;
        }
        else if (s == "assembled_outer_solve")
        {
            for (auto &kv : p->domains) kv.second.assembled_outer = value != 0;
        }
        else if (s == "shared_residual_norm")
        {
            for (auto &kv : p->domains) kv.second.shared_residual_norm = value != 0;
        }
        else if (s == "skip_last_basis_store")
        {
            if (p->subdomain) p->subdomain->skip_last_basis_store = value != 0;
        }
        else if (s == "early_gamma")
        {
            for (auto &kv : p->domains) kv.second.early_gamma = value != 0;
        }
        else if (s == "unit_stitch_in_place")
        {
            for (auto &kv : p->domains) kv.second.unit_stitch_in_place = value != 0;
        }
        else if (s == "lazy_steps")
        {
            for (auto &kv : p->domains) kv.second.lazy_steps = value != 0;
        }
        else if (s == "device_bookkeeping")
        {
            if (p->subdomain) p->subdomain->device_bookkeeping = value != 0;
            for (auto &kv : p->domains) kv.second.device_scalars = value != 0;
        }
        else if (s == "assembled_inner_solve")
        {
            if (p->subdomain) p->subdomain->assembled_inner = value != 0;
        }
        else if (s == "mfma_stiffness")
        {
            for (auto &kv : p->domains) kv.second.mfma_stiffness = value != 0;
            if (p->subdomain) p->subdomain->mfma_stiffness = value != 0;
        }
        else if (s == "sub_use_preconditioner")
        {
            if (!p->subdomain) return fail("problem was created without a Subdomain");
            p->subdomain->use_preconditioner = value == 1;
            p->subdomain->use_jacobi = value == 2;
        }
        else if (s == "affine_geometry")
        {
            // an option of this build: elements that are affine images of the reference cube do not stream their factor
            // arrays (fdd_stiffness_matrix_affine); only where the mesh's own arrays have that form (fddh_problem_affine_info)
            for (auto &kv : p->domains) kv.second.set_affine_geometry(value != 0);
            if (p->subdomain) p->subdomain->set_affine_geometry(value != 0);
        }
        else if (s == "amg_graph")
        {
            if (p->subdomain) p->subdomain->amg_hierarchy.use_graph = value != 0;
        }
        else if (s == "amg_fused_smoother")
        {
            if (p->subdomain) p->subdomain->amg_hierarchy.fused_smoother = value != 0;
        }
        else if (s == "amg_matrix_free_transfer")
        {
            // the interpolator of a geometric level applied from the lattice's weight table (fdd_lattice_prolong / _restrict)
            // instead of as two SpMVs; 0: the CSR interpolator on every level (the same operator, sums in the CSR order)
            if (p->subdomain) p->subdomain->amg_hierarchy.set_matrix_free_transfer(value != 0);
        }
        else if (s == "preconditioner_precision")
        {
            // the reference's PTYPE = Float (config.hpp:19-20): 64 or 32 for the whole inner solve (Krylov vectors, element
            // stiffness, gather, V-cycle)
            if (p->subdomain and not p->subdomain->set_precision(value)) return fail("preconditioner_precision is 64 or 32 (32: 3-D regions on the dof-space inner solve, Chebyshev order >= 2)");
        }
        else if (s == "amg_num_vcycles")
        {
            // subdomain.hpp:236 `num_vcycles` (run.py:154 rewrites the line and rebuilds; here a run-time switch)
            if (!p->subdomain) return fail("problem was created without a Subdomain");
            if (value < 1 || value > 16) return fail("amg_num_vcycles must lie in 1..16");
            p->subdomain->num_vcycles = value;
            p->subdomain->amg_hierarchy.set_num_vcycles(value);
        }
        else if (s == "amg_cheby_order")
        {
            // subdomain.hpp:237 `cheby_order` (run.py:155; clamped to 1..4 by subdomain.tpp:3477-3478).  The smoother's
            // coefficients are computed when the hierarchy is built: set it before (fddh_problem_amg_build / first solve)
            if (!p->subdomain) return fail("problem was created without a Subdomain");
            if (value < 1 || value > 4) return fail("amg_cheby_order must lie in 1..4 (subdomain.tpp:3477-3478)");
            if (p->subdomain->amg_hierarchy.ready() && value != p->subdomain->cheby_order) return fail("amg_cheby_order must be set before the hierarchy is built");
            p->subdomain->cheby_order = value;
        }
        else if (s == "amg_precision")
        {
            // AMG/config.hpp:4 `Float`: 64 (double) or 32 (float)
            if (p->subdomain and not p->subdomain->amg_hierarchy.set_precision(value)) return fail("amg_precision is 64 or 32 (32 needs a Chebyshev order of at least 2)");
        }
        else
            return fail("unknown flag '%s'", name);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_point_dofs(const fddh_problem *p, int *dof, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !dof) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        if (n != (int)p->subdomain->point_dof.size()) return fail("the subdomain region has %d points", (int)p->subdomain->point_dof.size());
        memcpy(dof, p->subdomain->point_dof.data(), (size_t)n * sizeof(int));
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_amg_add_level(fddh_problem *p, int n, const int *A_ptr, const int *A_col, const double *A_val, const double *D_val, const double *coefs, int num_coefs, int n_coarse, const int *P_ptr, const int *P_col, const double *P_val)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !A_ptr || !A_col || !A_val || !D_val || !coefs) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        Subdomain<SType> &s = *p->subdomain;
        if (s.amg_hierarchy.ready()) return fail("the AMG hierarchy is already finalized");
        if (num_coefs != s.cheby_order) return fail("cheby_order is %d, got %d coefficients", s.cheby_order, num_coefs);
        if (s.amg_hierarchy.levels.empty() && n != s.dofs()) return fail("the finest AMG level must have the subdomain's %d dofs, got %d", s.dofs(), n);
        if (!s.amg_hierarchy.levels.empty())
        {
            const amg::Level &prev = s.amg_hierarchy.levels.back();
            if (prev.P.num_rows == 0) return fail("the previous level was given without a prolongation, so it is the coarsest");
            if (prev.P.num_cols != n) return fail("level size %d does not match the previous prolongation's %d columns", n, prev.P.num_cols);
        }
        if ((P_ptr != nullptr) != (n_coarse > 0)) return fail("n_coarse > 0 exactly when a prolongation is given");
        s.amg_add_level(n, A_ptr, A_col, A_val, D_val, coefs, n_coarse, P_ptr, P_col, P_val);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_amg_finalize(fddh_problem *p)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        if (p->subdomain->amg_hierarchy.levels.empty()) return fail("no AMG level was added");
        if (p->subdomain->amg_hierarchy.levels.back().P.num_rows != 0) return fail("the last AMG level still has a prolongation: add its coarse level first");
        p->subdomain->amg_finalize();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_amg_build(fddh_problem *p, int coarsest_size, double strength, int smooth_prolongator, int verbose, int *num_levels)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        fdd::low_order::Options o;
        if (coarsest_size > 0) o.coarsest_size = coarsest_size;
        if (strength > 0.0) o.strength = strength;
        o.smooth_prolongator = smooth_prolongator != 0;
        const int nl = p->subdomain->amg_build(o, verbose != 0);
        if (num_levels) *num_levels = nl;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_amg_level_info(const fddh_problem *p, int level, int *n, int *nnz_A, int *n_coarse, int *nnz_P)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !p->subdomain) return fail("no Subdomain");
        const auto &lv = p->subdomain->amg_hierarchy.levels;
        if (level < 0 || level >= (int)lv.size()) return fail("the hierarchy has %d levels", (int)lv.size());
        if (n) *n = lv[level].n;
        if (nnz_A) *nnz_A = lv[level].A.num_nnz;
        if (n_coarse) *n_coarse = lv[level].P.num_cols;
        if (nnz_P) *nnz_P = lv[level].P.num_nnz;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_amg_level_transfer(const fddh_problem *p, int level, int *matrix_free)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !p->subdomain || !matrix_free) return fail("no Subdomain");
        const auto &H = p->subdomain->amg_hierarchy;
        if (level < 0 || level >= (int)H.levels.size()) return fail("the hierarchy has %d levels", (int)H.levels.size());
        *matrix_free = (H.levels[level].T.active && H.matrix_free_transfer) ? 1 : 0;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_amg_level_arrays(const fddh_problem *cp, int level, int *A_ptr, int *A_col, double *A_val, double *D_val, double *coefs, int *P_ptr, int *P_col, double *P_val)
{
    try
    {
        if (!cp || !cp->subdomain) return fail("no Subdomain");
        fddh_problem *p = const_cast<fddh_problem *>(cp);
        auto &lv = p->subdomain->amg_hierarchy.levels;
        if (level < 0 || level >= (int)lv.size()) return fail("the hierarchy has %d levels", (int)lv.size());
        amg::Level &L = lv[level];
        if (A_ptr) memcpy(A_ptr, L.A.ptr_hst.data(), L.A.ptr_hst.size() * sizeof(int));
        if (A_col) memcpy(A_col, L.A.col_hst.data(), L.A.col_hst.size() * sizeof(int));
        if (A_val) memcpy(A_val, L.A.val_hst.data(), L.A.val_hst.size() * sizeof(double));
        if (D_val) L.D_val.copyTo(D_val, (size_t)L.n * sizeof(double));
        if (coefs) memcpy(coefs, L.coefs.data(), L.coefs.size() * sizeof(double));
        if (L.P.num_rows > 0)
        {
            if (P_ptr) memcpy(P_ptr, L.P.ptr_hst.data(), L.P.ptr_hst.size() * sizeof(int));
            if (P_col) memcpy(P_col, L.P.col_hst.data(), L.P.col_hst.size() * sizeof(int));
            if (P_val) memcpy(P_val, L.P.val_hst.data(), L.P.val_hst.size() * sizeof(double));
        }
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

// z = low_order_preconditioner(r) on level-0 subdomain points (subdomain.tpp:3987-4159)
int fddh_problem_amg_apply(fddh_problem *p, const double *r, double *z)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !r || !z) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        if (!p->subdomain->amg_hierarchy.ready()) return fail("no finalized AMG hierarchy is attached");
        const size_t bytes = (size_t)p->fine().num_local_points * sizeof(double);
        p->a.copyFrom(r, bytes);
        p->subdomain->apply_low_order_preconditioner(p->b, p->a);
        p->b.copyTo(z, bytes);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_dssum(fddh_problem *p, double *out, const double *in, int apply_mask, int apply_weight)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !out || !in) return fail("null argument");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        p->a.copyFrom(in, bytes);
        d.direct_stiffness_summation(p->b, p->a, apply_mask != 0, apply_weight != 0);
        p->b.copyTo(out, bytes);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_stiffness(fddh_problem *p, double *out, const double *in, int apply_dssum)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !out || !in) return fail("null argument");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        p->a.copyFrom(in, bytes);
        d.stiffness_matrix(p->b, p->a, apply_dssum != 0);
        p->b.copyTo(out, bytes);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_residual_norm(fddh_problem *p, const double *r, double *norm)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !r || !norm) return fail("null argument");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        // ||r|| = sqrt(<r, r>) through the public pieces (Domain::residual_norm is private, domain.hpp:71)
        p->a.copyFrom(r, bytes);
        d.direct_stiffness_summation(p->b, p->a);
        fdd::memory ws = fdd::dev().malloc<double>(fdd_reduce_workspace_doubles());
        fdd::memory sc = fdd::dev().malloc<double>(1);
        FDD_CALL(fdd_dom_residual_norm(sc.as<double>(), ws.as<double>(), p->a.as<double>(), p->b.as<double>(), d.dirichlet_mask_memory().as<double>(), d.num_local_points, fdd::dev().stream));
        if (fdd::comm().size > 1) fdd::comm().allreduce_sum(sc.as<double>(), 1);
        double v = 0.0;
        sc.copyTo(&v, sizeof(double));
        ws.free();
        sc.free();
        *norm = std::sqrt(v);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_make_rhs(fddh_problem *p, int function_id, unsigned long long seed, double *u_star, double *f)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p) return fail("null argument");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        d.initial_function(p->a, function_id, seed); // poisson.cpp:211-213
        d.stiffness_matrix(p->b, p->a);              // poisson.cpp:219 (no dssum)
        if (u_star) p->a.copyTo(u_star, bytes);
        if (f) p->b.copyTo(f, bytes);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_make_rhs_from(fddh_problem *p, double *u_star_inout, double *f)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !u_star_inout) return fail("null argument");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        p->a.copyFrom(u_star_inout, bytes);
        d.direct_stiffness_summation(p->a, p->a, true, true); // domain.tpp:579
        d.stiffness_matrix(p->b, p->a);
        p->a.copyTo(u_star_inout, bytes);
        if (f) p->b.copyTo(f, bytes);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_solve(fddh_problem *p, int solver_id, const double *f, double *u, double *history, int history_cap, int *num_history, int *num_iterations)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !f || !u) return fail("null argument");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        p->a.copyFrom(f, bytes);

        if (p->subdomain)
        {
            if (solver_id == 0)
                d.flexible_conjugate_gradient(p->b, p->a, *p->subdomain);
            else
                d.generalized_minimum_residual(p->b, p->a, *p->subdomain);
        }
        else
        {
            if (solver_id == 0)
                d.flexible_conjugate_gradient(p->b, p->a, p->none);
            else
                d.generalized_minimum_residual(p->b, p->a, p->none);
        }

        p->b.copyTo(u, bytes);
        const int nh = (int)d.residual_history.size();
        if (history)
            for (int i = 0; i < nh && i < history_cap; i++) history[i] = d.residual_history[i];
        if (num_history) *num_history = nh;
        if (num_iterations) *num_iterations = d.num_iterations;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_solve_timed(fddh_problem *p, int solver_id, const double *f, double *u, double *history, int history_cap, int *num_history, int *num_iterations, double *seconds)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !f || !seconds) return fail("null argument");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        p->a.copyFrom(f, bytes);
        fdd::dev().finish();
        fdd::comm().barrier();
        const auto t0 = std::chrono::steady_clock::now();
        if (p->subdomain)
        {
            if (solver_id == 0)
                d.flexible_conjugate_gradient(p->b, p->a, *p->subdomain);
            else
                d.generalized_minimum_residual(p->b, p->a, *p->subdomain);
        }
        else
        {
            if (solver_id == 0)
                d.flexible_conjugate_gradient(p->b, p->a, p->none);
            else
                d.generalized_minimum_residual(p->b, p->a, p->none);
        }
        fdd::dev().finish();
        *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (u) p->b.copyTo(u, bytes);
        const int nh = (int)d.residual_history.size();
        if (history)
            for (int i = 0; i < nh && i < history_cap; i++) history[i] = d.residual_history[i];
        if (num_history) *num_history = nh;
        if (num_iterations) *num_iterations = d.num_iterations;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_precond_apply(fddh_problem *p, int type, const double *r, double *z, double *history, int history_cap, int *num_history)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !r || !z) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        Domain<SType> &d = p->fine();
        const size_t bytes = (size_t)d.num_local_points * sizeof(double);
        p->a.copyFrom(r, bytes);
        if (type == 0)
            p->subdomain->flexible_conjugate_gradient(p->b, p->a);
        else
            p->subdomain->generalized_minimum_residual(p->b, p->a);
        p->b.copyTo(z, bytes);
        p->subdomain->finish_history();
        const int nh = (int)p->subdomain->residual_history.size();
        if (history)
            for (int i = 0; i < nh && i < history_cap; i++) history[i] = p->subdomain->residual_history[i];
        if (num_history) *num_history = nh;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_op(fddh_problem *p, int op, const double *in, double *out)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !in || !out) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        Subdomain<PType> &s = *p->subdomain;
        const size_t bytes = (size_t)s.num_values * sizeof(double);
        if (op == 0)
        {
            // tree_operator: input is an outer (Domain) vector
            p->a.copyFrom(in, (size_t)p->fine().num_local_points * sizeof(double));
            s.apply_tree_operator(p->sb, p->a);
        }
        else
        {
            p->sa.copyFrom(in, bytes);
            if (op == 1)
                s.stiffness_matrix(p->sb, p->sa);
            else if (op == 2)
                s.direct_stiffness_summation(p->sb, p->sa);
            else
                return fail("unknown subdomain op %d", op);
        }
        p->sb.copyTo(out, bytes);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_dof_op(fddh_problem *p, int op, const double *in, double *out, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !in || !out) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        Subdomain<PType> &s = *p->subdomain;
        if (not s.dof_space_available()) return fail("the inner iteration of this problem does not run in dof space");
        if (n != s.dof_count()) return fail("dof vectors have %d entries, got %d", s.dof_count(), n);
        if (op == 0)
            s.host_operator_dofs(out, in);
        else if (op == 1)
        {
            p->a.copyFrom(in, (size_t)p->fine().num_local_points * sizeof(double));
            s.host_rhs_dofs(out, p->a);
        }
        else
            return fail("unknown dof-space op %d", op);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_jacobi_diagonal(fddh_problem *p, double *out, int n)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !out) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        const std::vector<double> &d = p->subdomain->jacobi_diagonal();
        if (n != p->subdomain->dofs()) return fail("the diagonal has %d entries, got %d", p->subdomain->dofs(), n);
        for (int i = 0; i < n; i++) out[i] = d[i];
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_sub_residual_norm(fddh_problem *p, const double *r, double *norm)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !r || !norm) return fail("null argument");
        if (!p->subdomain) return fail("problem was created without a Subdomain");
        p->sa.copyFrom(r, (size_t)p->subdomain->num_values * sizeof(double));
        p->subdomain->compute_residual_norm(*norm, p->sa);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_pcg_begin(fddh_problem *p, const double *f)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !f) return fail("null argument");
        Domain<SType> &d = p->fine();
        p->a.copyFrom(f, (size_t)d.num_local_points * sizeof(double));
        if (p->subdomain && d.use_preconditioner)
            d.fcg_begin(p->b, p->a, *p->subdomain);
        else
            d.fcg_begin(p->b, p->a, p->none);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_pcg_steps(fddh_problem *p, int steps, double *last_residual)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || steps < 0) return fail("bad argument");
        Domain<SType> &d = p->fine();
        double r = std::numeric_limits<double>::quiet_NaN();
        if (p->subdomain && d.use_preconditioner)
            r = d.fcg_steps(*p->subdomain, steps);
        else
            r = d.fcg_steps(p->none, steps);
        if (last_residual) *last_residual = r;
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_pcg_solution(fddh_problem *p, double *u)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !u) return fail("null argument");
        p->fine().fcg_finish();
        p->b.copyTo(u, (size_t)p->fine().num_local_points * sizeof(double));
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

// Average launch time of the assembly SpMVs on the problem's own matrices, timed with
// events on the stream (the "SpMV GB/s" half of the headline metric): which = 0: Q x
// (scatter, one non-zero per row), 1: Qt x (gather, 1-8 non-zeros per row).
int fddh_problem_spmv_time(fddh_problem *p, int which, int iterations, double *avg_us, double *algorithmic_bytes)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !avg_us || !algorithmic_bytes || iterations < 1 || which < 0 || which > 1) return fail("bad argument");
        Domain<SType> &d = p->fine();
        CSR_Matrix<SType> &A = (which == 0) ? d.scatter_matrix() : d.gather_matrix();
        fdd::memory x = fdd::dev().malloc<double>(std::max(A.num_cols, 1));
        fdd::memory y = fdd::dev().malloc<double>(std::max(A.num_rows, 1));
        FDD_CALL(fdd_set_to_value(x.as<double>(), 1.0, A.num_cols, 0, fdd::dev().stream));
        void *e0 = nullptr, *e1 = nullptr;
        FDD_CALL(fdd_event_create(&e0));
        FDD_CALL(fdd_event_create(&e1));
        for (int i = 0; i < 3; i++) A.multiply(y, x);
        FDD_CALL(fdd_event_record(e0, fdd::dev().stream));
        for (int i = 0; i < iterations; i++) A.multiply(y, x);
        FDD_CALL(fdd_event_record(e1, fdd::dev().stream));
        float ms = 0.0f;
        FDD_CALL(fdd_event_elapsed_ms(&ms, e0, e1));
        FDD_CALL(fdd_event_destroy(e0));
        FDD_CALL(fdd_event_destroy(e1));
        x.free();
        y.free();
        *avg_us = 1.0e3 * ms / iterations;
        *algorithmic_bytes = A.algorithmic_bytes(false);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_problem_comm_time(fddh_problem *p, int iterations, double *avg_us, double *bytes)
{
    try
    {
        if (int rc = rank_check(p)) return rc;
        if (!p || !avg_us || !bytes || iterations < 1) return fail("bad argument");
        for (int k = 0; k < FDDH_COMM_TIME_COUNT; k++) avg_us[k] = bytes[k] = 0.0;
        fdd::Comm &c = fdd::comm();
        if (c.size == 1) return 0;
        Domain<SType> &d = p->fine();
        fdd::memory scal = fdd::dev().malloc<double>(4);
        FDD_CALL(fdd_memset(scal.ptr(), 0, 4 * sizeof(double), fdd::dev().stream));
        const auto clock = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        auto timed = [&](const std::function<void()> &op) {
            op(); // warm-up (first use of a peer pair may set up a connection)
            fdd::dev().finish();
            c.barrier();
            const double t0 = clock();
            for (int i = 0; i < iterations; i++) op();
            fdd::dev().finish();
            return 1.0e6 * (clock() - t0) / iterations;
        };
        // every rank makes the same calls in the same order
        avg_us[0] = timed([&] { c.allreduce_sum(scal.as<double>(), 3); });
        bytes[0] = 3 * sizeof(double);
        avg_us[1] = timed([&] { d.comm_probe_interface(2, false); });
        bytes[1] = 2.0 * d.interface_slots_count() * sizeof(double);
        avg_us[4] = timed([&] { d.comm_probe_interface(2, true); });
        bytes[4] = d.comm_interface_neighbour_bytes(2);
        if (p->subdomain)
        {
            avg_us[2] = timed([&] { p->subdomain->comm_probe_coarse(); });
            bytes[2] = p->subdomain->comm_coarse_bytes();
            avg_us[3] = timed([&] { p->subdomain->comm_probe_ring(); });
            bytes[3] = p->subdomain->comm_ring_bytes();
            avg_us[5] = timed([&] { p->subdomain->comm_probe_ring_and_coarse(); });
            bytes[5] = p->subdomain->comm_ring_and_coarse_bytes();
        }
        scal.free();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_spmv_stencil_time(int m, int iterations, double *avg_us, double *algorithmic_bytes, long long *num_nnz)
{
    try
    {
        if (m < 2 || iterations < 1 || !avg_us || !algorithmic_bytes) return fail("bad argument");
        const long long n = (long long)m * m * m;
        if (n > 2000000000LL) return fail("grid too large");
        // rows in parallel ranges: counts first, then the entries (lexicographic neighbours = ascending columns)
        std::vector<int> ptr((size_t)n + 1, 0);
        auto span = [&](int i) { return (i == 0 || i == m - 1) ? 2 : 3; };
        for (int k = 0; k < m; k++)
            for (int j = 0; j < m; j++)
                for (int i = 0; i < m; i++) ptr[(size_t)i + (size_t)m * (j + (size_t)m * k) + 1] = span(i) * span(j) * span(k);
        long long total = 0;
        for (long long r = 0; r < n; r++)
        {
            total += ptr[r + 1];
            if (total > 2147483647LL) return fail("more than 2^31 non-zeros");
            ptr[r + 1] = (int)total;
        }
        std::vector<int> col((size_t)total);
        std::vector<double> val((size_t)total);
        fdd::low_order::parallel_ranges(n, fdd::low_order::range_parts(n), [&](long long r0, long long r1, int) {
            for (long long r = r0; r < r1; r++)
            {
                const int i = (int)(r % m), j = (int)((r / m) % m), k = (int)(r / ((long long)m * m));
                size_t at = (size_t)ptr[r];
                for (int dk = -1; dk <= 1; dk++)
                    for (int dj = -1; dj <= 1; dj++)
                        for (int di = -1; di <= 1; di++)
                        {
                            const int ii = i + di, jj = j + dj, kk = k + dk;
                            if (ii < 0 || jj < 0 || kk < 0 || ii >= m || jj >= m || kk >= m) continue;
                            const long long c = (long long)ii + (long long)m * (jj + (long long)m * kk);
                            col[at] = (int)c;
                            // trilinear stiffness stencil weights (8/3, 0, -1/6, -1/12 by neighbour class) times a seeded perturbation
                            const int cls = (di != 0) + (dj != 0) + (dk != 0);
                            const double base = (cls == 0) ? 8.0 / 3.0 : (cls == 1) ? -1.0e-3 : (cls == 2) ? -1.0 / 6.0 : -1.0 / 12.0;
                            val[at] = base * (1.0 + 1.0e-3 * (double)((r * 31 + c * 17) % 97));
                            at++;
                        }
            }
        });
        CSR_Matrix<SType> A;
        A.assemble_from_csr((int)n, (int)n, ptr.data(), col.data(), val.data());
        std::vector<int>().swap(col);
        std::vector<double>().swap(val);
        A.release_host();
        fdd::memory x = fdd::dev().malloc<double>((size_t)n);
        fdd::memory y = fdd::dev().malloc<double>((size_t)n);
        {
            std::vector<double> hx((size_t)n);
            for (long long r = 0; r < n; r++) hx[r] = 0.5 + 1.0e-3 * (double)((r * 7919) % 1009);
            x.copyFrom(hx.data(), (size_t)n * sizeof(double));
        }
        void *e0 = nullptr, *e1 = nullptr;
        FDD_CALL(fdd_event_create(&e0));
        FDD_CALL(fdd_event_create(&e1));
        for (int i = 0; i < 3; i++) A.multiply(y, x);
        FDD_CALL(fdd_event_record(e0, fdd::dev().stream));
        for (int i = 0; i < iterations; i++) A.multiply(y, x);
        FDD_CALL(fdd_event_record(e1, fdd::dev().stream));
        float ms = 0.0f;
        FDD_CALL(fdd_event_elapsed_ms(&ms, e0, e1));
        FDD_CALL(fdd_event_destroy(e0));
        FDD_CALL(fdd_event_destroy(e1));
        *avg_us = 1.0e3 * ms / iterations;
        *algorithmic_bytes = A.algorithmic_bytes(false);
        if (num_nnz) *num_nnz = total;
        x.free();
        y.free();
        A.ptr.free();
        A.col.free();
        A.val.free();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_profile_enable(int on)
{
    try
    {
        fdd::profiler().reset();
        fdd::profiler().enabled = on != 0;
        fdd::profiler().only.clear();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_profile_only(const char *kernel_key)
{
    try
    {
        fdd::profiler().only = kernel_key ? kernel_key : "";
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_profile_collect(char *json, size_t json_len)
{
    try
    {
        if (!json || json_len < 3) return fail("bad buffer");
        auto stats = fdd::profiler().collect();
        std::string s = "{";
        bool first = true;
        for (auto &kv : stats)
        {
            char item[512];
            snprintf(item, sizeof(item), "%s\"%s\": {\"count\": %lld, \"ms\": %.9g, \"bytes\": %.17g}", first ? "" : ", ", kv.first.c_str(), kv.second.count, kv.second.ms, kv.second.bytes);
            s += item;
            first = false;
        }
        s += "}";
        if (s.size() + 1 > json_len) return fail("profile JSON needs %zu bytes", s.size() + 1);
        memcpy(json, s.c_str(), s.size() + 1);
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_sync(void)
{
    try
    {
        fdd::dev().finish();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

int fddh_barrier(void)
{
    try
    {
        fdd::dev().finish();
        fdd::comm().barrier();
        fdd::dev().finish();
        return 0;
    }
    catch (const std::exception &e)
    {
        return fail("%s", e.what());
    }
}

} // extern "C"
