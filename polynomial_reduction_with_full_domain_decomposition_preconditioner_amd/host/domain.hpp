/*
 * domain.hpp -- Domain<DType>: the outer-solver host class of the reference
 * (domain.hpp:33-145, domain.tpp) with the same public surface --
 * initialize, initial_function, direct_stiffness_summation, stiffness_matrix,
 * flexible_conjugate_gradient<P>, generalized_minimum_residual<P> -- driving
 * the gfx950 kernels through the C-ABI instead of OCCA.
 *
 * What changed underneath (MI355X-first, same arithmetic):
 *   - stiffness_matrix is one fused launch (fdd_dom_stiffness_matrix) instead
 *     of two with a global scratch (domain.tpp:605-606);
 *   - every dot product finishes on the device; only the final scalar(s)
 *     cross PCIe (the reference copies one partial per 128 points and sums on
 *     the host, domain.tpp:924-926);
 *   - gslib's gs_add on the boundary prefix (domain.tpp:590-594: D2H, MPI,
 *     H2D) is a device-resident scatter -> all-reduce -> gather on a dense
 *     interface-slot vector (comm.hpp), MPI_Allreduce of scalars is an
 *     all-reduce of a device buffer;
 *   - mesh data can come from memory (MeshData) as well as from the
 *     Nek5000-export files the reference reads (domain.tpp:45-224).
 */
#ifndef FDD_DOMAIN_HPP
#define FDD_DOMAIN_HPP

#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "config.hpp"
#include "csr_matrix.hpp"
#include "element.hpp"
#include "gll.hpp"
#include "math.hpp"
#include "timer.hpp"

// The "no preconditioner" plugin: Domain's solvers take any type with these
// two methods (domain.tpp:639-642); with use_preconditioner == false neither
// is called.
struct NoPreconditioner
{
    void flexible_conjugate_gradient(fdd::memory &, fdd::memory &) {}
    void generalized_minimum_residual(fdd::memory &, fdd::memory &) {}
    // the surface Domain's assembled flexible CG looks at (never taken: can_assemble() is false)
    std::vector<int> point_dof;
    bool assembled_inner = false;
    bool can_assemble() const { return false; }
    int dofs() const { return 0; }
    void gmres_dofs(fdd::memory &, fdd::memory &, bool = true, bool = false) {}
    bool lazy_history = false;
    bool composite() const { return false; }
    bool composite_dof_space() const { return false; }
    int own_dofs() const { return 0; }
    fdd::memory new_dof_vector() { return fdd::dev().malloc<double>(1); }
    fdd::memory tree_points() { return fdd::dev().malloc<double>(1); }
    void comm_probe_coarse() {}
    void comm_probe_ring() {}
    double comm_coarse_bytes() const { return 0.0; }
    double comm_ring_bytes() const { return 0.0; }
    void gmres_composite_dofs(fdd::memory &, fdd::memory &, bool = true, bool = false, const double * = nullptr) {}
    const double *known_rhs_norm2_dev = nullptr;
    bool unit_norm_weight() const { return false; }
};

template <typename DType>
class Domain
{
  private:
    // Work arrays (domain.tpp:56-67)
    std::vector<fdd::memory> work_dev;

    // Dirichlet boundary conditions
    fdd::memory dirichlet_mask;

    // Assembly
    CSR_Matrix<DType> Q;
    CSR_Matrix<DType> Qt;
    fdd::memory assembled_weight;

    // Gather scatter (replaces gs_comm / gs_handle, domain.hpp:51-54)
    int num_bdary_nodes = 0;
    int num_interface_slots = 0;
    fdd::memory bdary_slot;       // int[num_bdary_nodes]: slot of each boundary node
    fdd::memory interface_slots;  // double[num_interface_slots]
    // Neighbour form of the same exchange (xGMI is point-to-point, every pair of GPUs has its own link): each peer gets only
    // the nodes the two ranks share, by one grouped send / receive, and every sharer adds a node's copies in ascending rank
    // order.  Buffer (entries of 1 or 2 interleaved vectors): [own prefix | parts to the peers | parts from the peers].
    struct InterfacePeer
    {
        int rank = 0;
        int first = 0; // first entry of this peer's part inside the send region (= inside the receive region)
        int count = 0; // shared nodes
    };
    std::vector<InterfacePeer> iface_peers;
    int iface_shared_entries = 0;          // sum of the peers' counts
    fdd::memory iface_buf;                 // double[2 * (num_bdary_nodes + 2 * iface_shared_entries)]
    fdd::memory iface_gather_index;        // int[num_bdary_nodes + iface_shared_entries]: prefix node of every own / sent entry
    fdd::memory iface_sum_ptr, iface_sum_col; // CSR over the boundary nodes: the entries of iface_buf a node's sum reads, by contributor rank
    std::vector<fdd::ExchangeOp> iface_ops;

    // Reductions
    fdd::memory reduce_ws;
    fdd::memory scalars; // small device buffer for final dot products

    // Solver
    fdd::memory r_k, r_kp1, q_k, z_k, p_k;
    std::vector<fdd::memory> V;
    std::vector<fdd::memory> Z;
    std::vector<fdd::memory> VA; // assembled copies of the outer GMRES basis
    int va_valid = 0;            // VA[0..va_valid) match V of the current cycle
    std::vector<std::vector<DType>> H;
    std::vector<DType> c_gmres;
    std::vector<DType> s_gmres;
    std::vector<DType> gamma;
    bool gmres_allocated = false;

    // state of a running flexible CG (fcg_begin / fcg_step)
    fdd::memory fcg_u;
    DType fcg_r_0_norm = 0.0;
    DType fcg_gamma_k = 0.0;
    int fcg_iter = 0;

    // ---- assembled flexible CG: every vector holds one value per local node ----
    // (see fcg_nodes_* below; built on first use)
    bool nodes_ready = false;
    bool fcg_nodes_active = false;
    bool fcg_norm_pending = false;
    int norm_parts = 1; // scalars[4..]: boundary-prefix and interior parts of the last enqueued residual norm
    fdd::memory norm_hist;               // device-side residual history of fcg_steps
    int norm_hist_cap = 0;
    bool norm_deferred = false;          // the saved prefix still waits for its exchange
    bool norm_reduce_pending = false;    // scalars[4..5] still hold this rank's parts only
    const double *norm_source = nullptr; // the vector whose norm is being taken
    fdd::memory nprefix;                 // its boundary prefix (exchanged copy)
    fdd::memory point_node_dev;           // int[num_local_points]: Q as an index array
    fdd::memory node_mask;                // Dirichlet mask per node
    fdd::memory node_stitch;              // local multiplicity * assembled weight * mask
    std::vector<DType> node_stitch_hst, node_mask_hst;
    bool norm_on_dof_slice = false;       // one rank, mask = indicator of the dof slice: the masked node norm is the slice's plain norm
    const double *pending_rhs_norm2 = nullptr; // what node_norm_enqueue returned for the residual the next precondition_nodes call receives
    bool stitch_is_one = false;           // every weight of the dof slice is exactly 1.0 (setup_dof_maps)
    fdd::memory node_of_dof, dof_of_node; // renumbering to / from the subdomain's dofs
    int nodes_sub_dofs = -1;
    int dof_shift = -1; // >= 0: subdomain dof d is node d + dof_shift (the numbering makes it so), no renumbering pass
    fdd::memory nu, nr, nr1, nq, nz, np, nt, sub_f, sub_u;
    // node-space outer GMRES (gmres_nodes): f^ = Qt f, the basis and the preconditioned basis as node vectors, with more
    // than one rank the basis once more with the interface prefix summed over the ranks, on a composite the basis on the
    // points as well (the degree tree restricts point data)
    fdd::memory nf;
    std::vector<fdd::memory> VN, ZN, VNA, VP;
    int gmres_nodes_vectors = 0;
    fdd::memory fcg_u_pts;
    bool composite_rhs_from_nodes = true; // composite preconditioner: own part of its right-hand side = the node residual (no second gather of the own points)
    fdd::memory rp;                 // composite preconditioner: the residual on the element-local points (the degree tree and the ring / superdomain exchange start from it)
    bool composite_precond = false; // the Subdomain is a full-domain-decomposition composite driven in dof space

    Math<DType> math;

    const double *G_ptrs[NUM_GEOM_FACTS];

    // From the all-gathered boundary ids (rank order, counts[r] per rank; every rank holds the same lists, so no further
    // communication): who shares which of this rank's boundary nodes.  Peer parts are ordered by global id on both sides.
    void build_interface_neighbour_plan(const std::vector<long long> &ids_by_rank, const std::vector<int> &counts, const std::vector<long long> &mine)
    {
        const int me = fdd::comm().rank, R = fdd::comm().size, nb = (int)mine.size();
        std::vector<std::pair<long long, int>> holders; // (global id, rank), sorted: the ranks of an id come out ascending
        holders.reserve(ids_by_rank.size());
        size_t at = 0;
        for (int r = 0; r < R; r++)
            for (int k = 0; k < counts[r]; k++, at++) holders.emplace_back(ids_by_rank[at], r);
        std::sort(holders.begin(), holders.end());
        // this rank's boundary nodes by global id
        std::vector<std::pair<long long, int>> own(nb);
        for (int b = 0; b < nb; b++) own[b] = {mine[b], b};
        std::sort(own.begin(), own.end());

        // per peer: the shared nodes in ascending global id (the walk over `own` is in that order)
        std::vector<std::vector<int>> shared(R);      // prefix node
        std::vector<std::vector<int>> sharers(nb);    // ranks holding node b, ascending (this rank included)
        for (int k = 0; k < nb; k++)
        {
            const long long g = own[k].first;
            const int b = own[k].second;
            auto lo = std::lower_bound(holders.begin(), holders.end(), std::make_pair(g, -1));
            for (auto it = lo; it != holders.end() and it->first == g; ++it)
            {
                sharers[b].push_back(it->second);
                if (it->second != me) shared[it->second].push_back(b);
            }
        }
        iface_peers.clear();
        iface_shared_entries = 0;
        std::vector<int> first_of(R, -1);
        for (int r = 0; r < R; r++)
        {
            if (shared[r].empty()) continue;
            InterfacePeer pr;
            pr.rank = r;
            pr.first = iface_shared_entries;
            pr.count = (int)shared[r].size();
            first_of[r] = pr.first;
            iface_shared_entries += pr.count;
            iface_peers.push_back(pr);
        }
        std::vector<int> gidx((size_t)nb + iface_shared_entries);
        for (int b = 0; b < nb; b++) gidx[b] = b;
        std::vector<std::vector<int>> pos(R); // position of node b inside peer r's part
        for (const InterfacePeer &pr : iface_peers)
        {
            pos[pr.rank].assign(nb, -1);
            for (int k = 0; k < pr.count; k++)
            {
                gidx[(size_t)nb + pr.first + k] = shared[pr.rank][k];
                pos[pr.rank][shared[pr.rank][k]] = k;
            }
        }
        std::vector<int> sptr(nb + 1, 0), scol;
        for (int b = 0; b < nb; b++)
        {
            for (int r : sharers[b]) scol.push_back(r == me ? b : nb + iface_shared_entries + first_of[r] + pos[r][b]);
            sptr[b + 1] = (int)scol.size();
        }
        iface_buf = fdd::dev().malloc<double>(2 * ((size_t)nb + 2 * (size_t)iface_shared_entries) + 2);
        iface_gather_index = fdd::dev().malloc<int>(std::max<size_t>(gidx.size(), 1));
        if (not gidx.empty()) iface_gather_index.copyFrom(gidx.data(), gidx.size() * sizeof(int));
        iface_sum_ptr = fdd::dev().malloc<int>(sptr.size());
        iface_sum_ptr.copyFrom(sptr.data(), sptr.size() * sizeof(int));
        iface_sum_col = fdd::dev().malloc<int>(std::max<size_t>(scol.size(), 1));
        if (not scol.empty()) iface_sum_col.copyFrom(scol.data(), scol.size() * sizeof(int));
    }

    // the grouped send / receive of the neighbour form for `nc` interleaved vectors
    void iface_exchange(int nc)
    {
        iface_ops.resize(iface_peers.size());
        double *base = iface_buf.as<double>();
        for (size_t k = 0; k < iface_peers.size(); k++)
        {
            const InterfacePeer &pr = iface_peers[k];
            fdd::ExchangeOp &op = iface_ops[k];
            op.peer = pr.rank;
            op.send = base + (size_t)nc * (num_bdary_nodes + pr.first);
            op.recv = base + (size_t)nc * (num_bdary_nodes + iface_shared_entries + pr.first);
            op.send_bytes = op.recv_bytes = (size_t)nc * pr.count * sizeof(double);
        }
        fdd::comm().exchange(iface_ops.data(), (int)iface_ops.size()); // every rank calls it, peers or not (one back-end meets world-wide)
    }

    void gs_add_boundary_neighbours(fdd::memory &a, fdd::memory *b)
    {
        void *stream = fdd::dev().stream;
        FDD_CALL(fdd_interface_gather(iface_buf.as<double>(), iface_gather_index.as<int>(), num_bdary_nodes + iface_shared_entries, a.as<double>(), b ? b->as<double>() : nullptr, stream));
        iface_exchange(b ? 2 : 1);
        FDD_CALL(fdd_interface_sum(a.as<double>(), b ? b->as<double>() : nullptr, iface_sum_ptr.as<int>(), iface_sum_col.as<int>(), num_bdary_nodes, iface_buf.as<double>(), stream));
    }

    void gs_add_boundary(fdd::memory &t)
    {
        if (fdd::comm().size == 1 or num_interface_slots == 0) return;
        if (neighbour_interface_exchange)
        {
            gs_add_boundary_neighbours(t, nullptr);
            return;
        }
        FDD_CALL(fdd_memset(interface_slots.ptr(), 0, (size_t)num_interface_slots * sizeof(double), fdd::dev().stream));
        FDD_CALL(fdd_interface_pack(interface_slots.as<double>(), bdary_slot.as<int>(), t.as<double>(), num_bdary_nodes, fdd::dev().stream));
        fdd::comm().allreduce_sum(interface_slots.as<double>(), num_interface_slots);
        FDD_CALL(fdd_interface_unpack(t.as<double>(), interface_slots.as<double>(), bdary_slot.as<int>(), num_bdary_nodes, fdd::dev().stream));
    }

    // two boundary prefixes in ONE exchange (one collective latency instead of two)
    void gs_add_boundary_pair(fdd::memory &a, fdd::memory &b)
    {
        if (fdd::comm().size == 1 or num_interface_slots == 0) return;
        if (neighbour_interface_exchange)
        {
            gs_add_boundary_neighbours(a, &b);
            return;
        }
        const int S = num_interface_slots;
        void *stream = fdd::dev().stream;
        FDD_CALL(fdd_memset(interface_slots.ptr(), 0, 2 * (size_t)S * sizeof(double), stream));
        FDD_CALL(fdd_interface_pack(interface_slots.as<double>(), bdary_slot.as<int>(), a.as<double>(), num_bdary_nodes, stream));
        FDD_CALL(fdd_interface_pack(interface_slots.as<double>() + S, bdary_slot.as<int>(), b.as<double>(), num_bdary_nodes, stream));
        fdd::comm().allreduce_sum(interface_slots.as<double>(), 2 * S);
        FDD_CALL(fdd_interface_unpack(a.as<double>(), interface_slots.as<double>(), bdary_slot.as<int>(), num_bdary_nodes, stream));
        FDD_CALL(fdd_interface_unpack(b.as<double>(), interface_slots.as<double>() + S, bdary_slot.as<int>(), num_bdary_nodes, stream));
    }

    // device scalars -> (all-reduce) -> host
    void fetch_scalars(DType *out, int n)
    {
        if (fdd::comm().size > 1) fdd::comm().allreduce_sum(scalars.as<double>(), n);
        scalars.copyTo(out, n * sizeof(DType));
    }

    void residual_norm(DType &r_norm, fdd::memory &r)
    {
        direct_stiffness_summation(work_dev[1], r);
        FDD_CALL(fdd_dom_residual_norm(scalars.as<double>(), reduce_ws.as<double>(), r.as<double>(), work_dev[1].as<double>(), dirichlet_mask.as<double>(), num_local_points, fdd::dev().stream));
        fetch_scalars(&r_norm, 1);
        r_norm = std::sqrt(r_norm);
    }

    void assembled_inner_product(DType &uv, fdd::memory &u, fdd::memory &v)
    {
        direct_stiffness_summation(work_dev[1], v);
        FDD_CALL(fdd_dom_inner_product(scalars.as<double>(), reduce_ws.as<double>(), u.as<double>(), work_dev[1].as<double>(), dirichlet_mask.as<double>(), num_local_points, fdd::dev().stream));
        fetch_scalars(&uv, 1);
    }

    void projection_inner_products(DType &gamma_k, DType &theta_k, fdd::memory &z, fdd::memory &r, fdd::memory &p, fdd::memory &q)
    {
        DType values[2];
        FDD_CALL(fdd_dom_projection_inner_products(scalars.as<double>(), reduce_ws.as<double>(), z.as<double>(), r.as<double>(), p.as<double>(), q.as<double>(), num_local_points, fdd::dev().stream));
        fetch_scalars(values, 2);
        gamma_k = values[0];
        theta_k = values[1];
    }

    void solution_and_residual_update(fdd::memory &u, fdd::memory &r1, fdd::memory &r, fdd::memory &p, fdd::memory &q, DType alpha_k)
    {
        FDD_CALL(fdd_dom_solution_and_residual_update(u.as<double>(), r1.as<double>(), r.as<double>(), p.as<double>(), q.as<double>(), alpha_k, num_local_points, fdd::dev().stream));
    }

    void inner_product_flexible(DType &theta_k, fdd::memory &r, fdd::memory &r1, fdd::memory &z)
    {
        FDD_CALL(fdd_dom_inner_product_flexible(scalars.as<double>(), reduce_ws.as<double>(), r.as<double>(), r1.as<double>(), z.as<double>(), num_local_points, fdd::dev().stream));
        fetch_scalars(&theta_k, 1);
    }

    void residual_and_search_update(fdd::memory &p, fdd::memory &r, fdd::memory &z, fdd::memory &r1, DType beta_k)
    {
        FDD_CALL(fdd_dom_residual_and_search_update(p.as<double>(), r.as<double>(), z.as<double>(), r1.as<double>(), beta_k, num_local_points, fdd::dev().stream));
    }

    template <typename PType>
    void apply_preconditioner(fdd::memory &z, fdd::memory &r, PType &subdomain)
    {
        if (use_preconditioner)
        {
            fdd_timer().start("subdomain.solver");
            if (preconditioner_type == 0)
                subdomain.flexible_conjugate_gradient(z, r);
            else
                subdomain.generalized_minimum_residual(z, r);
            fdd_timer().stop("subdomain.solver");

            fdd_timer().start("subdomain.stitching");
            direct_stiffness_summation(z, z, true, true);
            fdd_timer().stop("subdomain.stitching");
        }
        else
        {
            direct_stiffness_summation(z, r);
        }
    }

    // VA[i] = mask * QQt V[i], made once per basis vector (restructured outer GMRES)
    void gmres_cache_assembled(int upto)
    {
        if ((int)VA.size() != num_vectors + 1)
        {
            for (auto &m : VA) m.free();
            VA.resize(num_vectors + 1);
            for (auto &m : VA) m = fdd::dev().malloc<DType>(num_local_points);
            va_valid = 0;
        }
        for (int i = va_valid; i <= upto; i++) direct_stiffness_summation(VA[i], V[i]);
        va_valid = std::max(va_valid, upto + 1);
    }

    void allocate_gmres()
    {
        if (gmres_allocated) return;
        V.resize(num_vectors + 1);
        for (int i = 0; i < num_vectors + 1; i++) V[i] = fdd::dev().malloc<DType>(num_local_points);
        Z.resize(num_vectors);
        for (int i = 0; i < num_vectors; i++) Z[i] = fdd::dev().malloc<DType>(num_local_points);
        H.assign(num_vectors, std::vector<DType>(num_vectors, 0.0));
        c_gmres.assign(num_vectors, 0.0);
        s_gmres.assign(num_vectors, 0.0);
        gamma.assign(num_vectors + 1, 0.0);
        gmres_allocated = true;
    }

  public:
    // Member variables (domain.hpp:94-123)
    std::string directory;
    int poly_degree = 1;
    const char *data_type = "double";

    int num_total_elements = 0;
    long long num_total_points = 0;
    long long num_total_nodes = 0; // unique global nodes (the reference never assigns it, domain.hpp:100)

    int num_local_elements = 0;
    int num_local_points = 0;
    int num_local_nodes = 0;

    int num_elem_points = 0;

    MeshData<DType> mesh;
    std::vector<Element<DType>> elements;

    // Solver
    int num_blocks = 0;
    int num_iterations = 0;
    int num_vectors = 20;
    int max_iterations = 500;
    int preconditioner_type = 1;
    bool use_preconditioner = true;
    bool fused_dssum = true; // gather-scatter kernel instead of the reference's Qt / Q SpMV pair
    bool neighbour_interface_exchange = true; // gs_add on the boundary prefix: grouped sends / receives between the sharing ranks (false: the dense interface-slot all-reduce)
    bool restructured_outer = true; // outer GMRES: cached assembled basis, multi-dot, multi-axpy
    bool assembled_outer = true; // flexible CG on node vectors (one value per assembled node) when the configuration allows it
    bool device_scalars = true;  // node-space flexible CG: alpha / beta read by the update kernels from device memory, residual norm fetched late
    bool lazy_steps = true;      // fcg_steps: K iterations with one host synchronisation at the end
    bool shared_residual_norm = true; // one rank: the outer residual norm and the inner solve's first norm are the same sum over the dof slice, formed once (the iterates keep their bits; the recorded norm groups its terms differently)
    bool early_gamma = true;      // device scalars: the flexible dot also forms the next iteration's gamma = <z, r+> (same bits; the projection kernel is left with <p, q>)
    bool gamma_on_device = false;
    int gamma_slot = 0; // where the current gamma sits in `scalars`
    bool unit_stitch_in_place = true; // stitching weights of the dof slice all exactly 1 (one rank): the inner solve writes z~ in place (0: the multiplication by the ones, the reference's sequence)
    bool mfma_stiffness = true; // N >= 11: stiffness on the fp64 matrix cores (not bit-identical; 1e-12 tolerance)
    DType tolerance = 1.0e-07;
    std::vector<DType> residual_history; // what the reference prints per iteration

    // Operator
    fdd::memory D_hat;
    std::vector<double> D_hat_hst;
    fdd::memory geom_fact[NUM_GEOM_FACTS];

    Domain() {}
    Domain(char *directory_, int poly_degree_) { initialize(directory_, poly_degree_); }
    ~Domain() {}

    int boundary_nodes_count() const { return num_bdary_nodes; }
    int interface_slots_count() const { return num_interface_slots; }
    // one interface exchange of `vectors` (1 or 2) prefixes, as gs_add_boundary[_pair] issues it: the collective alone,
    // on the exchange buffer's current contents (bench.py's communication timings)
    void comm_probe_interface(int vectors, bool neighbours)
    {
        if (fdd::comm().size == 1 or num_interface_slots == 0) return;
        if (neighbours)
            iface_exchange(vectors == 2 ? 2 : 1);
        else
            fdd::comm().allreduce_sum(interface_slots.as<double>(), (size_t)num_interface_slots * (vectors == 2 ? 2 : 1));
    }
    // bytes this rank sends in one neighbour exchange of `vectors` prefixes
    double comm_interface_neighbour_bytes(int vectors) const { return (double)vectors * iface_shared_entries * sizeof(double); }
    CSR_Matrix<DType> &gather_matrix() { return Qt; }
    CSR_Matrix<DType> &scatter_matrix() { return Q; }
    fdd::memory &dirichlet_mask_memory() { return dirichlet_mask; }
    fdd::memory &assembled_weight_memory() { return assembled_weight; }

    // ---- mesh files: the format Domain::initialize reads (domain.tpp:45-224) ----
    static bool read_mesh_files(const char *dir, int poly_degree, int proc_id, MeshData<DType> &m)
    {
        char file_name[4096];
        int n_x, n_y, n_z;
        snprintf(file_name, sizeof(file_name), "%s/lx1_%d/size_%d.%d.dat", dir, poly_degree + 1, proc_id, poly_degree);
        FILE *fp = fopen(file_name, "r");
        if (!fp) return false;
        int got = fscanf(fp, "%d %d %d %d %d", &m.dim, &n_x, &n_y, &n_z, &m.num_local_elements);
        fclose(fp);
        if (got != 5) return false;
        m.poly_degree = poly_degree;
        const size_t P = (size_t)m.num_local_points();

        auto read_bin = [&](const char *stem, void *dst, size_t elem_size) -> bool {
            snprintf(file_name, sizeof(file_name), "%s/lx1_%d/%s_%d.%d.dat", dir, poly_degree + 1, stem, proc_id, poly_degree);
            FILE *f = fopen(file_name, "rb");
            if (!f) return false;
            size_t rd = fread(dst, elem_size, P, f);
            fclose(f);
            return rd == P;
        };

        bool ok = true;
        m.x.resize(P);
        ok = ok && read_bin("x", m.x.data(), sizeof(DType));
        if (m.dim >= 2)
        {
            m.y.resize(P);
            ok = ok && read_bin("y", m.y.data(), sizeof(DType));
        }
        if (m.dim >= 3)
        {
            m.z.resize(P);
            ok = ok && read_bin("z", m.z.data(), sizeof(DType));
        }
        m.glo_num.resize(P);
        ok = ok && read_bin("glo_num", m.glo_num.data(), sizeof(long long));
        m.node_degree.resize(P);
        ok = ok && read_bin("node_degree", m.node_degree.data(), sizeof(int));
        m.p_mask.resize(P);
        ok = ok && read_bin("p_mask", m.p_mask.data(), sizeof(DType));
        for (int g = 0; g < NUM_GEOM_FACTS; g++)
        {
            char stem[16];
            snprintf(stem, sizeof(stem), "g_%d", g + 1);
            m.g[g].resize(P);
            ok = ok && read_bin(stem, m.g[g].data(), sizeof(DType));
        }
        return ok;
    }

    static bool write_mesh_files(const char *dir, int proc_id, const MeshData<DType> &m)
    {
        char file_name[4096];
        const int N = m.poly_degree;
        const size_t P = (size_t)m.num_local_points();
        snprintf(file_name, sizeof(file_name), "%s/lx1_%d/size_%d.%d.dat", dir, N + 1, proc_id, N);
        FILE *fp = fopen(file_name, "w");
        if (!fp) return false;
        fprintf(fp, "%d %d %d %d %d\n", m.dim, N + 1, N + 1, (m.dim == 3) ? N + 1 : 1, m.num_local_elements);
        fclose(fp);
        auto write_bin = [&](const char *stem, const void *src, size_t elem_size) -> bool {
            snprintf(file_name, sizeof(file_name), "%s/lx1_%d/%s_%d.%d.dat", dir, N + 1, stem, proc_id, N);
            FILE *f = fopen(file_name, "wb");
            if (!f) return false;
            size_t wr = fwrite(src, elem_size, P, f);
            fclose(f);
            return wr == P;
        };
        bool ok = write_bin("x", m.x.data(), sizeof(DType));
        if (m.dim >= 2) ok = ok && write_bin("y", m.y.data(), sizeof(DType));
        if (m.dim >= 3) ok = ok && write_bin("z", m.z.data(), sizeof(DType));
        ok = ok && write_bin("glo_num", m.glo_num.data(), sizeof(long long));
        ok = ok && write_bin("node_degree", m.node_degree.data(), sizeof(int));
        ok = ok && write_bin("p_mask", m.p_mask.data(), sizeof(DType));
        for (int g = 0; g < NUM_GEOM_FACTS; g++)
        {
            char stem[16];
            snprintf(stem, sizeof(stem), "g_%d", g + 1);
            ok = ok && write_bin(stem, m.g[g].data(), sizeof(DType));
        }
        return ok;
    }

    void initialize(char *directory_, int poly_degree_)
    {
        directory = directory_;
        MeshData<DType> m;
        if (!read_mesh_files(directory_, poly_degree_, fdd::globals().proc_id, m))
        {
            pstdout("ERROR: There was a problem reading Nek5000 data\n");
            fprintf(stderr, "ERROR: There was a problem reading Nek5000 data in '%s' (lx1_%d)\n", directory_, poly_degree_ + 1);
            exit(EXIT_FAILURE);
        }
        initialize(std::move(m));
    }

    void initialize(MeshData<DType> mesh_)
    {
        mesh = std::move(mesh_);
        poly_degree = mesh.poly_degree;
        fdd::SetupTimer timing("Domain::initialize", fdd::comm().rank == 0);
        fdd::globals().dim = mesh.dim; // `dim` is a global set by the last mesh read (config.hpp:48)
        const int dim = mesh.dim;

        num_local_elements = mesh.num_local_elements;
        num_elem_points = mesh.num_elem_points();
        num_local_points = num_local_elements * num_elem_points;

        {
            double tot = (double)num_local_elements; // domain.tpp:49-50
            fdd::comm().allreduce_sum_host(&tot, 1);
            num_total_elements = (int)std::llround(tot);
            mesh.num_total_elements = num_total_elements;
            num_total_points = (long long)num_total_elements * num_elem_points;
        }

        // Work arrays
        work_dev.resize(3);
        for (int w = 0; w < 3; w++) work_dev[w] = fdd::dev().malloc<DType>(num_local_points);

        // Elements (light views)
        elements.clear();
        elements.reserve(num_local_elements);
        for (int e = 0; e < num_local_elements; e++)
        {
            elements.push_back(Element<DType>(e, dim, poly_degree));
            elements.back().bind(mesh);
        }

        dirichlet_mask = fdd::dev().malloc<DType>(num_local_points);
        dirichlet_mask.copyFrom(mesh.p_mask.data(), (size_t)num_local_points * sizeof(DType));

        for (int g = 0; g < NUM_GEOM_FACTS; g++)
        {
            geom_fact[g] = fdd::dev().malloc<DType>(num_local_points);
            geom_fact[g].copyFrom(mesh.g[g].data(), (size_t)num_local_points * sizeof(DType));
            G_ptrs[g] = geom_fact[g].as<double>();
        }

        timing.lap("elements, mask and factors to the device");
        // ---- communication / local numbering (domain.tpp:233-302) ----
        rstdout("Setting up domain stitching handle...\n");

        // One hash pass turns the 64-bit global ids into dense temporary ids (first appearance); the multiplicity
        // count, the four numbering passes and Q then work on plain arrays (the seven hash lookups per point of
        // the straightforward form were 40 of the 50 s of setup at N = 15).
        std::vector<int> tid(num_local_points);
        int num_tmp = 0;
        num_tmp = fdd::first_appearance_ids(mesh.glo_num.data(), (size_t)num_local_points, tid);
        timing.lap("temporary node ids");
        std::vector<int> local_node_degree(num_tmp, 0);
        for (int p = 0; p < num_local_points; p++) local_node_degree[tid[p]]++;

        std::vector<int> local_node_idx(num_tmp, -1);
        std::vector<long long> boundary_nodes;
        int count = 0;

        // Nodes whose local multiplicity differs from the global one (shared with other ranks) come first
        // (domain.tpp:249-281), the rest after them by first appearance.  Within that, Dirichlet nodes are moved to
        // the two ends -- shared Dirichlet | shared | interior | interior Dirichlet -- so that the Subdomain's dofs
        // (the unmasked nodes, in this order) are one contiguous slice of every node vector and the outer solve hands
        // its residual to the inner solve, and takes the correction back, without renumbering.  Nothing outside the
        // class sees the order except through Q / Qt.
        for (int pass = 0; pass < 4; pass++)
        {
            const bool want_shared = pass < 2;
            const bool want_dirichlet = (pass == 0 or pass == 3);
            for (int p = 0; p < num_local_points; p++)
            {
                const int t = tid[p];
                const bool shared = local_node_degree[t] != mesh.node_degree[p];
                const bool dirichlet = not(mesh.p_mask[p] > 0.0);
                if (shared != want_shared or dirichlet != want_dirichlet) continue;
                if (local_node_idx[t] < 0)
                {
                    if (shared) boundary_nodes.push_back(mesh.glo_num[p]);
                    local_node_idx[t] = count;
                    count++;
                }
            }
            if (pass == 1) num_bdary_nodes = count;
        }

        num_local_nodes = num_tmp;
        timing.lap("node numbering");

        // gs_setup (domain.tpp:283-284): dense interface slots shared by all ranks
        {
            std::vector<int> counts;
            std::vector<long long> all = fdd::comm().allgatherv_host(boundary_nodes, counts);
            if (fdd::comm().size > 1) build_interface_neighbour_plan(all, counts, boundary_nodes);
            std::sort(all.begin(), all.end());
            all.erase(std::unique(all.begin(), all.end()), all.end());
            num_interface_slots = (int)all.size();

            if (num_bdary_nodes > 0)
            {
                std::vector<int> slot(num_bdary_nodes);
                for (int b = 0; b < num_bdary_nodes; b++) slot[b] = (int)(std::lower_bound(all.begin(), all.end(), boundary_nodes[b]) - all.begin());
                bdary_slot = fdd::dev().malloc<int>(num_bdary_nodes);
                bdary_slot.copyFrom(slot.data(), (size_t)num_bdary_nodes * sizeof(int));
            }
            if (num_interface_slots > 0) interface_slots = fdd::dev().malloc<double>(2 * (size_t)num_interface_slots); // room for two prefixes in one exchange

            // unique global nodes = sum of owned nodes: interior ones + shared ones counted once
            double owned = (double)(num_local_nodes - num_bdary_nodes);
            fdd::comm().allreduce_sum_host(&owned, 1);
            num_total_nodes = (long long)std::llround(owned) + num_interface_slots;
        }

        timing.lap("interface slots");
        {
            // one unit entry per point, rows in order: what add_entry + assemble (domain.tpp:286-296) produce, written directly
            std::vector<int> q_ptr((size_t)num_local_points + 1);
            fdd::low_order::pod_vector<int> q_col((size_t)num_local_points);
            fdd::low_order::pod_vector<DType> q_val((size_t)num_local_points);
            fdd::low_order::parallel_ranges(num_local_points, fdd::low_order::range_parts(num_local_points), [&](long long p0, long long p1, int) {
                for (long long p = p0; p < p1; p++)
                {
                    q_ptr[p] = (int)p;
                    q_col[p] = local_node_idx[tid[p]];
                    q_val[p] = (DType)1.0;
                }
            });
            q_ptr[num_local_points] = num_local_points;
            Q.adopt_csr(num_local_points, num_local_nodes, std::move(q_ptr), std::move(q_col), std::move(q_val));
        }
        timing.lap("Q");
        Q.transpose(Qt);
        timing.lap("Qt");

        reduce_ws = fdd::dev().malloc<double>(fdd_reduce_workspace_doubles());
        scalars = fdd::dev().malloc<double>(8);

        assembled_weight = fdd::dev().malloc<DType>(num_local_nodes);
        math.set_to_value(work_dev[0], 1.0, num_local_points);
        Qt.multiply(assembled_weight, work_dev[0]);
        gs_add_boundary(assembled_weight);
        math.invert_vector_elements(assembled_weight, num_local_nodes);

        // Operator (domain.tpp:304-316)
        const int n = poly_degree + 1;
        std::vector<double> r_gll(n), w_gll(n);
        D_hat_hst.assign((size_t)n * n, 0.0);
        fdd::gll::zwgll(r_gll.data(), w_gll.data(), n);
        fdd::gll::dgll(D_hat_hst.data(), r_gll.data(), n);
        D_hat = fdd::dev().malloc<DType>((size_t)n * n);
        D_hat.copyFrom(D_hat_hst.data(), (size_t)n * n * sizeof(DType));

        // Solver vectors (GMRES bases are allocated on first use)
        r_k = fdd::dev().malloc<DType>(num_local_points);
        r_kp1 = fdd::dev().malloc<DType>(num_local_points);
        q_k = fdd::dev().malloc<DType>(num_local_points);
        z_k = fdd::dev().malloc<DType>(num_local_points);
        p_k = fdd::dev().malloc<DType>(num_local_points);

        num_blocks = (num_local_points + BLOCK_SIZE - 1) / BLOCK_SIZE;
    }

    // Replace D_hat (tests feed the reference's own table so that parity does
    // not depend on the last bits of the GLL nodes)
    void set_D_hat(const double *D, int n)
    {
        D_hat_hst.assign(D, D + (size_t)n * n);
        D_hat.copyFrom(D_hat_hst.data(), (size_t)n * n * sizeof(DType));
    }

    // domain.tpp:527-580
    void initial_function(fdd::memory &u, int function_id = 0, unsigned long long seed = 0)
    {
        std::vector<DType> w(num_local_points);
        const int dim = mesh.dim;
        // function_id 3/4 use rand()/RAND_MAX in the reference, unseeded
        // (domain.tpp:572-573); srand(seed) makes runs repeatable
        if (function_id == 3 or function_id == 4) srand((unsigned)seed);
        for (int p = 0; p < num_local_points; p++)
        {
            const DType x = mesh.x[p], y = mesh.y[p], z = (dim == 3) ? mesh.z[p] : 0.5;
            const DType sz = (dim == 3) ? std::sin(M_PI * z) : 1.0;
            DType v = 0.0;
            if (function_id == 0)
                v = std::sin(M_PI * x) * std::sin(M_PI * y) * sz;
            else if (function_id == 1)
                v = std::sin(M_PI * x) * std::sin(M_PI * y) * sz + std::sin(2.0 * M_PI * x) * std::sin(M_PI * y) * sz;
            else if (function_id == 2)
                v = std::exp(x) * std::sin(M_PI * x) * std::sin(M_PI * y) * sz;
            else if (function_id == 3)
                v = std::sin(M_PI * x) * std::sin(M_PI * y) * sz + (1.0 / 5.0) * ((DType)(rand()) / (DType)(RAND_MAX));
            else if (function_id == 4)
                v = (DType)(rand()) / (DType)(RAND_MAX);
            w[p] = v;
        }
        u.copyFrom(w.data(), (size_t)num_local_points * sizeof(DType));
        direct_stiffness_summation(u, u, true, true);
    }

    // domain.tpp:582-600
    void direct_stiffness_summation(fdd::memory &QQtu, fdd::memory &u, bool apply_dirichlet_mask = true, bool apply_assembled_weight = false)
    {
        if (fused_dssum and Qt.unit_values)
        {
            // One gather-scatter pass over the assembled nodes instead of the
            // Qt and Q SpMVs (same arithmetic, bit-identical; fdd_hip.h).  The
            // boundary prefix is gathered first, exchanged, and scattered last.
            const double *w = apply_assembled_weight ? assembled_weight.as<double>() : nullptr;
            const double *m = apply_dirichlet_mask ? dirichlet_mask.as<double>() : nullptr;
            // every rank of a multi-rank run enters the exchange, also one without shared nodes (an empty prefix):
            // the collectives must match across ranks (gs_add_boundary and node_norm_enqueue gate the same way)
            const bool multi = fdd::comm().size > 1 and num_interface_slots > 0;
            const int nb = multi ? num_bdary_nodes : 0;
            if (multi)
            {
                if (nb > 0) Qt.gather_scatter(nullptr, work_dev[0].as<double>(), u.as<double>(), w, nullptr, 0, nb, 1);
                gs_add_boundary(work_dev[0]);
            }
            Qt.gather_scatter(QQtu.as<double>(), nullptr, u.as<double>(), w, m, nb, num_local_nodes, 0);
            if (nb > 0) Qt.gather_scatter(QQtu.as<double>(), work_dev[0].as<double>(), nullptr, nullptr, m, 0, nb, 2);
            return;
        }

        if (apply_assembled_weight)
            Qt.multiply_weight(work_dev[0], u, assembled_weight);
        else
            Qt.multiply(work_dev[0], u);

        gs_add_boundary(work_dev[0]);

        if (apply_dirichlet_mask)
            Q.multiply_weight(QQtu, work_dev[0], dirichlet_mask);
        else
            Q.multiply(QQtu, work_dev[0]);
    }

    // domain.tpp:602-609
    void stiffness_matrix(fdd::memory &Au, fdd::memory &u, bool apply_dssum = false)
    {
        if (mesh.dim == 3 and poly_degree >= 11 and poly_degree <= 15 and mfma_stiffness and Au.ptr() != u.ptr())
        {
            // high order: the six contractions on the fp64 matrix cores (tolerance-level parity, fdd_hip.h)
            fdd::ProfileScope prof("mfma_stiffness_kernel", 64.0 * num_local_points);
            FDD_CALL(fdd_stiffness_matrix_mfma(Au.as<double>(), u.as<double>(), D_hat.as<double>(), G_ptrs, nullptr, num_local_elements, poly_degree, fdd::dev().stream));
        }
        else if (mesh.dim == 3 and poly_degree <= 15)
        {
            fdd::ProfileScope prof("fused_stiffness_kernel", 64.0 * num_local_points);
            FDD_CALL(fdd_dom_stiffness_matrix(Au.as<double>(), u.as<double>(), D_hat.as<double>(), G_ptrs, num_local_elements, poly_degree, fdd::dev().stream));
        }
        else if (mesh.dim == 2 and poly_degree <= 15)
        {
            fdd::ProfileScope prof("fused_stiffness_2d_kernel", 40.0 * num_local_points);
            FDD_CALL(fdd_stiffness_matrix_2d(Au.as<double>(), u.as<double>(), D_hat.as<double>(), G_ptrs, nullptr, num_local_elements, poly_degree, fdd::dev().stream));
        }
        else
        {
            double *GDu[3] = {work_dev[0].as<double>(), work_dev[1].as<double>(), work_dev[2].as<double>()};
            FDD_CALL(fdd_dom_stiffness_matrix_1(GDu, u.as<double>(), D_hat.as<double>(), G_ptrs, num_local_points, poly_degree, mesh.dim, fdd::dev().stream));
            FDD_CALL(fdd_dom_stiffness_matrix_2(Au.as<double>(), GDu, D_hat.as<double>(), num_local_points, poly_degree, mesh.dim, fdd::dev().stream));
        }

        if (apply_dssum) direct_stiffness_summation(Au, Au, true, false);
    }

    // ------------------------------------------------------------------
    // Flexible CG on assembled data.  The reference keeps every Krylov vector
    // per element-local point, but its recurrences only ever combine
    //   - continuous vectors (u, p, z: the same value on every copy of a node),
    //   - residual-type vectors (r, q) that enter through point-wise dots with
    //     continuous vectors, sum_p z_p r_p = sum_n z_n (Qt r)_n, through the
    //     residual norm <r, QQt r>, and through the preconditioner, which
    //     starts with Qt (subdomain.tpp:3996) itself.
    // So the iteration is carried on node vectors: u~, p~, z~ (one value per
    // node) and r^ = Qt r, q^ = Qt q (this rank's partial sums).  Per step: the
    // element stiffness reads Q p~ through the point -> node index array, one
    // gather Qt, and everything else is BLAS-1 on num_local_nodes values (0.68x
    // the points at N = 7) with no dssum pass; the inner solve takes r^ and
    // returns u~ directly in the subdomain's dof numbering.  Interface nodes
    // (the boundary prefix) are exchanged exactly where the reference's gs_op
    // sits: inside the residual norm and in the stitching.
    // Same recurrences as domain.tpp:611-725; iterates agree to rounding.
    // ------------------------------------------------------------------
    template <typename PType>
    bool can_fcg_nodes(PType &subdomain)
    {
        if (not(assembled_outer and Qt.unit_values and mesh.dim == 3 and poly_degree <= 15)) return false;
        if (not use_preconditioner) return true;
        if (preconditioner_type != 1) return false;
        if (subdomain.composite()) return subdomain.composite_dof_space() and own_nodes_have_dofs(subdomain);
        return subdomain.assembled_inner and subdomain.can_assemble() and (int)subdomain.point_dof.size() == num_local_points;
    }

    // composite region: every unmasked node of the rank carries one of the subdomain's leading (own) dofs -- always,
    // unless the region has no ring at the rank's own degree (subdomain_overlap 0), where own points hang
    template <typename PType>
    bool own_nodes_have_dofs(PType &subdomain)
    {
        const int nd = subdomain.own_dofs();
        for (int p = 0; p < num_local_points; p++)
        {
            const int d = subdomain.point_dof[p];
            if (mesh.p_mask[p] > 0.0 and (d < 0 or d >= nd)) return false;
        }
        return true;
    }

    void setup_nodes()
    {
        if (nodes_ready) return;
        const int nn = num_local_nodes;
        point_node_dev = fdd::dev().malloc<int>(std::max(num_local_points, 1));
        point_node_dev.copyFrom(Q.col_hst.data(), (size_t)num_local_points * sizeof(int)); // one entry per row of Q

        std::vector<DType> mask(nn, 1.0), stitch(nn), w(nn);
        for (int p = 0; p < num_local_points; p++) mask[Q.col_hst[p]] = mesh.p_mask[p];
        assembled_weight.copyTo(w.data(), (size_t)nn * sizeof(DType));
        for (int n = 0; n < nn; n++) stitch[n] = (DType)(Qt.ptr_hst[n + 1] - Qt.ptr_hst[n]) * w[n] * mask[n];
        node_mask = fdd::dev().malloc<DType>(std::max(nn, 1));
        node_mask.copyFrom(mask.data(), (size_t)nn * sizeof(DType));
        node_stitch = fdd::dev().malloc<DType>(std::max(nn, 1));
        node_stitch.copyFrom(stitch.data(), (size_t)nn * sizeof(DType));
        node_stitch_hst = stitch;
        node_mask_hst = mask;

        for (fdd::memory *v : {&nu, &nr, &nr1, &nq, &nz, &np, &nt}) *v = fdd::dev().malloc<DType>(std::max(nn, 1));
        nodes_ready = true;
    }

    template <typename PType>
    void setup_dof_maps(PType &subdomain)
    {
        // the dofs that sit on this rank's nodes: all of them for an own-elements region, the leading ones of a composite
        const int nd = subdomain.own_dofs();
        if (nodes_sub_dofs == nd and node_of_dof.ptr()) return;
        composite_precond = subdomain.composite();
        std::vector<int> n_of_d(std::max(nd, 1), -1), d_of_n(std::max(num_local_nodes, 1), -1);
        for (int p = 0; p < num_local_points; p++)
        {
            const int d = subdomain.point_dof[p];
            if (d >= 0 and d < nd)
            {
                n_of_d[d] = Q.col_hst[p];
                d_of_n[Q.col_hst[p]] = d;
            }
        }
        dof_shift = (nd > 0) ? n_of_d[0] : -1;
        for (int d = 0; d < nd and dof_shift >= 0; d++)
            if (n_of_d[d] != d + dof_shift) dof_shift = -1;
        if (dof_shift > 0 and (dof_shift & 1)) dof_shift = -1; // an odd shift would leave the slice 8-byte aligned only: the 16-byte kernels want more
        node_of_dof = fdd::dev().malloc<int>(std::max(nd, 1));
        node_of_dof.copyFrom(n_of_d.data(), (size_t)nd * sizeof(int));
        dof_of_node = fdd::dev().malloc<int>(std::max(num_local_nodes, 1));
        dof_of_node.copyFrom(d_of_n.data(), (size_t)num_local_nodes * sizeof(int));
        sub_f = fdd::dev().malloc<DType>(std::max(nd, 1));
        sub_u = subdomain.new_dof_vector(); // a composite keeps copies / hanging values behind its dofs
        if (composite_precond and not rp.ptr()) rp = subdomain.tree_points(); // lives in the Subdomain's tree vector: level 0 is read in place
        nodes_sub_dofs = nd;
        norm_on_dof_slice = fdd::comm().size == 1 and dof_shift >= 0 and not composite_precond and subdomain.unit_norm_weight() and (int)node_mask_hst.size() == num_local_nodes;
        for (int n = 0; n < num_local_nodes and norm_on_dof_slice; n++)
            if (node_mask_hst[n] != ((n >= dof_shift and n < dof_shift + nd) ? (DType)1.0 : (DType)0.0)) norm_on_dof_slice = false;
        stitch_is_one = dof_shift >= 0 and not composite_precond and (int)node_stitch_hst.size() >= dof_shift + nd;
        for (int d = 0; d < nd and stitch_is_one; d++)
            if (node_stitch_hst[(size_t)dof_shift + d] != (DType)1.0) stitch_is_one = false;
    }

    // t = Qt v: this rank's sums over the copies of every node (no exchange)
    void gather_nodes(fdd::memory &t, fdd::memory &v)
    {
        Qt.gather_scatter(nullptr, t.as<double>(), v.as<double>(), nullptr, nullptr, 0, num_local_nodes, 1);
    }

    // ---- affine elements (an option of this build; the reference always streams the six factor arrays) ----
    // Where every element of the mesh is an affine image of the reference cube (a box mesh), the factors of a point are
    // c_f(e) (w_i w_j) w_k: the kernel forms them from six numbers per element and does not read 48 of its 64 bytes per
    // point.  set_affine_geometry(true) checks the mesh's OWN factor arrays against that form on the device
    // (fdd_stiffness_affine_detect) and switches the node-space operator over only if every element passes.
    static constexpr double affine_tolerance = 64.0 * 2.220446049250313e-16;
    bool affine_geometry = false; // in use
    bool affine_checked = false;
    double affine_deviation = -1.0; // largest relative deviation found (-1: not checked)
    fdd::memory affine_c, affine_w;
    bool set_affine_geometry(bool on)
    {
        affine_geometry = false;
        if (not on) return true;
        if (mesh.dim != 3 or poly_degree > 15 or num_local_elements == 0) return false;
        if (not affine_checked)
        {
            const int n = poly_degree + 1;
            std::vector<double> z(n), w(n), dev_hst(num_local_elements);
            fdd::gll::zwgll(z.data(), w.data(), n);
            affine_w = fdd::dev().malloc<double>(n);
            affine_w.copyFrom(w.data(), (size_t)n * sizeof(double));
            affine_c = fdd::dev().malloc<double>((size_t)num_local_elements * NUM_GEOM_FACTS);
            fdd::memory dev_dev = fdd::dev().malloc<double>(num_local_elements);
            FDD_CALL(fdd_stiffness_affine_detect(affine_c.as<double>(), dev_dev.as<double>(), G_ptrs, nullptr, affine_w.as<double>(), num_local_elements, poly_degree, fdd::dev().stream));
            dev_dev.copyTo(dev_hst.data(), dev_hst.size() * sizeof(double));
            dev_dev.free();
            affine_deviation = 0.0;
            for (double x : dev_hst) affine_deviation = (x == x) ? std::max(affine_deviation, x) : 1.0;
            affine_checked = true;
        }
        affine_geometry = affine_deviation <= affine_tolerance;
        return affine_geometry;
    }

    // q (points) = A_local (Q p~)
    void stiffness_from_nodes(fdd::memory &q, fdd::memory &pn)
    {
        if (affine_geometry)
        {
            if (poly_degree >= 11 and mfma_stiffness)
            {
                fdd::ProfileScope prof("mfma_stiffness_kernel<gather,affine>", 12.0 * num_local_points + 8.0 * num_local_nodes);
                FDD_CALL(fdd_stiffness_matrix_mfma_affine(q.as<double>(), pn.as<double>(), nullptr, point_node_dev.as<int>(), D_hat.as<double>(), affine_c.as<double>(), affine_w.as<double>(), nullptr, num_local_elements, poly_degree, fdd::dev().stream));
                return;
            }
            fdd::ProfileScope prof("fused_stiffness_kernel<gather,affine>", 12.0 * num_local_points + 8.0 * num_local_nodes);
            FDD_CALL(fdd_stiffness_matrix_affine(q.as<double>(), pn.as<double>(), nullptr, point_node_dev.as<int>(), D_hat.as<double>(), affine_c.as<double>(), affine_w.as<double>(), nullptr, num_local_elements, poly_degree, fdd::dev().stream));
            return;
        }
        if (poly_degree >= 11 and mfma_stiffness)
        {
            fdd::ProfileScope prof("mfma_stiffness_kernel<gather>", 60.0 * num_local_points + 8.0 * num_local_nodes);
            FDD_CALL(fdd_stiffness_matrix_mfma_gather(q.as<double>(), pn.as<double>(), nullptr, point_node_dev.as<int>(), D_hat.as<double>(), G_ptrs, nullptr, num_local_elements, poly_degree, fdd::dev().stream));
            return;
        }
        fdd::ProfileScope prof("fused_stiffness_kernel<gather>", 60.0 * num_local_points + 8.0 * num_local_nodes);
        FDD_CALL(fdd_sub_stiffness_matrix_gather(q.as<double>(), pn.as<double>(), point_node_dev.as<int>(), D_hat.as<double>(), G_ptrs, nullptr, num_local_elements, poly_degree, fdd::dev().stream));
    }

    // sqrt(<r, QQt r>) (domain.tpp:916-931) from r^ = Qt r: sum_n r^_n * gs(r^)_n * mask_n.
    // Two halves: the reductions (+ all-reduce) are enqueued into scalars[4..5]; the value is
    // fetched when the host wants it, which may be after more work has been enqueued.
    // defer_exchange: the boundary prefix is only saved; its exchange rides with the stitching exchange of the
    // preconditioner that follows (precondition_nodes), which then finishes the norm (node_norm_finish)
    // Returns the device address of |r^ restricted to the dof slice|^2 when -- and only when -- the sum just enqueued IS that
    // number (one rank, shared_residual_norm): the caller hands it to the precondition_nodes call that receives the same,
    // unmodified rn (the inner solve's first norm); every other branch returns nullptr.
    const double *node_norm_enqueue(fdd::memory &rn, bool defer_exchange = false)
    {
        const int nn = num_local_nodes;
        // every rank of a multi-rank run takes the two-part path (a rank without shared nodes contributes an empty
        // prefix), so that all of them issue the same collectives
        const bool multi = fdd::comm().size > 1 and num_interface_slots > 0;
        const int nb = multi ? num_bdary_nodes : 0;
        double *out = scalars.as<double>() + 4;
        if (multi)
        {
            if (not nprefix.ptr()) nprefix = fdd::dev().malloc<DType>(std::max(num_bdary_nodes, 1));
            nprefix.copyFrom(rn, (size_t)nb * sizeof(DType));
            FDD_CALL(fdd_dom_residual_norm(out + 1, reduce_ws.as<double>(), rn.as<double>() + nb, rn.as<double>() + nb, node_mask.as<double>() + nb, nn - nb, fdd::dev().stream));
            norm_parts = 2;
            norm_source = rn.as<double>();
            if (defer_exchange)
            {
                norm_deferred = true;
                return nullptr;
            }
            gs_add_boundary(nprefix);
            node_norm_finish();
            return nullptr;
        }
        const double *known = nullptr;
        if (norm_on_dof_slice and shared_residual_norm)
        {
            // One rank, mask = 1 on the dof slice and 0 elsewhere: the masked sum over the nodes IS the plain sum over the
            // slice -- the sum the inner solve forms first (its right-hand side is this slice, read in place).  Formed once,
            // by the inner solve's own call, and handed to it (precondition_nodes): the iterates keep their bits, the
            // recorded norm differs from the masked form in how its terms are grouped.
            const double *self[1] = {rn.as<double>() + dof_shift};
            FDD_CALL(fdd_multi_weighted_inner_product_scaled(out, reduce_ws.as<double>(), self[0], self, nullptr, 1, nullptr, nodes_sub_dofs, fdd::dev().stream));
            known = out;
        }
        else
        {
            // sum r*r*mask with r read once (the arithmetic and the reduction tree of residual_norm_kernel, domain.okl:109-138)
            const double *self[1] = {rn.as<double>()};
            FDD_CALL(fdd_multi_weighted_inner_product(out, reduce_ws.as<double>(), rn.as<double>(), self, 1, node_mask.as<double>(), nn, fdd::dev().stream));
        }
        norm_parts = 1;
        if (fdd::comm().size > 1) fdd::comm().allreduce_sum(out, norm_parts);
        return known;
    }

    // the boundary part of the norm once the saved prefix has been exchanged, then the all-reduce of both parts
    void node_norm_finish()
    {
        double *out = scalars.as<double>() + 4;
        FDD_CALL(fdd_dom_residual_norm(out, reduce_ws.as<double>(), norm_source, nprefix.as<double>(), node_mask.as<double>(), num_bdary_nodes, fdd::dev().stream));
        if (norm_deferred and device_scalars)
            norm_reduce_pending = true; // its two scalars join the all-reduce of the flexible dot (fcg_nodes_step_direction)
        else
            fdd::comm().allreduce_sum(out, 2);
        norm_deferred = false;
    }

    DType node_norm_fetch()
    {
        DType v[2] = {0.0, 0.0};
        fdd::memory tail = scalars.slice(4, 2);
        tail.copyTo(v, norm_parts * sizeof(DType));
        return std::sqrt(v[0] + v[1]);
    }

    const double *node_norm(DType &r_norm, fdd::memory &rn)
    {
        const double *known = node_norm_enqueue(rn);
        r_norm = node_norm_fetch();
        return known;
    }

    // z~ = M^-1 r^ and the stitching (domain.tpp:639-645, 697-706)
    // rhs_norm2: device address of |rn's dof slice|^2 if the caller has just formed it from this very rn (node_norm_enqueue's
    // return value), else nullptr
    template <typename PType>
    void precondition_nodes(fdd::memory &zn, fdd::memory &rn, PType &subdomain, const double *rhs_norm2 = nullptr)
    {
        void *stream = fdd::dev().stream;
        if (use_preconditioner)
        {
            fdd_timer().start("subdomain.solver");
            if (composite_precond)
            {
                // full domain decomposition: the inner solve starts from the residual on the points (degree tree,
                // ring pull, coarse all-gather: Subdomain::tree_operator) and returns the composite's dof vector,
                // whose leading entries are this rank's nodes
                // (the own points' part of the right-hand side is the node residual itself when the dofs are a slice of the nodes)
                subdomain.gmres_composite_dofs(sub_u, rp, true, false, (dof_shift >= 0 and composite_rhs_from_nodes) ? rn.as<double>() + dof_shift : nullptr);
            }
            else if (dof_shift >= 0)
            {
                // the dofs are the node slice [dof_shift, dof_shift + dofs): the inner solve reads r^ in place -- and, where
                // every stitching weight of the slice is exactly 1 (one rank: multiplicity * 1/multiplicity), writes z~ in
                // place as well: the multiplication below would be x * 1.0
                fdd::memory f_slice = rn.slice(dof_shift, nodes_sub_dofs);
                // the norm of this very slice was formed a moment ago (node_norm_enqueue): the inner solve starts from it
                subdomain.known_rhs_norm2_dev = rhs_norm2;
                if (stitch_is_one and unit_stitch_in_place)
                {
                    fdd::memory z_slice = zn.slice(dof_shift, nodes_sub_dofs);
                    subdomain.gmres_dofs(z_slice, f_slice);
                }
                else
                    subdomain.gmres_dofs(sub_u, f_slice);
                subdomain.known_rhs_norm2_dev = nullptr;
            }
            else
            {
                FDD_CALL(fdd_gather_indexed(sub_f.as<double>(), rn.as<double>(), node_of_dof.as<int>(), nullptr, nodes_sub_dofs, stream));
                subdomain.gmres_dofs(sub_u, sub_f);
            }
            fdd_timer().stop("subdomain.solver");

            fdd_timer().start("subdomain.stitching");
            if (dof_shift >= 0 and not composite_precond and stitch_is_one and unit_stitch_in_place)
            {
                // z~ is already in place
            }
            else if (dof_shift >= 0)
            {
                // the Dirichlet ends of z~ were cleared once (fcg_nodes_begin) and nothing writes them: the shared ones
                // see only zeros in the exchange (every rank masks them)
                FDD_CALL(fdd_amg_vector_multiplication(zn.as<double>() + dof_shift, sub_u.as<double>(), node_stitch.as<double>() + dof_shift, nodes_sub_dofs, stream));
            }
            else
                FDD_CALL(fdd_gather_indexed(zn.as<double>(), sub_u.as<double>(), dof_of_node.as<int>(), node_stitch.as<double>(), num_local_nodes, stream));
            if (norm_deferred)
            {
                gs_add_boundary_pair(zn, nprefix);
                node_norm_finish();
            }
            else
                gs_add_boundary(zn);
            fdd_timer().stop("subdomain.stitching");
        }
        else
        {
            nt.copyFrom(rn, (size_t)num_local_nodes * sizeof(DType));
            if (norm_deferred)
            {
                gs_add_boundary_pair(nt, nprefix);
                node_norm_finish();
            }
            else
                gs_add_boundary(nt);
            FDD_CALL(fdd_amg_vector_multiplication(zn.as<double>(), nt.as<double>(), node_mask.as<double>(), num_local_nodes, stream));
        }
    }

    template <typename PType>
    void fcg_nodes_begin(fdd::memory &u, fdd::memory &f, PType &subdomain)
    {
        setup_nodes();
        if (use_preconditioner) setup_dof_maps(subdomain);
        fcg_u_pts = u;
        fcg_nodes_active = true;

        gather_nodes(nr, f);
        if (use_preconditioner and composite_precond) rp.copyFrom(f, (size_t)num_local_points * sizeof(DType)); // r = f on the points as well
        FDD_CALL(fdd_set_to_value(nu.as<double>(), 0.0, num_local_nodes, 0, fdd::dev().stream));
        FDD_CALL(fdd_set_to_value(nz.as<double>(), 0.0, num_local_nodes, 0, fdd::dev().stream));

        const double *r0_norm2 = node_norm(fcg_r_0_norm, nr);
        residual_history.push_back(fcg_r_0_norm);
        rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", 0, fcg_r_0_norm, 1.0);

        precondition_nodes(nz, nr, subdomain, r0_norm2);
        np.copyFrom(nz, (size_t)num_local_nodes * sizeof(DType));
        gamma_on_device = false; // the first iteration's projection kernel forms gamma itself
    }

    DType fcg_nodes_step_residual()
    {
        const int nn = num_local_nodes;
        void *stream = fdd::dev().stream;
        DType values[2], r_norm;

        fdd_timer().start("domain.operator_application");
        stiffness_from_nodes(q_k, np);
        gather_nodes(nq, q_k);
        fdd_timer().stop("domain.operator_application");

        if (device_scalars and gamma_on_device and early_gamma)
        {
            // gamma = <z, r> already sits in scalars[0]: the flexible dot of the previous iteration formed it from the
            // vectors it was reading anyway (fcg_nodes_step_direction); <p, q> is what is left of domain.okl:140-184
            FDD_CALL(fdd_sub_inner_product(scalars.as<double>() + 1, reduce_ws.as<double>(), np.as<double>(), nq.as<double>(), nn, stream));
            if (fdd::comm().size > 1) fdd::comm().allreduce_sum(scalars.as<double>() + 1, 1);
        }
        else
        {
            FDD_CALL(fdd_dom_projection_inner_products(scalars.as<double>(), reduce_ws.as<double>(), nz.as<double>(), nr.as<double>(), np.as<double>(), nq.as<double>(), nn, stream));
            if (device_scalars and fdd::comm().size > 1) fdd::comm().allreduce_sum(scalars.as<double>(), 2);
        }
        if (device_scalars)
        {
            // gamma = scalars[gamma_slot] (kept for beta), theta = scalars[1]: alpha never visits the host
            if (not(gamma_on_device and early_gamma)) gamma_slot = 0; // the projection kernel has just written it
            FDD_CALL(fdd_dom_solution_and_residual_update_dev(nu.as<double>(), nr1.as<double>(), nr.as<double>(), np.as<double>(), nq.as<double>(), scalars.as<double>() + gamma_slot, scalars.as<double>() + 1, nn, stream));
            if (use_preconditioner and composite_precond) // r+ = r - alpha q on the points too (domain.okl:191), alpha from device memory
                FDD_CALL(fdd_xmay_ratio_dev(rp.as<double>(), rp.as<double>(), scalars.as<double>() + gamma_slot, scalars.as<double>() + 1, q_k.as<double>(), num_local_points, stream));
            pending_rhs_norm2 = node_norm_enqueue(nr1, /*defer_exchange=*/true); // finished inside the preconditioner's exchange
            fcg_norm_pending = true;
            return std::numeric_limits<DType>::quiet_NaN(); // fcg_nodes_norm() has the value
        }
        fetch_scalars(values, 2);
        fcg_gamma_k = values[0];
        const DType alpha_k = fcg_gamma_k / values[1];

        FDD_CALL(fdd_dom_solution_and_residual_update(nu.as<double>(), nr1.as<double>(), nr.as<double>(), np.as<double>(), nq.as<double>(), alpha_k, nn, stream));
        if (use_preconditioner and composite_precond) FDD_CALL(fdd_vector_vector_addition(rp.as<double>(), 1.0, rp.as<double>(), -alpha_k, q_k.as<double>(), num_local_points, stream));

        pending_rhs_norm2 = node_norm(r_norm, nr1);
        residual_history.push_back(r_norm);
        rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", fcg_iter + 1, r_norm, r_norm / fcg_r_0_norm);
        return r_norm;
    }

    // the residual norm enqueued by the last fcg_nodes_step_residual (device_scalars mode)
    DType fcg_nodes_norm()
    {
        const DType r_norm = node_norm_fetch();
        fcg_norm_pending = false;
        residual_history.push_back(r_norm);
        rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", fcg_iter + 1, r_norm, r_norm / fcg_r_0_norm);
        return r_norm;
    }

    template <typename PType>
    void fcg_nodes_step_direction(PType &subdomain)
    {
        const int nn = num_local_nodes;
        DType theta_k;
        precondition_nodes(nz, nr1, subdomain, pending_rhs_norm2); // nr1 is untouched since its norm was enqueued (fcg_nodes_step_residual)
        pending_rhs_norm2 = nullptr;
        if (device_scalars)
        {
            // beta = scalars[3] / scalars[0] (theta / gamma), read by the update kernel; the residual norm's two
            // parts sit right behind it (scalars[4..5]) and are summed over the ranks in the same collective
            int theta_at = 3;
            if (early_gamma)
            {
                // {<z, r+> (the next iteration's gamma), theta} from one pass over r, r+, z.  The pair alternates between
                // scalars[2..3] and scalars[6..7] (the current gamma must outlive it: beta reads it below), either side of
                // the norm's two parts in scalars[4..5], so that one all-reduce still carries all four
                const int next = (gamma_slot == 2) ? 6 : 2;
                theta_at = next + 1;
                FDD_CALL(fdd_dom_inner_product_flexible_gamma(scalars.as<double>() + next, reduce_ws.as<double>(), nr.as<double>(), nr1.as<double>(), nz.as<double>(), nn, fdd::dev().stream));
                if (fdd::comm().size > 1)
                {
                    if (norm_reduce_pending)
                        fdd::comm().allreduce_sum(scalars.as<double>() + (next == 2 ? 2 : 4), 4);
                    else
                        fdd::comm().allreduce_sum(scalars.as<double>() + next, 2);
                }
                norm_reduce_pending = false;
                // p = z + beta p; "r = r+" (domain.okl:226-233) is a swap of the two node vectors, not a copy
                FDD_CALL(fdd_xpby_ratio_dev(np.as<double>(), nz.as<double>(), scalars.as<double>() + theta_at, scalars.as<double>() + gamma_slot, np.as<double>(), nn, fdd::dev().stream));
                gamma_slot = next;
            }
            else
            {
                FDD_CALL(fdd_dom_inner_product_flexible(scalars.as<double>() + 3, reduce_ws.as<double>(), nr.as<double>(), nr1.as<double>(), nz.as<double>(), nn, fdd::dev().stream));
                if (fdd::comm().size > 1) fdd::comm().allreduce_sum(scalars.as<double>() + 3, norm_reduce_pending ? 3 : 1);
                norm_reduce_pending = false;
                FDD_CALL(fdd_xpby_ratio_dev(np.as<double>(), nz.as<double>(), scalars.as<double>() + 3, scalars.as<double>() + gamma_slot, np.as<double>(), nn, fdd::dev().stream));
            }
            gamma_on_device = early_gamma;
            std::swap(nr, nr1);
        }
        else
        {
            gamma_on_device = false;
            FDD_CALL(fdd_dom_inner_product_flexible(scalars.as<double>(), reduce_ws.as<double>(), nr.as<double>(), nr1.as<double>(), nz.as<double>(), nn, fdd::dev().stream));
            fetch_scalars(&theta_k, 1);
            const DType beta_k = theta_k / fcg_gamma_k;
            FDD_CALL(fdd_dom_residual_and_search_update(np.as<double>(), nr.as<double>(), nz.as<double>(), nr1.as<double>(), beta_k, nn, fdd::dev().stream));
        }
        num_iterations++;
        fcg_iter++;
    }

    // u = Q u~: hand the solution back on the element-local points
    void fcg_finish()
    {
        if (not fcg_nodes_active) return;
        Q.multiply(fcg_u_pts, nu);
    }

    // domain.tpp:611-725.  The loop body is exposed as fcg_begin / fcg_step so
    // that a caller (bench.py) can time an exact number of iterations; the
    // method itself is begin + steps + the reference's stopping tests.
    template <typename PType>
    void fcg_begin(fdd::memory &u, fdd::memory &f, PType &subdomain)
    {
        residual_history.clear();
        num_iterations = 0;
        fcg_iter = 0;
        fcg_nodes_active = false;
        if (can_fcg_nodes(subdomain))
        {
            fcg_nodes_begin(u, f, subdomain);
            return;
        }
        fcg_u = u;

        fdd_timer().start("domain.vector_operations");
        FDD_CALL(fdd_dom_initialize_arrays(fcg_u.as<double>(), r_k.as<double>(), f.as<double>(), num_local_points, fdd::dev().stream));
        fdd_timer().stop("domain.vector_operations");

        fdd_timer().start("domain.residual_norm");
        residual_norm(fcg_r_0_norm, r_k);
        fdd_timer().stop("domain.residual_norm");

        residual_history.push_back(fcg_r_0_norm);
        rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", 0, fcg_r_0_norm, 1.0);

        apply_preconditioner(z_k, r_k, subdomain);

        fdd_timer().start("domain.vector_operations");
        p_k.copyFrom(z_k, (size_t)num_local_points * sizeof(DType));
        fdd_timer().stop("domain.vector_operations");

        num_iterations = 0;
        fcg_iter = 0;
    }

    // first half of an iteration: q = A p, alpha, u += alpha p, r+ = r - alpha q, ||r+||
    DType fcg_step_residual()
    {
        if (fcg_nodes_active) return fcg_nodes_step_residual();
        DType theta_k, r_norm;

        fdd_timer().start("domain.operator_application");
        stiffness_matrix(q_k, p_k);
        fdd_timer().stop("domain.operator_application");

        fdd_timer().start("domain.inner_products");
        projection_inner_products(fcg_gamma_k, theta_k, z_k, r_k, p_k, q_k);
        fdd_timer().stop("domain.inner_products");

        const DType alpha_k = fcg_gamma_k / theta_k;

        fdd_timer().start("domain.vector_operations");
        solution_and_residual_update(fcg_u, r_kp1, r_k, p_k, q_k, alpha_k);
        fdd_timer().stop("domain.vector_operations");

        fdd_timer().start("domain.residual_norm");
        residual_norm(r_norm, r_kp1);
        fdd_timer().stop("domain.residual_norm");

        residual_history.push_back(r_norm);
        rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", fcg_iter + 1, r_norm, r_norm / fcg_r_0_norm);
        return r_norm;
    }

    // second half: z = M^-1 r+ (+ stitching), beta, p = z + beta p, r = r+
    template <typename PType>
    void fcg_step_direction(PType &subdomain)
    {
        if (fcg_nodes_active)
        {
            fcg_nodes_step_direction(subdomain);
            return;
        }
        DType theta_k;

        apply_preconditioner(z_k, r_kp1, subdomain);

        fdd_timer().start("domain.inner_products");
        inner_product_flexible(theta_k, r_k, r_kp1, z_k);
        fdd_timer().stop("domain.inner_products");

        const DType beta_k = theta_k / fcg_gamma_k;

        fdd_timer().start("domain.vector_operations");
        residual_and_search_update(p_k, r_k, z_k, r_kp1, beta_k);
        fdd_timer().stop("domain.vector_operations");

        num_iterations++;
        fcg_iter++;
    }

    // one full PCG iteration without stopping tests (what bench.py calls a step)
    template <typename PType>
    DType fcg_step(PType &subdomain)
    {
        DType r_norm = fcg_step_residual();
        if (fcg_norm_pending)
        {
            // the norm's reductions are in the queue; the preconditioner is enqueued behind them
            // before the host waits for the value (the inner solve synchronises at its end anyway)
            fcg_step_direction(subdomain);
            fcg_iter--; // the history line belongs to the iteration just counted
            r_norm = fcg_nodes_norm();
            fcg_iter++;
            return r_norm;
        }
        fcg_step_direction(subdomain);
        return r_norm;
    }

    // K iterations without stopping tests (what bench.py times).  In the node-space / device-scalar mode nothing
    // in an iteration needs the host: the norms go to a device-side history and are read once at the end, the
    // inner solves run without their end-of-cycle read (Subdomain::lazy_history).  Same arithmetic, same order.
    template <typename PType>
    DType fcg_steps(PType &subdomain, int K)
    {
        DType r_norm = std::numeric_limits<DType>::quiet_NaN();
        if (K <= 0) return r_norm;
        if (not(fcg_nodes_active and device_scalars and lazy_steps))
        {
            for (int s = 0; s < K; s++) r_norm = fcg_step(subdomain);
            return r_norm;
        }
        if (norm_hist_cap < K)
        {
            norm_hist.free();
            norm_hist = fdd::dev().malloc<DType>(K);
            norm_hist_cap = K;
        }
        const bool saved = subdomain.lazy_history;
        subdomain.lazy_history = true;
        for (int s = 0; s < K; s++)
        {
            fcg_nodes_step_residual();
            fcg_nodes_step_direction(subdomain);
            FDD_CALL(fdd_sqrt_sum_dev(norm_hist.as<double>() + s, scalars.as<double>() + 4, norm_parts, fdd::dev().stream));
            fcg_norm_pending = false;
        }
        subdomain.lazy_history = saved;
        std::vector<DType> h(K);
        norm_hist.copyTo(h.data(), (size_t)K * sizeof(DType)); // the one synchronisation of the K iterations
        for (int s = 0; s < K; s++)
        {
            residual_history.push_back(h[s]);
            rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", fcg_iter - K + s + 1, h[s], h[s] / fcg_r_0_norm);
        }
        return h[K - 1];
    }

    template <typename PType>
    void flexible_conjugate_gradient(fdd::memory &u, fdd::memory &f, PType &subdomain, bool use_relative = true)
    {
        fcg_begin(u, f, subdomain);

        for (int iter = 0; iter < max_iterations; iter++)
        {
            DType r_norm = fcg_step_residual();
            bool direction_done = false;
            if (fcg_norm_pending)
            {
                // device_scalars: the second half of the iteration is enqueued before the host reads
                // the norm.  If the tests below stop the solve, that half was speculative: it touched
                // z, p, r only, never the solution, and its iteration count is taken back.
                fcg_step_direction(subdomain);
                direction_done = true;
                fcg_iter--;
                r_norm = fcg_nodes_norm();
                fcg_iter++;
            }

            bool stop = use_relative ? (r_norm / fcg_r_0_norm < tolerance) : (r_norm < tolerance);
            if (std::isnan(r_norm)) stop = true;
            if (stop)
            {
                if (direction_done)
                {
                    num_iterations--;
                    fcg_iter--;
                }
                break;
            }

            if (not direction_done) fcg_step_direction(subdomain);
        }
        fcg_finish();
    }

    // ------------------------------------------------------------------
    // Node-space flexible GMRES(m): the recurrences of domain.tpp:727-914 on one value per assembled node.  The outer
    // iteration sees its dual vectors (r, q, V_i) only through Qt -- <a, b> = sum_points a (QQt b) mask = sum_nodes
    // (Qt a) gs(Qt b) mask -- and its continuous ones (Z_j, u) only through Q, so basis, preconditioned basis, residual
    // and solution are node vectors (0.68 x the points at N = 7), an operator application is the gather-on-load stiffness
    // + one Qt gather, and no direct-stiffness summation is left: with more than one rank the interface prefix of a new
    // basis vector is exchanged once (VNA) and the norm exchanges its own.  Same Hessenberg entries, rotations, stopping
    // tests and iteration count as the point-space form to rounding (test_reference_shaped_and_restructured_paths_agree).
    // On a composite the basis is kept on the points as well: the preconditioner's degree tree restricts point data.
    // ------------------------------------------------------------------
    template <typename PType>
    void gmres_nodes(fdd::memory &u, fdd::memory &f, PType &subdomain, bool use_relative)
    {
        setup_nodes();
        if (use_preconditioner) setup_dof_maps(subdomain);
        const int nn = num_local_nodes, m = num_vectors;
        void *stream = fdd::dev().stream;
        const bool multi = fdd::comm().size > 1 and num_interface_slots > 0; // rank-uniform
        const bool points_too = use_preconditioner and composite_precond;
        if (gmres_nodes_vectors != m or (multi and VNA.empty()) or (points_too and VP.empty()))
        {
            for (auto *set : {&VN, &ZN, &VNA, &VP})
            {
                for (auto &v : *set) v.free();
                set->clear();
            }
            nf.free();
            nf = fdd::dev().malloc<DType>(std::max(nn, 1));
            VN.resize(m + 1);
            for (auto &v : VN) v = fdd::dev().malloc<DType>(std::max(nn, 1));
            ZN.resize(m);
            for (auto &v : ZN)
            {
                v = fdd::dev().malloc<DType>(std::max(nn, 1));
                FDD_CALL(fdd_set_to_value(v.as<double>(), 0.0, nn, 0, stream)); // the Dirichlet ends stay zero: nothing writes them
            }
            if (multi)
            {
                VNA.resize(m + 1);
                for (auto &v : VNA) v = fdd::dev().malloc<DType>(std::max(nn, 1));
            }
            if (points_too)
            {
                VP.resize(m + 1);
                for (auto &v : VP) v = fdd::dev().malloc<DType>(std::max(num_local_points, 1));
            }
            H.assign(m, std::vector<DType>(m, 0.0));
            c_gmres.assign(m, 0.0);
            s_gmres.assign(m, 0.0);
            gamma.assign(m + 1, 0.0);
            gmres_nodes_vectors = m;
        }
        residual_history.clear();

        // the assembled copy <., .> reads: gs over the ranks on the interface prefix, nothing to do on one rank
        auto assembled = [&](int i) -> fdd::memory & {
            if (not multi) return VN[i];
            VNA[i].copyFrom(VN[i], (size_t)nn * sizeof(DType));
            gs_add_boundary(VNA[i]);
            return VNA[i];
        };

        fdd_timer().start("domain.vector_operations");
        gather_nodes(nf, f); // f^ = Qt f
        FDD_CALL(fdd_set_to_value(nu.as<double>(), 0.0, nn, 0, stream));
        nr.copyFrom(nf, (size_t)nn * sizeof(DType));
        if (points_too) VP[0].copyFrom(f, (size_t)num_local_points * sizeof(DType));
        fdd_timer().stop("domain.vector_operations");

        DType r_norm, r_0_norm;
        fdd_timer().start("domain.residual_norm");
        node_norm(r_0_norm, nr);
        fdd_timer().stop("domain.residual_norm");
        residual_history.push_back(r_0_norm);
        rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", 0, r_0_norm, 1.0);

        bool converged = false;
        int iter = 0, j;
        DType alpha_j, beta_j, gamma_j, gamma_k;

        while (iter < max_iterations)
        {
            if (iter > 0)
            {
                fdd_timer().start("domain.operator_application");
                stiffness_from_nodes(q_k, nu);
                gather_nodes(nq, q_k);
                fdd_timer().stop("domain.operator_application");
                fdd_timer().start("domain.vector_operations");
                FDD_CALL(fdd_vector_vector_addition(nr.as<double>(), 1.0, nf.as<double>(), -1.0, nq.as<double>(), nn, stream));
                if (points_too) FDD_CALL(fdd_vector_vector_addition(VP[0].as<double>(), 1.0, f.as<double>(), -1.0, q_k.as<double>(), num_local_points, stream));
                fdd_timer().stop("domain.vector_operations");
                fdd_timer().start("domain.residual_norm");
                node_norm(r_norm, nr);
                fdd_timer().stop("domain.residual_norm");
                gamma[0] = r_norm;
            }
            else
                gamma[0] = r_0_norm;

            fdd_timer().start("domain.vector_operations");
            FDD_CALL(fdd_vector_scaling(VN[0].as<double>(), 1.0 / gamma[0], nr.as<double>(), nn, stream));
            if (points_too) FDD_CALL(fdd_vector_scaling(VP[0].as<double>(), 1.0 / gamma[0], VP[0].as<double>(), num_local_points, stream));
            fdd_timer().stop("domain.vector_operations");
            std::vector<const double *> va(m + 1, nullptr); // assembled basis of this cycle
            va[0] = assembled(0).template as<double>();

            for (j = 0; j < m; j++)
            {
                if (points_too) rp.copyFrom(VP[j], (size_t)num_local_points * sizeof(DType)); // the tree restricts V_j on the points
                precondition_nodes(ZN[j], VN[j], subdomain);

                fdd_timer().start("domain.operator_application");
                stiffness_from_nodes(q_k, ZN[j]);
                gather_nodes(nq, q_k);
                fdd_timer().stop("domain.operator_application");

                // classical Gram-Schmidt: every H[i][j] from the same q (domain.tpp:810-815), then the updates
                fdd_timer().start("domain.inner_products");
                std::vector<double> h(j + 1);
                for (int g0 = 0; g0 < j + 1; g0 += FDD_MULTI_MAX)
                {
                    const int cnt = std::min(FDD_MULTI_MAX, j + 1 - g0);
                    FDD_CALL(fdd_multi_weighted_inner_product(scalars.as<double>(), reduce_ws.as<double>(), nq.as<double>(), va.data() + g0, cnt, node_mask.as<double>(), nn, stream));
                    fetch_scalars(h.data() + g0, cnt);
                }
                for (int i = 0; i < j + 1; i++) H[i][j] = h[i];
                fdd_timer().stop("domain.inner_products");

                fdd_timer().start("domain.vector_operations");
                for (int g0 = 0; g0 < j + 1; g0 += FDD_MULTI_MAX)
                {
                    const int cnt = std::min(FDD_MULTI_MAX, j + 1 - g0);
                    const double *ptrs[FDD_MULTI_MAX], *pptrs[FDD_MULTI_MAX];
                    double coeffs[FDD_MULTI_MAX];
                    for (int i = 0; i < cnt; i++)
                    {
                        ptrs[i] = VN[g0 + i].template as<double>();
                        if (points_too) pptrs[i] = VP[g0 + i].template as<double>();
                        coeffs[i] = -H[g0 + i][j];
                    }
                    FDD_CALL(fdd_multi_axpy(nq.as<double>(), coeffs, ptrs, cnt, nn, stream));
                    if (points_too) FDD_CALL(fdd_multi_axpy(q_k.as<double>(), coeffs, pptrs, cnt, num_local_points, stream));
                }
                fdd_timer().stop("domain.vector_operations");

                for (int i = 0; i < j; i++)
                {
                    DType h_ij = H[i][j];
                    H[i][j] = c_gmres[i] * h_ij + s_gmres[i] * H[i + 1][j];
                    H[i + 1][j] = -s_gmres[i] * h_ij + c_gmres[i] * H[i + 1][j];
                }

                fdd_timer().start("domain.residual_norm");
                node_norm(alpha_j, nq);
                fdd_timer().stop("domain.residual_norm");

                if (std::abs(alpha_j) == 0.0)
                {
                    converged = true;
                    break;
                }

                beta_j = std::sqrt(H[j][j] * H[j][j] + alpha_j * alpha_j);
                gamma_j = 1.0 / beta_j;
                c_gmres[j] = H[j][j] * gamma_j;
                s_gmres[j] = alpha_j * gamma_j;
                H[j][j] = beta_j;
                gamma[j + 1] = -s_gmres[j] * gamma[j];
                gamma[j] = c_gmres[j] * gamma[j];

                r_norm = std::abs(gamma[j + 1]);
                residual_history.push_back(r_norm);
                rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", iter + 1, r_norm, r_norm / r_0_norm);

                if (use_relative ? (r_norm / r_0_norm < tolerance) : (r_norm < tolerance))
                {
                    converged = true;
                    break;
                }
                if (iter >= max_iterations or std::isnan(r_norm))
                {
                    converged = true;
                    break;
                }

                fdd_timer().start("domain.vector_operations");
                FDD_CALL(fdd_vector_scaling(VN[j + 1].as<double>(), 1.0 / alpha_j, nq.as<double>(), nn, stream));
                if (points_too) FDD_CALL(fdd_vector_scaling(VP[j + 1].as<double>(), 1.0 / alpha_j, q_k.as<double>(), num_local_points, stream));
                fdd_timer().stop("domain.vector_operations");
                if (j + 1 < m) va[j + 1] = assembled(j + 1).template as<double>(); // the last vector of a cycle is never projected on

                iter++;
            }

            if (j == m) j--;

            // back substitution stored into c_gmres (domain.tpp:891-899)
            for (int k = j; k >= 0; k--)
            {
                gamma_k = gamma[k];
                for (int i = j; i > k; i--) gamma_k -= H[k][i] * c_gmres[i];
                c_gmres[k] = gamma_k / H[k][k];
            }

            fdd_timer().start("domain.vector_operations");
            for (int g0 = 0; g0 < j + 1; g0 += FDD_MULTI_MAX)
            {
                const int cnt = std::min(FDD_MULTI_MAX, j + 1 - g0);
                const double *ptrs[FDD_MULTI_MAX];
                for (int i = 0; i < cnt; i++) ptrs[i] = ZN[g0 + i].template as<double>();
                FDD_CALL(fdd_multi_axpy(nu.as<double>(), c_gmres.data() + g0, ptrs, cnt, nn, stream));
            }
            fdd_timer().stop("domain.vector_operations");

            if (converged) break;
        }

        Q.multiply(u, nu); // the solution back on the element-local points
        num_iterations = iter;
    }

    // domain.tpp:727-914
    template <typename PType>
    void generalized_minimum_residual(fdd::memory &u, fdd::memory &f, PType &subdomain, bool use_relative = true)
    {
        if (restructured_outer and can_fcg_nodes(subdomain))
        {
            gmres_nodes(u, f, subdomain, use_relative);
            return;
        }
        allocate_gmres();
        residual_history.clear();

        fdd_timer().start("domain.vector_operations");
        fdd::memory &u_k = u;
        FDD_CALL(fdd_dom_initialize_arrays(u_k.as<double>(), r_k.as<double>(), f.as<double>(), num_local_points, fdd::dev().stream));
        fdd_timer().stop("domain.vector_operations");

        DType r_norm;
        DType r_0_norm;

        fdd_timer().start("domain.residual_norm");
        residual_norm(r_0_norm, r_k);
        fdd_timer().stop("domain.residual_norm");

        residual_history.push_back(r_0_norm);
        rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", 0, r_0_norm, 1.0);

        bool converged = false;
        int iter = 0;
        int j;

        DType alpha_j, beta_j, gamma_j, gamma_k;

        while (iter < max_iterations)
        {
            if (iter > 0)
            {
                fdd_timer().start("domain.operator_application");
                stiffness_matrix(r_k, u_k);
                fdd_timer().stop("domain.operator_application");

                fdd_timer().start("domain.vector_operations");
                math.vector_vector_addition(r_k, 1.0, f, -1.0, r_k, num_local_points);
                fdd_timer().stop("domain.vector_operations");

                fdd_timer().start("domain.residual_norm");
                residual_norm(r_norm, r_k);
                fdd_timer().stop("domain.residual_norm");

                gamma[0] = r_norm;
            }
            else
            {
                gamma[0] = r_0_norm;
            }

            fdd_timer().start("domain.vector_operations");
            math.vector_scaling(V[0], 1.0 / gamma[0], r_k, num_local_points);
            fdd_timer().stop("domain.vector_operations");
            va_valid = 0; // a new basis

            for (j = 0; j < num_vectors; j++)
            {
                apply_preconditioner(Z[j], V[j], subdomain);

                fdd_timer().start("domain.operator_application");
                stiffness_matrix(q_k, Z[j]);
                fdd_timer().stop("domain.operator_application");

                // classical Gram-Schmidt: every H[i][j] from the same q_k (domain.tpp:810-815)
                if (restructured_outer)
                {
                    // <q, V_i> = sum q * (QQt V_i) * mask with the assembled copy VA[i] = mask * QQt V_i cached when
                    // V_i was made (the reference redoes the dssum inside each of the (j+1)(j+2)/2 products);
                    // the (j+1) dots read q once per group of FDD_MULTI_MAX, the (j+1) updates are one pass per group
                    fdd_timer().start("domain.inner_products");
                    gmres_cache_assembled(j);
                    std::vector<double> h(j + 1);
                    for (int g0 = 0; g0 < j + 1; g0 += FDD_MULTI_MAX)
                    {
                        const int cnt = std::min(FDD_MULTI_MAX, j + 1 - g0);
                        const double *ptrs[FDD_MULTI_MAX];
                        for (int i = 0; i < cnt; i++) ptrs[i] = VA[g0 + i].template as<double>();
                        FDD_CALL(fdd_multi_weighted_inner_product(scalars.as<double>(), reduce_ws.as<double>(), q_k.as<double>(), ptrs, cnt, dirichlet_mask.as<double>(), num_local_points, fdd::dev().stream));
                        fetch_scalars(h.data() + g0, cnt);
                    }
                    for (int i = 0; i < j + 1; i++) H[i][j] = h[i];
                    fdd_timer().stop("domain.inner_products");

                    fdd_timer().start("domain.vector_operations");
                    for (int g0 = 0; g0 < j + 1; g0 += FDD_MULTI_MAX)
                    {
                        const int cnt = std::min(FDD_MULTI_MAX, j + 1 - g0);
                        const double *ptrs[FDD_MULTI_MAX];
                        double coeffs[FDD_MULTI_MAX];
                        for (int i = 0; i < cnt; i++)
                        {
                            ptrs[i] = V[g0 + i].template as<double>();
                            coeffs[i] = -H[g0 + i][j];
                        }
                        FDD_CALL(fdd_multi_axpy(q_k.as<double>(), coeffs, ptrs, cnt, num_local_points, fdd::dev().stream));
                    }
                    fdd_timer().stop("domain.vector_operations");
                }
                else
                {
                    for (int i = 0; i < j + 1; i++)
                    {
                        fdd_timer().start("domain.inner_products");
                        assembled_inner_product(H[i][j], q_k, V[i]);
                        fdd_timer().stop("domain.inner_products");
                    }

                    for (int i = 0; i < j + 1; i++)
                    {
                        fdd_timer().start("domain.vector_operations");
                        math.vector_vector_addition(q_k, 1.0, q_k, -H[i][j], V[i], num_local_points);
                        fdd_timer().stop("domain.vector_operations");
                    }
                }

                for (int i = 0; i < j; i++)
                {
                    DType h_ij = H[i][j];
                    H[i][j] = c_gmres[i] * h_ij + s_gmres[i] * H[i + 1][j];
                    H[i + 1][j] = -s_gmres[i] * h_ij + c_gmres[i] * H[i + 1][j];
                }

                fdd_timer().start("domain.residual_norm");
                residual_norm(alpha_j, q_k);
                fdd_timer().stop("domain.residual_norm");

                if (std::abs(alpha_j) == 0.0)
                {
                    converged = true;
                    break;
                }

                beta_j = std::sqrt(H[j][j] * H[j][j] + alpha_j * alpha_j);
                gamma_j = 1.0 / beta_j;
                c_gmres[j] = H[j][j] * gamma_j;
                s_gmres[j] = alpha_j * gamma_j;
                H[j][j] = beta_j;
                gamma[j + 1] = -s_gmres[j] * gamma[j];
                gamma[j] = c_gmres[j] * gamma[j];

                r_norm = std::abs(gamma[j + 1]);
                residual_history.push_back(r_norm);
                rstdout("Iter %2d: | residual_norm = %24.16g | relative_residual_norm = %24.16g | \n", iter + 1, r_norm, r_norm / r_0_norm);

                if (use_relative)
                {
                    if (r_norm / r_0_norm < tolerance)
                    {
                        converged = true;
                        break;
                    }
                }
                else
                {
                    if (r_norm < tolerance)
                    {
                        converged = true;
                        break;
                    }
                }

                if (iter >= max_iterations)
                {
                    converged = true;
                    break;
                }

                if (std::isnan(r_norm))
                {
                    converged = true;
                    break;
                }

                fdd_timer().start("domain.vector_operations");
                math.vector_scaling(V[j + 1], 1.0 / alpha_j, q_k, num_local_points);
                fdd_timer().stop("domain.vector_operations");

                iter++;
            }

            if (j == num_vectors) j--;

            // back substitution stored into c_gmres (domain.tpp:891-899)
            for (int k = j; k >= 0; k--)
            {
                gamma_k = gamma[k];
                for (int i = j; i > k; i--) gamma_k -= H[k][i] * c_gmres[i];
                c_gmres[k] = gamma_k / H[k][k];
            }

            if (restructured_outer)
            {
                fdd_timer().start("domain.vector_operations");
                for (int g0 = 0; g0 < j + 1; g0 += FDD_MULTI_MAX)
                {
                    const int cnt = std::min(FDD_MULTI_MAX, j + 1 - g0);
                    const double *ptrs[FDD_MULTI_MAX];
                    for (int i = 0; i < cnt; i++) ptrs[i] = Z[g0 + i].template as<double>();
                    FDD_CALL(fdd_multi_axpy(u_k.as<double>(), c_gmres.data() + g0, ptrs, cnt, num_local_points, fdd::dev().stream));
                }
                fdd_timer().stop("domain.vector_operations");
            }
            else
            {
                for (int i = 0; i < j + 1; i++)
                {
                    fdd_timer().start("domain.vector_operations");
                    math.vector_vector_addition(u_k, 1.0, u_k, c_gmres[i], Z[i], num_local_points);
                    fdd_timer().stop("domain.vector_operations");
                }
            }

            if (converged) break;
        }

        num_iterations = iter;
    }
};

#endif
