/*
 * comm.hpp -- the communication layer that stands where MPI + gslib stood.
 *
 * The reference's solve path has three kinds of exchange:
 *   MPI_Allreduce(SUM) of 1-2 scalars        (domain.tpp:929,946,969,995)
 *   gslib gs(gs_add) on the boundary prefix  (domain.tpp:590-594)
 *   MPI_Allgatherv of the coarse level       (subdomain.tpp:4620-4621)
 * all host-staged.  Here every exchange is a collective on DEVICE buffers,
 * one rank per GPU:
 *   - SingleComm: one rank, everything is a no-op / copy;
 *   - RcclComm:   RCCL over xGMI, called directly (librccl is dlopen'ed so the
 *                 kernel library carries no link-time dependency on it);
 *   - CallbackComm: the collectives are supplied by the embedding process
 *                 through C function pointers (torch.distributed: "nccl" on
 *                 GPUs, "gloo" in the CPU tests);
 *   - LocalComm:  several ranks inside ONE process, one host thread and one
 *                 stream each, sharing one GPU: collectives are device-to-device
 *                 copies and sums between the ranks' buffers (rehearsal of the
 *                 N-rank path on a one-GPU box; RCCL admits one rank per device).
 * gs_add becomes: scatter the prefix into a dense interface-slot vector,
 * all-reduce it, gather back (slots = sorted unique global ids that appear in
 * any rank's prefix).
 * The fourth exchange of the reference, the gslib "pull" of neighbour elements
 * at reduced degree into a rank's composite region (subdomain.tpp:601-642,
 * 4626), is a grouped send / receive between the ranks whose regions overlap:
 * Comm::exchange, one packed device buffer per peer and direction
 * (ncclSend/ncclRecv inside one group on xGMI; torch.distributed
 * batch_isend_irecv behind the callbacks).
 */
#ifndef FDD_COMM_HPP
#define FDD_COMM_HPP

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "fdd_device.hpp"

namespace fdd
{

// A failed collective does not end the process from inside the library: it throws, the C-ABI entry that was running
// returns non-zero and fddh_last_error() has the text (the driver prints it and exits with a failure code).
struct CommError : public std::runtime_error
{
    explicit CommError(const std::string &what) : std::runtime_error(what) {}
};

// one peer of a grouped point-to-point exchange: `send_bytes` from `send` go to `peer`, `recv_bytes` arrive from it
// in `recv` (device buffers; either side may be empty).  A peer is never the calling rank; it may appear more than once
// in a call, and its k-th op then pairs with the peer's k-th op for this rank (both sides list their parts in one order).
struct ExchangeOp
{
    int peer = 0;
    const void *send = nullptr;
    size_t send_bytes = 0;
    void *recv = nullptr;
    size_t recv_bytes = 0;
};

class Comm
{
  public:
    int rank = 0;
    int size = 1;

    virtual ~Comm() {}
    virtual const char *name() const = 0;
    // in-place sum over ranks of n doubles in device memory
    virtual void allreduce_sum(double *buf_dev, size_t n) = 0;
    // in-place max over ranks of n doubles in device memory
    virtual void allreduce_max(double *buf_dev, size_t n) = 0;
    // every rank contributes `bytes` from send_dev; recv_dev gets size*bytes in rank order
    virtual void allgather(const void *send_dev, void *recv_dev, size_t bytes) = 0;
    // grouped sends and receives on device buffers, ordered on the rank's stream like the collectives
    virtual void exchange(const ExchangeOp *ops, int n) = 0;
    virtual void barrier() = 0;

    // Point-to-point exchange of host byte strings (setup only): out[p] goes to rank p, the result holds what every
    // rank sent to this one.  Sizes are agreed on first (a dense size x size count matrix: setup runs at rank counts
    // where that is small); messages are staged through device scratch.
    std::vector<std::vector<char>> exchange_host(const std::vector<std::vector<char>> &out)
    {
        std::vector<std::vector<char>> in(size);
        if (size == 1)
        {
            in[0] = out[0];
            return in;
        }
        std::vector<double> counts((size_t)size * size, 0.0);
        for (int p = 0; p < size; p++) counts[(size_t)rank * size + p] = (double)out[p].size();
        allreduce_sum_host(counts.data(), counts.size());
        in[rank] = out[rank];
        std::vector<ExchangeOp> ops;
        std::vector<memory> sbuf(size), rbuf(size);
        for (int p = 0; p < size; p++)
        {
            if (p == rank) continue;
            const size_t ns = out[p].size(), nr = (size_t)counts[(size_t)p * size + rank];
            if (ns == 0 and nr == 0) continue;
            ExchangeOp op;
            op.peer = p;
            if (ns)
            {
                sbuf[p] = dev().malloc<char>(ns);
                sbuf[p].copyFrom(out[p].data(), ns);
                op.send = sbuf[p].ptr();
                op.send_bytes = ns;
            }
            if (nr)
            {
                rbuf[p] = dev().malloc<char>(nr);
                op.recv = rbuf[p].ptr();
                op.recv_bytes = nr;
            }
            ops.push_back(op);
        }
        exchange(ops.data(), (int)ops.size()); // also with no peers: every rank of the world takes part in the call
        for (const ExchangeOp &op : ops)
        {
            if (op.recv_bytes)
            {
                in[op.peer].resize(op.recv_bytes);
                rbuf[op.peer].copyTo(in[op.peer].data(), op.recv_bytes);
                rbuf[op.peer].free();
            }
            if (op.send_bytes) sbuf[op.peer].free();
        }
        return in;
    }

    // ---- host-side helpers for setup (tiny, staged through device scratch) ----
    void allreduce_sum_host(double *v, size_t n)
    {
        if (size == 1) return;
        memory tmp = dev().malloc<double>(n);
        tmp.copyFrom(v, n * sizeof(double));
        allreduce_sum(tmp.as<double>(), n);
        tmp.copyTo(v, n * sizeof(double));
        tmp.free();
    }

    void allreduce_max_host(double *v, size_t n)
    {
        if (size == 1) return;
        memory tmp = dev().malloc<double>(n);
        tmp.copyFrom(v, n * sizeof(double));
        allreduce_max(tmp.as<double>(), n);
        tmp.copyTo(v, n * sizeof(double));
        tmp.free();
    }

    // MPI_Allgatherv of int64 lists: returns the concatenation in rank order
    // and fills counts[rank]
    std::vector<long long> allgatherv_host(const std::vector<long long> &local, std::vector<int> &counts)
    {
        counts.assign(size, 0);
        if (size == 1)
        {
            counts[0] = (int)local.size();
            return local;
        }

        std::vector<double> cnt(size, 0.0);
        cnt[rank] = (double)local.size();
        allreduce_sum_host(cnt.data(), size);
        size_t max_count = 0;
        for (int p = 0; p < size; p++)
        {
            counts[p] = (int)cnt[p];
            max_count = std::max(max_count, (size_t)counts[p]);
        }
        if (max_count == 0) return std::vector<long long>();

        std::vector<long long> padded(max_count, 0);
        std::copy(local.begin(), local.end(), padded.begin());
        memory send = dev().malloc<long long>(max_count);
        memory recv = dev().malloc<long long>(max_count * size);
        send.copyFrom(padded.data(), max_count * sizeof(long long));
        allgather(send.ptr(), recv.ptr(), max_count * sizeof(long long));
        std::vector<long long> all(max_count * size);
        recv.copyTo(all.data(), all.size() * sizeof(long long));
        send.free();
        recv.free();

        std::vector<long long> out;
        for (int p = 0; p < size; p++) out.insert(out.end(), all.begin() + p * max_count, all.begin() + p * max_count + counts[p]);
        return out;
    }
};

class SingleComm : public Comm
{
  public:
    const char *name() const override { return "single"; }
    void allreduce_sum(double *, size_t) override {}
    void allreduce_max(double *, size_t) override {}
    void allgather(const void *send_dev, void *recv_dev, size_t bytes) override
    {
        if (send_dev != recv_dev) FDD_CALL(fdd_memcpy_d2d(recv_dev, send_dev, bytes, dev().stream));
    }
    void exchange(const ExchangeOp *, int n) override
    {
        if (n > 0) throw CommError("point-to-point exchange on a single-rank communicator");
    }
    void barrier() override {}
};

// Collectives supplied by the embedding process (C function pointers).  Each
// returns 0 on success.  Pointers are device pointers of this rank (host
// pointers in the CPU test build); the callee must order the collective after
// work already queued on the rank's stream and before work queued later.
struct CommCallbacks
{
    void *ctx;
    int (*allreduce_sum_f64)(void *ctx, void *buf, long long n);
    int (*allreduce_max_f64)(void *ctx, void *buf, long long n);
    int (*allgather_bytes)(void *ctx, const void *send, void *recv, long long bytes);
    int (*barrier)(void *ctx);
    // n peers: send_bytes[i] from send[i] to peers[i], recv_bytes[i] from peers[i] into recv[i] (one group)
    int (*exchange_bytes)(void *ctx, int n, const int *peers, const void *const *send, const long long *send_bytes, void *const *recv, const long long *recv_bytes);
};

class CallbackComm : public Comm
{
    CommCallbacks cb_;

    static void ok(int rc, const char *what)
    {
        if (rc != 0) throw CommError(std::string("communication callback ") + what + " failed (code " + std::to_string(rc) + ")");
    }

  public:
    CallbackComm(int rank_, int size_, const CommCallbacks &cb) : cb_(cb)
    {
        rank = rank_;
        size = size_;
    }
    const char *name() const override { return "callback"; }
    void allreduce_sum(double *buf, size_t n) override { ok(cb_.allreduce_sum_f64(cb_.ctx, buf, (long long)n), "allreduce_sum_f64"); }
    void allreduce_max(double *buf, size_t n) override { ok(cb_.allreduce_max_f64(cb_.ctx, buf, (long long)n), "allreduce_max_f64"); }
    void allgather(const void *send, void *recv, size_t bytes) override { ok(cb_.allgather_bytes(cb_.ctx, send, recv, (long long)bytes), "allgather_bytes"); }
    void exchange(const ExchangeOp *ops, int n) override
    {
        if (n <= 0) return;
        if (not cb_.exchange_bytes) throw CommError("the communication callbacks have no exchange_bytes entry (needed by the composite region)");
        std::vector<int> peers(n);
        std::vector<const void *> sp(n);
        std::vector<void *> rp(n);
        std::vector<long long> sb(n), rb(n);
        for (int i = 0; i < n; i++)
        {
            peers[i] = ops[i].peer;
            sp[i] = ops[i].send;
            sb[i] = (long long)ops[i].send_bytes;
            rp[i] = ops[i].recv;
            rb[i] = (long long)ops[i].recv_bytes;
        }
        ok(cb_.exchange_bytes(cb_.ctx, n, peers.data(), sp.data(), sb.data(), rp.data(), rb.data()), "exchange_bytes");
    }
    void barrier() override { ok(cb_.barrier(cb_.ctx), "barrier"); }
};

// RCCL called directly.  Only the handful of entry points used are resolved.
class RcclComm : public Comm
{
  public:
    static constexpr int kUniqueIdBytes = 128; // NCCL_UNIQUE_ID_BYTES

  private:
    struct UniqueId
    {
        char internal[kUniqueIdBytes];
    };
    typedef int (*GetUniqueId_t)(UniqueId *);
    typedef int (*CommInitRank_t)(void **, int, UniqueId, int);
    typedef int (*CommDestroy_t)(void *);
    typedef int (*AllReduce_t)(const void *, void *, size_t, int, int, void *, void *);
    typedef int (*AllGather_t)(const void *, void *, size_t, int, void *, void *);
    typedef int (*SendRecv_t)(void *, size_t, int, int, void *, void *); // ncclSend (const buffer) / ncclRecv
    typedef int (*Group_t)(void);
    typedef const char *(*GetErrorString_t)(int);

    void *lib_ = nullptr;
    void *comm_ = nullptr;
    memory token_;
    GetUniqueId_t get_unique_id_ = nullptr;
    CommInitRank_t comm_init_rank_ = nullptr;
    CommDestroy_t comm_destroy_ = nullptr;
    AllReduce_t all_reduce_ = nullptr;
    AllGather_t all_gather_ = nullptr;
    SendRecv_t send_ = nullptr, recv_ = nullptr;
    Group_t group_start_ = nullptr, group_end_ = nullptr;
    GetErrorString_t get_error_string_ = nullptr;

    enum
    {
        kSum = 0,
        kMax = 2,
        kInt8 = 0,
        kFloat64 = 8
    }; // rccl.h ncclRedOp_t / ncclDataType_t

    void ok(int rc, const char *what)
    {
        if (rc != 0) throw CommError(std::string("RCCL ") + what + " failed: " + (get_error_string_ ? get_error_string_(rc) : "?"));
    }

    void load()
    {
        if (lib_) return;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names)
        {
            lib_ = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib_) break;
        }
        if (!lib_) throw CommError(std::string("cannot load librccl: ") + dlerror());
        get_unique_id_ = (GetUniqueId_t)dlsym(lib_, "ncclGetUniqueId");
        comm_init_rank_ = (CommInitRank_t)dlsym(lib_, "ncclCommInitRank");
        comm_destroy_ = (CommDestroy_t)dlsym(lib_, "ncclCommDestroy");
        all_reduce_ = (AllReduce_t)dlsym(lib_, "ncclAllReduce");
        all_gather_ = (AllGather_t)dlsym(lib_, "ncclAllGather");
        send_ = (SendRecv_t)dlsym(lib_, "ncclSend");
        recv_ = (SendRecv_t)dlsym(lib_, "ncclRecv");
        group_start_ = (Group_t)dlsym(lib_, "ncclGroupStart");
        group_end_ = (Group_t)dlsym(lib_, "ncclGroupEnd");
        get_error_string_ = (GetErrorString_t)dlsym(lib_, "ncclGetErrorString");
        if (!get_unique_id_ || !comm_init_rank_ || !all_reduce_ || !all_gather_ || !send_ || !recv_ || !group_start_ || !group_end_)
            throw CommError("librccl lacks a required entry point");
    }

  public:
    RcclComm() { load(); }
    ~RcclComm() override
    {
        if (comm_ && comm_destroy_) comm_destroy_(comm_);
    }
    const char *name() const override { return "rccl"; }

    // rank 0 creates the id; the launcher ships the 128 bytes to every rank
    void unique_id(char *out128)
    {
        UniqueId id;
        ok(get_unique_id_(&id), "ncclGetUniqueId");
        memcpy(out128, id.internal, kUniqueIdBytes);
    }

    void init(const char *id128, int rank_, int size_)
    {
        UniqueId id;
        memcpy(id.internal, id128, kUniqueIdBytes);
        rank = rank_;
        size = size_;
        ok(comm_init_rank_(&comm_, size_, id, rank_), "ncclCommInitRank");
        token_ = dev().malloc<double>(1);
    }

    void allreduce_sum(double *buf, size_t n) override { ok(all_reduce_(buf, buf, n, kFloat64, kSum, comm_, dev().stream), "ncclAllReduce"); }
    void allreduce_max(double *buf, size_t n) override { ok(all_reduce_(buf, buf, n, kFloat64, kMax, comm_, dev().stream), "ncclAllReduce"); }
    void allgather(const void *send, void *recv, size_t bytes) override { ok(all_gather_(send, recv, bytes, kInt8, comm_, dev().stream), "ncclAllGather"); }
    // one group: every send and receive of the call progresses together over the xGMI links (all peers are one hop)
    void exchange(const ExchangeOp *ops, int n) override
    {
        if (n <= 0) return;
        ok(group_start_(), "ncclGroupStart");
        // a failing call must not leave the communicator in group mode: remember the first error, always close the
        // group, throw afterwards
        int first_rc = 0;
        const char *first_what = nullptr;
        for (int i = 0; i < n and first_rc == 0; i++)
        {
            if (ops[i].send_bytes)
            {
                const int rc = send_(const_cast<void *>(ops[i].send), ops[i].send_bytes, kInt8, ops[i].peer, comm_, dev().stream);
                if (rc != 0 and first_rc == 0) first_rc = rc, first_what = "ncclSend";
            }
            if (ops[i].recv_bytes and first_rc == 0)
            {
                const int rc = recv_(ops[i].recv, ops[i].recv_bytes, kInt8, ops[i].peer, comm_, dev().stream);
                if (rc != 0 and first_rc == 0) first_rc = rc, first_what = "ncclRecv";
            }
        }
        const int end_rc = group_end_();
        if (first_rc != 0) ok(first_rc, first_what);
        ok(end_rc, "ncclGroupEnd");
    }
    void barrier() override
    {
        FDD_CALL(fdd_memset(token_.ptr(), 0, sizeof(double), dev().stream));
        allreduce_sum(token_.as<double>(), 1);
        dev().finish();
    }
};


// ---------------------------------------------------------------------------------------------------------------
// Several ranks in one process.  The per-rank state of the host layer (device stream, communicator, globals, timer,
// profiler) is thread_local: a rank is a host thread, as in the reference (one MPI rank = one host thread), and
// nothing stops several of them from living in one process and sharing a GPU.  RCCL cannot connect them (one rank per
// device and communicator), so this back-end moves the data itself: every collective is
//   finish my stream | publish my pointers | meet | copy / sum from the peers' DEVICE buffers on my stream | finish | meet
// Sums run in rank order on every rank: all ranks get bit-identical results, like a ring all-reduce's.
// Used by the 4- and 8-rank tests and the bench rehearsal on the one-GPU box; not a performance path.
// ---------------------------------------------------------------------------------------------------------------
struct LocalWorld
{
    const int size;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    bool failed = false;
    std::vector<void *> buf;                     // all-reduce operands
    std::vector<const void *> send;              // all-gather operands
    std::vector<std::vector<ExchangeOp>> ops;    // exchange operands
    int timeout_seconds = 300;

    explicit LocalWorld(int n) : size(n), buf(n, nullptr), send(n, nullptr), ops(n) {}

    void fail()
    {
        std::lock_guard<std::mutex> lock(m);
        failed = true;
        cv.notify_all();
    }

    // every rank arrives; a rank that never does (it failed elsewhere) turns into an error after the timeout, not a hang
    void meet()
    {
        std::unique_lock<std::mutex> lock(m);
        if (failed) throw CommError("local world: a peer rank failed");
        const long gen = generation;
        if (++arrived == size)
        {
            arrived = 0;
            generation++;
            cv.notify_all();
            return;
        }
        const bool ok = cv.wait_for(lock, std::chrono::seconds(timeout_seconds), [&] { return generation != gen or failed; });
        if (not ok or failed)
        {
            failed = true;
            cv.notify_all();
            throw CommError(ok ? "local world: a peer rank failed" : "local world: timed out waiting for a peer rank (collective mismatch?)");
        }
    }
};

class LocalComm : public Comm
{
    std::shared_ptr<LocalWorld> w_;
    memory tmp_;
    size_t tmp_n_ = 0;

    double *scratch(size_t n)
    {
        if (n > tmp_n_)
        {
            tmp_.free();
            tmp_ = dev().malloc<double>(n);
            tmp_n_ = n;
        }
        return tmp_.as<double>();
    }

    void reduce(double *buf, size_t n, bool max)
    {
        if (n == 0) return;
        void *stream = dev().stream;
        dev().finish();
        w_->buf[rank] = buf;
        w_->meet();
        double *t = scratch(n);
        try
        {
            if (max)
            {
                // small vectors only (timers, setup flags): through the host
                std::vector<double> acc(n), one(n);
                for (int r = 0; r < size; r++)
                {
                    FDD_CALL(fdd_memcpy_d2h(one.data(), w_->buf[r], n * sizeof(double), stream));
                    for (size_t i = 0; i < n; i++) acc[i] = (r == 0) ? one[i] : std::max(acc[i], one[i]);
                }
                FDD_CALL(fdd_memcpy_h2d(t, acc.data(), n * sizeof(double), stream));
            }
            else
            {
                FDD_CALL(fdd_memcpy_d2d(t, w_->buf[0], n * sizeof(double), stream));
                for (int r = 1; r < size; r++) FDD_CALL(fdd_vector_vector_addition(t, 1.0, t, 1.0, (const double *)w_->buf[r], (int)n, stream));
            }
            dev().finish();
        }
        catch (...)
        {
            w_->fail();
            throw;
        }
        w_->meet(); // every rank has read every operand
        FDD_CALL(fdd_memcpy_d2d(buf, t, n * sizeof(double), stream));
    }

  public:
    LocalComm(std::shared_ptr<LocalWorld> world, int rank_) : w_(std::move(world))
    {
        rank = rank_;
        size = w_->size;
    }
    const char *name() const override { return "local"; }
    void allreduce_sum(double *buf, size_t n) override { reduce(buf, n, false); }
    void allreduce_max(double *buf, size_t n) override { reduce(buf, n, true); }
    void allgather(const void *send_dev, void *recv_dev, size_t bytes) override
    {
        void *stream = dev().stream;
        dev().finish();
        w_->send[rank] = send_dev;
        w_->meet();
        try
        {
            for (int r = 0; r < size; r++)
                if (bytes) FDD_CALL(fdd_memcpy_d2d((char *)recv_dev + (size_t)r * bytes, w_->send[r], bytes, stream));
            dev().finish();
        }
        catch (...)
        {
            w_->fail(); // the peers are on their way to the second meeting point: an error there, not a wait
            throw;
        }
        w_->meet();
    }
    void exchange(const ExchangeOp *ops, int n) override
    {
        // like the other back-ends this is collective among the ranks that exchange; here every rank of the world
        // must call it (a rank without peers passes n = 0), because the meeting point is world-wide
        void *stream = dev().stream;
        dev().finish();
        w_->ops[rank].assign(ops, ops + std::max(n, 0));
        w_->meet();
        std::string error;
        try
        {
            for (int i = 0; i < n and error.empty(); i++)
            {
                // a peer may appear more than once in a call (ring part + coarse part): the k-th op for a peer pairs with that
                // peer's k-th op for this rank, as grouped ncclSend / ncclRecv to one peer match in order
                int nth = 0;
                for (int j = 0; j < i; j++)
                    if (ops[j].peer == ops[i].peer) nth++;
                if (ops[i].recv_bytes == 0) continue;
                const ExchangeOp *theirs = nullptr;
                int seen = 0;
                for (const ExchangeOp &o : w_->ops[ops[i].peer])
                    if (o.peer == rank and seen++ == nth) theirs = &o;
                if (theirs == nullptr or theirs->send_bytes != ops[i].recv_bytes)
                    error = "local world: rank " + std::to_string(rank) + " expects " + std::to_string(ops[i].recv_bytes) + " bytes from rank " + std::to_string(ops[i].peer) + ", which sends " + std::to_string(theirs ? theirs->send_bytes : 0);
                else
                    FDD_CALL(fdd_memcpy_d2d(ops[i].recv, theirs->send, ops[i].recv_bytes, stream));
            }
            if (error.empty()) dev().finish();
        }
        catch (...)
        {
            w_->fail();
            throw;
        }
        if (not error.empty())
        {
            w_->fail();
            throw CommError(error);
        }
        w_->meet();
    }
    void barrier() override
    {
        dev().finish();
        w_->meet();
    }
};

// the communicator of the calling rank (= host thread)
inline Comm *&comm_ptr()
{
    static thread_local Comm *c = new SingleComm();
    return c;
}

inline Comm &comm() { return *comm_ptr(); }

inline void set_comm(Comm *c)
{
    Comm *&p = comm_ptr();
    if (p != c) delete p;
    p = c;
}

} // namespace fdd

#endif
