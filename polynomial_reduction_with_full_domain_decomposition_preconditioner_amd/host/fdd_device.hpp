/*
 * fdd_device.hpp -- the thin layer that stands where OCCA stood.
 *
 * fdd::memory mirrors the slice of occa::memory the reference's host classes
 * use (copyFrom / copyTo / slice / ptr / free; usage inventory in SURVEY.md
 * section 8(b)) and fdd::device_t mirrors occa::device (malloc<T>, finish),
 * both over the C-ABI of include/fdd_hip.h.  Slices are pointer arithmetic.
 * A failing C-ABI call prints and exits, the reference's own error style
 * (csr_matrix.tpp:72-76).
 */
#ifndef FDD_DEVICE_HPP
#define FDD_DEVICE_HPP

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "fdd_hip.h"

namespace fdd
{

inline void check(int rc, const char *what)
{
    if (rc != 0)
    {
        fprintf(stderr, "ERROR: %s failed (code %d): %s\n", what, rc, fdd_last_error());
        exit(EXIT_FAILURE);
    }
}

#define FDD_CALL(expr) ::fdd::check((expr), #expr)

struct device_t
{
    void *stream = nullptr; // hipStream_t all kernels of this rank run on
    bool owns_stream = false; // created by fddh_init(own_stream): released by fddh_rank_finalize
    bool initialised = false; // fddh_init ran on this thread (a null stream alone cannot tell: it is also the legacy default stream)

    template <typename T>
    class memory malloc(size_t n);

    void finish() { FDD_CALL(fdd_stream_sync(stream)); }
};

// the device handle of the calling rank: one rank = one host thread (the reference's model), so the state is
// thread_local and several ranks may live in one process (LocalComm, comm.hpp)
inline device_t &dev()
{
    static thread_local device_t d;
    return d;
}

class memory
{
  private:
    char *base_ = nullptr;
    size_t count_ = 0; // elements
    size_t elem_ = 1;  // bytes per element
    bool owner_ = false;

  public:
    memory() {}
    memory(void *p, size_t count, size_t elem, bool owner) : base_((char *)p), count_(count), elem_(elem), owner_(owner) {}

    void *ptr() const { return base_; }
    template <typename T>
    T *as() const
    {
        return reinterpret_cast<T *>(base_);
    }
    size_t size() const { return count_; }
    bool isInitialized() const { return base_ != nullptr; }

    // host -> device (blocking, like occa::memory::copyFrom(const void*, bytes))
    void copyFrom(const void *src, size_t bytes) { FDD_CALL(fdd_memcpy_h2d(base_, src, bytes, dev().stream)); }
    // device -> device
    void copyFrom(const memory &src, size_t bytes) { FDD_CALL(fdd_memcpy_d2d(base_, src.base_, bytes, dev().stream)); }
    // device -> host (blocking)
    void copyTo(void *dst, size_t bytes) const
    {
        if (bytes <= 256) // reduction results: pinned staging inside the library
            FDD_CALL(fdd_fetch_scalars(dst, base_, bytes, dev().stream));
        else
            FDD_CALL(fdd_memcpy_d2h(dst, base_, bytes, dev().stream));
    }
    // device -> device
    void copyTo(memory dst, size_t bytes) const { FDD_CALL(fdd_memcpy_d2d(dst.base_, base_, bytes, dev().stream)); }

    // element units, as occa::memory::slice(offset, count) (subdomain.tpp:3945-3946)
    memory slice(size_t offset, size_t count) const { return memory(base_ + offset * elem_, count, elem_, false); }

    void free()
    {
        if (owner_ && base_) FDD_CALL(fdd_free(base_));
        base_ = nullptr;
        count_ = 0;
        owner_ = false;
    }
};

template <typename T>
inline memory device_t::malloc(size_t n)
{
    void *p = nullptr;
    FDD_CALL(fdd_malloc(&p, n * sizeof(T)));
    return memory(p, n, sizeof(T), true);
}

// Per-kernel timing with HIP events on the rank's own stream (what bench.py's
// `roofline` object reports).  Off by default; when on, two event records
// bracket each instrumented launch -- no synchronisation until collect().
// Keys are the device kernel families as rocprofv3 names them, so the averages
// can be checked against a --kernel-trace --stats run of the same command.
class KernelProfiler
{
  public:
    struct Stat
    {
        long long count = 0;
        double ms = 0.0;
        double bytes = 0.0; // algorithmic bytes, summed over launches
    };

  private:
    struct Rec
    {
        int key;
        double bytes;
        void *e0;
        void *e1;
    };
    std::vector<std::string> keys_;
    std::vector<Rec> recs_;
    std::vector<void *> pool_;
    size_t used_ = 0;

    void *event()
    {
        if (used_ == pool_.size())
        {
            void *e = nullptr;
            FDD_CALL(fdd_event_create(&e));
            pool_.push_back(e);
        }
        return pool_[used_++];
    }

    int key_id(const char *key)
    {
        for (size_t i = 0; i < keys_.size(); i++)
            if (keys_[i] == key) return (int)i;
        keys_.push_back(key);
        return (int)keys_.size() - 1;
    }

    std::vector<char> open_; // per open scope: was it recorded

  public:
    bool enabled = false;
    std::string only; // when set, only this kernel family is timed (two event records cost a few microseconds each)

    void begin(const char *key, double bytes)
    {
        if (!enabled) return;
        const bool take = only.empty() or only == key;
        open_.push_back(take ? 1 : 0);
        if (!take) return;
        Rec r{key_id(key), bytes, event(), event()};
        FDD_CALL(fdd_event_record(r.e0, dev().stream));
        recs_.push_back(r);
    }

    void end()
    {
        if (!enabled or open_.empty()) return;
        const bool took = open_.back() != 0;
        open_.pop_back();
        if (took) FDD_CALL(fdd_event_record(recs_.back().e1, dev().stream));
    }

    void reset()
    {
        recs_.clear();
        open_.clear();
        used_ = 0;
    }

    // synchronises, folds every recorded launch into per-key totals, clears
    std::vector<std::pair<std::string, Stat>> collect()
    {
        dev().finish();
        std::vector<Stat> stats(keys_.size());
        for (const Rec &r : recs_)
        {
            float ms = 0.0f;
            FDD_CALL(fdd_event_elapsed_ms(&ms, r.e0, r.e1));
            stats[r.key].count++;
            stats[r.key].ms += ms;
            stats[r.key].bytes += r.bytes;
        }
        reset();
        std::vector<std::pair<std::string, Stat>> out;
        for (size_t i = 0; i < keys_.size(); i++)
            if (stats[i].count) out.push_back(std::make_pair(keys_[i], stats[i]));
        return out;
    }
};

inline KernelProfiler &profiler()
{
    static thread_local KernelProfiler p;
    return p;
}

struct ProfileScope
{
    ProfileScope(const char *key, double bytes) { profiler().begin(key, bytes); }
    ~ProfileScope() { profiler().end(); }
};

} // namespace fdd

#endif
