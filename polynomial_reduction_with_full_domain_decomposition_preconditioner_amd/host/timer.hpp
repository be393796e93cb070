/*
 * timer.hpp -- Timer<DType>: named wall-clock regions with the reference's
 * interface (timer.hpp:15-38: initialize / start / stop / reset / total).
 *
 * What is deliberately NOT inherited: the reference puts
 * a device.finish() + MPI_Barrier at every start and a device.finish() +
 * MPI_Allreduce(MAX) over a num_procs-long array at every stop
 * (timer.tpp:47-68), which serialises the solve it measures, and it hashes
 * `const char*` keys by address (timer.hpp:20-22).  Here keys are strings,
 * regions synchronise the stream only when `enabled` (off by default, so the
 * solve path carries no host syncs for timing), there is no barrier inside a
 * region, and the max over ranks is taken once, in total(key, "max").
 */
#ifndef FDD_TIMER_HPP
#define FDD_TIMER_HPP

#include <chrono>
#include <string>
#include <unordered_map>

#include "config.hpp"

template <typename DType = double>
class Timer
{
  private:
    typedef std::chrono::high_resolution_clock clock;
    std::unordered_map<std::string, clock::time_point> t_start;
    std::unordered_map<std::string, DType> t_total;

  public:
    bool enabled = false;

    Timer() {}
    ~Timer() {}

    void initialize() { t_total.clear(); }

    void start(const char *key, bool sync = true)
    {
        if (!enabled) return;
        if (sync) fdd::dev().finish();
        t_start[key] = clock::now();
    }

    void stop(const char *key, bool sync = true)
    {
        if (!enabled) return;
        if (sync) fdd::dev().finish();
        auto it = t_start.find(key);
        if (it == t_start.end()) return;
        t_total[key] += std::chrono::duration<DType>(clock::now() - it->second).count();
    }

    void reset(const char *key) { t_total[key] = 0.0; }

    DType total(const char *key)
    {
        auto it = t_total.find(key);
        return (it == t_total.end()) ? (DType)0.0 : it->second;
    }

    // aggregation over ranks: "max" (the reference's table, poisson.cpp:256) or "sum"
    DType total(const char *key, const char *aggregation)
    {
        double v = (double)total(key);
        if (std::string(aggregation) == "max")
            fdd::comm().allreduce_max_host(&v, 1);
        else
            fdd::comm().allreduce_sum_host(&v, 1);
        return (DType)v;
    }
};

inline Timer<double> &fdd_timer()
{
    static thread_local Timer<double> t;
    return t;
}

// (no `timer` macro: a header a maintainer drops into the reference's include path must not redefine a common identifier;
// the host classes call fdd_timer() where the reference has its global `timer`, config.hpp:50)

#endif
