// Host-side setup on the cores of the rank: contiguous index ranges on std::thread.  Every user keeps the order the
// serial loop would produce (pieces concatenated in range order, a thread placing only ITS rows while scanning in order),
// so no result depends on the thread count.
#ifndef FDD_HOST_PARALLEL_HPP
#define FDD_HOST_PARALLEL_HPP

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace fdd
{

// FDD_SETUP_TIMING=1: host time of the setup's phases, printed by the rank that is told to (`print`)
struct SetupTimer
{
    const char *scope;
    bool on;
    double mark;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    SetupTimer(const char *scope_, bool print) : scope(scope_), on(print and getenv("FDD_SETUP_TIMING") != nullptr), mark(now()) {}
    void lap(const char *what)
    {
        const double t = now();
        if (on) printf("%s %-36s %8.3f s\n", scope, what, t - mark);
        mark = t;
    }
};

// Distinct 64-bit keys (any value but LLONG_MIN) -> consecutive slots 0, 1, 2, ... in order of first appearance: open
// addressing with linear probing.  The setup's numbering passes look tens of millions of global ids up; a node-based
// std::unordered_map spends its time allocating.
class KeySlots
{
    std::vector<long long> table_key; // LLONG_MIN = empty
    std::vector<int> table_slot;
    std::vector<long long> keys;
    size_t mask;

  public:
    explicit KeySlots(size_t expected)
    {
        size_t cap = 16;
        while (2 * cap < 3 * expected + 4) cap <<= 1; // load factor <= 2/3
        table_key.assign(cap, LLONG_MIN);
        table_slot.assign(cap, -1);
        mask = cap - 1;
        keys.reserve(expected);
    }
    int find_or_insert(long long key)
    {
        size_t h = (size_t)(((unsigned long long)key * 0x9E3779B97F4A7C15ull) >> 17) & mask;
        while (table_key[h] != LLONG_MIN)
        {
            if (table_key[h] == key) return table_slot[h];
            h = (h + 1) & mask;
        }
        if (3 * (keys.size() + 1) > 2 * table_key.size())
        {
            grow();
            return find_or_insert(key);
        }
        table_key[h] = key;
        table_slot[h] = (int)keys.size();
        keys.push_back(key);
        return table_slot[h];
    }
    // more keys than announced: twice the table, every key back in (their slots stay)
    void grow()
    {
        const size_t cap = 2 * table_key.size();
        table_key.assign(cap, LLONG_MIN);
        table_slot.assign(cap, -1);
        mask = cap - 1;
        for (size_t s = 0; s < keys.size(); s++)
        {
            size_t h = (size_t)(((unsigned long long)keys[s] * 0x9E3779B97F4A7C15ull) >> 17) & mask;
            while (table_key[h] != LLONG_MIN) h = (h + 1) & mask;
            table_key[h] = keys[s];
            table_slot[h] = (int)s;
        }
    }
    long long key_of_slot(int slot) const { return keys[slot]; }
    int size() const { return (int)keys.size(); }
};

namespace low_order
{

// Setup runs on the host cores of the rank (one rank per GPU: 16 of them on a one-GPU box): contiguous index
// ranges on std::thread, results concatenated in range order, so every output is the one the serial loop gives.
// FDD_HOST_THREADS overrides the count (1 = serial).
inline int host_threads()
{
    if (const char *e = getenv("FDD_HOST_THREADS")) return std::max(1, atoi(e));
    unsigned hw = std::thread::hardware_concurrency();
    // several ranks on the node (LOCAL_WORLD_SIZE from the launcher): each takes its share of the cores
    if (const char *e = getenv("LOCAL_WORLD_SIZE"))
    {
        const int local = atoi(e);
        if (local > 1) hw = std::max(1u, hw / (unsigned)local);
    }
    return (int)std::min(16u, std::max(1u, hw));
}

// f(begin, end, part) for `parts` contiguous ranges covering [0, n)
template <typename F>
inline void parallel_ranges(long long n, int parts, F f)
{
    if (parts <= 1 or n < 2 * parts)
    {
        f(0LL, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < parts; t++) pool.emplace_back(f, n * t / parts, n * (t + 1) / parts, t);
    for (std::thread &th : pool) th.join();
}

inline int range_parts(long long n)
{
    const int t = host_threads();
    return (n < 2 * t) ? 1 : t;
}

// std::vector whose resize(n) leaves new elements uninitialised: the large index / value arrays of the setup are
// written in full by the threads right after they are sized, and a serial zero-fill of a gigabyte (plus its page
// faults on one core) costs more than the parallel loop that follows it.
template <typename T>
struct default_init_allocator : std::allocator<T>
{
    template <typename U>
    struct rebind
    {
        using other = default_init_allocator<U>;
    };
    default_init_allocator() = default;
    template <typename U>
    default_init_allocator(const default_init_allocator<U> &) {}
    template <typename U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... Args>
    void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
template <typename T>
using pod_vector = std::vector<T, default_init_allocator<T>>;

// out = the pieces one after the other (piece t produced by range t), copied by the threads; the pieces are released
template <typename T, typename Out, typename Piece>
inline void concatenate_into(Out &out, std::vector<Piece> &pieces)
{
    const int parts = (int)pieces.size();
    std::vector<size_t> offset((size_t)parts + 1, 0);
    for (int t = 0; t < parts; t++) offset[t + 1] = offset[t] + pieces[t].size();
    out.resize(offset[parts]);
    std::vector<std::thread> pool;
    for (int t = 0; t < parts; t++)
        pool.emplace_back([&, t] {
            if (not pieces[t].empty()) std::memcpy(out.data() + offset[t], pieces[t].data(), pieces[t].size() * sizeof(T));
            Piece().swap(pieces[t]);
        });
    for (std::thread &th : pool) th.join();
}
template <typename Out, typename Piece>
inline void concatenate(Out &out, std::vector<Piece> &pieces)
{
    concatenate_into<typename Out::value_type>(out, pieces);
}

} // namespace low_order

// ids[p] = the rank of keys[p] among the DISTINCT keys in order of first appearance (0, 1, 2, ...); returns their number.
// What one pass with a KeySlots gives, on the host threads: thread t owns the keys whose hash falls in its share, finds
// for each the first point that carries it (scanning the points in order), a prefix sum over the points that ARE such
// first carriers numbers them, and every point takes the number of its key's first carrier.
inline int first_appearance_ids(const long long *keys, size_t n, std::vector<int> &ids)
{
    ids.resize(n);
    const int T = low_order::range_parts((long long)n);
    if (T <= 1)
    {
        KeySlots first(n);
        for (size_t p = 0; p < n; p++) ids[p] = first.find_or_insert(keys[p]);
        return first.size();
    }
    std::vector<int> rep(n); // the first point with the same key
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++)
            pool.emplace_back([&, t] {
                KeySlots mine(n / (size_t)T + n / (size_t)(4 * T) + 16);
                std::vector<int> first_point;
                first_point.reserve(n / (size_t)T + 16);
                for (size_t p = 0; p < n; p++)
                {
                    const unsigned long long h = (unsigned long long)keys[p] * 0xD6E8FEB86659FD93ull;
                    if ((int)((h >> 32) % (unsigned long long)T) != t) continue;
                    const int s = mine.find_or_insert(keys[p]);
                    if (s == (int)first_point.size()) first_point.push_back((int)p);
                    rep[p] = first_point[s];
                }
            });
        for (std::thread &th : pool) th.join();
    }
    // number the first carriers in point order: counts per range, offsets, then the numbers
    std::vector<long long> range_count((size_t)T + 1, 0);
    low_order::parallel_ranges((long long)n, T, [&](long long p0, long long p1, int part) {
        long long c = 0;
        for (long long p = p0; p < p1; p++) c += (rep[p] == (int)p);
        range_count[(size_t)part + 1] = c;
    });
    for (int t = 0; t < T; t++) range_count[(size_t)t + 1] += range_count[t];
    low_order::parallel_ranges((long long)n, T, [&](long long p0, long long p1, int part) {
        int next = (int)range_count[part];
        for (long long p = p0; p < p1; p++)
            if (rep[p] == (int)p) ids[p] = next++;
    });
    low_order::parallel_ranges((long long)n, T, [&](long long p0, long long p1, int) {
        for (long long p = p0; p < p1; p++)
            if (rep[p] != (int)p) ids[p] = ids[rep[p]];
    });
    return (int)range_count[T];
}

} // namespace fdd

#endif
