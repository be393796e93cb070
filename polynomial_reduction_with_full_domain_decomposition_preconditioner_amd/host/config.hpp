/*
 * config.hpp -- per-rank (= per host thread) state shared by the host classes, the role of the
 * reference's config.hpp (globals dim / proc_id / num_procs / device, the
 * rstdout / pstdout macros and quit()).  The device and communicator objects
 * live in fdd_device.hpp / comm.hpp.
 */
#ifndef FDD_CONFIG_HPP
#define FDD_CONFIG_HPP

#include <cstdio>
#include <cstdlib>

#include "comm.hpp"
#include "fdd_device.hpp"

#ifndef BLOCK_SIZE
#define BLOCK_SIZE 128 /* AMG/config.hpp:5 -- only the num_blocks bookkeeping uses it here */
#endif

#ifndef NUM_GEOM_FACTS
#define NUM_GEOM_FACTS 6 /* element.hpp:10-12 */
#endif

namespace fdd
{

struct globals_t
{
    int dim = 3;
    int proc_id = 0;
    int num_procs = 1;
    bool print = true;          // rank-0 residual history lines (rstdout)
    FILE *pstdout_file = nullptr; // per-rank log (config.hpp:27), optional
};

inline globals_t &globals()
{
    static thread_local globals_t g; // per rank = per host thread
    return g;
}

} // namespace fdd

#define rstdout(...)                                                   \
    {                                                                  \
        if (fdd::globals().proc_id == 0 && fdd::globals().print)       \
        {                                                              \
            printf(__VA_ARGS__);                                       \
            fflush(stdout);                                            \
        }                                                              \
    }

#define pstdout(...)                                                   \
    {                                                                  \
        if (fdd::globals().pstdout_file)                               \
        {                                                              \
            fprintf(fdd::globals().pstdout_file, __VA_ARGS__);         \
            fflush(fdd::globals().pstdout_file);                       \
        }                                                              \
    }

namespace fdd
{
// the reference's quit() exits with SUCCESS after finalising MPI/HYPRE (config.hpp:57-62)
inline void quit()
{
    dev().finish();
    exit(EXIT_SUCCESS);
}
} // namespace fdd

#endif
