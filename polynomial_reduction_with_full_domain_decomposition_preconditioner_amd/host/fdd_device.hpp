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

    template <typename T>
    class memory malloc(size_t n);

    void finish() { FDD_CALL(fdd_stream_sync(stream)); }
};

inline device_t &dev()
{
    static device_t d;
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
    void copyTo(void *dst, size_t bytes) const { FDD_CALL(fdd_memcpy_d2h(dst, base_, bytes, dev().stream)); }
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

} // namespace fdd

#endif
