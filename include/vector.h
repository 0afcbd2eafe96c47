// vector.h — device buffer RAII over the C ABI.  Same name and member functions as the
// reference's CudaVector<T> (include/vector.h:34-169) so its call sites port 1:1, but
//   * move-only (the reference is implicitly copyable => double cudaFree),
//   * every runtime call is checked (the reference checks none),
//   * copyFrom(T*, n) accepts n <= size and throws std::runtime_error beyond it, as the reference
//     does (:142-153).
// CudaVector(T* v, n) follows what the reference's constructor DOES (:124-127: allocate n elements
// and copy them from HOST pointer v), not its doc comment ("wraps an existing device pointer", :51-55);
// the non-owning view of device memory is CudaVector<T>::wrap(devicePtr, n).
#ifndef RMD_VECTOR_H
#define RMD_VECTOR_H

#include <cstddef>
#include <stdexcept>
#include <vector>

#include "utils.h"

template <typename T>
using CpuVector = std::vector<T>;

template <typename T>
struct CudaVector {
private:
    T* data_p = nullptr;
    size_t size_p = 0;
    bool owner = true;

public:
    CudaVector() = default;
    // (not `explicit`, like the reference's constructors at include/vector.h:119-130: a size or a host vector converts)
    CudaVector(size_t size) : size_p(size) { rmdCheck(rmd_malloc((void**)&data_p, size * sizeof(T)), "CudaVector"); }
    CudaVector(const T* hostData, size_t size) : CudaVector(size) { copyFrom(hostData, size); }   // reference :124-127
    CudaVector(const CpuVector<T>& v) : CudaVector(v.data(), v.size()) {}                         // reference :130
    // non-owning view of `size` elements of DEVICE memory (the destructor leaves them alone)
    static CudaVector wrap(T* devicePtr, size_t size)
    {
        CudaVector v;
        v.data_p = devicePtr; v.size_p = size; v.owner = false;
        return v;
    }
    CudaVector(const CudaVector&) = delete;
    CudaVector& operator=(const CudaVector&) = delete;
    CudaVector(CudaVector&& o) noexcept : data_p(o.data_p), size_p(o.size_p), owner(o.owner) { o.data_p = nullptr; o.size_p = 0; }
    CudaVector& operator=(CudaVector&& o) noexcept
    {
        if (this != &o) { release(); data_p = o.data_p; size_p = o.size_p; owner = o.owner; o.data_p = nullptr; o.size_p = 0; }
        return *this;
    }
    ~CudaVector() { release(); }

    void resize(size_t size)
    {
        release();
        size_p = size; owner = true;
        rmdCheck(rmd_malloc((void**)&data_p, size * sizeof(T)), "CudaVector::resize");
    }
    size_t size() const { return size_p; }
    T* data() { return data_p; }
    const T* data() const { return data_p; }

    void copyFrom(const CpuVector<T>& v)
    {
        if (v.size() != size_p) throw std::runtime_error("CudaVector::copyFrom: size mismatch");
        rmdCheck(rmd_memcpy_h2d(data_p, v.data(), size_p * sizeof(T)), "CudaVector::copyFrom");
    }
    void copyFromAsync(const CpuVector<T>& v, void* stream = nullptr)
    {
        if (v.size() != size_p) throw std::runtime_error("CudaVector::copyFromAsync: size mismatch");
        rmdCheck(rmd_memcpy_h2d_async(data_p, v.data(), size_p * sizeof(T), stream), "CudaVector::copyFromAsync");
    }
    void copyFrom(const T* host, size_t n)                                      // reference :142-146: n <= size
    {
        if (n > size_p) throw std::runtime_error("CudaVector::copyFrom: size mismatch");
        rmdCheck(rmd_memcpy_h2d(data_p, host, n * sizeof(T)), "CudaVector::copyFrom");
    }
    void copyFromAsync(const T* host, size_t n, void* stream = nullptr)         // reference :149-153
    {
        if (n > size_p) throw std::runtime_error("CudaVector::copyFromAsync: size mismatch");
        rmdCheck(rmd_memcpy_h2d_async(data_p, host, n * sizeof(T), stream), "CudaVector::copyFromAsync");
    }
    void copyToAsync(T* host, void* stream = nullptr) const                     // reference :161-163
    {
        rmdCheck(rmd_memcpy_d2h_async(host, data_p, size_p * sizeof(T), stream), "CudaVector::copyToAsync");
    }
    void copyTo(CpuVector<T>& v) const
    {
        v.resize(size_p);
        rmdCheck(rmd_memcpy_d2h(v.data(), data_p, size_p * sizeof(T)), "CudaVector::copyTo");
    }
    void copyToAsync(CpuVector<T>& v, void* stream = nullptr) const
    {
        v.resize(size_p);
        rmdCheck(rmd_memcpy_d2h_async(v.data(), data_p, size_p * sizeof(T), stream), "CudaVector::copyToAsync");
    }
    void copyTo(T* host) const { rmdCheck(rmd_memcpy_d2h(host, data_p, size_p * sizeof(T)), "CudaVector::copyTo"); }
    void fill(int byteValue, void* stream = nullptr) { rmdCheck(rmd_memset(data_p, byteValue, size_p * sizeof(T), stream), "CudaVector::fill"); }

private:
    void release()
    {
        if (owner && data_p) rmd_free(data_p);
        data_p = nullptr; size_p = 0;
    }
};

#endif
