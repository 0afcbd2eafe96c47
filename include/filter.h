// filter.h — the reference's filter entry points (include/filter.cuh:11-28) for C++ host code.
// FilterParams keeps the reference layout and default member initialisers; the two kernels keep
// their names as inline wrappers over the extern "C" launchers, so a call site
//     filterKernelBaseline<<<grid, block, smem>>>(frame, params);      (reference src/test.cu:73-75)
// becomes
//     filterKernelBaseline(frame, params);                             (geometry chosen inside)
// Errors throw std::runtime_error, which the harness reports as "Fail with ..." like the
// reference's harness does for its own exceptions (src/test.cu:40-42).
#ifndef RMD_FILTER_H
#define RMD_FILTER_H

#include <cstddef>

#include "extended_math.h"
#include "gbuffer.h"
#include "image.h"
#include "utils.h"
#include "vector.h"

struct FilterParams {
    enum FilterType { AVERAGE, GAUSSIAN, CROSS, WAVELET } type;
    int depth;
    int level;
    int radius;
    float sigmaSpace;
    float sigmaColor;
    float sigmaAlbedo;
    float sigmaNormal;

    bool cacheInput = true;
    bool cacheBuffer = true;
};

static_assert(sizeof(FilterParams) == sizeof(rmd_filter_params) && sizeof(FilterParams) == 36, "FilterParams must match rmd_filter_params");
static_assert(offsetof(FilterParams, radius) == 12 && offsetof(FilterParams, sigmaNormal) == 28 &&
              offsetof(FilterParams, cacheInput) == 32 && offsetof(FilterParams, cacheBuffer) == 33,
              "FilterParams offsets must match the reference");

inline rmd_filter_params toAbi(const FilterParams& p)
{
    rmd_filter_params r;
    r.type = (int)p.type; r.depth = p.depth; r.level = p.level; r.radius = p.radius;
    r.sigmaSpace = p.sigmaSpace; r.sigmaColor = p.sigmaColor; r.sigmaAlbedo = p.sigmaAlbedo; r.sigmaNormal = p.sigmaNormal;
    r.cacheInput = p.cacheInput; r.cacheBuffer = p.cacheBuffer;
    return r;
}

inline void filterKernelBaseline(GBuffer frame, const FilterParams params, void* stream = nullptr)
{
    rmdCheck(rmd_filter_baseline(toAbi(frame), toAbi(params), stream), "filterKernelBaseline");
}

inline void filterKernelTiled(GBuffer frame, const FilterParams params, void* stream = nullptr)
{
    rmdCheck(rmd_filter_tiled(toAbi(frame), toAbi(params), stream), "filterKernelTiled");
}

#endif
