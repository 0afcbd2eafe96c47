// gbuffer.h — frame descriptor structs of the reference (include/gbuffer.h:6-33).
// GBuffer keeps the reference layout (56 bytes: shape@0 render@8 denoised@16 normal@24
// albedo@32 buffer@40) and is what the filter launchers take by value.  CudaGBuffer is declared
// but never defined in the reference (:20-33); it is implemented here
// (raymarchdenoisercuda_amd/host/gbuffer.cpp): device planes + asynchronous upload of
// render/<scene>/<frame>/{render,albedo,normal}.png through pinned host staging.
#ifndef RMD_GBUFFER_H
#define RMD_GBUFFER_H

#include <cstddef>
#include <string>

#include "image.h"
#include "vector.h"

struct GBuffer {
    int2 shape;

    uchar4* render;
    uchar4* denoised;
    uchar4* normal;
    uchar4* albedo;
    uchar4* buffer[2];
};

static_assert(sizeof(GBuffer) == sizeof(rmd_gbuffer) && sizeof(GBuffer) == 56, "GBuffer must match rmd_gbuffer");
static_assert(offsetof(GBuffer, render) == 8 && offsetof(GBuffer, denoised) == 16 && offsetof(GBuffer, normal) == 24 &&
              offsetof(GBuffer, albedo) == 32 && offsetof(GBuffer, buffer) == 40, "GBuffer offsets must match the reference");

inline rmd_gbuffer toAbi(const GBuffer& g)
{
    rmd_gbuffer r;
    r.shape.x = g.shape.x; r.shape.y = g.shape.y;
    r.render = (rmd_uchar4*)g.render; r.denoised = (rmd_uchar4*)g.denoised;
    r.normal = (rmd_uchar4*)g.normal; r.albedo = (rmd_uchar4*)g.albedo;
    r.buffer[0] = (rmd_uchar4*)g.buffer[0]; r.buffer[1] = (rmd_uchar4*)g.buffer[1];
    return r;
}

// reference include/gbuffer.h:16-18: the host-side Images shadow the device pointers of the base struct
struct CPUGBuffer : GBuffer {
    Image render, albedo, normal;
};

struct CudaGBuffer : GBuffer {
    CudaVector<uchar4> renderVec, albedoVec, normalVec, denoisedVec;
    CudaVector<uchar4> bufferVec;             // both ping-pong planes, back to back
    uchar4* denoisedCPU = nullptr;            // host copy filled by download()
    byte* stage = nullptr;                    // pinned host staging of the three input planes (3 x W x H x 4 bytes)
    size_t stageBytes = 0;
    void* uploadDone = nullptr;               // event recorded behind the last upload

    CudaGBuffer();
    explicit CudaGBuffer(int2 shape);
    ~CudaGBuffer();
    CudaGBuffer(const CudaGBuffer&) = delete;
    CudaGBuffer& operator=(const CudaGBuffer&) = delete;

    void allocate(int2 shape);
    // Loads <filepath>/render.png, albedo.png, normal.png (RGBA8) and uploads them ASYNCHRONOUSLY on `stream`
    // (the hook the reference declares at include/gbuffer.h:32 takes a stream for exactly this): the decoded
    // pixels go through pinned staging owned by this object, the call returns once the copies are queued,
    // work queued on `stream` afterwards sees the planes.  waitUpload() blocks until they have landed; the
    // next openImages() and the destructor wait by themselves.
    void openImages(std::string filepath, void* stream = nullptr);
    void waitUpload();
    // copies `denoised` back to denoisedCPU (allocated on first use) and returns it
    uchar4* download();
};

#endif
