// gbuffer.cpp — CudaGBuffer, which the reference declares (include/gbuffer.h:20-33) but never
// defines: device planes for one frame + upload of the render/albedo/normal PNGs.
#include "gbuffer.h"

#include <cstdlib>
#include <cstring>

CudaGBuffer::CudaGBuffer() : GBuffer{} {}

CudaGBuffer::CudaGBuffer(int2 s) : GBuffer{} { allocate(s); }

CudaGBuffer::~CudaGBuffer()
{
    if (uploadDone) { rmd_event_synchronize(uploadDone); rmd_event_destroy(uploadDone); }
    if (stage) rmd_host_free_pinned(stage);
    free(denoisedCPU);
}

void CudaGBuffer::waitUpload()
{
    if (uploadDone) rmdCheck(rmd_event_synchronize(uploadDone), "CudaGBuffer::waitUpload");
}

void CudaGBuffer::allocate(int2 s)
{
    const size_t n = (size_t)s.x * s.y;
    shape = s;
    renderVec.resize(n); albedoVec.resize(n); normalVec.resize(n); denoisedVec.resize(n);
    bufferVec.resize(2 * n);
    render = renderVec.data(); albedo = albedoVec.data(); normal = normalVec.data(); denoised = denoisedVec.data();
    buffer[0] = bufferVec.data(); buffer[1] = bufferVec.data() + n;
    free(denoisedCPU);
    denoisedCPU = nullptr;
}

void CudaGBuffer::openImages(std::string filepath, void* stream)
{
    if (!filepath.empty() && filepath.back() != '/') filepath += '/';
    Image r(filepath + "render.png", 4), a(filepath + "albedo.png", 4), n(filepath + "normal.png", 4);
    if (a.shape.x != r.shape.x || a.shape.y != r.shape.y || n.shape.x != r.shape.x || n.shape.y != r.shape.y)
        throw std::runtime_error("CudaGBuffer::openImages: planes in '" + filepath + "' differ in size");
    // first: the previous asynchronous upload may still read the staging buffer and write the device planes that
    // allocate() is about to free
    waitUpload();
    if (shape.x != r.shape.x || shape.y != r.shape.y) allocate(int2{ r.shape.x, r.shape.y });
    const size_t bytes = (size_t)shape.x * shape.y * 4;
    if (stageBytes < 3 * bytes) {
        if (stage) rmd_host_free_pinned(stage);
        stage = nullptr; stageBytes = 0;
        rmdCheck(rmd_host_alloc_pinned((void**)&stage, 3 * bytes), "openImages(pinned staging)");
        stageBytes = 3 * bytes;
    }
    if (!uploadDone) rmdCheck(rmd_event_create(&uploadDone), "openImages(event)");
    // decoded pixels -> pinned staging (the Images die at scope exit, the staging lives with the object)
    memcpy(stage, r.data, bytes);
    memcpy(stage + bytes, a.data, bytes);
    memcpy(stage + 2 * bytes, n.data, bytes);
    rmdCheck(rmd_memcpy_h2d_async(render, stage, bytes, stream), "openImages(render)");
    rmdCheck(rmd_memcpy_h2d_async(albedo, stage + bytes, bytes, stream), "openImages(albedo)");
    rmdCheck(rmd_memcpy_h2d_async(normal, stage + 2 * bytes, bytes, stream), "openImages(normal)");
    rmdCheck(rmd_event_record(uploadDone, stream), "openImages(record)");      // no blocking sync here
}

uchar4* CudaGBuffer::download()
{
    const size_t n = (size_t)shape.x * shape.y;
    if (!denoisedCPU) denoisedCPU = (uchar4*)malloc(n * sizeof(uchar4));
    rmdCheck(rmd_memcpy_d2h(denoisedCPU, denoised, n * sizeof(uchar4)), "CudaGBuffer::download");
    return denoisedCPU;
}
