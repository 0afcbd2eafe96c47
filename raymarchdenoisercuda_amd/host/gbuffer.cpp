// gbuffer.cpp — CudaGBuffer, which the reference declares (include/gbuffer.h:20-33) but never
// defines: device planes for one frame + upload of the render/albedo/normal PNGs.
#include "gbuffer.h"

#include <cstdlib>
#include <cstring>

CudaGBuffer::CudaGBuffer() : GBuffer{} {}

CudaGBuffer::CudaGBuffer(int2 s) : GBuffer{} { allocate(s); }

CudaGBuffer::~CudaGBuffer() { free(denoisedCPU); }

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
    if (shape.x != r.shape.x || shape.y != r.shape.y) allocate(int2{ r.shape.x, r.shape.y });
    const size_t bytes = (size_t)shape.x * shape.y * 4;
    rmdCheck(rmd_memcpy_h2d_async(render, r.data, bytes, stream), "openImages(render)");
    rmdCheck(rmd_memcpy_h2d_async(albedo, a.data, bytes, stream), "openImages(albedo)");
    rmdCheck(rmd_memcpy_h2d_async(normal, n.data, bytes, stream), "openImages(normal)");
    rmdCheck(rmd_stream_sync(stream), "openImages(sync)");   // the Images die at scope exit
}

uchar4* CudaGBuffer::download()
{
    const size_t n = (size_t)shape.x * shape.y;
    if (!denoisedCPU) denoisedCPU = (uchar4*)malloc(n * sizeof(uchar4));
    rmdCheck(rmd_memcpy_d2h(denoisedCPU, denoised, n * sizeof(uchar4)), "CudaGBuffer::download");
    return denoisedCPU;
}
