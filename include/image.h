// image.h — host image struct, field-for-field the reference's (include/image.h:16-75:
// int3 shape {width, height, channels} @0, byte* data @16; 24 bytes).  PNG I/O is a small
// zlib-based codec (raymarchdenoisercuda_amd/host/image.cpp) instead of the vendored stb:
// any conforming decoder yields the same bytes.  Load/save failures throw std::runtime_error
// as in the reference (src/image.cpp:38-39,43-51).
#ifndef RMD_IMAGE_H
#define RMD_IMAGE_H

#include <cstddef>
#include <string>

#include "utils.h"

struct Image {
    int3 shape;    // width, height, channels
    byte* data;    // shape.x * shape.y * shape.z bytes, row-major, interleaved

    Image();
    explicit Image(int3 shape);                 // allocates (zero-filled)
    Image(byte* data, int3 shape);              // deep copy (the reference aliases and later double-frees, src/image.cpp:27-31)
    Image(std::string filename, int channels);  // PNG -> `channels` bytes per pixel (1..4), alpha = 255 when added
    Image(const Image&) = delete;
    Image& operator=(const Image&) = delete;
    Image(Image&& o) noexcept;
    Image& operator=(Image&& o) noexcept;
    ~Image();

    void save(std::string filename);
    static void save(std::string filename, byte* data, int3 shape);
};

static_assert(sizeof(Image) == 24 && offsetof(Image, data) == 16, "Image layout must match the reference");

#endif
