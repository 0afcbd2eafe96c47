// utils.h — host utilities of the drop-in (role of reference include/utils.h + src/utils.cpp).
// Host code never sees HIP device syntax: kernels live in librmd.so behind include/rmd_api.h.
#ifndef RMD_UTILS_H
#define RMD_UTILS_H

#include <hip/hip_vector_types.h>   // int2 / int3 / uchar4 / float4 PODs (same size/alignment as CUDA's)
#include <chrono>
#include <iostream>
#include <stdexcept>
#include <string>

#include "rmd_api.h"

typedef unsigned char byte;          // reference include/utils.h:15

// The reference declares CHECK_CUDA for the driver API and never uses it (include/utils.h:17-24);
// here every C-ABI call goes through rmdCheck so failures reach the harness as
// std::runtime_error, the only error channel it understands (reference src/test.cu:40-42).
inline void rmdCheck(int rc, const char* what = "librmd")
{
    if (rc != 0)
        throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + rmd_last_error_string());
}

// reference src/utils.cpp:5-15
inline void printGPUProperties() { rmdCheck(rmd_print_device_properties(), "printGPUProperties"); }

#endif
