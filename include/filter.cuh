// filter.cuh — kept so that `#include "filter.cuh"` (reference src/test.cu:2, src/filter.cu:1)
// still resolves; everything lives in filter.h.
#include "filter.h"
