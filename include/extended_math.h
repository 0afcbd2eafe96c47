// extended_math.h — the three index helpers the hot path uses in the reference
// (include/extended_math.h:54-68), for host code.  Its uchar3/float3 helpers are dead code in the
// reference (and its uchar3 operator- adds, :6-8) and are not carried over.
#ifndef RMD_EXTENDED_MATH_H
#define RMD_EXTENDED_MATH_H

#include "utils.h"

inline int totalSize(int2 shape) { return shape.x * shape.y; }
inline int totalSize(int3 shape) { return shape.x * shape.y * shape.z; }
inline int inRange(int2 pos, int2 shape) { return pos.x >= 0 && pos.x < shape.x && pos.y >= 0 && pos.y < shape.y; }
inline int flattenIndex(int2 p, int2 shape) { return p.y * shape.x + p.x; }

#endif
