// floor(x / d) without the ~20-instruction hardware-assisted division sequence, for divisors that are kernel arguments (T, T2, F: the same for every lane):
// the host passes m = ceil(2^32 / d) = (2^32 + e) / d with 0 <= e < d; umulhi(x, m) = floor(x / d + x e / (d 2^32)) and the second term is < x / 2^32 < 1, so the
// estimate is never low and at most one high for every 32-bit x, and one compare-and-decrement finishes it (q d does not wrap for x + d < 2^32: indices here are < 2^27).
#pragma once
#include <hip/hip_runtime.h>

namespace rtfs {
inline unsigned div_magic_of(int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }
__device__ __forceinline__ int div_magic(unsigned x, unsigned d, unsigned m) {
    const unsigned q = __umulhi(x, m);
    return (int)(q * d > x ? q - 1 : q);
}
}  // namespace rtfs
