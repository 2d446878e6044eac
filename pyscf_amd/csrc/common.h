// Shared helpers for the gfx950 kernels of libpyscf_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

namespace pamd {

// Last error message (thread local), readable through PAMD_last_error().
extern thread_local char g_errmsg[512];

inline int set_error(int code, const char *what, const char *file, int line)
{
    snprintf(g_errmsg, sizeof(g_errmsg), "%s (%s:%d)", what, file, line);
    return code;
}

#define PAMD_CHECK_HIP(expr)                                                            \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return pamd::set_error(-(int)_e - 1000, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define PAMD_CHECK_LAUNCH() PAMD_CHECK_HIP(hipGetLastError())

#define PAMD_REQUIRE(cond, msg)                                                         \
    do {                                                                                \
        if (!(cond)) return pamd::set_error(-1, msg, __FILE__, __LINE__);               \
    } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// v_mfma_f64_16x16x4_f64:  D(16x16) += A(16x4) * B(4x16)
//   A operand: lane l holds A[m = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][n = l & 15]
//   D operand: lane l, reg r holds D[m = (l >> 4) + 4 r][n = l & 15]
__device__ __forceinline__ double4_t mfma_f64_16x16x4(double a, double b, double4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace pamd
