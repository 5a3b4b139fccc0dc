// Shared helpers for the tdr HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define TDR_OK 0
#define TDR_ERR_ARG (-1)
#define TDR_ERR_HIP (-2)
#define TDR_ERR_UNSUPPORTED (-3)

void tdr_set_error(const char* fmt, ...);

#define TDR_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) {                             \
            tdr_set_error(__VA_ARGS__);            \
            return TDR_ERR_ARG;                    \
        }                                          \
    } while (0)

#define TDR_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            tdr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
            return TDR_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

struct TdrConvDesc;
int tdr_conv_forward_bx3(const TdrConvDesc* d, void* stream);   // tdr_conv_bx3.hip

static inline int tdr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
