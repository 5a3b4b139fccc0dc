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

// Launch-heuristic tuning / A-B switches (tile configurations, workgroup targets, ring depths, measurement forms of a kernel) exist only in
// TUNING builds of the library (`make -C csrc variant VFILE=<file> VFLAGS=-DTDR_TUNING_KNOBS VOUT=...`): the shipped library reads none of
// them -- every call returns "unset" and the heuristic's default stands.  tdr_tuning_build() tells a caller which kind it has loaded.
#include <stdlib.h>
#ifdef TDR_TUNING_KNOBS
static inline const char* tdr_tune_env(const char* name) { return getenv(name); }
#else
static inline const char* tdr_tune_env(const char*) { return nullptr; }
#endif

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// wave sum on the VALU only (DPP lane permutes inside each row of 16, then one v_readlane per row): no LDS traffic
// and short dependency chains, for kernels that reduce many values per wave.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_perm<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_perm<0x141>(v);    // row_half_mirror
    v += dpp_perm<0x140>(v);    // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 2-way fp16 split of two fp32 values: h = rn_f16(x) packed by v_cvt_pk_f16_f32, m = rn_f16(x - h) straight from the packed heads
// with v_fma_mixlo / mixhi_f16 (f16 source x -1 + fp32 source: x - h is exact in fp32, so the one rounding of the mix instruction
// IS rn_f16(x - h)) -- 1.5 VALU per value where convert / convert back / subtract / convert costs 3.5 - 4, same bits.  The values
// are pinned in VGPRs first: head and residual must start from the SAME fp32 value (a product left free may be rounded from its
// exact form for one and from the rounded one for the other, tdr_nafblock.hip).
typedef float tdr_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 tdr_f16x2 __attribute__((ext_vector_type(2)));
// PIN = false only for values that come straight from a load (nothing the compiler could fold into the conversion): the pin costs register copies
template <bool PIN = true>
__device__ __forceinline__ void tdr_split2_f16(float x0, float x1, unsigned& h, unsigned& m) {
    if constexpr (PIN) asm volatile("" : "+v"(x0), "+v"(x1));
    const tdr_f32x2 xv = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, tdr_f16x2));
    const float neg1 = -1.0f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(m) : "v"(h), "v"(neg1), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(h), "v"(neg1), "v"(x1));
}

// 3-way bf16 split of two fp32 values: h = rn_bf16(x), m = rn_bf16(x - h), l = rn_bf16(x - h - m), packed pairwise (x0 in the low
// half).  The same bits as the scalar split3 of tdr_conv_bx3.hip / tdr_pack.h (round-to-nearest-even conversions, exact fp32
// subtractions): v_cvt_pk_bf16_f32 packs a plane, a shift / mask gives the plane back as fp32 -- 11 VALU per pair.  x = h + m + l
// carries 24+ significant bits for any fp32 exponent (no fp16 window, no pre-scale).
typedef __bf16 tdr_bf16x2 __attribute__((ext_vector_type(2)));
template <bool PIN = true>
__device__ __forceinline__ void tdr_split3_bf16(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    if constexpr (PIN) asm volatile("" : "+v"(x0), "+v"(x1));
    const tdr_f32x2 xv = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(xv, tdr_bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const tdr_f32x2 rv = {r0, r1};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, tdr_bf16x2));
    const tdr_f32x2 sv = {r0 - __builtin_bit_cast(float, m << 16), r1 - __builtin_bit_cast(float, m & 0xffff0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(sv, tdr_bf16x2));
}
