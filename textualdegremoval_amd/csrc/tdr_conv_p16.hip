// 3x3 / stride 1 / pad 1 implicit-GEMM convolution on PRE-SPLIT operands ("P16" tensors), gfx950 only.
//
// The fp16-split arithmetic of tdr_conv_bx3.hip (x = h + m, h = rn_f16(x), m = rn_f16(x - h); a*b ~ am*bh + ah*bm + ah*bh on
// v_mfma_f32_32x32x16_f16, fp32 accumulation) re-did the split in every consumer: dword loads per (lane, channel), ~8 VALU per
// element, VGPR staging across the MFMA block.  Here the PRODUCER writes the pair once, in the layout the matrix pipe consumes:
//
//   P16 tensor of an fp32 [N][C][H][W] activation (C % 16 == 0):  uint4 slots  [N][C/8][plane 2][H+2][W+2]
//   slot = 8 consecutive channels of ONE pixel as 8 x f16; plane 0 = heads h, plane 1 = residuals m; the one-pixel border is zero
//   (it IS the convolution's zero padding).  Bytes per element: 4, the same as the fp32 value the pair encodes.
//   An element that is EXACTLY zero is stored with the head -0.0 (p16_head): "x > 0" is then the head's sign bit being clear even for
//   0 < x < 2^-25, whose head rounds to +0 -- the ReLU masks of the backward pass are read from the pair planes (p16_positive), and
//   with a plain fp16(x) > 0 test a few such elements per tensor lost their gradient term (profiles/r4/diag_p16_seq.log).
//
// With that layout a B fragment (lane = pixel, 8 channels) is one 16-byte slot, a halo-tile row is LW contiguous slots in
// memory, and BOTH operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU): the halo tile of the next
// 16-channel group in NB pieces spread over the taps of the current group, the packed weight fragments of tap s + LA (the hx2
// pack of tdr_pack_weights_hx2 is already lane-linear: one 1 KiB piece per (m-tile, plane)) into a ring of LA + 1 slots.
// The main loop has no global loads the compiler knows of: waits are counted by hand (s_waitcnt vmcnt(N) with N = the pieces
// issued after the one the next step needs, never 0) in front of one raw s_barrier per (group, tap) step.
//
// Accumulation order per output element is the one of conv_bx3_kernel<..., SCH_HX2>: groups of 16 channels ascending, taps
// ascending, products mh, hm, hh -- results are bit-identical to that kernel on the same operands.
//
// Replaces (reference): the 3x3 convolutions of the MASA encoder's ResidualBlocks, forward and data gradient
// (models/archs/network_nafnet_guided_arch.py:44-59,110-143 and their autograd).
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include "tdr_common.h"
#include "../../include/tdr.h"

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

namespace {

// Two plane formats share the slot layout [N][C/8][plane NS][H+2][W+2] (16-byte slots of 8 channels of one pixel):
//   PF_PAIR (wp_fmt 2): NS = 2 fp16 planes (h, m), products mh hm hh -- the fp16-window format described above (TDR_MATH=hx2);
//   PF_TRI  (wp_fmt 1): NS = 3 bf16 planes (h, m, l), products lh hl mm mh hm hh (v_mfma_f32_32x32x16_bf16) -- the reference-
//            arithmetic format (TDR_MATH=bx3).  h = rn_bf16(x), m = rn_bf16(x - h), l = rn_bf16(x - h - m): 8 + 8 + 8 significand
//            bits with fp32's exponent, so for every normal fp32 x the three planes hold x EXACTLY (h + m + l == x): a triple-plane
//            tensor is the fp32 tensor at 6 bytes per element, no window, no loss scale, and the forward residual stream can
//            live in it.  Same sign convention for exact zeros (head -0.0).  Results are bit-identical to
//            conv_bx3_kernel<..., SCH_BX3> on the fp32 tensors (same products, same accumulation order).
enum { PF_TRI = 1, PF_PAIR = 2 };
template <int PF> struct PlaneFmt { static constexpr int NS = PF == PF_TRI ? 3 : 2; };

// head plane value of x: rn_f16(x), except that an exact zero becomes -0.0 (see the header: the sign bit of the head is "x <= 0")
__device__ __forceinline__ _Float16 p16_head(float x) {
    const _Float16 h = (_Float16)x;
    return x == 0.f ? __builtin_bit_cast(_Float16, (unsigned short)0x8000) : h;
}
__device__ __forceinline__ bool p16_positive(_Float16 head) { return (__builtin_bit_cast(unsigned short, head) & 0x8000u) == 0; }
// four fp32 values -> their head (p16_head) and residual slots halves: tdr_split2_f16 pairs + the sign bit of exact zeros
__device__ __forceinline__ void p16_split4(float x0, float x1, float x2, float x3, uint2& h, uint2& m) {
    unsigned h0, h1, m0, m1;
    tdr_split2_f16(x0, x1, h0, m0);
    tdr_split2_f16(x2, x3, h1, m1);
    h0 |= (x0 == 0.f ? 0x8000u : 0u) | (x1 == 0.f ? 0x80000000u : 0u);     // +-0 -> -0.0
    h1 |= (x2 == 0.f ? 0x8000u : 0u) | (x3 == 0.f ? 0x80000000u : 0u);
    h = make_uint2(h0, h1);
    m = make_uint2(m0, m1);
}

// the triple-plane counterpart: tdr_split3_bf16 pairs + the sign bit of exact zeros
__device__ __forceinline__ void p24_split4(float x0, float x1, float x2, float x3, uint2& h, uint2& m, uint2& l) {
    unsigned h0, h1, m0, m1, l0, l1;
    tdr_split3_bf16(x0, x1, h0, m0, l0);
    tdr_split3_bf16(x2, x3, h1, m1, l1);
    h0 |= (x0 == 0.f ? 0x8000u : 0u) | (x1 == 0.f ? 0x80000000u : 0u);     // +-0 -> -0.0
    h1 |= (x2 == 0.f ? 0x8000u : 0u) | (x3 == 0.f ? 0x80000000u : 0u);
    h = make_uint2(h0, h1);
    m = make_uint2(m0, m1);
    l = make_uint2(l0, l1);
}
// four packed 16-bit plane elements (a uint2) -> fp32
template <int PF>
__device__ __forceinline__ void plane4_to_f32(uint2 v, float (&o)[4]) {
    if constexpr (PF == PF_TRI) {
        o[0] = __builtin_bit_cast(float, v.x << 16); o[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
        o[2] = __builtin_bit_cast(float, v.y << 16); o[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
    } else {
        const pf16x4 h = __builtin_bit_cast(pf16x4, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (float)h[e];
    }
}

struct P16Args {
    const uint4* in; long in_ns;       // P16 input, slots per image
    int Cin, H, W, Hp, Wp;
    const uint4* wp; int MT;           // hx2 weight pack [group][tap][m-tile][plane][lane]
    int Cout;
    const float* bias;
    const float* res32; long res32_ns;
    const uint4* res16; long res16_ns;
    const float* mask32; long mask32_ns;
    const uint4* mask16; long mask16_ns;
    int relu;
    float* out32; long out32_ns;
    uint4* out16; long out16_ns;
    int tiles_x, tiles_y, mtiles;
};

#define P16_GLDS(gptr, lptr)                                                                          \
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(gptr),          \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// counted wait + workgroup barrier in ONE asm statement: the compiler neither sees the LDS-DMA pieces in its own vmcnt
// bookkeeping nor may it move LDS accesses across this point ("memory").  lgkmcnt(0): this wave's fragment reads of the step
// are complete (they were consumed by the MFMAs above) before any wave may overwrite their ring slot.
template <int N>
__device__ __forceinline__ void wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- epilogue shared by the kernels of this file (accumulator layout: lane (j, kk) holds pixel j of row tn, channels
// mb + (r&3) + 8*(r>>2), mb = .. + 4*kk): bias, residual (fp32 tensor or pair planes), ReLU, mask (fp32 tensor or the sign bit of
// the head plane), then the fp32 NCHW store and / or the pair-plane store incl. the zero border of the output tensor.
template <int TM, int TN, int PF = PF_PAIR>
__device__ __forceinline__ void p16_epilogue(const P16Args& a, f32x16 (&acc)[TM][TN], int n, int m0, int wm, int wn, int oy0, int ox0, int j, int kk) {
    constexpr int NS = PlaneFmt<PF>::NS;
    const long PS = (long)a.Hp * a.Wp;
    const long HW = (long)a.H * a.W;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int oy = oy0 + wn * TN + tn, ox = ox0 + j;
        const bool pvalid = oy < a.H && ox < a.W;
        const long pix = pvalid ? (long)oy * a.W + ox : 0;
        const long pslot = pvalid ? (long)(oy + 1) * a.Wp + ox + 1 : 0;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            __builtin_amdgcn_sched_barrier(0);     // one tile at a time: hoisting every tile's operand loads spills
            const int mt0 = m0 + (wm * TM + tm) * 32;
            const int mb = mt0 + 4 * kk;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[tm][tn][r];
            if (a.bias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += a.bias[min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
            }
            if (a.res32) {
                float tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tv[r] = a.res32[(long)n * a.res32_ns + (long)min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1) * HW + pix];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += tv[r];
            }
            if (a.res16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oc = min((mt0 >> 3) + q, (a.Cout >> 3) - 1);
                    const char* p = reinterpret_cast<const char*>(a.res16 + (long)n * a.res16_ns + (long)oc * NS * PS + pslot) + kk * 8;
                    float h[4], m[4];
                    plane4_to_f32<PF>(*reinterpret_cast<const uint2*>(p), h);
                    plane4_to_f32<PF>(*reinterpret_cast<const uint2*>(p + PS * 16), m);
                    if constexpr (PF == PF_TRI) {
                        float l[4];
                        plane4_to_f32<PF>(*reinterpret_cast<const uint2*>(p + 2 * PS * 16), l);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q + e] += (h[e] + m[e]) + l[e];      // the planes' sum is the stored fp32 value, exactly
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 * q + e] += h[e] + m[e];
                    }
                }
            }
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (a.mask32) {
                float tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tv[r] = a.mask32[(long)n * a.mask32_ns + (long)min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1) * HW + pix];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = tv[r] > 0.f ? v[r] : 0.f;
            }
            if (a.mask16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oc = min((mt0 >> 3) + q, (a.Cout >> 3) - 1);
                    const char* p = reinterpret_cast<const char*>(a.mask16 + (long)n * a.mask16_ns + (long)oc * NS * PS + pslot) + kk * 8;
                    const pf16x4 h = *reinterpret_cast<const pf16x4*>(p);         // (sign bits only: the same test for both formats)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * q + e] = p16_positive(h[e]) ? v[4 * q + e] : 0.f;
                }
            }
            if (a.out32) {
                float* op = a.out32 + (long)n * a.out32_ns + pix;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (pvalid && m < a.Cout) op[(long)m * HW] = v[r];
                }
            }
            if (a.out16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!pvalid || mt0 + 8 * q >= a.Cout) continue;
                    uint2 pl[NS];                                // every plane from the same fp32 value (tdr_split2_f16 / tdr_split3_bf16)
                    if constexpr (PF == PF_TRI) p24_split4(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], pl[0], pl[1], pl[2]);
                    else p16_split4(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], pl[0], pl[1]);
                    char* base = reinterpret_cast<char*>(a.out16 + (long)n * a.out16_ns + (long)((mt0 >> 3) + q) * NS * PS) + kk * 8;
#pragma unroll
                    for (int s = 0; s < NS; ++s) *reinterpret_cast<uint2*>(base + (s * PS + pslot) * 16) = pl[s];
                    // the zero border of the output tensor is written by the tiles that touch it
                    const pf16x4 z = {0, 0, 0, 0};
                    const bool top = oy == 0, bot = oy == a.H - 1, lef = ox == 0, rig = ox == a.W - 1;
                    if (top | bot | lef | rig) {
                        const long Wp = a.Wp;
                        auto zero_at = [&](long sl) {
#pragma unroll
                            for (int s = 0; s < NS; ++s) *reinterpret_cast<pf16x4*>(base + (s * PS + sl) * 16) = z;
                        };
                        if (top) zero_at(ox + 1);
                        if (bot) zero_at((long)(a.H + 1) * Wp + ox + 1);
                        if (lef) zero_at((long)(oy + 1) * Wp);
                        if (rig) zero_at((long)(oy + 1) * Wp + a.W + 1);
                        if (top && lef) zero_at(0);
                        if (top && rig) zero_at(a.W + 1);
                        if (bot && lef) zero_at((long)(a.H + 1) * Wp);
                        if (bot && rig) zero_at((long)(a.H + 1) * Wp + a.W + 1);
                    }
                }
            }
        }
    }
}

// the same epilogue one half-octet at a time (a handful of live registers): for the thin-level kernels, which keep the weight
// fragments of all steps resident; 5 - 10 % slower on the main kernel (less memory-level parallelism), hence separate
template <int TM, int TN>
__device__ __forceinline__ void p16_epilogue_lean(const P16Args& a, f32x16 (&acc)[TM][TN], int n, int m0, int wm, int wn, int oy0, int ox0, int j, int kk) {
    const long PS = (long)a.Hp * a.Wp;
    const long HW = (long)a.H * a.W;
    // One half-octet (the 4 consecutive channels mt0 + 8q + 4kk .. of this lane) at a time, fenced from the next: a handful of live
    // registers.  Hoisting the operand loads of a whole 32 x 32 tile (or of every tile) spilled hundreds of bytes per lane in the
    // kernels that keep large accumulator / weight sets resident.
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int oy = oy0 + wn * TN + tn, ox = ox0 + j;
        const bool pvalid = oy < a.H && ox < a.W;
        const long pix = pvalid ? (long)oy * a.W + ox : 0;
        const long pslot = pvalid ? (long)(oy + 1) * a.Wp + ox + 1 : 0;
        const bool top = oy == 0, bot = oy == a.H - 1, lef = ox == 0, rig = ox == a.W - 1;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mt0 = m0 + (wm * TM + tm) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_barrier(0);
                const int c0 = mt0 + 8 * q + 4 * kk;                          // first of the lane's 4 channels
                const int oc = min((mt0 >> 3) + q, (a.Cout >> 3) - 1);        // its octet (pair-plane operands)
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[tm][tn][4 * q + e];
                if (a.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += a.bias[min(c0 + e, a.Cout - 1)];
                }
                if (a.res32) {
                    float tv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) tv[e] = a.res32[(long)n * a.res32_ns + (long)min(c0 + e, a.Cout - 1) * HW + pix];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += tv[e];
                }
                if (a.res16) {
                    const char* p = reinterpret_cast<const char*>(a.res16 + (long)n * a.res16_ns + (long)oc * 2 * PS + pslot) + kk * 8;
                    const pf16x4 h = *reinterpret_cast<const pf16x4*>(p);
                    const pf16x4 m = *reinterpret_cast<const pf16x4*>(p + PS * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)h[e] + (float)m[e];
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (a.mask32) {
                    float tv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) tv[e] = a.mask32[(long)n * a.mask32_ns + (long)min(c0 + e, a.Cout - 1) * HW + pix];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = tv[e] > 0.f ? v[e] : 0.f;
                }
                if (a.mask16) {
                    const char* p = reinterpret_cast<const char*>(a.mask16 + (long)n * a.mask16_ns + (long)oc * 2 * PS + pslot) + kk * 8;
                    const pf16x4 h = *reinterpret_cast<const pf16x4*>(p);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = p16_positive(h[e]) ? v[e] : 0.f;
                }
                if (a.out32) {
                    float* op = a.out32 + (long)n * a.out32_ns + pix;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (pvalid && c0 + e < a.Cout) op[(long)(c0 + e) * HW] = v[e];
                }
                if (a.out16 && pvalid && mt0 + 8 * q < a.Cout) {
                    uint2 h, m;                                  // head and residual from the same fp32 value (tdr_split2_f16)
                    p16_split4(v[0], v[1], v[2], v[3], h, m);
                    char* base = reinterpret_cast<char*>(a.out16 + (long)n * a.out16_ns + (long)((mt0 >> 3) + q) * 2 * PS) + kk * 8;
                    *reinterpret_cast<uint2*>(base + pslot * 16) = h;
                    *reinterpret_cast<uint2*>(base + (PS + pslot) * 16) = m;
                    // the zero border of the output tensor is written by the tiles that touch it
                    if (top | bot | lef | rig) {
                        const pf16x4 z = {0, 0, 0, 0};
                        const long Wp = a.Wp;
                        auto zero_at = [&](long sl) {
                            *reinterpret_cast<pf16x4*>(base + sl * 16) = z;
                            *reinterpret_cast<pf16x4*>(base + (PS + sl) * 16) = z;
                        };
                        if (top) zero_at(ox + 1);
                        if (bot) zero_at((long)(a.H + 1) * Wp + ox + 1);
                        if (lef) zero_at((long)(oy + 1) * Wp);
                        if (rig) zero_at((long)(oy + 1) * Wp + a.W + 1);
                        if (top && lef) zero_at(0);
                        if (top && rig) zero_at(a.W + 1);
                        if (bot && lef) zero_at((long)(a.H + 1) * Wp);
                        if (bot && rig) zero_at((long)(a.H + 1) * Wp + a.W + 1);
                    }
                }
            }
        }
    }
}

// Workgroup = NW = WM x WN waves (4 or 8), tile = BM output channels x TH rows x 32 columns; wave (wm, wn) owns m-tiles wm*TM..
// and rows wn*TN..  (a 32-pixel sub-tile is one image row segment, so every tap shift of a B fragment is a contiguous run of
// slots).  PIPE: the fragments of step s + 1 are read from LDS while the MFMAs of step s run (register double buffer), so a
// wave keeps the matrix pipe busy on its own; the weight ring is then LA + 1 = 4 slots with a run-time slot index.
// PF: plane format of the input / weight pack (PF_PAIR: 2 fp16 planes, 3 products; PF_TRI: 3 bf16 planes, 6 products).
template <int TM, int TN, int WM, int WN, bool PIPE, int ABL = 0, bool ILV = false, int PF = PF_PAIR>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8) ? 1 : 2) void conv3x3_p16_kernel(P16Args a) {
    constexpr int NS = PlaneFmt<PF>::NS;
    constexpr int NPR = PF == PF_TRI ? 6 : 3;              // matrix products per fp32 product
    constexpr int NW = WM * WN, NT = 64 * NW;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int BM = 32 * TM * WM, TH = TN * WN;
    constexpr int LH = TH + 2, LW = 34, TS = LH * LW;      // halo tile slots per (octet, plane)
    constexpr int NBW = (2 * NS * TS + NT - 1) / NT;       // halo pieces per wave per group (2 octets x NS planes)
    constexpr int BREG = NBW * NT;                         // slots per halo buffer (padded to whole pieces)
    constexpr int NAI = (BM / 32) * NS;                    // weight pieces per step
    constexpr int NAW = (NAI + NW - 1) / NW;               // per wave (the first NAI % NW waves issue one more than the others)
    constexpr int NAX = NAI % NW;                          // 0: every wave issues NAW pieces
    constexpr int LA = PIPE ? 3 : 2;
    constexpr int R = LA + 1;
    // halo pieces must be issued early enough to be covered by the wait that publishes them: taps 0 .. NTB-1, HPT pieces per tap
    constexpr int NTB = PIPE ? 6 : 7;
    constexpr int HPT = (NBW + NTB - 1) / NTB;

    extern __shared__ __attribute__((aligned(1024))) uint4 smem4[];
    uint4* sB = smem4;                   // [2][BREG]
    uint4* sA = smem4 + 2 * BREG;        // [R][NAI * 64]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, kk = lane >> 5;
    int logical;
    {   // XCD-aware block order (as conv_bx3_kernel): the m-tiles of one pixel tile run back to back on one XCD
        const int T = gridDim.x, b = blockIdx.x;
        const int q = T >> 3, r = T & 7, xcd = b & 7, slot = b >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int mtile = logical % a.mtiles;
    int pt = logical / a.mtiles;
    const int tx = pt % a.tiles_x; pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    const int n = pt / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * 32;
    const int m0 = mtile * BM;
    const long PS = (long)a.Hp * a.Wp;
    const int ngroups = a.Cin >> 4;
    const int S = ngroups * 9;

    // ---- LDS-DMA sources.  Halo piece k of this wave covers flat slots (k*NW + wave)*64 + lane of [octet][plane][LH][LW]
    unsigned boff[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int f = min((k * NW + wave) * 64 + lane, 2 * NS * TS - 1);
        const int op = f / TS, s = f - op * TS;
        const int r = s / LW, c = s - r * LW;
        const int gy = min(oy0 + r, a.Hp - 1), gx = min(ox0 + c, a.Wp - 1);
        boff[k] = (unsigned)((op * PS + (long)gy * a.Wp + gx) * 16);
    }
    const char* bsrc = reinterpret_cast<const char*>(a.in + (long)n * a.in_ns);     // + g * 2 * NS * 16 * PS bytes per group
    const long bstep = 2 * NS * 16 * PS;
    unsigned aoff[NAW];
#pragma unroll
    for (int i = 0; i < NAW; ++i) {
        const int idx = min(i * NW + wave, NAI - 1);
        const int mt = min(mtile * (BM / 32) + (idx / NS), a.MT - 1);
        aoff[i] = (unsigned)(((mt * NS + (idx % NS)) * 64 + lane) * 16);
    }
    const char* asrc = reinterpret_cast<const char*>(a.wp);
    const long astep = (long)a.MT * (NS * 1024);                                     // bytes per (group, tap)
    const bool a_extra = NAX == 0 || wave < NAX;                                     // this wave issues NAW (else NAW - 1) weight pieces per step

    // ABL (timing ablations, profiles/probe_conv_p16.py; results are wrong): 1 no LDS-DMA in the loop, 2 no fragment reads in the
    // loop, 4 no barrier, 8 no MFMAs, 16 no epilogue
    auto issue_a = [&](int step, int slot) {
        if ((ABL & 1) && step >= LA) return;
        const char* base = asrc + (long)min(step, S - 1) * astep;
#pragma unroll
        for (int i = 0; i < NAW; ++i)
            if (NAX == 0 || i < NAW - 1 || a_extra) P16_GLDS(base + aoff[i], sA + slot * (NAI * 64) + (i * NW + wave) * 64);
    };
    auto issue_b = [&](int g, int buf, int k) {
        if ((ABL & 1) && g > 0) return;
        const char* base = bsrc + (long)min(g, ngroups - 1) * bstep;
        P16_GLDS(base + boff[k], sB + buf * BREG + (k * NW + wave) * 64);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // ---- prologue: halo tile of group 0, weight pieces of steps 0 .. LA-1
#pragma unroll
    for (int k = 0; k < NBW; ++k) issue_b(0, 0, k);
#pragma unroll
    for (int d = 0; d < LA; ++d) issue_a(d, d);
    wait_barrier<0>();

    const uint4* pa0 = sA + (wm * TM * NS) * 64 + lane;
    const uint4* pb0 = sB + kk * NS * TS + (wn * TN) * LW + j;
    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};                          // pair: mh hm hh
    constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};        // triple: lh hl mm mh hm hh (small cross terms first)

    pu32x4 af[TM][NS], bf[TN][NS];
    auto read_frags = [&](pu32x4 (&fa)[TM][NS], pu32x4 (&fb)[TN][NS], int slot, int buf, int tap) {
        const uint4* pa = pa0 + slot * (NAI * 64);
        const uint4* pb = pb0 + buf * BREG + (tap / 3) * LW + (tap % 3);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int s = 0; s < NS; ++s) fa[tm][s] = __builtin_bit_cast(pu32x4, pa[(tm * NS + s) * 64]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int s = 0; s < NS; ++s) fb[tn][s] = __builtin_bit_cast(pu32x4, pb[s * TS + tn * LW]);
    };
    auto mma_step = [&](const pu32x4 (&fa)[TM][NS], const pu32x4 (&fb)[TN][NS]) {
        if constexpr (!ILV) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < NPR; ++q)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if constexpr (PF == PF_TRI)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8, fa[tm][SA[q]]), __builtin_bit_cast(pbf16x8, fb[tn][SB[q]]), acc[tm][tn], 0, 0, 0);
                    else
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8, fa[tm][HA[q]]), __builtin_bit_cast(pf16x8, fb[tn][HB[q]]), acc[tm][tn], 0, 0, 0);
                }
        if constexpr (!ILV) __builtin_amdgcn_s_setprio(0);
    };

    if constexpr (PIPE) read_frags(af, bf, 0, 0, 0);

    // counted wait: pieces this wave issued after the ones the next step needs (own weight pieces of this step + the halo pieces
    // of this tap and the previous one) stay in flight
    auto step_wait = [&](auto nbc) {
        constexpr int nb = decltype(nbc)::value;
        if constexpr (ABL & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if constexpr (ABL & 1) wait_barrier<0>();
        else if constexpr (NAX == 0) wait_barrier<NAW + nb>();
        else { if (a_extra) wait_barrier<NAW + nb>(); else wait_barrier<NAW - 1 + nb>(); }
    };

    auto step = [&](auto tapc, int g, int buf) {
        constexpr int tap = decltype(tapc)::value;
        constexpr int hb_now = (tap * HPT < NBW) ? ((NBW - tap * HPT) < HPT ? (NBW - tap * HPT) : HPT) : 0;       // halo pieces issued at this tap
        constexpr int ptap = (tap + 8) % 9;
        constexpr int hb_prev = (ptap * HPT < NBW) ? ((NBW - ptap * HPT) < HPT ? (NBW - ptap * HPT) : HPT) : 0;   // .. at the previous step
        const int s = g * 9 + tap;
        issue_a(s + LA, PIPE ? ((s + LA) & 3) : ((tap + LA) % 3));
#pragma unroll
        for (int h = 0; h < hb_now; ++h) issue_b(g + 1, buf ^ 1, tap * HPT + h);
        if constexpr (PIPE) {
            // fragments of step s + 1 (its weight slot and halo buffer were published by the previous barrier)
            pu32x4 afn[TM][NS], bfn[TN][NS];
            constexpr int ntap = (tap + 1) % 9;
            if (!(ABL & 2)) read_frags(afn, bfn, (s + 1) & 3, tap == 8 ? (buf ^ 1) : buf, ntap);
            else {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int q = 0; q < NS; ++q) afn[tm][q] = af[tm][q];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int q = 0; q < NS; ++q) bfn[tn][q] = bf[tn][q];
            }
            if (!(ABL & 8)) mma_step(af, bf);
            else { asm volatile("" : "+v"(af[0][0]), "+v"(bf[0][0])); }
            // issue order inside the step: one LDS-DMA piece or one fragment read in the shadow of each MFMA (they belong to later
            // steps, so nothing below depends on them) instead of a load block in front of an MFMA block -- the waves that share
            // a SIMD run in phase, so a block of loads is time in which NO wave of the SIMD has an MFMA to issue
            if constexpr (ILV) {
                constexpr int NM = TM * TN * NPR, ND = (TM + TN) * NS, NV = NAW + hb_now;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // MFMA
                    if (i < NV) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);      // VMEM (LDS-DMA piece)
                    else if (i - NV < ND) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                }
            }
            // outstanding afterwards: pieces younger than the weight pieces of step s + 2 (issued at step s - 1)
            step_wait(std::integral_constant<int, hb_prev + hb_now>{});
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int q = 0; q < NS; ++q) af[tm][q] = afn[tm][q];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int q = 0; q < NS; ++q) bf[tn][q] = bfn[tn][q];
        } else {
            if (!(ABL & 2) || s == 0) read_frags(af, bf, (tap % R), buf, tap);
            if (!(ABL & 8)) mma_step(af, bf);
            else { asm volatile("" : "+v"(af[0][0]), "+v"(bf[0][0])); }
            // pieces issued after the weight pieces of step s + 1 (issued at step s - 1)
            step_wait(std::integral_constant<int, hb_prev + hb_now>{});
        }
    };
    for (int g = 0; g < ngroups; ++g) {
        const int buf = g & 1;
        step(std::integral_constant<int, 0>{}, g, buf); step(std::integral_constant<int, 1>{}, g, buf);
        step(std::integral_constant<int, 2>{}, g, buf); step(std::integral_constant<int, 3>{}, g, buf);
        step(std::integral_constant<int, 4>{}, g, buf); step(std::integral_constant<int, 5>{}, g, buf);
        step(std::integral_constant<int, 6>{}, g, buf); step(std::integral_constant<int, 7>{}, g, buf);
        step(std::integral_constant<int, 8>{}, g, buf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the clamped tail pieces have landed: LDS is free

    if constexpr (ABL & 16) {
        float sum = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[tm][tn][r];
        if (sum == 1.2345f && a.out32) a.out32[0] = sum;
        return;
    }
    p16_epilogue<TM, TN, PF>(a, acc, n, m0, wm, wn, oy0, ox0, j, kk);
}

template <int TM, int TN, int WM, int WN, bool PIPE, int ABL = 0, bool ILV = false, int PF = PF_PAIR>
int launch_p16(const P16Args& a0, int N, hipStream_t st) {
    constexpr int NS = PlaneFmt<PF>::NS;
    constexpr int NW = WM * WN, BM = 32 * TM * WM, TH = TN * WN;
    constexpr int TS = (TH + 2) * 34, NBW = (2 * NS * TS + 64 * NW - 1) / (64 * NW), BREG = NBW * 64 * NW, NAI = (BM / 32) * NS;
    constexpr int R = PIPE ? 4 : 3;
    P16Args a = a0;
    a.tiles_x = tdr_cdiv(a.W, 32);
    a.tiles_y = tdr_cdiv(a.H, TH);
    a.mtiles = tdr_cdiv(a.Cout, BM);
    const size_t lds = (size_t)(2 * BREG + R * NAI * 64) * 16;
    static_assert((size_t)(2 * BREG + R * NAI * 64) * 16 <= 160 * 1024, "tile configuration does not fit the 160 KiB of LDS");
    dim3 grid((unsigned)((long)a.tiles_x * a.tiles_y * a.mtiles * N));
    auto kern = conv3x3_p16_kernel<TM, TN, WM, WN, PIPE, ABL, ILV, PF>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a);
    TDR_LAUNCH_CHECK("conv3x3_p16_kernel");
    return TDR_OK;
}

// ---- thin levels (Cin <= 32, Cout <= 32: the C = 32 level of the MASA encoder at full resolution: 18 (group, tap) steps, one m-tile).
// The whole weight pack (<= 36 KiB) is loaded into LDS ONCE per workgroup; workgroups are persistent and walk pixel tiles of
// TH x 32 outputs; the halo tile of ALL channels of a pixel tile (2 groups x 4 planes) is loaded in one go, so a tile is
// [barrier, LDS-DMA of the next halo tile issued, epilogue stores of the previous tile, wait, barrier, 18 barrier-free steps].
// The launch is HBM-bound (4 B in + 4 B out per element and 0.3 kflop per byte): what matters is that the stream never stops --
// two workgroups per CU alternate between their load and MFMA phases.
template <int TN, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_p16_thin_kernel(P16Args a, int ntiles) {
    constexpr int TH = 4 * TN, LH = TH + 2, LW = 34, TS = LH * LW;
    constexpr int NBW = (4 * TS + 255) / 256, BREG = 4 * TS;            // exact: the last piece is exec-masked, no padding
    constexpr int MAXG = 2;
    extern __shared__ __attribute__((aligned(1024))) uint4 smem4[];
    uint4* sB = smem4;                         // [MAXG][BREG]
    uint4* sA = smem4 + MAXG * BREG;           // [ngroups * 9][2 planes][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const long PS = (long)a.Hp * a.Wp;
    const int ngroups = a.Cin >> 4, S = ngroups * 9;
    // weights: S steps x 2 pieces (m-tile 0 of every (group, tap)); pieces i = wave, wave + 4, ..
    for (int i = wave; i < 2 * S; i += 4) {
        const char* src = reinterpret_cast<const char*>(a.wp) + ((long)(i >> 1) * a.MT * 2 + (i & 1)) * 1024 + lane * 16;
        P16_GLDS(src, sA + i * 64);
    }
    // halo pieces: tile-independent decomposition of the flat slot index
    int prc[NBW];
    unsigned plin[NBW];
    bool pok[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int f0 = (k * 4 + wave) * 64 + lane;
        pok[k] = f0 < 4 * TS;
        const int f = min(f0, 4 * TS - 1);
        const int op = f / TS, sl = f - op * TS;
        const int r = sl / LW, c = sl - r * LW;
        prc[k] = (r << 8) | c;
        plin[k] = (unsigned)(op * PS * 16);
    }
    const uint4* pa0 = sA + lane;
    const uint4* pb0 = sB + kk * 2 * TS + (wave * TN) * LW + j;
    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
    const int tiles_xy = a.tiles_x * a.tiles_y;

    auto issue_tile = [&](int t) {
        const int n = t / tiles_xy, r0 = t - n * tiles_xy;
        const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
        const char* base = reinterpret_cast<const char*>(a.in + (long)n * a.in_ns);
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            const int gy = min(ty * TH + (prc[k] >> 8), a.Hp - 1), gx = min(tx * 32 + (prc[k] & 255), a.Wp - 1);
            const unsigned off = plin[k] + (unsigned)((gy * a.Wp + gx) * 16);
            if (pok[k])                          // lanes past the tile write nothing (LDS-DMA honours EXEC)
                for (int g = 0; g < ngroups; ++g) P16_GLDS(base + (long)g * 64 * PS + off, sB + g * BREG + (k * 4 + wave) * 64);
        }
    };

    f32x16 acc[1][TN];
    int tprev = -1;
    int t = blockIdx.x;
    if (t < ntiles) issue_tile(t);
    for (; t < ntiles; t += gridDim.x) {
        // this tile's halo (and, first time, the weights) have landed; the stores of tile t - 2 strides are long done
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tprev >= 0 && !(ABL & 16)) {      // the previous tile's stores go out now and drain under this tile's MFMAs
            const int n = tprev / tiles_xy, r0 = tprev - n * tiles_xy;
            const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
            p16_epilogue_lean<1, TN>(a, acc, n, 0, 0, wave, ty * TH, tx * 32, j, kk);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][tn][r] = 0.f;
        for (int g = 0; g < ngroups; ++g) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const uint4* pa = pa0 + (g * 9 + tap) * 128;
                const uint4* pb = pb0 + g * BREG + (tap / 3) * LW + (tap % 3);
                pf16x8 af[2], bf[TN][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) af[s2] = __builtin_bit_cast(pf16x8, pa[s2 * 64]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) bf[tn][s2] = __builtin_bit_cast(pf16x8, pb[s2 * TS + tn * LW]);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[HA[q]], bf[tn][HB[q]], acc[0][tn], 0, 0, 0);
            }
        }
        // every wave is done reading the halo tile: the next tile's pieces go out
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int tnext = t + gridDim.x;
        if (tnext < ntiles && !(ABL & 1)) issue_tile(tnext);
        tprev = t;
    }
    if (tprev >= 0) {
        const int n = tprev / tiles_xy, r0 = tprev - n * tiles_xy;
        const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
        p16_epilogue_lean<1, TN>(a, acc, n, 0, 0, wave, ty * TH, tx * 32, j, kk);
    }
}

template <int TN, int ABL = 0>
int launch_p16_thin(const P16Args& a0, int N, hipStream_t st) {
    constexpr int TH = 4 * TN, TS = (TH + 2) * 34, BREG = 4 * TS;
    P16Args a = a0;
    a.tiles_x = tdr_cdiv(a.W, 32);
    a.tiles_y = tdr_cdiv(a.H, TH);
    a.mtiles = 1;
    const int ngroups = a.Cin >> 4;
    const long ntiles = (long)a.tiles_x * a.tiles_y * N;
    const size_t lds = (size_t)(2 * BREG + ngroups * 9 * 128) * 16;
    static const int wgs = tdr_tune_env("TDR_P16_THIN_WGS") ? atoi(tdr_tune_env("TDR_P16_THIN_WGS")) : 512;      // 2 resident workgroups x 256 CUs
    const int grid = (int)(ntiles < wgs ? ntiles : wgs);
    auto kern = conv3x3_p16_thin_kernel<TN, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a, (int)ntiles);
    TDR_LAUNCH_CHECK("conv3x3_p16_thin_kernel");
    return TDR_OK;
}

// ---- thin levels, second structure: ONE 8-wave workgroup per CU, the weight fragments of all (group, tap) steps in REGISTERS (one
// m-tile: 18 x 2 fragments = 144 VGPRs, loaded once per workgroup -- the fragment reads of the weights were half of the LDS traffic
// of an LDS-read-bound loop), the whole-K halo tile double buffered in LDS.  Per tile: [stores of tile t-1] -> [LDS-DMA of tile
// t+1 into the other buffer] -> [18 barrier-free steps on tile t] -> wait + barrier: loads and stores are both issued a whole
// compute phase before anything waits for them.  LDS-DMA by inline asm (hipcc would put vmcnt(0) in front of the fragment reads).
__device__ __forceinline__ void p16_glds_asm(const char* src, const uint4* dst) {
    const unsigned l = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(l) : "memory");
}

template <int TN, int ABL = 0>
__global__ __launch_bounds__(512) void conv3x3_p16_thin8_kernel(P16Args a, int ntiles) {
    constexpr int TH = 8 * TN, LH = TH + 2, LW = 34, TS = LH * LW;
    constexpr int NBW = (4 * TS + 511) / 512, BREG = 4 * TS;
    extern __shared__ __attribute__((aligned(1024))) uint4 smem4[];      // [2 buffers][2 groups][BREG]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const long PS = (long)a.Hp * a.Wp;
    const int ngroups = a.Cin >> 4;
    pf16x8 aw[18][2];
#pragma unroll
    for (int s2 = 0; s2 < 18; ++s2) {
        const int sc = min(s2, ngroups * 9 - 1);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            aw[s2][pl] = __builtin_bit_cast(pf16x8, a.wp[((long)sc * a.MT * 2 + pl) * 64 + lane]);
    }
    int prc[NBW];
    unsigned plin[NBW];
    bool pok[NBW];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int f0 = (k * 8 + wave) * 64 + lane;
        pok[k] = f0 < 4 * TS;
        const int f = min(f0, 4 * TS - 1);
        const int op = f / TS, sl = f - op * TS;
        const int r = sl / LW, c = sl - r * LW;
        prc[k] = (r << 8) | c;
        plin[k] = (unsigned)(op * PS * 16);
    }
    const uint4* pb0 = smem4 + kk * 2 * TS + (wave * TN) * LW + j;
    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
    const int tiles_xy = a.tiles_x * a.tiles_y;

    auto issue_tile = [&](int t, int buf) {
        const int n = t / tiles_xy, r0 = t - n * tiles_xy;
        const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
        const char* base = reinterpret_cast<const char*>(a.in + (long)n * a.in_ns);
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            const int gy = min(ty * TH + (prc[k] >> 8), a.Hp - 1), gx = min(tx * 32 + (prc[k] & 255), a.Wp - 1);
            const unsigned off = plin[k] + (unsigned)((gy * a.Wp + gx) * 16);
            if (pok[k])
                for (int g = 0; g < ngroups; ++g)
                    p16_glds_asm(base + (long)g * 64 * PS + off, smem4 + (buf * 2 + g) * BREG + (k * 8 + wave) * 64);
        }
    };
    auto tile_epilogue = [&](int t, f32x16 (&acc)[1][TN]) {
        const int n = t / tiles_xy, r0 = t - n * tiles_xy;
        const int ty = r0 / a.tiles_x, tx = r0 - ty * a.tiles_x;
        p16_epilogue_lean<1, TN>(a, acc, n, 0, 0, wave, ty * TH, tx * 32, j, kk);
    };

    f32x16 acc[1][TN];
    int tprev = -1, it = 0;
    int t = blockIdx.x;
    if (t < ntiles) issue_tile(t, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (; t < ntiles; t += gridDim.x, ++it) {
        const int buf = it & 1;
        if (tprev >= 0 && !(ABL & 16)) tile_epilogue(tprev, acc);
        const int tnext = t + gridDim.x;
        if (tnext < ntiles && !(ABL & 1)) issue_tile(tnext, buf ^ 1);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][tn][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            if (g < ngroups) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const uint4* pb = pb0 + (buf * 2 + g) * BREG + (tap / 3) * LW + (tap % 3);
                    pf16x8 bf[TN][2];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) bf[tn][s2] = __builtin_bit_cast(pf16x8, pb[s2 * TS + tn * LW]);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aw[g * 9 + tap][HA[q]], bf[tn][HB[q]], acc[0][tn], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        tprev = t;
    }
    if (tprev >= 0) tile_epilogue(tprev, acc);
}

template <int TN, int ABL = 0>
int launch_p16_thin8(const P16Args& a0, int N, hipStream_t st) {
    constexpr int TH = 8 * TN, TS = (TH + 2) * 34, BREG = 4 * TS;
    P16Args a = a0;
    a.tiles_x = tdr_cdiv(a.W, 32);
    a.tiles_y = tdr_cdiv(a.H, TH);
    a.mtiles = 1;
    const long ntiles = (long)a.tiles_x * a.tiles_y * N;
    const size_t lds = (size_t)(4 * BREG) * 16;
    static const int wgs = tdr_tune_env("TDR_P16_THIN8_WGS") ? atoi(tdr_tune_env("TDR_P16_THIN8_WGS")) : 256;      // one resident workgroup per CU
    const int grid = (int)(ntiles < wgs ? ntiles : wgs);
    auto kern = conv3x3_p16_thin8_kernel<TN, ABL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a, (int)ntiles);
    TDR_LAUNCH_CHECK("conv3x3_p16_thin8_kernel");
    return TDR_OK;
}

// ---- fp32 NCHW <-> plane tensors
template <int PF>
__global__ void p16_from_f32_kernel(const float* __restrict__ src, long src_ns, int C, int H, int W, uint4* __restrict__ dst) {
    constexpr int NS = PlaneFmt<PF>::NS;
    const int Hp = H + 2, Wp = W + 2;
    const long PS = (long)Hp * Wp, HW = (long)H * W;
    const int G = C >> 3;
    const long total = (long)G * PS;
    const int n = blockIdx.y;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int oc = (int)(i / PS);
        const long s = i - oc * PS;
        const int y = (int)(s / Wp), x = (int)(s - (long)y * Wp);
        const bool in = y >= 1 && y <= H && x >= 1 && x <= W;
        const float* p = src + (long)n * src_ns + (long)oc * 8 * HW + (in ? (long)(y - 1) * W + x - 1 : 0);
        uint4* o = dst + ((long)n * G + oc) * NS * PS + s;
        if constexpr (PF == PF_TRI) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = in ? p[e * HW] : 0.f;
            uint2 h0, m0, l0, h1, m1, l1;
            p24_split4(v[0], v[1], v[2], v[3], h0, m0, l0);
            p24_split4(v[4], v[5], v[6], v[7], h1, m1, l1);
            if (!in) { h0 = make_uint2(0, 0); h1 = h0; }          // (the border is plain +0, as the convolution epilogue writes it)
            o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            o[PS] = make_uint4(m0.x, m0.y, m1.x, m1.y);
            o[2 * PS] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        } else {
            pf16x8 h, m;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = in ? p[e * HW] : 0.f;
                asm volatile("" : "+v"(v));
                const _Float16 hh = (_Float16)v;
                h[e] = in ? p16_head(v) : hh;             // (the border is plain +0, as the convolution epilogue writes it)
                m[e] = (_Float16)(v - (float)hh);
            }
            o[0] = __builtin_bit_cast(uint4, h);
            o[PS] = __builtin_bit_cast(uint4, m);
        }
    }
}

template <int PF>
__global__ void p16_to_f32_kernel(const uint4* __restrict__ src, int C, int H, int W, float* __restrict__ dst, long dst_ns) {
    constexpr int NS = PlaneFmt<PF>::NS;
    const int Hp = H + 2, Wp = W + 2;
    const long PS = (long)Hp * Wp, HW = (long)H * W;
    const int G = C >> 3;
    const long total = (long)G * HW;
    const int n = blockIdx.y;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int oc = (int)(i / HW);
        const long s = i - oc * HW;
        const int y = (int)(s / W), x = (int)(s - (long)y * W);
        const uint4* p = src + ((long)n * G + oc) * NS * PS + (long)(y + 1) * Wp + x + 1;
        float* o = dst + (long)n * dst_ns + (long)(oc * 8) * HW + s;
        if constexpr (PF == PF_TRI) {
            const uint4 h = p[0], m = p[PS], l = p[2 * PS];
            const unsigned hh[4] = {h.x, h.y, h.z, h.w}, mm[4] = {m.x, m.y, m.z, m.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[(long)(2 * e) * HW] = (__builtin_bit_cast(float, hh[e] << 16) + __builtin_bit_cast(float, mm[e] << 16)) + __builtin_bit_cast(float, ll[e] << 16);
                o[(long)(2 * e + 1) * HW] = (__builtin_bit_cast(float, hh[e] & 0xffff0000u) + __builtin_bit_cast(float, mm[e] & 0xffff0000u)) +
                                            __builtin_bit_cast(float, ll[e] & 0xffff0000u);
            }
        } else {
            const pf16x8 h = __builtin_bit_cast(pf16x8, p[0]), m = __builtin_bit_cast(pf16x8, p[PS]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[(long)e * HW] = (float)h[e] + (float)m[e];
        }
    }
}

}  // namespace

extern "C" int64_t tdr_p16_bytes_fmt(int N, int C, int H, int W, int fmt) {
    return (int64_t)N * (C / 8) * (fmt == PF_TRI ? 3 : 2) * (H + 2) * (W + 2) * 16;
}
extern "C" int64_t tdr_p16_bytes(int N, int C, int H, int W) { return tdr_p16_bytes_fmt(N, C, H, W, PF_PAIR); }

extern "C" int tdr_p16_from_f32_fmt(const float* src, int64_t src_ns, int N, int C, int H, int W, void* dst, int fmt, void* stream) {
    TDR_REQUIRE(src && dst && N > 0, "tdr_p16_from_f32: bad argument");
    TDR_REQUIRE(C % 16 == 0, "tdr_p16_from_f32: C = %d is not a multiple of 16", C);
    TDR_REQUIRE(fmt == PF_TRI || fmt == PF_PAIR, "tdr_p16_from_f32: plane format %d (1: bf16 triple, 2: fp16 pair)", fmt);
    const long total = (long)(C / 8) * (H + 2) * (W + 2);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (fmt == PF_TRI) hipLaunchKernelGGL(p16_from_f32_kernel<PF_TRI>, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, src, (long)src_ns, C, H, W, (uint4*)dst);
    else hipLaunchKernelGGL(p16_from_f32_kernel<PF_PAIR>, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, src, (long)src_ns, C, H, W, (uint4*)dst);
    TDR_LAUNCH_CHECK("p16_from_f32_kernel");
    return TDR_OK;
}
extern "C" int tdr_p16_from_f32(const float* src, int64_t src_ns, int N, int C, int H, int W, void* dst, void* stream) {
    return tdr_p16_from_f32_fmt(src, src_ns, N, C, H, W, dst, PF_PAIR, stream);
}

extern "C" int tdr_p16_to_f32_fmt(const void* src, int N, int C, int H, int W, float* dst, int64_t dst_ns, int fmt, void* stream) {
    TDR_REQUIRE(src && dst && N > 0, "tdr_p16_to_f32: bad argument");
    TDR_REQUIRE(C % 16 == 0, "tdr_p16_to_f32: C = %d is not a multiple of 16", C);
    TDR_REQUIRE(fmt == PF_TRI || fmt == PF_PAIR, "tdr_p16_to_f32: plane format %d (1: bf16 triple, 2: fp16 pair)", fmt);
    const long total = (long)(C / 8) * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (fmt == PF_TRI) hipLaunchKernelGGL(p16_to_f32_kernel<PF_TRI>, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, C, H, W, dst, (long)dst_ns);
    else hipLaunchKernelGGL(p16_to_f32_kernel<PF_PAIR>, dim3(blocks, N), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, C, H, W, dst, (long)dst_ns);
    TDR_LAUNCH_CHECK("p16_to_f32_kernel");
    return TDR_OK;
}
extern "C" int tdr_p16_to_f32(const void* src, int N, int C, int H, int W, float* dst, int64_t dst_ns, void* stream) {
    return tdr_p16_to_f32_fmt(src, N, C, H, W, dst, dst_ns, PF_PAIR, stream);
}

static int p16_env_cfg() {
    const int c = tdr_tune_env("TDR_P16_CFG") ? atoi(tdr_tune_env("TDR_P16_CFG")) : 0;
    return (((c >= 100 && c < 300) || c >= 400) && !tdr_tune_env("TDR_PROBES")) ? 0 : c;
}
static int g_p16_cfg = p16_env_cfg();
extern "C" int tdr_conv3x3_p16_force_cfg(int cfg) {
    // 100 .. 299 are timing ablations that compute WRONG results (profiles/probe_conv_p16.py): never reachable from a product process
    static const bool probes = tdr_tune_env("TDR_PROBES") != nullptr;
    TDR_REQUIRE(probes || cfg < 100 || (cfg >= 300 && cfg < 400), "tdr_conv3x3_p16_force_cfg: configuration %d is a timing ablation (set TDR_PROBES=1)", cfg);
    g_p16_cfg = cfg;
    return TDR_OK;
}

extern "C" int tdr_conv3x3_p16(const TdrConvP16Desc* d, void* stream) {
    TDR_REQUIRE(d && d->in && d->wp && (d->out32 || d->out16), "tdr_conv3x3_p16: null pointer");
    TDR_REQUIRE(d->Cin % 16 == 0 && d->Cin >= 16, "tdr_conv3x3_p16: Cin = %d is not a multiple of 16", d->Cin);
    TDR_REQUIRE(!d->out16 || d->Cout % 16 == 0, "tdr_conv3x3_p16: P16 output needs Cout %% 16 == 0 (got %d)", d->Cout);
    TDR_REQUIRE((!d->res16 && !d->mask16) || d->Cout % 8 == 0, "tdr_conv3x3_p16: P16 residual / mask need Cout %% 8 == 0");
    TDR_REQUIRE(d->wp_fmt == 2 || d->wp_fmt == 1, "tdr_conv3x3_p16: weights must be the 2-way fp16 split pack (wp_fmt 2, pair planes) or the 3-way bf16 split pack (wp_fmt 1, triple planes)");
    const int NSd = d->wp_fmt == 1 ? 3 : 2;       // planes of every plane tensor of this call
    P16Args a;
    a.in = (const uint4*)d->in; a.in_ns = (long)(d->Cin / 8) * NSd * (d->H + 2) * (d->W + 2);
    a.Cin = d->Cin; a.H = d->H; a.W = d->W; a.Hp = d->H + 2; a.Wp = d->W + 2;
    a.wp = (const uint4*)d->wp; a.MT = d->Mpad >> 5; a.Cout = d->Cout;
    a.bias = d->bias;
    a.res32 = d->res32; a.res32_ns = d->res32_ns;
    a.res16 = (const uint4*)d->res16; a.res16_ns = (long)(d->Cout / 8) * NSd * (d->H + 2) * (d->W + 2);
    a.mask32 = d->mask32; a.mask32_ns = d->mask32_ns;
    a.mask16 = (const uint4*)d->mask16; a.mask16_ns = a.res16_ns;
    a.relu = d->relu;
    a.out32 = d->out32; a.out32_ns = d->out32_ns;
    a.out16 = (uint4*)d->out16; a.out16_ns = a.res16_ns;
    a.tiles_x = a.tiles_y = a.mtiles = 0;
    hipStream_t st = (hipStream_t)stream;
    const int N = d->N;
    auto blocks = [&](int bm, int th) { return (long)tdr_cdiv(d->Cout, bm) * tdr_cdiv(d->H, th) * tdr_cdiv(d->W, 32) * N; };
    int cfg = g_p16_cfg;
    if (d->wp_fmt == 1) {
        // triple planes (TDR_MATH=bx3).  Tile configurations (forced ones: 301 ..): see profiles/r5/probe_p24*.log
        constexpr int T = PF_TRI;
        if (cfg < 300) cfg = 0;
        if (cfg == 0) {
            // profiles/r5/probe_p24_v1.log / probe_p24_abl.log (N = 8, C = 64 .. 512 at 256^2 .. 32^2): the pure MFMA stream of a launch is
            // 126 - 140 us (232 GFLOP of products at the sustained ~1.75 PFLOP/s), LDS-DMA issue adds 13 - 24, fragment reads 6 - 11, the
            // epilogue 10 - 54 (it is HBM-write-bound at C = 64 and nothing overlaps it with one workgroup per CU).  Per level the best:
            // C <= 64: 64 x (4 x 32) with two workgroups per CU (220 vs 238 us); 128 x (8 x 32) on 8 waves where that still fills the
            // chip (176 / 160 vs 191 / 170 us at C = 128 / 256); else 64 x (8 x 32) (C = 512 @ 32^2: 160 us, 256 workgroups)
            if (d->Cout <= 32) cfg = 308;
            else if (d->Cout <= 64) cfg = 303;
            else if (d->Cout % 128 == 0 && blocks(128, 8) >= 256) cfg = 302;
            else cfg = 301;
        }
        switch (cfg) {
            //                         TM TN WM WN PIPE ABL ILV
            case 301: return launch_p16<2, 2, 1, 4, true, 0, true, T>(a, N, st);     //  64 x (8 x 32), 4 waves (89 KiB: one workgroup per CU)
            case 302: return launch_p16<2, 2, 2, 4, true, 0, true, T>(a, N, st);     // 128 x (8 x 32), 8 waves
            case 303: return launch_p16<1, 2, 2, 2, true, 0, true, T>(a, N, st);     //  64 x (4 x 32), 4 waves (two workgroups per CU)
            case 304: return launch_p16<2, 2, 2, 2, true, 0, true, T>(a, N, st);     // 128 x (4 x 32), 4 waves
            case 306: return launch_p16<2, 2, 4, 2, true, 0, true, T>(a, N, st);     // 256 x (4 x 32), 8 waves
            case 307: return launch_p16<2, 1, 2, 4, true, 0, true, T>(a, N, st);     // 128 x (4 x 32), 8 waves
            case 308: return launch_p16<1, 2, 1, 4, true, 0, true, T>(a, N, st);     //  32 x (8 x 32), 4 waves (the C = 32 level)
            case 309: return launch_p16<1, 1, 1, 4, true, 0, true, T>(a, N, st);     //  32 x (4 x 32), 4 waves
            case 311: return launch_p16<2, 2, 1, 4, true, 0, false, T>(a, N, st);    // 301 without the interleaved issue order
            case 312: return launch_p16<2, 2, 2, 4, true, 0, false, T>(a, N, st);
            case 321: return launch_p16<2, 2, 1, 4, false, 0, false, T>(a, N, st);   // 301 without pipelined fragments
            // timing ablations of 301 / 302 (wrong results; TDR_PROBES only): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no barrier, 8 no MFMAs, 16 no epilogue
            case 401: return launch_p16<2, 2, 1, 4, true, 1, true, T>(a, N, st);
            case 402: return launch_p16<2, 2, 1, 4, true, 2, true, T>(a, N, st);
            case 403: return launch_p16<2, 2, 1, 4, true, 3, true, T>(a, N, st);
            case 404: return launch_p16<2, 2, 1, 4, true, 4, true, T>(a, N, st);
            case 407: return launch_p16<2, 2, 1, 4, true, 7, true, T>(a, N, st);
            case 408: return launch_p16<2, 2, 1, 4, true, 8, true, T>(a, N, st);
            case 416: return launch_p16<2, 2, 1, 4, true, 16, true, T>(a, N, st);
            case 423: return launch_p16<2, 2, 1, 4, true, 23, true, T>(a, N, st);
            case 431: return launch_p16<2, 2, 1, 4, true, 31, true, T>(a, N, st);
            case 501: return launch_p16<2, 2, 2, 4, true, 1, true, T>(a, N, st);
            case 507: return launch_p16<2, 2, 2, 4, true, 7, true, T>(a, N, st);
            case 516: return launch_p16<2, 2, 2, 4, true, 16, true, T>(a, N, st);
            case 523: return launch_p16<2, 2, 2, 4, true, 23, true, T>(a, N, st);
            default: break;
        }
        tdr_set_error("tdr_conv3x3_p16: unknown triple-plane tile configuration %d", cfg);
        return TDR_ERR_ARG;
    }
    if (cfg >= 300) cfg = 0;
    if (cfg >= 32 && cfg <= 36 && d->Cin <= 32 && d->Cout <= 32)
        return cfg == 32 ? launch_p16_thin8<1>(a, N, st) : (cfg == 33 ? launch_p16_thin8<2>(a, N, st) : (cfg == 34 ? launch_p16_thin8<1, 1>(a, N, st) :
               (cfg == 35 ? launch_p16_thin8<1, 16>(a, N, st) : launch_p16_thin8<1, 17>(a, N, st))));
    if (cfg >= 130 && cfg <= 132 && d->Cin <= 32 && d->Cout <= 32)      // timing ablations of the thin kernel (wrong results)
        return cfg == 130 ? launch_p16_thin<2, 1>(a, N, st) : (cfg == 131 ? launch_p16_thin<2, 16>(a, N, st) : launch_p16_thin<2, 17>(a, N, st));
    if ((cfg == 0 || cfg == 30 || cfg == 31) && d->Cin <= 32 && d->Cout <= 32) {
        // thin level: weights-stationary persistent kernel (TN = 2: 8 x 32 pixel tiles; cfg 31: 4 x 32).  Measured at 32 -> 32 @512^2, N = 8
        // (profiles/r4/probe_p16_thin*.log): 182 - 195 us against 208 - 212 us of the fp32-tensor kernel and a ~125 us HBM floor (624 MB);
        // the 8-wave / weights-in-registers structure (cfg 32 / 33) hides the loads (108 us without its epilogue) but not yet the stores.
        return cfg == 31 ? launch_p16_thin<1>(a, N, st) : launch_p16_thin<2>(a, N, st);
    }
    if (cfg >= 30 && cfg <= 36) cfg = 0;          // forced thin configuration on a shape it does not take
    if (cfg == 0) {
        // profiles/r4/probe_p16_v3.log (N = 8, C = 64 .. 512 at 256^2 .. 32^2): the 64 x (8 x 32) tile with pipelined fragments and
        // LDS-DMA / fragment reads interleaved with the MFMAs is the best or within 2 % of the best at every level
        cfg = 16;
        (void)blocks;
    }
    switch (cfg) {
        //                       TM TN WM WN PIPE
        case 1: return launch_p16<4, 2, 1, 4, false>(a, N, st);   // 128 x (8 x 32), 4 waves
        case 2: return launch_p16<2, 2, 2, 2, false>(a, N, st);   // 128 x (4 x 32)
        case 3: return launch_p16<2, 2, 1, 4, false>(a, N, st);   //  64 x (8 x 32)
        case 4: return launch_p16<1, 2, 2, 2, false>(a, N, st);   //  64 x (4 x 32)
        case 5: return launch_p16<4, 1, 1, 4, false>(a, N, st);   // 128 x (4 x 32), all waves side by side in pixels
        case 6: return launch_p16<2, 2, 1, 4, true>(a, N, st);    //  64 x (8 x 32), pipelined fragments
        case 7: return launch_p16<2, 2, 2, 2, true>(a, N, st);    // 128 x (4 x 32), pipelined
        case 8: return launch_p16<2, 2, 2, 4, false>(a, N, st);   // 128 x (8 x 32), 8 waves
        case 9: return launch_p16<2, 2, 2, 4, true>(a, N, st);    // 128 x (8 x 32), 8 waves, pipelined
        case 10: return launch_p16<2, 4, 2, 4, false>(a, N, st);  // 128 x (16 x 32), 8 waves
        case 11: return launch_p16<2, 4, 2, 4, true>(a, N, st);   // 128 x (16 x 32), 8 waves, pipelined
        case 12: return launch_p16<2, 2, 4, 2, true>(a, N, st);   // 256 x (4 x 32), 8 waves, pipelined
        case 13: return launch_p16<4, 2, 1, 4, true>(a, N, st);   // 128 x (8 x 32), 4 waves, pipelined
        case 16: return launch_p16<2, 2, 1, 4, true, 0, true>(a, N, st);    //  64 x (8 x 32), pipelined + interleaved
        case 17: return launch_p16<2, 2, 2, 2, true, 0, true>(a, N, st);    // 128 x (4 x 32)
        case 19: return launch_p16<2, 2, 2, 4, true, 0, true>(a, N, st);    // 128 x (8 x 32), 8 waves
        case 22: return launch_p16<2, 2, 4, 2, true, 0, true>(a, N, st);    // 256 x (4 x 32), 8 waves
        case 201: return launch_p16<2, 2, 1, 4, true, 1, true>(a, N, st);
        case 202: return launch_p16<2, 2, 1, 4, true, 2, true>(a, N, st);
        case 203: return launch_p16<2, 2, 1, 4, true, 3, true>(a, N, st);
        case 204: return launch_p16<2, 2, 1, 4, true, 4, true>(a, N, st);
        case 207: return launch_p16<2, 2, 1, 4, true, 7, true>(a, N, st);
        case 216: return launch_p16<2, 2, 1, 4, true, 16, true>(a, N, st);
        case 223: return launch_p16<2, 2, 1, 4, true, 23, true>(a, N, st);
        case 101: return launch_p16<2, 2, 1, 4, false, 1>(a, N, st);
        case 102: return launch_p16<2, 2, 1, 4, false, 2>(a, N, st);
        case 103: return launch_p16<2, 2, 1, 4, false, 3>(a, N, st);
        case 104: return launch_p16<2, 2, 1, 4, false, 4>(a, N, st);
        case 107: return launch_p16<2, 2, 1, 4, false, 7>(a, N, st);
        case 108: return launch_p16<2, 2, 1, 4, false, 8>(a, N, st);
        case 116: return launch_p16<2, 2, 1, 4, false, 16>(a, N, st);
        case 123: return launch_p16<2, 2, 1, 4, false, 23>(a, N, st);
        case 131: return launch_p16<2, 2, 1, 4, false, 31>(a, N, st);
        default: break;
    }
    tdr_set_error("tdr_conv3x3_p16: unknown tile configuration %d", cfg);
    return TDR_ERR_ARG;
}
