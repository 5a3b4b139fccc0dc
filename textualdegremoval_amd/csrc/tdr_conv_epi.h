// Shared argument block and output epilogues of the implicit-GEMM convolution kernels
// (tdr_conv_mfma.hip: exact fp32 MFMA; tdr_conv_bx3.hip: 3-way bf16 split MFMA).  Both produce
// 32x32 accumulator tiles in the gfx950 C/D layout: lane (j = lane&31, kk = lane>>5) holds pixel j
// of sub-tile tn and rows (r&3) + 8*(r>>2) + 4*kk.
#pragma once
#include "tdr_common.h"

namespace {

enum { EPI_STD = 0, EPI_GATEBWD = 1, EPI_PSHUF = 2 };

struct ConvArgs {
    const float* in; long in_ns; int Cin, H, W;
    const float* wp; long wp_ns; int Mpad, Cout;
    float* out; long out_ns; int OH, OW;
    int pad, tw_log2, tiles_x, mtiles;
    const float* kscale; long kscale_ns;
    long gate_off;
    const float* bias; long bias_ns;
    const float* scale; long scale_ns;
    const float* bias2; long bias2_ns; float bias2_mul;
    const float* res; long res_ns;
    const float* mask; long mask_ns;
    const float* aux; long aux_ns;
    int relu;
    int vec_epi;      // 1: rows are 16-byte aligned (OW % 4 == 0, aligned bases): LDS-transposed float4 epilogue
    int single_buf;   // split-bf16 kernel: one LDS operand buffer (short K loops; more workgroups per CU)
    int scheme;       // 0: 3-way bf16 split (6 products), 1: 2-way fp16 split (3 products; operands in fp16 range)
};

// acc[TM][TN]: wave (wm, wn) owns output-channel tiles wm*TM.. and pixel sub-tiles wn*TN..
template <int TM, int TN, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], int n, int m0, int wm, int wn,
                                              int oy0, int ox0, int j, int kk) {
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2;
    const long HWo = (long)a.OH * a.OW;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int t = wn * TN + tn;
        const int oy = oy0 + t * SR + (j >> a.tw_log2), ox = ox0 + (j & (TW - 1));
        const bool pvalid = oy < a.OH && ox < a.OW;
        const long pix = pvalid ? (long)oy * a.OW + ox : 0;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mb = m0 + (wm * TM + tm) * 32 + 4 * kk;
            if (EPI == EPI_PSHUF) {
                // rows 4q..4q+3 (q = r>>2) of this lane are the 2x2 sub-pixels of channel (mb+8q)/4
                const long OW2 = 2L * a.OW;
                const long p2 = pvalid ? (2L * oy) * OW2 + 2L * ox : 0;
                float2 r0[4], r1[4];
                if (a.res) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = min(mb + 8 * q, a.Cout - 4) >> 2;
                        const float* rp = a.res + (long)n * a.res_ns + (long)c * 4 * HWo + p2;
                        r0[q] = *reinterpret_cast<const float2*>(rp);
                        r1[q] = *reinterpret_cast<const float2*>(rp + OW2);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mb + 8 * q;
                    float v0 = acc[tm][tn][4 * q + 0], v1 = acc[tm][tn][4 * q + 1];
                    float v2 = acc[tm][tn][4 * q + 2], v3 = acc[tm][tn][4 * q + 3];
                    if (a.res) { v0 += r0[q].x; v1 += r0[q].y; v2 += r1[q].x; v3 += r1[q].y; }
                    if (pvalid && m < a.Cout) {
                        float* o = a.out + (long)n * a.out_ns + (long)(m >> 2) * 4 * HWo + p2;
                        *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                        *reinterpret_cast<float2*>(o + OW2) = make_float2(v2, v3);
                    }
                }
            } else if (EPI == EPI_GATEBWD) {
                const float* ax = a.aux + (long)n * a.aux_ns;
                const long half = (long)a.Cout * HWo;
                float a0[16], a1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mc = min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1);
                    const long o = (long)mc * HWo + pix;
                    a0[r] = ax[o];
                    a1[r] = ax[o + half];
                }
                float* op = a.out + (long)n * a.out_ns;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (pvalid && m < a.Cout) {
                        const long o = (long)m * HWo + pix;
                        const float v = acc[tm][tn][r];
                        op[o] = v * a1[r];
                        op[o + half] = v * a0[r];
                    }
                }
            } else {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[tm][tn][r];
                if (a.bias) {
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) tv[r] = a.bias[(long)n * a.bias_ns + min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += tv[r];
                }
                if (a.scale) {
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) tv[r] = a.scale[(long)n * a.scale_ns + min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] *= tv[r];
                }
                if (a.bias2) {
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) tv[r] = a.bias2[(long)n * a.bias2_ns + min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += a.bias2_mul * tv[r];
                }
                if (a.res) {
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tv[r] = a.res[(long)n * a.res_ns + (long)min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1) * HWo + pix];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += tv[r];
                }
                if (a.relu == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
                } else if (a.relu == 2) {               // exact (erf) GELU: ViT MLP
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.f + erff(v[r] * 0.70710678118654752f));
                } else if (a.relu == 3) {               // quick_gelu x*sigmoid(1.702x): OpenAI CLIP MLP
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = v[r] / (1.f + expf(-1.702f * v[r]));
                }
                if (a.mask) {
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tv[r] = a.mask[(long)n * a.mask_ns + (long)min(mb + (r & 3) + 8 * (r >> 2), a.Cout - 1) * HWo + pix];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = tv[r] > 0.f ? v[r] : 0.f;
                }
                float* op = a.out + (long)n * a.out_ns + pix;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (pvalid && m < a.Cout) op[(long)m * HWo] = v[r];
                }
            }
        }
    }
}


// Vector epilogue: each 32x32 accumulator tile goes through a wave-private LDS patch (32 rows x 36 floats) so
// that a lane ends up with 4 consecutive pixels of one output channel: the tile leaves as 4 global_store_dwordx4
// per lane (8 x 128-byte row segments per wave instruction) instead of 16 dword stores (2 segments per
// instruction), and residual / mask / gate operands arrive as float4 loads.  DS operations of one wave execute
// in order, so the patch needs no barrier.  Requires a.vec_epi (all rows 16-byte aligned); STD and GATEBWD only.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void conv_epilogue_vec(const ConvArgs& a, f32x16 (&acc)[TM][TN], int n, int m0, int wm, int wn,
                                                  int oy0, int ox0, int lane, float* sw) {
    static_assert(EPI == EPI_STD || EPI == EPI_GATEBWD, "vector epilogue: STD / GATEBWD");
    const int j = lane & 31, kk = lane >> 5;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2;
    const long HWo = (long)a.OH * a.OW;
    const int c4 = (lane & 7) * 4;            // first of this lane's 4 pixels inside the 32-pixel sub-tile
    const int r0 = lane >> 3;                 // rows r0 + 8 i
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int t = wn * TN + tn;
        const int oy = oy0 + t * SR + (c4 >> a.tw_log2), ox = ox0 + (c4 & (TW - 1));
        const bool pvalid = oy < a.OH && ox < a.OW;
        const long pix = pvalid ? (long)oy * a.OW + ox : 0;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sw[((r & 3) + 8 * (r >> 2) + 4 * kk) * 36 + j] = acc[tm][tn][r];
            f32x4 v[4];
            int mrow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = *reinterpret_cast<const f32x4*>(sw + (r0 + 8 * i) * 36 + c4);
                mrow[i] = m0 + (wm * TM + tm) * 32 + r0 + 8 * i;
            }
            if (EPI == EPI_GATEBWD) {
                const float* ax = a.aux + (long)n * a.aux_ns;
                const long half = (long)a.Cout * HWo;
                f32x4 a0[4], a1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long o = (long)min(mrow[i], a.Cout - 1) * HWo + pix;
                    a0[i] = *reinterpret_cast<const f32x4*>(ax + o);
                    a1[i] = *reinterpret_cast<const f32x4*>(ax + o + half);
                }
                float* op = a.out + (long)n * a.out_ns;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (pvalid && mrow[i] < a.Cout) {
                        const long o = (long)mrow[i] * HWo + pix;
                        *reinterpret_cast<f32x4*>(op + o) = v[i] * a1[i];
                        *reinterpret_cast<f32x4*>(op + o + half) = v[i] * a0[i];
                    }
            } else {
                f32x4 rv[4], mv[4];
                if (a.res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        rv[i] = *reinterpret_cast<const f32x4*>(a.res + (long)n * a.res_ns + (long)min(mrow[i], a.Cout - 1) * HWo + pix);
                }
                if (a.mask) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        mv[i] = *reinterpret_cast<const f32x4*>(a.mask + (long)n * a.mask_ns + (long)min(mrow[i], a.Cout - 1) * HWo + pix);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int mc = min(mrow[i], a.Cout - 1);
                    f32x4 x = v[i];
                    if (a.bias) x += a.bias[(long)n * a.bias_ns + mc];
                    if (a.scale) x *= a.scale[(long)n * a.scale_ns + mc];
                    if (a.bias2) x += a.bias2_mul * a.bias2[(long)n * a.bias2_ns + mc];
                    if (a.res) x += rv[i];
                    if (a.relu == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
                    } else if (a.relu == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = 0.5f * x[e] * (1.f + erff(x[e] * 0.70710678118654752f));
                    } else if (a.relu == 3) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = x[e] / (1.f + expf(-1.702f * x[e]));
                    }
                    if (a.mask) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = mv[i][e] > 0.f ? x[e] : 0.f;
                    }
                    if (pvalid && mrow[i] < a.Cout)
                        *reinterpret_cast<f32x4*>(a.out + (long)n * a.out_ns + (long)mrow[i] * HWo + pix) = x;
                }
            }
        }
    }
    (void)z4;
}

}  // namespace
