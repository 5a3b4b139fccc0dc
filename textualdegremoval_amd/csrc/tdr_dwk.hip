// Generic grouped "depthwise-like" convolutions of DRSformer-ref's mixed-scale feed-forward (network_drsformer_guided_arch*.py
// :216-253): K x K (K = 3, 5), stride 1, pad K/2, no bias, groups = Cout with `mult` = 1 or 2 input planes per output plane
// (dwconv3x3 / dwconv5x5: mult 1 over 2h planes; dwconv3x3_1 / dwconv5x5_1: Conv2d(2h, h, groups=h), mult 2), optional ReLU:
//     y[n][c] = act( sum_{i < mult} w[c][i] (*) x[n][c * mult + i] )
// HBM-bound stencils: each input plane goes through LDS once per launch (tile + halo); the weight gradient writes per-tile
// partials that a second kernel sums in a fixed order (deterministic).
#include <stdlib.h>
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled kernels, K in {1, 3, 5, 7}, dilation 1 or 2 (MEFC's dil_conv ops): a workgroup owns a 64 x 16 output tile of one
// (image, output plane); the input tile + halo of each of its `mult` input planes goes through LDS once, a thread computes 4
// adjacent outputs from a (4 + K - 1)-wide register window per kernel row.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TW_ = 64, TH_ = 16;

template <int K, int DIL>
__device__ __forceinline__ void load_tile(float* __restrict__ t, const float* __restrict__ plane, const float* __restrict__ maskp,
                                          int y0, int x0, int H, int W) {
    constexpr int LW = TW_ + (K - 1) * DIL, LH = TH_ + (K - 1) * DIL, P = (K / 2) * DIL;
    for (int i = threadIdx.x; i < LW * LH; i += 256) {
        const int r = i / LW, c = i - r * LW;
        const int y = y0 + r - P, x = x0 + c - P;
        float v = 0.f;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            v = plane[(long)y * W + x];
            if (maskp && !(maskp[(long)y * W + x] > 0.f)) v = 0.f;
        }
        t[i] = v;
    }
}

// FLIP = false: y[c] = act(b + sum_q w[c][q] (*) x[c*mult+q])            grid (tiles, Cout, N)
// FLIP = true : dx[c*mult+q] = w[c][q]^flip (*) (dy[c] masked by yact)   (same tile of g serves both q)
template <int K, int DIL, bool FLIP>
__global__ __launch_bounds__(256) void dwk_tiled_kernel(const float* __restrict__ in, long in_ns, const float* __restrict__ maskp_,
                                                       long mask_ns, const float* __restrict__ w, const float* __restrict__ b,
                                                       int Cout, int mult, int H, int W, int tiles_x, int relu,
                                                       float* __restrict__ out, long out_ns) {
    constexpr int LW = TW_ + (K - 1) * DIL;
    __shared__ float tile[(TW_ + (K - 1) * DIL) * (TH_ + (K - 1) * DIL)];
    const int c = blockIdx.y, n = blockIdx.z;
    const int ty0 = (blockIdx.x / tiles_x) * TH_, tx0 = (blockIdx.x % tiles_x) * TW_;
    const int ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 4;
    const long HW = (long)H * W;
    float acc[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q][j] = 0.f;
    for (int q = 0; q < mult; ++q) {
        const int pin = FLIP ? c : c * mult + q;
        if (!FLIP || q == 0) {
            __syncthreads();
            load_tile<K, DIL>(tile, in + (long)n * in_ns + (long)pin * HW, (FLIP && maskp_) ? maskp_ + (long)n * mask_ns + (long)c * HW : nullptr,
                         ty0, tx0, H, W);
            __syncthreads();
        }
        const float* wp = w + ((long)c * mult + q) * K * K;
        float* a = FLIP ? acc[q] : acc[0];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            float win[4 + (K - 1) * DIL];
#pragma unroll
            for (int i = 0; i < 4 + (K - 1) * DIL; ++i) win[i] = tile[(ly + ky * DIL) * LW + lx + i];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float wv = FLIP ? wp[(K - 1 - ky) * K + (K - 1 - kx)] : wp[ky * K + kx];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += wv * win[kx * DIL + j];
            }
        }
    }
    const int y = ty0 + ly;
    if (y >= H) return;
    if constexpr (FLIP) {
        for (int q = 0; q < mult; ++q) {
            float* op = out + (long)n * out_ns + ((long)c * mult + q) * HW + (long)y * W;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (tx0 + lx + j < W) op[tx0 + lx + j] = acc[q][j];
        }
    } else {
        const float bv = b ? b[c] : 0.f;
        float* op = out + (long)n * out_ns + (long)c * HW + (long)y * W;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (tx0 + lx + j < W) {
                const float v = acc[0][j] + bv;
                op[tx0 + lx + j] = relu ? fmaxf(v, 0.f) : v;
            }
    }
}

// weight gradient, tiled: partial[ci][n * tiles + tile][K*K] (+ bias partial as element K*K), then a fixed-order finish
template <int K, int DIL>
__global__ __launch_bounds__(256) void dwk_wgrad_tiled_kernel(const float* __restrict__ dy, long dy_ns, const float* __restrict__ yact,
                                                             long y_ns, const float* __restrict__ x, long x_ns, int mult, int H,
                                                             int W, int tiles_x, int tiles_y, int tpb, float* __restrict__ part) {
    constexpr int LW = TW_ + (K - 1) * DIL, KK = K * K;
    __shared__ float tile[(TW_ + (K - 1) * DIL) * (TH_ + (K - 1) * DIL)];
    __shared__ float red[4][KK + 1];
    const int ci = blockIdx.y, n = blockIdx.z, c = ci / mult;
    const int tx0 = (blockIdx.x % tiles_x) * TW_, tg = blockIdx.x / tiles_x;
    const int ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 4;
    const long HW = (long)H * W;
    float acc[KK + 1];
#pragma unroll
    for (int t = 0; t <= KK; ++t) acc[t] = 0.f;
    // a workgroup walks `tpb` vertically adjacent tiles and reduces once: the K*K + 1 wave reductions per tile were most of the
    // time of this kernel (K = 7: 568 us per launch at 8 x 48 x 256 x 256 with one tile per workgroup)
    for (int tt = tg * tpb; tt < min(tiles_y, (tg + 1) * tpb); ++tt) {
        const int ty0 = tt * TH_;
        load_tile<K, DIL>(tile, x + (long)n * x_ns + (long)ci * HW, nullptr, ty0, tx0, H, W);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        const int y = ty0 + ly;
        if (y < H) {
            const float* gp = dy + (long)n * dy_ns + (long)c * HW + (long)y * W;
            const float* ap = yact ? yact + (long)n * y_ns + (long)c * HW + (long)y * W : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = tx0 + lx + j;
                if (xx < W) g[j] = (!ap || ap[xx] > 0.f) ? gp[xx] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            float win[4 + (K - 1) * DIL];
#pragma unroll
            for (int i = 0; i < 4 + (K - 1) * DIL; ++i) win[i] = tile[(ly + ky * DIL) * LW + lx + i];
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
                acc[ky * K + kx] += (g[0] * win[kx * DIL] + g[1] * win[kx * DIL + 1]) + (g[2] * win[kx * DIL + 2] + g[3] * win[kx * DIL + 3]);
        }
        acc[KK] += (g[0] + g[1]) + (g[2] + g[3]);
        __syncthreads();                                      // the tile is reloaded
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t <= KK; ++t) {
        const float s = wave_sum(acc[t]);
        if (lane == 0) red[wv][t] = s;
    }
    __syncthreads();
    if (threadIdx.x <= KK) {
        const long slot = ((long)ci * gridDim.z + n) * gridDim.x + blockIdx.x;
        part[slot * (KK + 1) + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 5 x 5, dilation 1: data gradient and weight gradient in ONE pass over dy and x (the register-window scheme of tdr_dwsg.hip's
// one-pass backward).  A thread owns a 4-column strip of one input plane and walks `rpt` rows with two 5-row windows (x and the
// ReLU-masked dy) of 4 + 4 columns; the two columns either side come from the neighbouring lanes by DPP shuffles, or from memory
// at wave / row-block edges.  Rows are requested one iteration ahead.  Per row: dx = w^T (*) dy from the dy window, dW += centre
// dy row x the x window.  The tiled pair above reads dy twice and moves both operands through LDS; this reads each once.
// ---------------------------------------------------------------------------------------------------------------
struct Row8 { float v[8]; };
struct Raw8 { f32x4 m; float l0, l1, r0, r1; };

__device__ __forceinline__ Raw8 load_raw8(const float* __restrict__ plane, const float* __restrict__ maskp, int y, int x0, int H, int W,
                                          bool on, bool left_lane, bool right_lane) {
    Raw8 r;
    r.m = f32x4{0.f, 0.f, 0.f, 0.f};
    r.l0 = r.l1 = r.r0 = r.r1 = 0.f;
    if (on && y >= 0 && y < H) {
        const float* row = plane + (long)y * W;
        r.m = *reinterpret_cast<const f32x4*>(row + x0);
        if (!left_lane && x0 >= 4) { r.l0 = row[x0 - 2]; r.l1 = row[x0 - 1]; }
        if (!right_lane && x0 + 4 < W) { r.r0 = row[x0 + 4]; r.r1 = row[x0 + 5]; }          // W % 4 == 0: x0 + 5 < W too
        if (maskp) {
            const float* mr = maskp + (long)y * W;
            const f32x4 k = *reinterpret_cast<const f32x4*>(mr + x0);
#pragma unroll
            for (int e = 0; e < 4; ++e) r.m[e] = k[e] > 0.f ? r.m[e] : 0.f;
            if (!left_lane && x0 >= 4) { r.l0 = mr[x0 - 2] > 0.f ? r.l0 : 0.f; r.l1 = mr[x0 - 1] > 0.f ? r.l1 : 0.f; }
            if (!right_lane && x0 + 4 < W) { r.r0 = mr[x0 + 4] > 0.f ? r.r0 : 0.f; r.r1 = mr[x0 + 5] > 0.f ? r.r1 : 0.f; }
        }
    }
    return r;
}
__device__ __forceinline__ Row8 finish_row8(const Raw8& r, bool left_lane, bool right_lane) {
    float l0 = __shfl_up(r.m[2], 1, 64), l1 = __shfl_up(r.m[3], 1, 64);
    float r0 = __shfl_down(r.m[0], 1, 64), r1 = __shfl_down(r.m[1], 1, 64);
    if (!left_lane) { l0 = r.l0; l1 = r.l1; }
    if (!right_lane) { r0 = r.r0; r1 = r.r1; }
    Row8 o;
    o.v[0] = l0; o.v[1] = l1; o.v[2] = r.m[0]; o.v[3] = r.m[1]; o.v[4] = r.m[2]; o.v[5] = r.m[3]; o.v[6] = r0; o.v[7] = r1;
    return o;
}

struct Dw5Args {
    const float* dy; long dy_ns;
    const float* yact; long y_ns;
    const float* x; long x_ns;
    const float* w;
    float* dx; long dx_ns;
    float* part;
    int mult, H, W, tprw_log2, rpt, ncb;
    int accumulate;      // dx += instead of dx = (the sum of two branches' input gradients without a separate add pass)
};

__global__ __launch_bounds__(256) void dwk5_bwd_fused_kernel(Dw5Args a) {
    constexpr int K = 5, R = 2, KK = 25;
    __shared__ float red[4][KK + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int pl = blockIdx.y, n = blockIdx.z, co = pl / a.mult, H = a.H, W = a.W;
    const int TPRW = 1 << a.tprw_log2;
    const int cg = tid & (TPRW - 1), strip = tid >> a.tprw_log2;
    const int bx = blockIdx.x % a.ncb, by = blockIdx.x / a.ncb;
    const int x0 = (bx * TPRW + cg) * 4;
    const int ybeg = (by * (256 >> a.tprw_log2) + strip) * a.rpt;
    const bool active = x0 < W && ybeg < H;
    const bool left_lane = lane != 0 && cg != 0;
    const bool right_lane = lane != 63 && cg != TPRW - 1;
    const long HW = (long)H * W;
    const float* xp = a.x + (long)n * a.x_ns + (long)pl * HW;
    const float* dyp = a.dy + (long)n * a.dy_ns + (long)co * HW;
    const float* mp = a.yact ? a.yact + (long)n * a.y_ns + (long)co * HW : nullptr;
    float* dxp = a.dx + (long)n * a.dx_ns + (long)pl * HW;
    float w[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) w[i] = a.w[(long)pl * KK + i];
    float acc[KK + 1];
#pragma unroll
    for (int i = 0; i <= KK; ++i) acc[i] = 0.f;

    Row8 xw[K], gw[K];
#pragma unroll
    for (int k = 0; k < K - 1; ++k) {                            // rows ybeg - 2 .. ybeg + 1
        xw[k] = finish_row8(load_raw8(xp, nullptr, ybeg - R + k, x0, H, W, active, left_lane, right_lane), left_lane, right_lane);
        gw[k] = finish_row8(load_raw8(dyp, mp, ybeg - R + k, x0, H, W, active, left_lane, right_lane), left_lane, right_lane);
    }
    Raw8 nx = load_raw8(xp, nullptr, ybeg + R, x0, H, W, active, left_lane, right_lane);
    Raw8 ng = load_raw8(dyp, mp, ybeg + R, x0, H, W, active, left_lane, right_lane);
    for (int i = 0; i < a.rpt; ++i) {
        const int yc = ybeg + i;                                 // centre row of both windows
        xw[K - 1] = finish_row8(nx, left_lane, right_lane);
        gw[K - 1] = finish_row8(ng, left_lane, right_lane);
        const bool more = active && i + 1 < a.rpt;
        nx = load_raw8(xp, nullptr, yc + R + 1, x0, H, W, more, left_lane, right_lane);
        ng = load_raw8(dyp, mp, yc + R + 1, x0, H, W, more, left_lane, right_lane);
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] += w[KK - 1 - (ky * K + kx)] * gw[ky].v[e + kx];                   // transposed conv: flipped taps
                    acc[ky * K + kx] += gw[R].v[e + R] * xw[ky].v[e + kx];
                }
        acc[KK] += (gw[R].v[2] + gw[R].v[3]) + (gw[R].v[4] + gw[R].v[5]);
        if (active && yc < H) {
            f32x4 v = {o[0], o[1], o[2], o[3]};
            if (a.accumulate) v += *reinterpret_cast<const f32x4*>(dxp + (long)yc * W + x0);
            *reinterpret_cast<f32x4*>(dxp + (long)yc * W + x0) = v;
        }
#pragma unroll
        for (int k = 0; k < K - 1; ++k) { xw[k] = xw[k + 1]; gw[k] = gw[k + 1]; }
    }
#pragma unroll
    for (int i = 0; i <= KK; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[tid >> 6][i] = s;
    }
    __syncthreads();
    if (tid <= KK) {
        const long slot = ((long)pl * gridDim.z + n) * gridDim.x + blockIdx.x;
        a.part[slot * (KK + 1) + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// one workgroup per input plane: fixed-order sum of its nparts partial rows
__global__ __launch_bounds__(256) void dwk_wgrad_finish_kernel(const float* __restrict__ part, int nparts, int KK, int mult,
                                                              float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[256];
    const int ci = blockIdx.x;
    for (int t = 0; t <= KK; ++t) {
        float s = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 256) s += part[((long)ci * nparts + i) * (KK + 1) + t];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (t < KK) dw[(long)ci * KK + t] = red[0];
            else if (db && ci % mult == 0) db[ci / mult] = red[0];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small pieces of DRSformer-ref's MEFC sub-network (network_drsformer_guided_arch.py:371-548)
// ---------------------------------------------------------------------------------------------------------------
// nn.AvgPool2d(3, stride=1, padding=1, count_include_pad=False) and its adjoint
__global__ void avgpool3_kernel(const float* __restrict__ in, int H, int W, long total, int adjoint, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const float* p = in + (i - (long)y * W - x);
        float s = 0.f;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                if (adjoint) {       // dx[p] = sum_{q in N(p)} dy[q] / count(q)
                    const int cy = min(yy + 1, H - 1) - max(yy - 1, 0) + 1, cx = min(xx + 1, W - 1) - max(xx - 1, 0) + 1;
                    s += p[(long)yy * W + xx] / (float)(cy * cx);
                } else {
                    s += p[(long)yy * W + xx];
                }
            }
        }
        if (!adjoint) {
            const int cy = min(y + 1, H - 1) - max(y - 1, 0) + 1, cx = min(x + 1, W - 1) - max(x - 1, 0) + 1;
            s /= (float)(cy * cx);
        }
        out[i] = s;
    }
}

// W % 4 == 0: a thread owns 4 adjacent outputs of one row -- three float4 rows + two edge values each instead of 36 scalar loads,
// no per-element integer division (row sums first, then the three rows: rounding differs from the scalar kernel in the last bit).
__global__ __launch_bounds__(256) void avgpool3_vec_kernel(const float* __restrict__ in, int H, int W, long planes, int adjoint,
                                                          float* __restrict__ out) {
    const int qpr = W >> 2;                                   // quads per row
    const long quads = planes * H * qpr;
    for (long q = blockIdx.x * 256L + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
        const int qx = (int)(q % qpr);
        const long row = q / qpr;
        const int y = (int)(row % H), x0 = qx * 4;
        const float* base = in + (row - y) * W;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            const float* r = base + (long)yy * W;
            const f32x4 m = *reinterpret_cast<const f32x4*>(r + x0);
            float v[6] = {x0 > 0 ? r[x0 - 1] : 0.f, m[0], m[1], m[2], m[3], x0 + 4 < W ? r[x0 + 4] : 0.f};
            if (adjoint) {
                const float cy = (float)(min(yy + 1, H - 1) - max(yy - 1, 0) + 1);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int xx = x0 - 1 + k;
                    const int cx = min(xx + 1, W - 1) - max(xx - 1, 0) + 1;
                    v[k] = (xx >= 0 && xx < W) ? v[k] / (cy * (float)cx) : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += (v[e] + v[e + 1]) + v[e + 2];
        }
        if (!adjoint) {
            const int cy = min(y + 1, H - 1) - max(y - 1, 0) + 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = x0 + e, cx = min(x + 1, W - 1) - max(x - 1, 0) + 1;
                s[e] /= (float)(cy * cx);
            }
        }
        *reinterpret_cast<f32x4*>(out + row * W + x0) = f32x4{s[0], s[1], s[2], s[3]};
    }
}

// y[n][o] = act(b[o] + sum_i W[o][i] x[n][i]); one wave per output element
__global__ __launch_bounds__(64) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                             const float* __restrict__ b, int Cin, int Cout, int relu,
                                                             float* __restrict__ y) {
    const int o = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    float s = 0.f;
    for (int i = lane; i < Cin; i += 64) s += W[(long)o * Cin + i] * x[(long)n * Cin + i];
    s = wave_sum(s);
    if (lane == 0) {
        s += b ? b[o] : 0.f;
        y[(long)n * Cout + o] = relu ? fmaxf(s, 0.f) : s;
    }
}
// single workgroup: g = dy (* [y > 0]); dW[o][i] = sum_n g[n][o] x[n][i]; db[o] = sum_n g[n][o]; dx[n][i] = sum_o g[n][o] W[o][i]
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                              const float* __restrict__ x, const float* __restrict__ W, int N,
                                                              int Cin, int Cout, float* __restrict__ dx, float* __restrict__ dW,
                                                              float* __restrict__ db) {
    extern __shared__ float g[];          // [N][Cout]
    for (int i = threadIdx.x; i < N * Cout; i += 256) g[i] = (!yact || yact[i] > 0.f) ? dy[i] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < Cout * Cin; i += 256) {
        const int o = i / Cin, c = i - o * Cin;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += g[n * Cout + o] * x[(long)n * Cin + c];
        dW[i] = s;
    }
    for (int o = threadIdx.x; o < Cout; o += 256) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += g[n * Cout + o];
        if (db) db[o] = s;
    }
    for (int i = threadIdx.x; i < N * Cin; i += 256) {
        const int n = i / Cin, c = i - n * Cin;
        float s = 0.f;
        for (int o = 0; o < Cout; ++o) s += g[n * Cout + o] * W[(long)o * Cin + c];
        dx[i] = s;
    }
}

// softmax over rows of length L (L <= 64), forward and backward (one thread per row)
__global__ void softmax_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy, int rows, int L, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = x + (long)r * L;
    if (!dy) {
        float m = -INFINITY, z = 0.f;
        for (int i = 0; i < L; ++i) m = fmaxf(m, p[i]);
        for (int i = 0; i < L; ++i) z += expf(p[i] - m);
        for (int i = 0; i < L; ++i) out[(long)r * L + i] = expf(p[i] - m) / z;
    } else {                 // x = the softmax output here
        float dot = 0.f;
        for (int i = 0; i < L; ++i) dot += p[i] * dy[(long)r * L + i];
        for (int i = 0; i < L; ++i) out[(long)r * L + i] = p[i] * (dy[(long)r * L + i] - dot);
    }
}

// dst[n][:len] = src[n][:len] * w[n * w_stride]
__global__ void scale_copy_kernel(const float* __restrict__ src, long src_ns, const float* __restrict__ w, int w_stride, long len,
                                  long total, float* __restrict__ dst, long dst_ns) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / len, e = i - n * len;
        dst[n * dst_ns + e] = src[n * src_ns + e] * w[n * w_stride];
    }
}

__global__ void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = fmaxf(a[i] + b[i], 0.f);
}

constexpr int RD_BLOCKS = 64;
// part[n][b] = sum over block b's slice of a[n] . b[n]; finish sums the RD_BLOCKS partials in order
__global__ __launch_bounds__(256) void rows_dot_kernel(const float* __restrict__ a, long a_ns, const float* __restrict__ b, long b_ns,
                                                      long len, float* __restrict__ part) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const long per = (len + RD_BLOCKS - 1) / RD_BLOCKS, e0 = blockIdx.x * per, e1 = e0 + per < len ? e0 + per : len;
    float s = 0.f;
    const float* pa = a + n * a_ns;
    const float* pb = b + n * b_ns;
    if ((per & 3) == 0 && (((uintptr_t)pa | (uintptr_t)pb) & 15) == 0) {      // 16-byte path: 4 x fewer, independent loads
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (long e = e0 + threadIdx.x * 4L; e + 3 < e1; e += 1024) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(pa + e), v = *reinterpret_cast<const f32x4*>(pb + e);
            acc += u * v;
        }
        s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        for (long e = e0 + ((e1 - e0) & ~3L) + threadIdx.x; e < e1; e += 256) s += pa[e] * pb[e];
    } else {
        for (long e = e0 + threadIdx.x; e < e1; e += 256) s += pa[e] * pb[e];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)n * RD_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void rows_dot_finish_kernel(const float* __restrict__ part, int N, int out_stride, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int b = 0; b < RD_BLOCKS; ++b) s += part[(long)n * RD_BLOCKS + b];
    out[(long)n * out_stride] = s;
}

inline int dgrid(long total, int cap = 16384) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

#define DWK_CASES(X)  X(1, 1) X(3, 1) X(5, 1) X(7, 1) X(3, 2) X(5, 2) X(7, 2)

extern "C" int tdr_dwk_fwd(const float* x, int64_t x_ns, const float* w, const float* b, int N, int Cout, int mult, int H, int W, int K,
                           int dil, int relu, float* y, int64_t y_ns, void* stream) {
    TDR_REQUIRE(x && w && y && N > 0 && Cout > 0 && (mult == 1 || mult == 2) && (K == 1 || K == 3 || K == 5 || K == 7) && (dil == 1 || (dil == 2 && K > 1)),
                "tdr_dwk_fwd: bad argument (mult 1|2, K 1|3|5|7, dil 1|2; got mult=%d K=%d dil=%d)", mult, K, dil);
    const int tiles_x = tdr_cdiv(W, TW_), tiles = tiles_x * tdr_cdiv(H, TH_);
#define X(K_, D_) if (K == K_ && dil == D_) hipLaunchKernelGGL((dwk_tiled_kernel<K_, D_, false>), dim3(tiles, Cout, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, nullptr, 0L, w, b, Cout, mult, H, W, tiles_x, relu, y, (long)y_ns);
    DWK_CASES(X)
#undef X
    TDR_LAUNCH_CHECK("dwk_tiled_fwd");
    return TDR_OK;
}

extern "C" int64_t tdr_dwk_bwd_ws_floats(int N, int Cout, int mult, int H, int W, int K) {
    return (int64_t)Cout * mult * N * tdr_cdiv(W, TW_) * tdr_cdiv(H, TH_) * (K * K + 1);
}

extern "C" int tdr_dwk_bwd(const float* dy, int64_t dy_ns, const float* yact, int64_t y_ns, const float* x, int64_t x_ns, const float* w,
                           int N, int Cout, int mult, int H, int W, int K, int dil, float* dx, int64_t dx_ns, float* dw, float* db,
                           float* ws, void* stream) {
    return tdr_dwk_bwd_acc(dy, dy_ns, yact, y_ns, x, x_ns, w, N, Cout, mult, H, W, K, dil, dx, dx_ns, 0, dw, db, ws, stream);
}

// accumulate != 0: dx += (one-pass 5x5 kernel only; tdr_dwk_bwd_can_accumulate tells whether these arguments take it)
extern "C" int tdr_dwk_bwd_can_accumulate(int W, int K, int dil, int64_t dy_ns, int64_t y_ns, int64_t x_ns, int64_t dx_ns) {
    return K == 5 && dil == 1 && W % 4 == 0 && ((dy_ns | y_ns | x_ns | dx_ns) & 3) == 0 && tdr_tune_env("TDR_DWK_TILED") == nullptr;
}

extern "C" int tdr_dwk_bwd_acc(const float* dy, int64_t dy_ns, const float* yact, int64_t y_ns, const float* x, int64_t x_ns,
                               const float* w, int N, int Cout, int mult, int H, int W, int K, int dil, float* dx, int64_t dx_ns,
                               int accumulate, float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(dy && x && w && dx && dw && ws && N > 0 && Cout > 0 && (mult == 1 || mult == 2) && (K == 1 || K == 3 || K == 5 || K == 7) &&
                    (dil == 1 || (dil == 2 && K > 1)), "tdr_dwk_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = tdr_cdiv(W, TW_), tiles_y = tdr_cdiv(H, TH_), tiles = tiles_x * tiles_y;
    const int tpb = 8, wtiles = tiles_x * tdr_cdiv(tiles_y, tpb);                          // weight-gradient workgroups per plane
    static const bool tiled_only = tdr_tune_env("TDR_DWK_TILED") != nullptr;                     // A/B aid: the two-kernel backward
    if (K == 5 && dil == 1 && W % 4 == 0 && !tiled_only && ((dy_ns | y_ns | x_ns | dx_ns) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
          reinterpret_cast<uintptr_t>(yact)) & 15) == 0) {
        int groups = W / 4, lg = 0;
        while ((1 << lg) < groups && lg < 8) ++lg;
        const int spb = 256 >> lg, ncb = tdr_cdiv(groups, 1 << lg);
        int rpt = tdr_cdiv(H, spb);
        if (rpt > 32) rpt = 32;
        const int nb = ncb * tdr_cdiv(H, spb * rpt);
        if (nb <= tiles) {                                                              // (the partials fit the tiled layout's scratch)
            Dw5Args a{dy, (long)dy_ns, yact, (long)y_ns, x, (long)x_ns, w, dx, (long)dx_ns, ws, mult, H, W, lg, rpt, ncb, accumulate};
            hipLaunchKernelGGL(dwk5_bwd_fused_kernel, dim3(nb, Cout * mult, N), dim3(256), 0, st, a);
            hipLaunchKernelGGL(dwk_wgrad_finish_kernel, dim3(Cout * mult), dim3(256), 0, st, ws, N * nb, K * K, mult, dw, db);
            TDR_LAUNCH_CHECK("dwk5_bwd_fused");
            return TDR_OK;
        }
    }
    TDR_REQUIRE(!accumulate, "tdr_dwk_bwd_acc: accumulate needs the one-pass 5x5 kernel (see tdr_dwk_bwd_can_accumulate)");
#define X(K_, D_)                                                                                                                          \
    if (K == K_ && dil == D_) {                                                                                                            \
        hipLaunchKernelGGL((dwk_tiled_kernel<K_, D_, true>), dim3(tiles, Cout, N), dim3(256), 0, st, dy, (long)dy_ns, yact, (long)y_ns, w, \
                           nullptr, Cout, mult, H, W, tiles_x, 0, dx, (long)dx_ns);                                                        \
        hipLaunchKernelGGL((dwk_wgrad_tiled_kernel<K_, D_>), dim3(wtiles, Cout * mult, N), dim3(256), 0, st, dy, (long)dy_ns, yact,        \
                           (long)y_ns, x, (long)x_ns, mult, H, W, tiles_x, tiles_y, tpb, ws);                                              \
    }
    DWK_CASES(X)
#undef X
    hipLaunchKernelGGL(dwk_wgrad_finish_kernel, dim3(Cout * mult), dim3(256), 0, st, ws, N * wtiles, K * K, mult, dw, db);
    TDR_LAUNCH_CHECK("dwk_tiled_bwd");
    return TDR_OK;
}

extern "C" int tdr_avgpool3(const float* in, int planes, int H, int W, int adjoint, float* out, void* stream) {
    TDR_REQUIRE(in && out && planes > 0 && H > 0 && W > 0, "tdr_avgpool3: bad argument");
    const long total = (long)planes * H * W;
    if (W % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
        hipLaunchKernelGGL(avgpool3_vec_kernel, dim3(dgrid(total / 4)), dim3(256), 0, (hipStream_t)stream, in, H, W, (long)planes, adjoint, out);
    else
        hipLaunchKernelGGL(avgpool3_kernel, dim3(dgrid(total)), dim3(256), 0, (hipStream_t)stream, in, H, W, total, adjoint, out);
    TDR_LAUNCH_CHECK("avgpool3");
    return TDR_OK;
}

extern "C" int tdr_linear_small_fwd(const float* x, const float* W, const float* b, int N, int Cin, int Cout, int relu, float* y,
                                    void* stream) {
    TDR_REQUIRE(x && W && y && N > 0 && Cin > 0 && Cout > 0, "tdr_linear_small_fwd: bad argument");
    hipLaunchKernelGGL(linear_small_fwd_kernel, dim3(Cout, N), dim3(64), 0, (hipStream_t)stream, x, W, b, Cin, Cout, relu, y);
    TDR_LAUNCH_CHECK("linear_small_fwd");
    return TDR_OK;
}

extern "C" int tdr_linear_small_bwd(const float* dy, const float* yact, const float* x, const float* W, int N, int Cin, int Cout,
                                    float* dx, float* dW, float* db, void* stream) {
    TDR_REQUIRE(dy && x && W && dx && dW && N > 0 && (long)N * Cout <= 16384, "tdr_linear_small_bwd: bad argument (N * Cout <= 16384)");
    hipLaunchKernelGGL(linear_small_bwd_kernel, dim3(1), dim3(256), (size_t)N * Cout * sizeof(float), (hipStream_t)stream, dy, yact, x, W, N,
                       Cin, Cout, dx, dW, db);
    TDR_LAUNCH_CHECK("linear_small_bwd");
    return TDR_OK;
}

extern "C" int tdr_softmax_rows(const float* x, const float* dy, int rows, int L, float* out, void* stream) {
    TDR_REQUIRE(x && out && rows > 0 && L > 0 && L <= 64, "tdr_softmax_rows: bad argument (L <= 64)");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(tdr_cdiv(rows, 64)), dim3(64), 0, (hipStream_t)stream, x, dy, rows, L, out);
    TDR_LAUNCH_CHECK("softmax_rows");
    return TDR_OK;
}

extern "C" int tdr_scale_copy(const float* src, int64_t src_ns, const float* w, int w_stride, int N, int64_t len, float* dst,
                              int64_t dst_ns, void* stream) {
    TDR_REQUIRE(src && w && dst && N > 0 && len > 0, "tdr_scale_copy: bad argument");
    const long total = (long)N * len;
    hipLaunchKernelGGL(scale_copy_kernel, dim3(dgrid(total)), dim3(256), 0, (hipStream_t)stream, src, (long)src_ns, w, w_stride, (long)len,
                       total, dst, (long)dst_ns);
    TDR_LAUNCH_CHECK("scale_copy");
    return TDR_OK;
}

extern "C" int tdr_rows_dot(const float* a, int64_t a_ns, const float* b, int64_t b_ns, int N, int64_t len, float* out, int out_stride,
                            float* ws, void* stream) {
    TDR_REQUIRE(a && b && out && ws && N > 0 && len > 0, "tdr_rows_dot: bad argument (ws: 64 * N floats)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rows_dot_kernel, dim3(RD_BLOCKS, N), dim3(256), 0, st, a, (long)a_ns, b, (long)b_ns, (long)len, ws);
    hipLaunchKernelGGL(rows_dot_finish_kernel, dim3(tdr_cdiv(N, 64)), dim3(64), 0, st, ws, N, out_stride, out);
    TDR_LAUNCH_CHECK("rows_dot");
    return TDR_OK;
}

extern "C" int tdr_add_relu(const float* a, const float* b, int64_t numel, float* out, void* stream) {
    TDR_REQUIRE(a && b && out && numel > 0, "tdr_add_relu: bad argument");
    hipLaunchKernelGGL(add_relu_kernel, dim3(dgrid(numel)), dim3(256), 0, (hipStream_t)stream, a, b, (long)numel, out);
    TDR_LAUNCH_CHECK("add_relu");
    return TDR_OK;
}
