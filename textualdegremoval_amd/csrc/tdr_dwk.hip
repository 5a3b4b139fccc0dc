// Generic grouped "depthwise-like" convolutions of DRSformer-ref's mixed-scale feed-forward (network_drsformer_guided_arch*.py
// :216-253): K x K (K = 3, 5), stride 1, pad K/2, no bias, groups = Cout with `mult` = 1 or 2 input planes per output plane
// (dwconv3x3 / dwconv5x5: mult 1 over 2h planes; dwconv3x3_1 / dwconv5x5_1: Conv2d(2h, h, groups=h), mult 2), optional ReLU:
//     y[n][c] = act( sum_{i < mult} w[c][i] (*) x[n][c * mult + i] )
// Straightforward streaming kernels (HBM-bound stencils: each input plane is read once per launch, neighbours come from
// L1/L2); the weight gradient is one workgroup per (c, i) plane pair with a fixed-order in-block reduction (deterministic).
// First implementation of this family: correctness and the C ABI first, tiling later.
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

__global__ void dwk_fwd_kernel(const float* __restrict__ x, long x_ns, const float* __restrict__ w, const float* __restrict__ b, int Cout, int mult, int H, int W,
                               int K, int relu, long total, float* __restrict__ y, long y_ns) {
    const int pad = K / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % W);
        long r = i / W;
        const int py = (int)(r % H); r /= H;
        const int c = (int)(r % Cout);
        const long n = r / Cout;
        float s = b ? b[c] : 0.f;
        for (int q = 0; q < mult; ++q) {
            const float* xp = x + n * x_ns + ((long)c * mult + q) * H * W;
            const float* wp = w + ((long)c * mult + q) * K * K;
            for (int ky = 0; ky < K; ++ky) {
                const int yy = py + ky - pad;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < K; ++kx) {
                    const int xx = px + kx - pad;
                    if (xx >= 0 && xx < W) s += wp[ky * K + kx] * xp[(long)yy * W + xx];
                }
            }
        }
        y[n * y_ns + ((long)c * H + py) * W + px] = relu ? fmaxf(s, 0.f) : s;
    }
}

// dx[n][c*mult + q][p] = sum_taps w[c][q][ky][kx] * g[n][c][p - (tap - pad)],  g = dy (* [y > 0])
__global__ void dwk_bwd_data_kernel(const float* __restrict__ dy, long dy_ns, const float* __restrict__ yact, long y_ns,
                                    const float* __restrict__ w, int Cout, int mult, int H, int W, int K, long total,
                                    float* __restrict__ dx, long dx_ns) {
    const int pad = K / 2;
    const int Cin = Cout * mult;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % W);
        long r = i / W;
        const int py = (int)(r % H); r /= H;
        const int ci = (int)(r % Cin);
        const long n = r / Cin;
        const int c = ci / mult;
        const float* gp = dy + n * dy_ns + (long)c * H * W;
        const float* ap = yact ? yact + n * y_ns + (long)c * H * W : nullptr;
        const float* wp = w + (long)ci * K * K;
        float s = 0.f;
        for (int ky = 0; ky < K; ++ky) {
            const int yy = py - (ky - pad);
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int xx = px - (kx - pad);
                if (xx < 0 || xx >= W) continue;
                const long o = (long)yy * W + xx;
                const float g = (!ap || ap[o] > 0.f) ? gp[o] : 0.f;
                s += wp[ky * K + kx] * g;
            }
        }
        dx[n * dx_ns + ((long)ci * H + py) * W + px] = s;
    }
}

// one workgroup per input plane ci (= c * mult + q): dw[ci][tap] = sum_{n, p} g[n][c][p] * x[n][ci][p + tap - pad]
template <int K>
__global__ __launch_bounds__(256) void dwk_bwd_weight_kernel(const float* __restrict__ dy, long dy_ns, const float* __restrict__ yact,
                                                            long y_ns, const float* __restrict__ x, long x_ns, int N, int mult, int H,
                                                            int W, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[4][K * K];
    __shared__ float redb[4];
    float accb = 0.f;
    const int ci = blockIdx.x, c = ci / mult;
    constexpr int pad = K / 2;
    const long HW = (long)H * W;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
    for (long i = threadIdx.x; i < N * HW; i += 256) {
        const long n = i / HW;
        const long p = i - n * HW;
        const int py = (int)(p / W), px = (int)(p - (long)py * W);
        float g = dy[n * dy_ns + c * HW + p];
        if (yact && !(yact[n * y_ns + c * HW + p] > 0.f)) g = 0.f;
        accb += g;
        const float* xp = x + n * x_ns + (long)ci * HW;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yy = py + ky - pad;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int xx = px + kx - pad;
                const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                acc[ky * K + kx] += ok ? g * xp[(long)yy * W + xx] : 0.f;
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const float s = wave_sum(acc[t]);
        if (lane == 0) red[wv][t] = s;
    }
    {
        const float sb = wave_sum(accb);
        if (lane == 0) redb[wv] = sb;
    }
    __syncthreads();
    if (db && threadIdx.x == 0 && ci == c * mult) db[c] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
    if (threadIdx.x < K * K) dw[(long)ci * K * K + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled versions for K = 3 / 5 (the shapes DRSformer-ref uses): a workgroup owns a 64 x 16 output tile of one
// (image, output plane); the input tile + halo of each of its `mult` input planes goes through LDS once, a thread computes 4
// adjacent outputs from a (4 + K - 1)-wide register window per kernel row.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TW_ = 64, TH_ = 16;

template <int K>
__device__ __forceinline__ void load_tile(float* __restrict__ t, const float* __restrict__ plane, const float* __restrict__ maskp,
                                          int y0, int x0, int H, int W) {
    constexpr int LW = TW_ + K - 1, LH = TH_ + K - 1, P = K / 2;
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
template <int K, bool FLIP>
__global__ __launch_bounds__(256) void dwk_tiled_kernel(const float* __restrict__ in, long in_ns, const float* __restrict__ maskp_,
                                                       long mask_ns, const float* __restrict__ w, const float* __restrict__ b,
                                                       int Cout, int mult, int H, int W, int tiles_x, int relu,
                                                       float* __restrict__ out, long out_ns) {
    constexpr int LW = TW_ + K - 1;
    __shared__ float tile[(TW_ + K - 1) * (TH_ + K - 1)];
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
            load_tile<K>(tile, in + (long)n * in_ns + (long)pin * HW, (FLIP && maskp_) ? maskp_ + (long)n * mask_ns + (long)c * HW : nullptr,
                         ty0, tx0, H, W);
            __syncthreads();
        }
        const float* wp = w + ((long)c * mult + q) * K * K;
        float* a = FLIP ? acc[q] : acc[0];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            float win[4 + K - 1];
#pragma unroll
            for (int i = 0; i < 4 + K - 1; ++i) win[i] = tile[(ly + ky) * LW + lx + i];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float wv = FLIP ? wp[(K - 1 - ky) * K + (K - 1 - kx)] : wp[ky * K + kx];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += wv * win[kx + j];
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
template <int K>
__global__ __launch_bounds__(256) void dwk_wgrad_tiled_kernel(const float* __restrict__ dy, long dy_ns, const float* __restrict__ yact,
                                                             long y_ns, const float* __restrict__ x, long x_ns, int mult, int H,
                                                             int W, int tiles_x, float* __restrict__ part) {
    constexpr int LW = TW_ + K - 1, KK = K * K;
    __shared__ float tile[(TW_ + K - 1) * (TH_ + K - 1)];
    __shared__ float red[4][KK + 1];
    const int ci = blockIdx.y, n = blockIdx.z, c = ci / mult;
    const int ty0 = (blockIdx.x / tiles_x) * TH_, tx0 = (blockIdx.x % tiles_x) * TW_;
    const int ly = threadIdx.x >> 4, lx = (threadIdx.x & 15) * 4;
    const long HW = (long)H * W;
    load_tile<K>(tile, x + (long)n * x_ns + (long)ci * HW, nullptr, ty0, tx0, H, W);
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
    float acc[KK + 1];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        float win[4 + K - 1];
#pragma unroll
        for (int i = 0; i < 4 + K - 1; ++i) win[i] = tile[(ly + ky) * LW + lx + i];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc[ky * K + kx] = (g[0] * win[kx] + g[1] * win[kx + 1]) + (g[2] * win[kx + 2] + g[3] * win[kx + 3]);
    }
    acc[KK] = (g[0] + g[1]) + (g[2] + g[3]);
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

inline int dgrid(long total, int cap = 16384) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

extern "C" int tdr_dwk_fwd(const float* x, int64_t x_ns, const float* w, const float* b, int N, int Cout, int mult, int H, int W, int K,
                           int relu, float* y, int64_t y_ns, void* stream) {
    TDR_REQUIRE(x && w && y && N > 0 && Cout > 0 && (mult == 1 || mult == 2) && (K == 1 || K == 3 || K == 5 || K == 7),
                "tdr_dwk_fwd: bad argument (mult 1|2, K 1|3|5|7; got mult=%d K=%d)", mult, K);
    if (K == 3 || K == 5) {
        const int tiles_x = tdr_cdiv(W, TW_), tiles = tiles_x * tdr_cdiv(H, TH_);
        if (K == 3) hipLaunchKernelGGL((dwk_tiled_kernel<3, false>), dim3(tiles, Cout, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, nullptr, 0L, w, b, Cout, mult, H, W, tiles_x, relu, y, (long)y_ns);
        else hipLaunchKernelGGL((dwk_tiled_kernel<5, false>), dim3(tiles, Cout, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, nullptr, 0L, w, b, Cout, mult, H, W, tiles_x, relu, y, (long)y_ns);
        TDR_LAUNCH_CHECK("dwk_tiled_fwd");
        return TDR_OK;
    }
    const long total = (long)N * Cout * H * W;
    hipLaunchKernelGGL(dwk_fwd_kernel, dim3(dgrid(total)), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, w, b, Cout, mult, H, W, K, relu,
                       total, y, (long)y_ns);
    TDR_LAUNCH_CHECK("dwk_fwd");
    return TDR_OK;
}

extern "C" int64_t tdr_dwk_bwd_ws_floats(int N, int Cout, int mult, int H, int W, int K) {
    if (K != 3 && K != 5) return 0;
    return (int64_t)Cout * mult * N * tdr_cdiv(W, TW_) * tdr_cdiv(H, TH_) * (K * K + 1);
}

extern "C" int tdr_dwk_bwd(const float* dy, int64_t dy_ns, const float* yact, int64_t y_ns, const float* x, int64_t x_ns, const float* w,
                           int N, int Cout, int mult, int H, int W, int K, float* dx, int64_t dx_ns, float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(dy && x && w && dx && dw && N > 0 && Cout > 0 && (mult == 1 || mult == 2) && (K == 1 || K == 3 || K == 5 || K == 7),
                "tdr_dwk_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if ((K == 3 || K == 5) && ws) {
        const int tiles_x = tdr_cdiv(W, TW_), tiles = tiles_x * tdr_cdiv(H, TH_);
        if (K == 3) {
            hipLaunchKernelGGL((dwk_tiled_kernel<3, true>), dim3(tiles, Cout, N), dim3(256), 0, st, dy, (long)dy_ns, yact, (long)y_ns, w, nullptr, Cout, mult, H, W, tiles_x, 0, dx, (long)dx_ns);
            hipLaunchKernelGGL(dwk_wgrad_tiled_kernel<3>, dim3(tiles, Cout * mult, N), dim3(256), 0, st, dy, (long)dy_ns, yact, (long)y_ns, x, (long)x_ns, mult, H, W, tiles_x, ws);
        } else {
            hipLaunchKernelGGL((dwk_tiled_kernel<5, true>), dim3(tiles, Cout, N), dim3(256), 0, st, dy, (long)dy_ns, yact, (long)y_ns, w, nullptr, Cout, mult, H, W, tiles_x, 0, dx, (long)dx_ns);
            hipLaunchKernelGGL(dwk_wgrad_tiled_kernel<5>, dim3(tiles, Cout * mult, N), dim3(256), 0, st, dy, (long)dy_ns, yact, (long)y_ns, x, (long)x_ns, mult, H, W, tiles_x, ws);
        }
        hipLaunchKernelGGL(dwk_wgrad_finish_kernel, dim3(Cout * mult), dim3(256), 0, st, ws, N * tiles, K * K, mult, dw, db);
        TDR_LAUNCH_CHECK("dwk_tiled_bwd");
        return TDR_OK;
    }
    const long total = (long)N * Cout * mult * H * W;
    hipLaunchKernelGGL(dwk_bwd_data_kernel, dim3(dgrid(total)), dim3(256), 0, st, dy, (long)dy_ns, yact, (long)y_ns, w, Cout, mult, H, W, K,
                       total, dx, (long)dx_ns);
#define DWK_W(K_) hipLaunchKernelGGL(dwk_bwd_weight_kernel<K_>, dim3(Cout * mult), dim3(256), 0, st, dy, (long)dy_ns, yact, (long)y_ns, x, (long)x_ns, N, mult, H, W, dw, db)
    if (K == 1) DWK_W(1); else if (K == 3) DWK_W(3); else if (K == 5) DWK_W(5); else DWK_W(7);
#undef DWK_W
    TDR_LAUNCH_CHECK("dwk_bwd");
    return TDR_OK;
}
