// Inference-time pieces of the validation path, gfx950.
//
// (1) TLSC local average pooling -- `AvgPool2d.forward` of models/archs/nafnet_local_arch.py:10-75 (fast_imp = False): the global
//     average pool of a NAFBlock's SCA branch becomes a k1 x k2 box mean (k fixed at 1.5 x the training feature size), computed
//     by the reference through a 2-D integral image in fp32 and replicate-padded back to H x W.  Here: a vertical sliding sum
//     (lanes along W, coalesced; double accumulator) and a horizontal pass from a per-row prefix in LDS (double) -- no fp32
//     cancellation of two large integral-image entries, so the result is the exact box mean to fp32 rounding.
// (2) Y-channel SSIM in float64 -- `_ssim_cly` (metrics/psnr_ssim.py:184-222): the five 11 x 11 Gaussian-filtered fields, the SSIM
//     map and its mean are evaluated in double like the reference (cv2.filter2D on float64 images, BORDER_REPLICATE).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

// tmp[p][y'][x] = sum_{r < k1} in[p][y' + r][x],  y' in [0, H - k1]
__global__ void tlsc_vsum_kernel(const float* __restrict__ in, int H, int W, int k1, float* __restrict__ tmp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const long p = blockIdx.y;
    const float* src = in + p * (long)H * W + x;
    float* dst = tmp + p * (long)(H - k1 + 1) * W + x;
    double s = 0.0;
    for (int y = 0; y < H; ++y) {
        s += (double)src[(long)y * W];
        if (y >= k1) s -= (double)src[(long)(y - k1) * W];
        if (y >= k1 - 1) dst[(long)(y - k1 + 1) * W] = (float)s;
    }
}

// out[p][Y][X] = box mean at (clamp(Y - padt), clamp(X - padl)): one workgroup per (plane, output row)
__global__ __launch_bounds__(256) void tlsc_hmean_kernel(const float* __restrict__ tmp, int H, int W, int k1, int k2, float* __restrict__ out) {
    extern __shared__ double pre[];                 // [W + 1] prefix of the source row, then 256 chunk sums
    double* chunk = pre + W + 1;
    const int Y = blockIdx.x, tid = threadIdx.x;
    const long p = blockIdx.y;
    const int hv = H - k1 + 1, wv = W - k2 + 1;
    const int padt = (H - hv) / 2, padl = (W - wv) / 2;
    const int ys = min(max(Y - padt, 0), hv - 1);
    const float* row = tmp + (p * hv + ys) * (long)W;
    const int per = (W + 255) / 256, x0 = tid * per, x1 = min(x0 + per, W);
    double s = 0.0;
    for (int x = x0; x < x1; ++x) s += (double)row[x];
    chunk[tid] = s;
    __syncthreads();
    if (tid == 0) {
        double a = 0.0;
        for (int i = 0; i < 256; ++i) { const double c = chunk[i]; chunk[i] = a; a += c; }
    }
    __syncthreads();
    double a = chunk[tid];
    if (tid == 0) pre[0] = 0.0;
    for (int x = x0; x < x1; ++x) { a += (double)row[x]; pre[x + 1] = a; }
    __syncthreads();
    const double inv = 1.0 / ((double)k1 * (double)k2);
    float* o = out + (p * H + Y) * (long)W;
    for (int X = tid; X < W; X += 256) {
        const int xs = min(max(X - padl, 0), wv - 1);
        o[X] = (float)((pre[xs + k2] - pre[xs]) * inv);
    }
}

// ---- float64 SSIM of one channel (BORDER_REPLICATE): a workgroup owns a 16 x 16 tile, stages the 26 x 26 haloed tile of both images
// as doubles, filters the five fields separably (cv2.filter2D applies the outer product window; separable evaluation differs from it
// only by double rounding), and writes one partial sum.
constexpr int ST = 16, SR = 5, SE = ST + 2 * SR;
struct SsimYArgs { const float* a; const float* b; int H, W; double c1, c2; double g[11]; double* partial; };

__global__ __launch_bounds__(256) void ssim_y64_kernel(SsimYArgs p) {
    __shared__ double s0[5][SE][SE + 1];
    __shared__ double s1[5][SE][ST + 1];
    __shared__ double red[4];
    const int tid = threadIdx.x, x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
    for (int i = tid; i < SE * SE; i += 256) {
        const int r = i / SE, c = i - r * SE;
        const int gy = min(max(y0 + r - SR, 0), p.H - 1), gx = min(max(x0 + c - SR, 0), p.W - 1);
        const double a = (double)p.a[(long)gy * p.W + gx], b = (double)p.b[(long)gy * p.W + gx];
        s0[0][r][c] = a; s0[1][r][c] = b; s0[2][r][c] = a * a; s0[3][r][c] = b * b; s0[4][r][c] = a * b;
    }
    __syncthreads();
    for (int i = tid; i < 5 * SE * ST; i += 256) {          // along W
        const int f = i / (SE * ST), rem = i - f * (SE * ST), r = rem / ST, c = rem - r * ST;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 11; ++k) v += p.g[k] * s0[f][r][c + k];
        s1[f][r][c] = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    double m[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 11; ++k) v += p.g[k] * s1[f][ty + k][tx];
        m[f] = v;
    }
    double val = 0.0;
    if (y0 + ty < p.H && x0 + tx < p.W) {
        const double mu1 = m[0], mu2 = m[1];
        const double s11 = m[2] - mu1 * mu1, s22 = m[3] - mu2 * mu2, s12 = m[4] - mu1 * mu2;
        val = ((2 * mu1 * mu2 + p.c1) * (2 * s12 + p.c2)) / ((mu1 * mu1 + mu2 * mu2 + p.c1) * (s11 + s22 + p.c2));
    }
    // fixed-order block reduction
    for (int o = 32; o > 0; o >>= 1) val += __shfl_xor(val, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) p.partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void ssim_y64_finish_kernel(const double* __restrict__ partial, int n, double inv, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += partial[i];
        out[0] = s * inv;
    }
}

}  // namespace

extern "C" int64_t tdr_local_avgpool_ws_floats(int planes, int H, int W, int k1) {
    const int kk = k1 < H ? k1 : H;
    return (int64_t)planes * (H - kk + 1) * W;
}

extern "C" int tdr_local_avgpool(const float* in, int planes, int H, int W, int k1, int k2, float* ws, float* out, void* stream) {
    TDR_REQUIRE(in && ws && out && planes > 0 && H > 0 && W > 0 && k1 > 0 && k2 > 0, "tdr_local_avgpool: bad argument");
    TDR_REQUIRE(W <= 8192, "tdr_local_avgpool: W = %d > 8192 (row prefix lives in LDS)", W);
    const int kk1 = k1 < H ? k1 : H, kk2 = k2 < W ? k2 : W;            // k = min(size, kernel_size) (nafnet_local_arch.py:62)
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tlsc_vsum_kernel, dim3(tdr_cdiv(W, 256), planes), dim3(256), 0, st, in, H, W, kk1, ws);
    TDR_LAUNCH_CHECK("tlsc_vsum_kernel");
    const size_t lds = (size_t)(W + 1 + 256) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tlsc_hmean_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(tlsc_hmean_kernel, dim3(H, planes), dim3(256), lds, st, ws, H, W, kk1, kk2, out);
    TDR_LAUNCH_CHECK("tlsc_hmean_kernel");
    return TDR_OK;
}

extern "C" int64_t tdr_ssim_y64_ws_doubles(int H, int W) { return (int64_t)tdr_cdiv(H, 16) * tdr_cdiv(W, 16) + 1; }

extern "C" int tdr_ssim_y64(const float* img1, const float* img2, int H, int W, double* ws, double* out, void* stream) {
    TDR_REQUIRE(img1 && img2 && ws && out && H > 0 && W > 0, "tdr_ssim_y64: bad argument");
    SsimYArgs p;
    p.a = img1; p.b = img2; p.H = H; p.W = W;
    p.c1 = (0.01 * 255) * (0.01 * 255); p.c2 = (0.03 * 255) * (0.03 * 255);
    double sum = 0.0;                                 // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised
    for (int i = 0; i < 11; ++i) { p.g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += p.g[i]; }
    for (int i = 0; i < 11; ++i) p.g[i] /= sum;
    p.partial = ws;
    const dim3 grid(tdr_cdiv(W, 16), tdr_cdiv(H, 16));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_y64_kernel, grid, dim3(256), 0, st, p);
    TDR_LAUNCH_CHECK("ssim_y64_kernel");
    hipLaunchKernelGGL(ssim_y64_finish_kernel, dim3(1), dim3(64), 0, st, ws, (int)(grid.x * grid.y), 1.0 / ((double)H * W), out);
    TDR_LAUNCH_CHECK("ssim_y64_finish_kernel");
    return TDR_OK;
}
