// Operators of the reference's un-guided SFNet that the rest of the library does not have (models/archs/sfnet_arch_utils.py:76-265,
// models/archs/network_sfnet_guided_arch.py:200-214,366-407), forward and backward, training-mode semantics:
//   exact-erf GELU after a convolution (BasicConv, :76-98)                        tdr_gelu_fwd / _bwd
//   F.interpolate(scale_factor=0.5), nearest (:368-369)                           tdr_subsample2
//   InstanceNorm2d(affine) (SCM, :208)                                            tdr_instnorm_fwd / _bwd
//   Gap (:101-117) and Patch_ap (:239-265): out = x * A[c, region] + mean_region(x) * B[c, region] over the whole plane (Gap) or its four
//   quadrants (Patch_ap), channel slices in place                                 tdr_region_affine_fwd / _bwd
//   dynamic_filter (:152-192) + SFconv (:195-236):
//       per-plane mean                                                             (tdr_plane_mean, csrc/tdr_prompt.hip)
//       the vector pipeline on [N, c] pooled vectors -- 1x1 conv -> BatchNorm over the batch (running buffers moved) -> softmax over the
//       k*k taps; fc -> fcs[0], fcs[1] -> softmax over all 2c entries -- and its backward, ONE workgroup each (N <= 16, c <= 256)
//                                                                                  tdr_sf_dyn_vec_fwd / _bwd
//       low = k x k stencil of the reflection-padded plane with per-(image, group) taps; mix = x * a_high + low * (a_low - a_high)
//                                                                                  tdr_sf_dynfilt_fwd
//       backward: per-(image, channel) sums for d a_high / d a_low, per-(image, group, tap) sums for d taps, and the data gradient
//       (gather over the reflected stencil -- deterministic, no atomics)           tdr_sf_dynfilt_bwd_reduce / _bwd_dx
//   ConvTranspose2d(4, stride 2, padding 1) (:87) as a 3x3 / pad 1 convolution with 4 Cout output channels + PixelShuffle(2): the weight
//   re-tiling and its transpose for the gradient                                   tdr_convt4_weight_to_3x3 / _grad_from_3x3
// These are HBM-bound point-wise / stencil / reduction kernels: coalesced along W, one plane (or region) per workgroup, wave-level
// reductions; the convolutions themselves run on the library's MFMA kernels.
#include "tdr_common.h"
#include "tdr_erf.h"
#include "../../include/tdr.h"

namespace {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_1ulp(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float x) {
    return 0.5f * (1.0f + erf_1ulp(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// sum over the 256 threads of a block (fixed order)
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// y = gelu(z), z = x + bias[channel] (bias optional; z written back when z_out is given -- in place over x is fine)
__global__ void gelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias, int C, int HW, float* __restrict__ z_out,
                                float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float z = x[i];
    if (bias) z += bias[(i / HW) % C];
    if (z_out) z_out[i] = z;
    y[i] = gelu_f(z);
}
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dz, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dz[i] = dy[i] * gelu_d(z[i]);
}
__global__ void subsample2_kernel(const float* __restrict__ x, int H, int W, float* __restrict__ y) {
    const int OH = H / 2, OW = W / 2;
    const long p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < OH * OW) y[p * OH * OW + i] = x[p * (long)H * W + (long)(2 * (i / OW)) * W + 2 * (i % OW)];
}

// ---- InstanceNorm2d: one workgroup per (n, c) plane
__global__ __launch_bounds__(256) void instnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                          float eps, int C, int HW, float* __restrict__ y, float* __restrict__ mu,
                                                          float* __restrict__ rs) {
    __shared__ float red[4];
    const long p = blockIdx.x;
    const int c = (int)(p % C);
    const float* xp = x + p * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) s += xp[i];
    const float m = block_sum256(s, red) / HW;
    float v = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) { const float d = xp[i] - m; v += d * d; }
    const float r = 1.f / sqrtf(block_sum256(v, red) / HW + eps);
    if (threadIdx.x == 0) { mu[p] = m; rs[p] = r; }
    const float g = w[c] * r, o = b[c] - m * g;
    for (int i = threadIdx.x; i < HW; i += 256) y[p * HW + i] = xp[i] * g + o;
}
// dx = rs * w * (dy - mean(dy) - xhat * mean(dy * xhat)); per-plane partials of dw = sum dy * xhat, db = sum dy
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mu,
                                                          const float* __restrict__ rs, const float* __restrict__ w, int C, int HW,
                                                          float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4];
    const long p = blockIdx.x;
    const int c = (int)(p % C);
    const float m = mu[p], r = rs[p];
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float d = dy[p * HW + i];
        s1 += d;
        s2 += d * (x[p * HW + i] - m) * r;
    }
    const float a1 = block_sum256(s1, red), a2 = block_sum256(s2, red);
    if (threadIdx.x == 0) { part[2 * p] = a2; part[2 * p + 1] = a1; }
    const float g = w[c] * r, m1 = a1 / HW, m2 = a2 / HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float xh = (x[p * HW + i] - m) * r;
        dx[p * HW + i] = g * (dy[p * HW + i] - m1 - xh * m2);
    }
}
// o0[j] = sum_n part[n][j][0], o1[j] = sum_n part[n][j][1]     (fixed order over n)
__global__ void pair_over_batch_kernel(const float* __restrict__ part, int N, int J, float* __restrict__ o0, float* __restrict__ o1) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    float a = 0.f, b = 0.f;
    for (int n = 0; n < N; ++n) { a += part[((long)n * J + j) * 2]; b += part[((long)n * J + j) * 2 + 1]; }
    o0[j] = a;
    o1[j] = b;
}

// dA[j] = sum_n part[n][j][0], dB[j] = sum_n part[n][j][1] (fixed order) -> d ph = dA - dB, d pl = dB   (A = ph + shift, B = pl - A)
__global__ void region_affine_finish_kernel(const float* __restrict__ part, int N, int J, float* __restrict__ dph, float* __restrict__ dpl) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    float a = 0.f, b = 0.f;
    for (int n = 0; n < N; ++n) { a += part[((long)n * J + j) * 2]; b += part[((long)n * J + j) * 2 + 1]; }
    dph[j] = a - b;
    dpl[j] = b;
}

// ---- region affine: workgroup = (region, channel, image); regions = q x q equal blocks of the plane
__global__ __launch_bounds__(256) void region_affine_fwd_kernel(const float* __restrict__ x, long x_ns, const float* __restrict__ ph,
                                                               const float* __restrict__ pl, float shift, int q, int C, int H, int W,
                                                               float* __restrict__ y, long y_ns, float* __restrict__ mean) {
    __shared__ float red[4];
    const int reg = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
    const int rh = H / q, rw = W / q, y0 = (reg / q) * rh, x0 = (reg % q) * rw;
    const float* xp = x + (long)n * x_ns + (long)c * H * W;
    float s = 0.f;
    for (int i = threadIdx.x; i < rh * rw; i += 256) s += xp[(long)(y0 + i / rw) * W + x0 + i % rw];
    const float m = block_sum256(s, red) / (rh * rw);
    const int k = c * q * q + reg;
    if (threadIdx.x == 0) mean[((long)n * C + c) * q * q + reg] = m;
    const float a = ph[k] + shift, mb = m * (pl[k] - a);          // A = ph + shift, B = pl - A
    float* yp = y + (long)n * y_ns + (long)c * H * W;
    for (int i = threadIdx.x; i < rh * rw; i += 256) {
        const long o = (long)(y0 + i / rw) * W + x0 + i % rw;
        yp[o] = xp[o] * a + mb;
    }
}
// dx = dy * A + mean_region(dy) * B ; partials per (n, c, region): [sum dy * x, mean_region(x) * sum dy]
__global__ __launch_bounds__(256) void region_affine_bwd_kernel(const float* __restrict__ dy, long dy_ns, const float* __restrict__ x, long x_ns,
                                                               const float* __restrict__ ph, const float* __restrict__ pl, float shift,
                                                               const float* __restrict__ mean, int q, int C, int H, int W,
                                                               float* __restrict__ dx, long dx_ns, float* __restrict__ part) {
    __shared__ float red[4];
    const int reg = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
    const int rh = H / q, rw = W / q, y0 = (reg / q) * rh, x0 = (reg % q) * rw;
    const float* xp = x + (long)n * x_ns + (long)c * H * W;
    const float* dp = dy + (long)n * dy_ns + (long)c * H * W;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < rh * rw; i += 256) {
        const long o = (long)(y0 + i / rw) * W + x0 + i % rw;
        s1 += dp[o] * xp[o];
        s2 += dp[o];
    }
    const float a1 = block_sum256(s1, red), a2 = block_sum256(s2, red);
    const long pi = ((long)n * C + c) * q * q + reg;
    if (threadIdx.x == 0) { part[2 * pi] = a1; part[2 * pi + 1] = mean[pi] * a2; }
    const int k = c * q * q + reg;
    const float a = ph[k] + shift, mb = a2 / (rh * rw) * (pl[k] - a);
    float* op = dx + (long)n * dx_ns + (long)c * H * W;
    for (int i = threadIdx.x; i < rh * rw; i += 256) {
        const long o = (long)(y0 + i / rw) * W + x0 + i % rw;
        op[o] = dp[o] * a + mb;
    }
}

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// low = stencil(x; taps[n][g]) with reflection padding; mix = x * ah + low * (al - ah)
__global__ __launch_bounds__(256) void dynfilt_fwd_kernel(const float* __restrict__ x, long x_ns, const float* __restrict__ taps, const float* __restrict__ ah,
                                                         const float* __restrict__ al, int C, int cg, int H, int W, int k,
                                                         float* __restrict__ low, float* __restrict__ mix) {
    __shared__ float tp[25];
    const int c = blockIdx.y, n = blockIdx.z, G = C / cg, g = c / cg, p = k / 2;
    if (threadIdx.x < k * k) tp[threadIdx.x] = taps[((long)n * G + g) * k * k + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int yy = i / W, xx = i % W;
    const float* xp = x + (long)n * x_ns + (long)c * H * W;
    float a = 0.f;
    for (int t = 0; t < k * k; ++t) a += tp[t] * xp[(long)reflect(yy + t / k - p, H) * W + reflect(xx + t % k - p, W)];
    const long o = ((long)n * C + c) * H * W + i;
    low[o] = a;
    const float h = ah[(long)n * C + c], l = al[(long)n * C + c];
    mix[o] = xp[i] * h + a * (l - h);
}
// per (n, c): dah = sum dmix * (x - low), dal = sum dmix * low
__global__ __launch_bounds__(256) void dynfilt_bwd_att_kernel(const float* __restrict__ dmix, const float* __restrict__ x, long x_ns,
                                                             const float* __restrict__ low, int C, int HW, float* __restrict__ dah,
                                                             float* __restrict__ dal) {
    __shared__ float red[4];
    const int c = blockIdx.x, n = blockIdx.y;
    const float* xp = x + (long)n * x_ns + (long)c * HW;
    const long o = ((long)n * C + c) * HW;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float d = dmix[o + i], lw = low[o + i];
        s1 += d * (xp[i] - lw);
        s2 += d * lw;
    }
    const float a1 = block_sum256(s1, red), a2 = block_sum256(s2, red);
    if (threadIdx.x == 0) { dah[(long)n * C + c] = a1; dal[(long)n * C + c] = a2; }
}
// per (n, g, t): dtaps = sum_{c in g} (al - ah)[n][c] * sum_px dmix[c][px] * xpad[c][px + t]
__global__ __launch_bounds__(256) void dynfilt_bwd_taps_kernel(const float* __restrict__ dmix, const float* __restrict__ x, long x_ns,
                                                              const float* __restrict__ ah, const float* __restrict__ al, int C, int cg,
                                                              int H, int W, int k, float* __restrict__ dtaps) {
    __shared__ float red[4];
    const int t = blockIdx.x, g = blockIdx.y, n = blockIdx.z, G = C / cg, p = k / 2;
    const int dy = t / k - p, dx = t % k - p;
    float s = 0.f;
    for (int cc = 0; cc < cg; ++cc) {
        const int c = g * cg + cc;
        const float sc = al[(long)n * C + c] - ah[(long)n * C + c];
        const float* xp = x + (long)n * x_ns + (long)c * H * W;
        const float* dp = dmix + ((long)n * C + c) * H * W;
        float sl = 0.f;
        for (int i = threadIdx.x; i < H * W; i += 256)
            sl += dp[i] * xp[(long)reflect(i / W + dy, H) * W + reflect(i % W + dx, W)];
        s += sc * sl;
    }
    const float a = block_sum256(s, red);
    if (threadIdx.x == 0) dtaps[((long)n * G + g) * k * k + t] = a;
}
// dx = dmix * ah + (al - ah) * gather_t taps[t] * dmix[source pixels that read this pixel through tap t] + dap / HW
// the pre-images of input row i under tap offset d (p = k / 2): y = i - d; and, through the reflection, y = -i - d (1 <= i <= p) and
// y = 2 (H - 1) - i - d (H - 1 - p <= i <= H - 2) -- each kept when it is a row of the image whose tap really leaves the image there
__global__ __launch_bounds__(256) void dynfilt_bwd_dx_kernel(const float* __restrict__ dmix, const float* __restrict__ taps, const float* __restrict__ ah,
                                                            const float* __restrict__ al, const float* __restrict__ dap, int C, int cg,
                                                            int H, int W, int k, float* __restrict__ dx, long dx_ns) {
    __shared__ float tp[25];
    const int c = blockIdx.y, n = blockIdx.z, G = C / cg, g = c / cg, p = k / 2;
    if (threadIdx.x < k * k) tp[threadIdx.x] = taps[((long)n * G + g) * k * k + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int yi = i / W, xi = i % W;
    const float* dp = dmix + ((long)n * C + c) * H * W;
    float acc = 0.f;
    for (int ty = 0; ty < k; ++ty) {
        const int d = ty - p;
        int ys[3], ny = 0;
        { const int y = yi - d; if (y >= 0 && y < H) ys[ny++] = y; }
        if (yi >= 1 && yi <= p) { const int y = -yi - d; if (y >= 0 && y < H && y + d < 0) ys[ny++] = y; }
        if (yi >= H - 1 - p && yi <= H - 2) { const int y = 2 * (H - 1) - yi - d; if (y >= 0 && y < H && y + d >= H) ys[ny++] = y; }
        for (int tx = 0; tx < k; ++tx) {
            const int e = tx - p;
            int xs[3], nx = 0;
            { const int xq = xi - e; if (xq >= 0 && xq < W) xs[nx++] = xq; }
            if (xi >= 1 && xi <= p) { const int xq = -xi - e; if (xq >= 0 && xq < W && xq + e < 0) xs[nx++] = xq; }
            if (xi >= W - 1 - p && xi <= W - 2) { const int xq = 2 * (W - 1) - xi - e; if (xq >= 0 && xq < W && xq + e >= W) xs[nx++] = xq; }
            float sl = 0.f;
            for (int a = 0; a < ny; ++a)
                for (int b = 0; b < nx; ++b) sl += dp[(long)ys[a] * W + xs[b]];
            acc += tp[ty * k + tx] * sl;
        }
    }
    const float h = ah[(long)n * C + c], l = al[(long)n * C + c];
    dx[(long)n * dx_ns + (long)c * H * W + i] = dp[i] * h + (l - h) * acc + dap[(long)n * C + c] / (H * W);
}

// ---- the vector pipeline of dynamic_filter + SFconv on pooled vectors: ONE workgroup (N * max(GK, 2c) values)
struct DynVecArgs {
    const float *ap, *wconv, *bn_w, *bn_b, *fc_w, *fc_b, *f0_w, *f0_b, *f1_w, *f1_b;
    float *run_mean, *run_var; long long* nbt;
    float *taps, *ah, *al;                 // outputs [N][GK], [N][c], [N][c]
    float *xhat, *rstd, *z, *att;          // saved: [N][GK], [GK], [N][d], [N][2c]
    int N, c, GK, KK, d;
    float eps, mom;
    int use_running;
};
#define FOR_T(i, n) for (int i = threadIdx.x; i < (n); i += blockDim.x)
__global__ __launch_bounds__(256) void dyn_vec_fwd_kernel(DynVecArgs a) {
    const int N = a.N, c = a.c, GK = a.GK, KK = a.KK, d = a.d;
    // 1x1 conv (no bias) -> xhat holds the pre-norm value for now
    FOR_T(i, N * GK) {
        const int n = i / GK, j = i % GK;
        float s = 0.f;
        for (int ci = 0; ci < c; ++ci) s += a.wconv[(long)j * c + ci] * a.ap[(long)n * c + ci];
        a.xhat[i] = s;
    }
    __syncthreads();
    // BatchNorm over the batch (biased variance normalises; the running buffers move towards the batch mean / UNBIASED variance)
    FOR_T(j, GK) {
        float m = 0.f, v = 0.f;
        if (a.use_running) {                                   // module.eval(): the running statistics normalise, nothing is updated
            m = a.run_mean[j];
            v = a.run_var[j];
        } else {
            for (int n = 0; n < N; ++n) m += a.xhat[(long)n * GK + j];
            m /= N;
            for (int n = 0; n < N; ++n) { const float e = a.xhat[(long)n * GK + j] - m; v += e * e; }
            v /= N;
        }
        const float r = 1.f / sqrtf(v + a.eps);
        a.rstd[j] = r;
        if (!a.use_running) {
            a.run_mean[j] = (1.f - a.mom) * a.run_mean[j] + a.mom * m;
            a.run_var[j] = (1.f - a.mom) * a.run_var[j] + a.mom * v * N / (N > 1 ? N - 1 : 1);
        }
        for (int n = 0; n < N; ++n) {
            const float xh = (a.xhat[(long)n * GK + j] - m) * r;
            a.xhat[(long)n * GK + j] = xh;
            a.taps[(long)n * GK + j] = xh * a.bn_w[j] + a.bn_b[j];          // (logits for now)
        }
    }
    if (threadIdx.x == 0 && a.nbt && !a.use_running) *a.nbt += 1;
    __syncthreads();
    // softmax over the KK taps of every (image, group)
    FOR_T(i, N * GK / KK) {
        float* t = a.taps + (long)i * KK;
        float mx = t[0];
        for (int q = 1; q < KK; ++q) mx = fmaxf(mx, t[q]);
        float s = 0.f;
        for (int q = 0; q < KK; ++q) { t[q] = __expf(t[q] - mx); s += t[q]; }
        for (int q = 0; q < KK; ++q) t[q] /= s;
    }
    // SFconv: z = fc(ap)
    FOR_T(i, N * d) {
        const int n = i / d, m = i % d;
        float s = a.fc_b[m];
        for (int ci = 0; ci < c; ++ci) s += a.fc_w[(long)m * c + ci] * a.ap[(long)n * c + ci];
        a.z[i] = s;
    }
    __syncthreads();
    FOR_T(i, N * 2 * c) {
        const int n = i / (2 * c), e = i % (2 * c);
        const float* w = e < c ? a.f0_w + (long)e * d : a.f1_w + (long)(e - c) * d;
        float s = e < c ? a.f0_b[e] : a.f1_b[e - c];
        for (int m = 0; m < d; ++m) s += w[m] * a.z[(long)n * d + m];
        a.att[i] = s;
    }
    __syncthreads();
    FOR_T(n, N) {                                              // softmax over ALL 2c entries (nn.Softmax(dim=1) on the concatenation)
        float* t = a.att + (long)n * 2 * c;
        float mx = t[0];
        for (int q = 1; q < 2 * c; ++q) mx = fmaxf(mx, t[q]);
        float s = 0.f;
        for (int q = 0; q < 2 * c; ++q) { t[q] = __expf(t[q] - mx); s += t[q]; }
        for (int q = 0; q < 2 * c; ++q) {
            t[q] /= s;
            if (q < c) a.ah[(long)n * c + q] = t[q]; else a.al[(long)n * c + q - c] = t[q];
        }
    }
}

struct DynVecBwdArgs {
    const float *ap, *wconv, *bn_w, *fc_w, *f0_w, *f1_w;
    const float *taps, *xhat, *rstd, *z, *att;
    const float *dtaps, *dah, *dal;        // [N][GK], [N][c], [N][c]
    float *dap;                            // [N][c]
    float *g_wconv, *g_bn_w, *g_bn_b, *g_fc_w, *g_fc_b, *g_f0_w, *g_f0_b, *g_f1_w, *g_f1_b;
    float *s_dl, *s_dz, *s_dlf;            // scratch [N][2c], [N][d], [N][GK]
    int N, c, GK, KK, d;
};
__global__ __launch_bounds__(256) void dyn_vec_bwd_kernel(DynVecBwdArgs a) {
    const int N = a.N, c = a.c, GK = a.GK, KK = a.KK, d = a.d;
    // softmax over 2c: dlogit = att * (datt - sum att * datt)
    FOR_T(n, N) {
        float dot = 0.f;
        for (int q = 0; q < 2 * c; ++q) dot += a.att[(long)n * 2 * c + q] * (q < c ? a.dah[(long)n * c + q] : a.dal[(long)n * c + q - c]);
        for (int q = 0; q < 2 * c; ++q)
            a.s_dl[(long)n * 2 * c + q] = a.att[(long)n * 2 * c + q] * ((q < c ? a.dah[(long)n * c + q] : a.dal[(long)n * c + q - c]) - dot);
    }
    // taps softmax: dlogit = taps * (dtaps - sum taps * dtaps) per (n, g)  -> s_dlf (gradient of the BatchNorm output y)
    FOR_T(i, N * GK / KK) {
        float dot = 0.f;
        for (int q = 0; q < KK; ++q) dot += a.taps[(long)i * KK + q] * a.dtaps[(long)i * KK + q];
        for (int q = 0; q < KK; ++q) a.s_dlf[(long)i * KK + q] = a.taps[(long)i * KK + q] * (a.dtaps[(long)i * KK + q] - dot);
    }
    __syncthreads();
    // fcs backward
    FOR_T(i, N * d) {
        const int n = i / d, m = i % d;
        float s = 0.f;
        for (int ci = 0; ci < c; ++ci)
            s += a.f0_w[(long)ci * d + m] * a.s_dl[(long)n * 2 * c + ci] + a.f1_w[(long)ci * d + m] * a.s_dl[(long)n * 2 * c + c + ci];
        a.s_dz[i] = s;
    }
    FOR_T(i, 2 * c * d) {
        const int e = i / d, m = i % d;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += a.s_dl[(long)n * 2 * c + e] * a.z[(long)n * d + m];
        (e < c ? a.g_f0_w : a.g_f1_w)[(long)(e < c ? e : e - c) * d + m] = s;
    }
    FOR_T(e, 2 * c) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += a.s_dl[(long)n * 2 * c + e];
        (e < c ? a.g_f0_b : a.g_f1_b)[e < c ? e : e - c] = s;
    }
    // BatchNorm backward (per channel j over the batch)
    FOR_T(j, GK) {
        float sdy = 0.f, sdyx = 0.f;
        for (int n = 0; n < N; ++n) { const float dy = a.s_dlf[(long)n * GK + j]; sdy += dy; sdyx += dy * a.xhat[(long)n * GK + j]; }
        a.g_bn_w[j] = sdyx;
        a.g_bn_b[j] = sdy;
        const float gw = a.bn_w[j], r = a.rstd[j];
        for (int n = 0; n < N; ++n) {
            const float dxh = a.s_dlf[(long)n * GK + j] * gw;
            a.s_dlf[(long)n * GK + j] = r * (dxh - gw * sdy / N - a.xhat[(long)n * GK + j] * gw * sdyx / N);
        }
    }
    __syncthreads();
    FOR_T(i, d * c) {
        const int m = i / c, ci = i % c;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += a.s_dz[(long)n * d + m] * a.ap[(long)n * c + ci];
        a.g_fc_w[i] = s;
    }
    FOR_T(m, d) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += a.s_dz[(long)n * d + m];
        a.g_fc_b[m] = s;
    }
    FOR_T(i, GK * c) {
        const int j = i / c, ci = i % c;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += a.s_dlf[(long)n * GK + j] * a.ap[(long)n * c + ci];
        a.g_wconv[i] = s;
    }
    FOR_T(i, N * c) {
        const int n = i / c, ci = i % c;
        float s = 0.f;
        for (int m = 0; m < d; ++m) s += a.fc_w[(long)m * c + ci] * a.s_dz[(long)n * d + m];
        for (int j = 0; j < GK; ++j) s += a.wconv[(long)j * c + ci] * a.s_dlf[(long)n * GK + j];
        a.dap[i] = s;
    }
}

// ConvTranspose2d(4, 2, 1) weight [Cin][Cout][4][4] <-> the 3x3 / pad 1 convolution weight [4 Cout][Cin][3][3] whose output channel
// co * 4 + a * 2 + b is output parity (a, b) (then PixelShuffle(2)): parity 0 reads window rows (0, 1) with kernel rows (3, 1), parity 1
// reads window rows (1, 2) with kernel rows (2, 0); same for columns.  The other 5 of the 9 taps of a row are zero.
__device__ __forceinline__ int convt_k(int par, int r) { return par == 0 ? (r == 0 ? 3 : (r == 1 ? 1 : -1)) : (r == 1 ? 2 : (r == 2 ? 0 : -1)); }
__global__ void convt4_to_3x3_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ w3, int reverse) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long tot = (long)4 * Cout * Cin * 9;
    if (i >= tot) return;
    const int s = (int)(i % 3), r = (int)(i / 3 % 3), ci = (int)(i / 9 % Cin);
    const int m = (int)(i / 9 / Cin), co = m / 4, a = (m >> 1) & 1, b = m & 1;
    const int kr = convt_k(a, r), ks = convt_k(b, s);
    if (!reverse) {
        w3[i] = (kr < 0 || ks < 0) ? 0.f : w[(((long)ci * Cout + co) * 4 + kr) * 4 + ks];
    } else if (kr >= 0 && ks >= 0) {
        const_cast<float*>(w)[(((long)ci * Cout + co) * 4 + kr) * 4 + ks] = w3[i];       // every (kr, ks) has exactly one (a, r, b, s)
    }
}
__global__ void repeat4_kernel(const float* __restrict__ b, int Cout, float* __restrict__ b4, int reverse) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cout) return;
    if (!reverse) { for (int q = 0; q < 4; ++q) b4[4 * i + q] = b[i]; }
    else const_cast<float*>(b)[i] = (b4[4 * i] + b4[4 * i + 1]) + (b4[4 * i + 2] + b4[4 * i + 3]);
}

// ---- mode[0] == 'test' (inference with TLSC pooling, sfnet_arch_utils.py:108-113, :226-229, :247-250): the pooled operand of Gap / Patch_ap /
// SFconv is a per-pixel box-mean MAP (tdr_local_avgpool, csrc/tdr_tlsc.hip) instead of one number per plane.  Forward only.
// region planes of a dense-NCHW view as their own tensor: out [N][(c q + p1) q + p2][H / q][W / q] = x[n][c][p1 H / q + i][p2 W / q + j]
// (q = 1: a dense copy of a channel slice; q = 2: Patch_ap's `b c (p1 w1) (p2 w2) -> b (c p1 p2) w1 w2`)
__global__ __launch_bounds__(256) void region_split_kernel(const float* __restrict__ x, long x_ns, int q, int C, int H, int W, float* __restrict__ out) {
    const int Hq = H / q, Wq = W / q;
    const int J = blockIdx.y, n = blockIdx.z;
    const int c = J / (q * q), p1 = (J / q) % q, p2 = J % q;
    const float* src = x + (long)n * x_ns + (long)c * H * W + (long)p1 * Hq * W + (long)p2 * Wq;
    float* dst = out + ((long)n * C * q * q + J) * Hq * Wq;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hq * Wq; i += gridDim.x * 256) dst[i] = src[(long)(i / Wq) * W + i % Wq];
}
// y = m * pl[J] + (x - m) * (ph[J] + shift), m = the box-mean map of the pixel's region plane (the reference's own operation order:
// Gap x_d * fscale_d + (x - x_d) * (fscale_h + 1), :117-119; Patch_ap (patch_x - low) * h + low * l, :262-263)
__global__ __launch_bounds__(256) void local_affine_kernel(const float* __restrict__ x, long x_ns, const float* __restrict__ m, const float* __restrict__ ph,
                                                           const float* __restrict__ pl, float shift, int q, int C, int H, int W,
                                                           float* __restrict__ y, long y_ns) {
    const int Hq = H / q, Wq = W / q;
    const int J = blockIdx.y, n = blockIdx.z;
    const int c = J / (q * q), p1 = (J / q) % q, p2 = J % q;
    const long off = (long)c * H * W + (long)p1 * Hq * W + (long)p2 * Wq;
    const float* src = x + (long)n * x_ns + off;
    float* dst = y + (long)n * y_ns + off;
    const float* mm = m + ((long)n * C * q * q + J) * Hq * Wq;
    const float a = ph[J] + shift, b = pl[J];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Hq * Wq; i += gridDim.x * 256) {
        const long o = (long)(i / Wq) * W + i % Wq;
        const float lo = mm[i];
        dst[o] = lo * b + (src[o] - lo) * a;
    }
}
// SFconv's `emerge = low + high` with high = x - low (:218; one rounding away from x, kept as the reference computes it)
__global__ __launch_bounds__(256) void emerge_kernel(const float* __restrict__ x, long x_ns, const float* __restrict__ low, int CHW, float* __restrict__ out) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < CHW; i += gridDim.x * 256) {
        const float l = low[(long)n * CHW + i];
        out[(long)n * CHW + i] = l + (x[(long)n * x_ns + i] - l);
    }
}
// per pixel: softmax over the 2c logits [lh[:, p] ; ll[:, p]] (nn.Softmax(dim=1) on the concatenation, :226-227), mix = (x - low) * a_high + low * a_low
__global__ __launch_bounds__(256) void softmax_mix_kernel(const float* __restrict__ x, long x_ns, const float* __restrict__ low, const float* __restrict__ lh,
                                                          const float* __restrict__ ll, int c, int HW, float* __restrict__ mix) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* ph = lh + (long)n * c * HW + p;
    const float* pl = ll + (long)n * c * HW + p;
    float mx = ph[0];
    for (int k = 0; k < c; ++k) mx = fmaxf(mx, fmaxf(ph[(long)k * HW], pl[(long)k * HW]));
    float s = 0.f;
    for (int k = 0; k < c; ++k) s += __expf(ph[(long)k * HW] - mx) + __expf(pl[(long)k * HW] - mx);
    const float inv = 1.f / s;
    for (int k = 0; k < c; ++k) {
        const float lo = low[((long)n * c + k) * HW + p];
        const float hi = x[(long)n * x_ns + (long)k * HW + p] - lo;
        mix[((long)n * c + k) * HW + p] = hi * (__expf(ph[(long)k * HW] - mx) * inv) + lo * (__expf(pl[(long)k * HW] - mx) * inv);
    }
}

}  // namespace

extern "C" int tdr_gelu_fwd(const float* x, const float* bias, int C, int HW, float* z_out, float* y, int64_t n, void* stream) {
    TDR_REQUIRE(x && y && n > 0 && (!bias || (C > 0 && HW > 0)), "tdr_gelu_fwd: bad argument");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(tdr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, C, HW, z_out, y, (long)n);
    TDR_LAUNCH_CHECK("gelu_fwd_kernel");
    return TDR_OK;
}
extern "C" int tdr_gelu_bwd(const float* dy, const float* z, float* dz, int64_t n, void* stream) {
    TDR_REQUIRE(dy && z && dz && n > 0, "tdr_gelu_bwd: bad argument");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(tdr_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, z, dz, (long)n);
    TDR_LAUNCH_CHECK("gelu_bwd_kernel");
    return TDR_OK;
}
extern "C" int tdr_subsample2(const float* x, int planes, int H, int W, float* y, void* stream) {
    TDR_REQUIRE(x && y && planes > 0 && H >= 2 && W >= 2, "tdr_subsample2: bad argument");
    hipLaunchKernelGGL(subsample2_kernel, dim3(tdr_cdiv((long)(H / 2) * (W / 2), 256), planes), dim3(256), 0, (hipStream_t)stream, x, H, W, y);
    TDR_LAUNCH_CHECK("subsample2_kernel");
    return TDR_OK;
}
extern "C" int tdr_instnorm_fwd(const float* x, const float* w, const float* b, float eps, int N, int C, int HW, float* y, float* mu,
                                float* rs, void* stream) {
    TDR_REQUIRE(x && w && b && y && mu && rs && N > 0 && C > 0 && HW > 0, "tdr_instnorm_fwd: bad argument");
    hipLaunchKernelGGL(instnorm_fwd_kernel, dim3(N * C), dim3(256), 0, (hipStream_t)stream, x, w, b, eps, C, HW, y, mu, rs);
    TDR_LAUNCH_CHECK("instnorm_fwd_kernel");
    return TDR_OK;
}
// ws: 2 * N * C floats
extern "C" int tdr_instnorm_bwd(const float* dy, const float* x, const float* mu, const float* rs, const float* w, int N, int C, int HW,
                                float* dx, float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(dy && x && mu && rs && w && dx && dw && db && ws, "tdr_instnorm_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(instnorm_bwd_kernel, dim3(N * C), dim3(256), 0, st, dy, x, mu, rs, w, C, HW, dx, ws);
    hipLaunchKernelGGL(pair_over_batch_kernel, dim3(tdr_cdiv(C, 256)), dim3(256), 0, st, ws, N, C, dw, db);       // ws [N][C][2]
    TDR_LAUNCH_CHECK("instnorm_bwd_kernel");
    return TDR_OK;
}
extern "C" int tdr_region_affine_fwd(const float* x, int64_t x_ns, const float* ph, const float* pl, float shift, int q, int N, int C, int H,
                                     int W, float* y, int64_t y_ns, float* mean, void* stream) {
    TDR_REQUIRE(x && ph && pl && y && mean && (q == 1 || q == 2) && H % q == 0 && W % q == 0, "tdr_region_affine_fwd: bad argument");
    hipLaunchKernelGGL(region_affine_fwd_kernel, dim3(q * q, C, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, ph, pl, shift, q, C, H, W, y,
                       (long)y_ns, mean);
    TDR_LAUNCH_CHECK("region_affine_fwd_kernel");
    return TDR_OK;
}
// dph, dpl: [C * q * q]; ws: 2 * N * C * q * q floats
extern "C" int tdr_region_affine_bwd(const float* dy, int64_t dy_ns, const float* x, int64_t x_ns, const float* ph, const float* pl, float shift,
                                     const float* mean, int q, int N, int C, int H, int W, float* dx, int64_t dx_ns, float* dph, float* dpl,
                                     float* ws, void* stream) {
    TDR_REQUIRE(dy && x && ph && pl && mean && dx && dph && dpl && ws && (q == 1 || q == 2), "tdr_region_affine_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int J = C * q * q;
    hipLaunchKernelGGL(region_affine_bwd_kernel, dim3(q * q, C, N), dim3(256), 0, st, dy, (long)dy_ns, x, (long)x_ns, ph, pl, shift, mean, q, C, H,
                       W, dx, (long)dx_ns, ws);
    hipLaunchKernelGGL(region_affine_finish_kernel, dim3(tdr_cdiv(J, 256)), dim3(256), 0, st, ws, N, J, dph, dpl);
    TDR_LAUNCH_CHECK("region_affine_bwd_kernel");
    return TDR_OK;
}
extern "C" int tdr_sf_dyn_vec_fwd(const TdrSfDynVecDesc* d, void* stream) {
    TDR_REQUIRE(d && d->ap && d->wconv && d->bn_w && d->bn_b && d->run_mean && d->run_var && d->fc_w && d->fc_b && d->f0_w && d->f0_b &&
                    d->f1_w && d->f1_b && d->taps && d->ah && d->al && d->xhat && d->rstd && d->z && d->att,
                "tdr_sf_dyn_vec_fwd: null pointer");
    TDR_REQUIRE(d->N > 0 && d->N <= 64 && d->c > 0 && d->GK % d->KK == 0 && (d->KK == 9 || d->KK == 25), "tdr_sf_dyn_vec_fwd: bad shape");
    DynVecArgs a;
    a.ap = d->ap; a.wconv = d->wconv; a.bn_w = d->bn_w; a.bn_b = d->bn_b; a.fc_w = d->fc_w; a.fc_b = d->fc_b; a.f0_w = d->f0_w; a.f0_b = d->f0_b;
    a.f1_w = d->f1_w; a.f1_b = d->f1_b; a.run_mean = d->run_mean; a.run_var = d->run_var; a.nbt = reinterpret_cast<long long*>(d->nbt);
    a.taps = d->taps; a.ah = d->ah; a.al = d->al; a.xhat = d->xhat; a.rstd = d->rstd; a.z = d->z; a.att = d->att;
    a.N = d->N; a.c = d->c; a.GK = d->GK; a.KK = d->KK; a.d = d->d; a.eps = d->eps; a.mom = d->momentum;
    a.use_running = d->use_running;
    hipLaunchKernelGGL(dyn_vec_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("dyn_vec_fwd_kernel");
    return TDR_OK;
}
extern "C" int tdr_sf_dyn_vec_bwd(const TdrSfDynVecBwdDesc* d, void* stream) {
    TDR_REQUIRE(d && d->ap && d->wconv && d->bn_w && d->fc_w && d->f0_w && d->f1_w && d->taps && d->xhat && d->rstd && d->z && d->att && d->dtaps &&
                    d->dah && d->dal && d->dap && d->g_wconv && d->g_bn_w && d->g_bn_b && d->g_fc_w && d->g_fc_b && d->g_f0_w && d->g_f0_b &&
                    d->g_f1_w && d->g_f1_b && d->ws,
                "tdr_sf_dyn_vec_bwd: null pointer");
    DynVecBwdArgs a;
    a.ap = d->ap; a.wconv = d->wconv; a.bn_w = d->bn_w; a.fc_w = d->fc_w; a.f0_w = d->f0_w; a.f1_w = d->f1_w;
    a.taps = d->taps; a.xhat = d->xhat; a.rstd = d->rstd; a.z = d->z; a.att = d->att; a.dtaps = d->dtaps; a.dah = d->dah; a.dal = d->dal;
    a.dap = d->dap; a.g_wconv = d->g_wconv; a.g_bn_w = d->g_bn_w; a.g_bn_b = d->g_bn_b; a.g_fc_w = d->g_fc_w; a.g_fc_b = d->g_fc_b;
    a.g_f0_w = d->g_f0_w; a.g_f0_b = d->g_f0_b; a.g_f1_w = d->g_f1_w; a.g_f1_b = d->g_f1_b;
    a.N = d->N; a.c = d->c; a.GK = d->GK; a.KK = d->KK; a.d = d->d;
    a.s_dl = d->ws; a.s_dz = d->ws + (long)d->N * 2 * d->c; a.s_dlf = a.s_dz + (long)d->N * d->d;      // ws: N * (2c + d + GK) floats
    hipLaunchKernelGGL(dyn_vec_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("dyn_vec_bwd_kernel");
    return TDR_OK;
}
extern "C" int tdr_sf_dynfilt_fwd(const float* x, int64_t x_ns, const float* taps, const float* ah, const float* al, int N, int C, int groups,
                                  int H, int W, int k, float* low, float* mix, void* stream) {
    TDR_REQUIRE(x && taps && ah && al && low && mix && (k == 3 || k == 5) && C % groups == 0 && H > k / 2 && W > k / 2,
                "tdr_sf_dynfilt_fwd: bad argument");
    hipLaunchKernelGGL(dynfilt_fwd_kernel, dim3(tdr_cdiv((long)H * W, 256), C, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, taps, ah, al,
                       C, C / groups, H, W, k, low, mix);
    TDR_LAUNCH_CHECK("dynfilt_fwd_kernel");
    return TDR_OK;
}
extern "C" int tdr_sf_dynfilt_bwd_reduce(const float* dmix, const float* x, int64_t x_ns, const float* low, const float* ah, const float* al,
                                         int N, int C, int groups, int H, int W, int k, float* dah, float* dal, float* dtaps, void* stream) {
    TDR_REQUIRE(dmix && x && low && ah && al && dah && dal && dtaps && (k == 3 || k == 5) && C % groups == 0, "tdr_sf_dynfilt_bwd_reduce: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dynfilt_bwd_att_kernel, dim3(C, N), dim3(256), 0, st, dmix, x, (long)x_ns, low, C, H * W, dah, dal);
    hipLaunchKernelGGL(dynfilt_bwd_taps_kernel, dim3(k * k, groups, N), dim3(256), 0, st, dmix, x, (long)x_ns, ah, al, C, C / groups, H, W, k, dtaps);
    TDR_LAUNCH_CHECK("dynfilt_bwd_reduce");
    return TDR_OK;
}
extern "C" int tdr_sf_dynfilt_bwd_dx(const float* dmix, const float* taps, const float* ah, const float* al, const float* dap, int N, int C,
                                     int groups, int H, int W, int k, float* dx, int64_t dx_ns, void* stream) {
    TDR_REQUIRE(dmix && taps && ah && al && dap && dx && (k == 3 || k == 5) && C % groups == 0, "tdr_sf_dynfilt_bwd_dx: bad argument");
    hipLaunchKernelGGL(dynfilt_bwd_dx_kernel, dim3(tdr_cdiv((long)H * W, 256), C, N), dim3(256), 0, (hipStream_t)stream, dmix, taps, ah, al, dap, C,
                       C / groups, H, W, k, dx, (long)dx_ns);
    TDR_LAUNCH_CHECK("dynfilt_bwd_dx_kernel");
    return TDR_OK;
}
extern "C" int tdr_convt4_weight_to_3x3(const float* w, const float* b, int Cin, int Cout, float* w3, float* b4, void* stream) {
    TDR_REQUIRE(w && w3 && Cin > 0 && Cout > 0, "tdr_convt4_weight_to_3x3: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(convt4_to_3x3_kernel, dim3(tdr_cdiv((long)36 * Cout * Cin, 256)), dim3(256), 0, st, w, Cin, Cout, w3, 0);
    if (b && b4) hipLaunchKernelGGL(repeat4_kernel, dim3(tdr_cdiv(Cout, 256)), dim3(256), 0, st, b, Cout, b4, 0);
    TDR_LAUNCH_CHECK("convt4_to_3x3_kernel");
    return TDR_OK;
}
extern "C" int tdr_convt4_grad_from_3x3(const float* dw3, const float* db4, int Cin, int Cout, float* dw, float* db, void* stream) {
    TDR_REQUIRE(dw3 && dw && Cin > 0 && Cout > 0, "tdr_convt4_grad_from_3x3: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(convt4_to_3x3_kernel, dim3(tdr_cdiv((long)36 * Cout * Cin, 256)), dim3(256), 0, st, dw, Cin, Cout, const_cast<float*>(dw3), 1);
    if (db && db4) hipLaunchKernelGGL(repeat4_kernel, dim3(tdr_cdiv(Cout, 256)), dim3(256), 0, st, db, Cout, const_cast<float*>(db4), 1);
    TDR_LAUNCH_CHECK("convt4_grad_from_3x3");
    return TDR_OK;
}

extern "C" int tdr_sf_region_split(const float* x, int64_t x_ns, int q, int N, int C, int H, int W, float* out, void* stream) {
    TDR_REQUIRE(x && out && (q == 1 || q == 2) && H % q == 0 && W % q == 0 && N > 0 && C > 0, "tdr_sf_region_split: bad argument");
    hipLaunchKernelGGL(region_split_kernel, dim3(tdr_cdiv((long)(H / q) * (W / q), 1024), C * q * q, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns,
                       q, C, H, W, out);
    TDR_LAUNCH_CHECK("region_split_kernel");
    return TDR_OK;
}
extern "C" int tdr_sf_local_affine(const float* x, int64_t x_ns, const float* m, const float* ph, const float* pl, float shift, int q, int N, int C,
                                   int H, int W, float* y, int64_t y_ns, void* stream) {
    TDR_REQUIRE(x && m && ph && pl && y && (q == 1 || q == 2) && H % q == 0 && W % q == 0, "tdr_sf_local_affine: bad argument");
    hipLaunchKernelGGL(local_affine_kernel, dim3(tdr_cdiv((long)(H / q) * (W / q), 1024), C * q * q, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns,
                       m, ph, pl, shift, q, C, H, W, y, (long)y_ns);
    TDR_LAUNCH_CHECK("local_affine_kernel");
    return TDR_OK;
}
extern "C" int tdr_sf_emerge(const float* x, int64_t x_ns, const float* low, int N, int C, int HW, float* out, void* stream) {
    TDR_REQUIRE(x && low && out && N > 0 && C > 0 && HW > 0, "tdr_sf_emerge: bad argument");
    hipLaunchKernelGGL(emerge_kernel, dim3(tdr_cdiv((long)C * HW, 1024), N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, low, C * HW, out);
    TDR_LAUNCH_CHECK("emerge_kernel");
    return TDR_OK;
}
extern "C" int tdr_sf_softmax_mix(const float* x, int64_t x_ns, const float* low, const float* lh, const float* ll, int N, int C, int HW, float* mix,
                                  void* stream) {
    TDR_REQUIRE(x && low && lh && ll && mix && N > 0 && C > 0 && HW > 0, "tdr_sf_softmax_mix: bad argument");
    hipLaunchKernelGGL(softmax_mix_kernel, dim3(tdr_cdiv(HW, 256), N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, low, lh, ll, C, HW, mix);
    TDR_LAUNCH_CHECK("softmax_mix_kernel");
    return TDR_OK;
}
