// HBM-bound streaming kernels of the NAFNet-ref train step (gfx950).
// All tensors fp32 NCHW; lanes run along W/HW so every wave-instruction touches
// contiguous 256-B segments.  Reductions are two-stage with fixed order
// (deterministic, no float atomics).
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

// ===========================================================================
// LayerNorm2d  (models/archs/nafnet_arch_utils.py:264-300)
// block = 64 pixels x SLICES channel slices; thread keeps its CPT channels in
// registers (two-pass variance like the reference: mean first, then (x-mu)^2).
// ===========================================================================
template <int SLICES, int CPT>
__global__ __launch_bounds__(64 * SLICES) void ln_fwd_kernel(const float* __restrict__ x, long x_ns,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float eps, int center, int C, int HW,
                                                            float* __restrict__ y, float* __restrict__ mu,
                                                            float* __restrict__ rstd) {
    __shared__ float red[SLICES][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int px = blockIdx.x * 64 + lane, n = blockIdx.y;
    const bool pok = px < HW;
    const float* xn = x + (long)n * x_ns + (pok ? px : HW - 1);
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) v[i] = xn[(long)min(slice + SLICES * i, C - 1) * HW];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        v[i] = (pok && slice + SLICES * i < C) ? v[i] : 0.f;
        s += v[i];
    }
    red[slice][lane] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < SLICES; ++k) tot += red[k][lane];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + SLICES * i;
        const float d = v[i] - mean;
        if (c < C) q += d * d;
    }
    red[slice][lane] = q;
    __syncthreads();
    float vt = 0.f;
#pragma unroll
    for (int k = 0; k < SLICES; ++k) vt += red[k][lane];
    const float rs = 1.0f / sqrtf(vt / (float)C + eps);
    if (!pok) return;
    float* yn = y + ((long)n * C) * HW + px;
    const float mo = center ? mean : 0.f;          // BiasFree_LayerNorm scales the uncentred x
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + SLICES * i;
        if (c < C) yn[(long)c * HW] = (v[i] - mo) * rs * w[c] + (b ? b[c] : 0.f);
    }
    if (slice == 0) {
        mu[(long)n * HW + px] = mean;
        rstd[(long)n * HW + px] = rs;
    }
}

// wide C over FEW pixels (the ViT / CLIP token LayerNorms: C = 768 .. 1280 over ~1 000 tokens): 64-pixel tiles give a dozen
// workgroups on a 256-CU chip and each walks C / 16 rows serially (33 us for [1280 x 1152]).  Here a workgroup takes 16 pixels x 64
// channel slices (64-byte row segments), a thread keeps its C / 64 values in registers: 4x the workgroups, one read of x.
template <int CPT>
__global__ __launch_bounds__(1024) void ln_fwd_narrow_kernel(const float* __restrict__ x, long x_ns, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, int center, int C, int HW,
                                                            float* __restrict__ y, float* __restrict__ mu, float* __restrict__ rstd) {
    __shared__ float red[16][16];
    const int p16 = threadIdx.x & 15, slice = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int px = blockIdx.x * 16 + p16, n = blockIdx.y;
    const bool pok = px < HW;
    const float* xn = x + (long)n * x_ns + (pok ? px : HW - 1);
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + 64 * i;
        v[i] = c < C ? xn[(long)c * HW] : 0.f;
        s += v[i];
    }
    auto block_sum = [&](float t) {
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        __syncthreads();
        if ((threadIdx.x & 63) < 16) red[wave][p16] = t;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k][p16];
        return tot;
    };
    const float mean = block_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const float d = (slice + 64 * i < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rs = 1.0f / sqrtf(block_sum(q) / (float)C + eps);
    if (!pok) return;
    float* yn = y + ((long)n * C) * HW + px;
    const float mo = center ? mean : 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + 64 * i;
        if (c < C) yn[(long)c * HW] = (v[i] - mo) * rs * w[c] + (b ? b[c] : 0.f);
    }
    if (slice == 0) {
        mu[(long)n * HW + px] = mean;
        rstd[(long)n * HW + px] = rs;
    }
}

// any C: re-reads x (L2-resident for the small deep maps this serves); 16 channel slices of 64 pixels
__global__ __launch_bounds__(1024) void ln_fwd_generic_kernel(const float* __restrict__ x, long x_ns,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             float eps, int center, int C, int HW,
                                                             float* __restrict__ y, float* __restrict__ mu,
                                                             float* __restrict__ rstd) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int px = blockIdx.x * 64 + lane, n = blockIdx.y;
    const bool pok = px < HW;
    const float* xn = x + (long)n * x_ns + (pok ? px : HW - 1);
    auto total = [&]() {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        return t;
    };
    float s = 0.f;
#pragma unroll 8
    for (int c = slice; c < C; c += 16) s += xn[(long)c * HW];
    red[slice][lane] = s;
    __syncthreads();
    const float mean = total() / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll 8
    for (int c = slice; c < C; c += 16) {
        const float d = xn[(long)c * HW] - mean;
        q += d * d;
    }
    red[slice][lane] = q;
    __syncthreads();
    const float rs = 1.0f / sqrtf(total() / (float)C + eps);
    if (!pok) return;
    float* yn = y + ((long)n * C) * HW + px;
    const float mo = center ? mean : 0.f;
#pragma unroll 8
    for (int c = slice; c < C; c += 16) yn[(long)c * HW] = (xn[(long)c * HW] - mo) * rs * w[c] + (b ? b[c] : 0.f);
    if (slice == 0) {
        mu[(long)n * HW + px] = mean;
        rstd[(long)n * HW + px] = rs;
    }
}

// backward, register path.  grid.x blocks walk pixel tiles (n, 64 px); per-channel
// sum(go*yhat), sum(go) are kept in registers and reduced across lanes once.
// Register budget: 5 arrays of CPT floats (x->yhat, go, acc_w, acc_b, add).
template <int SLICES, int CPT>
__global__ __launch_bounds__(64 * SLICES) void ln_bwd_kernel(
    const float* __restrict__ go, const float* __restrict__ x, long x_ns, const float* __restrict__ mu,
    const float* __restrict__ rstd, const float* __restrict__ w, const float* __restrict__ add, long add_ns, int add_C,
    int center, int N, int C, int HW, float* __restrict__ gx, float* __restrict__ part /*[grid][2][C]*/) {
    __shared__ float red[2][SLICES][64];
    // wave-uniform channel slice (readfirstlane): channel row offsets become scalar address arithmetic
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = (HW + 63) / 64;
    float aw[CPT], ab[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) { aw[i] = 0.f; ab[i] = 0.f; }
    // 32-bit element offsets from uniform base pointers (saddr + voffset addressing): the
    // 64-bit per-channel addresses of four tensors would otherwise eat 8*CPT registers.
    const unsigned uHW = (unsigned)HW;
    for (int t = blockIdx.x; t < N * tiles; t += gridDim.x) {
        const int n = t / tiles, px = (t % tiles) * 64 + lane;
        const bool pok = px < HW;
        const unsigned pxc = pok ? px : HW - 1;            // clamped: loads stay unconditional
        const float m = mu[(long)n * HW + pxc], rs = pok ? rstd[(long)n * HW + pxc] : 0.f;
        const float* xb = x + (long)n * x_ns;
        const float* gb_ = go + (long)n * C * HW;
        float* ob = gx + (long)n * C * HW;
        float xv[CPT], g0[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const unsigned c = min(slice + SLICES * i, C - 1);
            xv[i] = xb[c * uHW + pxc];
            g0[i] = gb_[c * uHW + pxc];
        }
        // BiasFree (center == 0): y = x*rs*w, so the sums run over the uncentred yu = yh + mu*rs and the
        // mean(g) term vanishes; the (x-mu)*rs factor of d rs/dx stays centred.
        const float off = center ? 0.f : m * rs;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = slice + SLICES * i;
            const bool ok = pok && c < C;
            const float gg = ok ? g0[i] : 0.f;
            const float yh = ok ? (xv[i] - m) * rs : 0.f;
            const float yu = ok ? yh + off : 0.f;
            const float gv = gg * w[min(c, C - 1)];
            xv[i] = yh; g0[i] = gv;
            s1 += gv;
            s2 += gv * yu;
            aw[i] += gg * yu;
            ab[i] += gg;
        }
        red[0][slice][lane] = s1;
        red[1][slice][lane] = s2;
        __syncthreads();
        float S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int k = 0; k < SLICES; ++k) { S1 += red[0][k][lane]; S2 += red[1][k][lane]; }
        __syncthreads();
        const float mg = center ? S1 / (float)C : 0.f, mgy = S2 / (float)C;
        float av[CPT];
        if (add) {
            const float* ab_ = add + (long)n * add_ns;
#pragma unroll
            for (int i = 0; i < CPT; ++i) av[i] = ab_[(unsigned)min(slice + SLICES * i, add_C - 1) * uHW + pxc];
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = slice + SLICES * i;
            float v = rs * (g0[i] - xv[i] * mgy - mg);
            if (add && c < add_C) v += av[i];
            if (pok && c < C) ob[(unsigned)c * uHW + pxc] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + SLICES * i;
        const float sw = wave_sum(aw[i]), sb = wave_sum(ab[i]);
        if (lane == 0 && c < C) {
            part[((long)blockIdx.x * 2 + 0) * C + c] = sw;
            part[((long)blockIdx.x * 2 + 1) * C + c] = sb;
        }
    }
}

// any C: 64 pixels x 16 channel slices; two passes over the channels (the second re-reads the
// 64-pixel x C tile from L2).  Loads are unconditional (clamped pixel index).
// PART: the second pass also reduces this block's 64 pixels of sum(go*yhat), sum(go) per channel into
// part[block][2][C] (instead of a separate ln_param_grad_kernel pass over go and x).
template <bool PART>
__global__ __launch_bounds__(1024) void ln_bwd_generic_kernel(
    const float* __restrict__ go, const float* __restrict__ x, long x_ns, const float* __restrict__ mu,
    const float* __restrict__ rstd, const float* __restrict__ w, const float* __restrict__ add, long add_ns, int add_C,
    int center, int C, int HW, float* __restrict__ gx, float* __restrict__ part) {
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int px = blockIdx.x * 64 + lane, n = blockIdx.y;
    const bool pok = px < HW;
    const unsigned pxc = pok ? px : HW - 1, uHW = (unsigned)HW;
    const float m = mu[(long)n * HW + pxc], rs = rstd[(long)n * HW + pxc];
    const float* xb = x + (long)n * x_ns;
    const float* gb_ = go + (long)n * C * HW;
    const float off = center ? 0.f : m * rs;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int c = slice; c < C; c += 16) {
        const float g = gb_[(unsigned)c * uHW + pxc] * w[c];
        const float yh = (xb[(unsigned)c * uHW + pxc] - m) * rs;
        s1 += g; s2 += g * (yh + off);
    }
    red[0][slice][lane] = s1; red[1][slice][lane] = s2;
    __syncthreads();
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { S1 += red[0][k][lane]; S2 += red[1][k][lane]; }
    const float mg = center ? S1 / (float)C : 0.f, mgy = S2 / (float)C;
    const float* ab_ = add ? add + (long)n * add_ns : xb;
    float* ob = gx + (long)n * C * HW;
    for (int c0 = slice; c0 < C; c0 += 64) {        // four channels per trip: 12 loads in flight, then the reductions
        float g0[4], xv[4], av[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = min(c0 + 16 * k, C - 1);
            g0[k] = gb_[(unsigned)c * uHW + pxc];
            xv[k] = xb[(unsigned)c * uHW + pxc];
            av[k] = ab_[(unsigned)min(c, add ? add_C - 1 : C - 1) * uHW + pxc];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + 16 * k;
            if (c < C) {                              // wave-uniform
                const float g = g0[k] * w[c];
                const float yh = (xv[k] - m) * rs;
                float v = rs * (g - yh * mgy - mg);
                if (add && c < add_C) v += av[k];
                if (pok) ob[(unsigned)c * uHW + pxc] = v;
                if (PART) {
                    const float sw = wave_sum(pok ? g0[k] * (yh + off) : 0.f), sb = wave_sum(pok ? g0[k] : 0.f);
                    if (lane == 0) {
                        float* pb = part + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 2) * C;
                        pb[c] = sw;
                        pb[C + c] = sb;
                    }
                }
            }
        }
    }
}

// single pass for C <= 16*CPT: the 64-pixel x C tile of go and x lives in registers (CPT channels per thread, all
// 2*CPT loads in flight at once), with the per-block parameter-gradient partials of the PART variant above.
template <int CPT>
__global__ __launch_bounds__(1024) void ln_bwd_cached_kernel(
    const float* __restrict__ go, const float* __restrict__ x, long x_ns, const float* __restrict__ mu,
    const float* __restrict__ rstd, const float* __restrict__ w, const float* __restrict__ add, long add_ns, int add_C,
    int center, int C, int HW, float* __restrict__ gx, float* __restrict__ part) {
    __shared__ float red[2][16][64];
    // the channel slice through readfirstlane: channel row offsets are then scalar, every access is
    // (scalar row base) + (one 32-bit pixel offset register) instead of a 64-bit address per channel
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = blockIdx.x * 64 + lane, n = blockIdx.y;
    const bool pok = px < HW;
    const unsigned pxc = pok ? px : HW - 1;
    const float* xb = x + (long)n * x_ns;
    const float* gb_ = go + (long)n * C * HW;
    float g0[CPT], xv[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const long row = (long)min(slice + 16 * i, C - 1) * HW;
        g0[i] = (gb_ + row)[pxc];
        xv[i] = (xb + row)[pxc];
    }
    const float m = mu[(long)n * HW + pxc], rs = rstd[(long)n * HW + pxc];
    const float off = center ? 0.f : m * rs;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {            // branch-free: channels past C carry clamped loads and a zero weight
        const int c = slice + 16 * i;
        const float wc = c < C ? w[min(c, C - 1)] : 0.f;
        g0[i] = c < C ? g0[i] : 0.f;
        xv[i] = (xv[i] - m) * rs;                // yhat
        const float g = g0[i] * wc;
        s1 += g; s2 += g * (xv[i] + off);
    }
    red[0][slice][lane] = s1; red[1][slice][lane] = s2;
    __syncthreads();
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { S1 += red[0][k][lane]; S2 += red[1][k][lane]; }
    const float mg = center ? S1 / (float)C : 0.f, mgy = S2 / (float)C;
    const float* ab_ = add ? add + (long)n * add_ns : xb;
    float* ob = gx + (long)n * C * HW;
    float* pb = part + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 2) * C;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + 16 * i, cc = min(c, C - 1);
        const long row = (long)cc * HW;
        float v = rs * (g0[i] * w[cc] - xv[i] * mgy - mg);
        const float av = (ab_ + (long)min(cc, add ? add_C - 1 : C - 1) * HW)[pxc];
        if (add && c < add_C) v += av;
        const float sw = wave_sum_dpp(pok ? g0[i] * (xv[i] + off) : 0.f), sb = wave_sum_dpp(pok ? g0[i] : 0.f);
        if (c < C) {                              // wave-uniform
            if (pok) (ob + row)[pxc] = v;
            if (lane == 0) { pb[c] = sw; pb[C + c] = sb; }
        }
    }
}

// generic path parameter grads: block (c, split) walks channel c's pixels; part[split][2][C]
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ go, const float* __restrict__ x,
                                                           long x_ns, const float* __restrict__ mu,
                                                           const float* __restrict__ rstd, int center, int N, int C,
                                                           int HW, float* __restrict__ part) {
    __shared__ float red[2][4];
    const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const long total = (long)N * HW;
    float sw = 0.f, sb = 0.f;
    for (long i = (long)split * 256 + threadIdx.x; i < total; i += (long)nsplit * 256) {
        const int n = (int)(i / HW), px = (int)(i % HW);
        const float g0 = go[((long)n * C + c) * HW + px];
        const float yh = (x[(long)n * x_ns + (long)c * HW + px] - (center ? mu[i] : 0.f)) * rstd[i];
        sw += g0 * yh; sb += g0;
    }
    sw = wave_sum(sw); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sw; red[1][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long)split * 2 + 0) * C + c] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[((long)split * 2 + 1) * C + c] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// LayerNorm parameter gradients: part [nparts][2][C] -> o0[c] = sum_k part[k][0][c], o1[c] = sum_k part[k][1][c]
// (blockIdx.y picks the half; one launch for both)
template <int KL>
__global__ __launch_bounds__(64 * KL) void pair_sum_partials_kernel(const float* __restrict__ part, int nparts, int C,
                                                                    float* __restrict__ o0, float* __restrict__ o1) {
    __shared__ float red[KL][64];
    const int lane = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int ec = e < C ? e : C - 1;
    const float* p = part + (long)blockIdx.y * C + ec;
    const long stride = 2L * C;
    float s0 = 0.f, s1 = 0.f;
    int k = kl;
    for (; k + KL < nparts; k += 2 * KL) { s0 += p[(long)k * stride]; s1 += p[(long)(k + KL) * stride]; }
    if (k < nparts) s0 += p[(long)k * stride];
    red[kl][lane] = s0 + s1;
    __syncthreads();
    if (kl == 0 && e < C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < KL; ++q) t += red[q][lane];
        (blockIdx.y ? o1 : o0)[e] = t;
    }
}

// the same reduction for MANY problems in one launch: table rows {part, o0, o1, nparts, C} (64-bit words: device pointers and the problem's
// shape), blockIdx.z = problem, grid.x sized for the widest one; per problem exactly pair_sum_partials_kernel's summation order (bit-identical)
template <int KL>
__global__ __launch_bounds__(64 * KL) void pair_sum_partials_multi_kernel(const long long* __restrict__ tab) {
    __shared__ float red[KL][64];
    const long long* row = tab + 5L * blockIdx.z;
    const int nparts = (int)row[3], C = (int)row[4];
    if ((int)blockIdx.x * 64 >= C) return;                   // (uniform per workgroup: before any barrier)
    const float* part = reinterpret_cast<const float*>(row[0]);
    float* o = reinterpret_cast<float*>(blockIdx.y ? row[2] : row[1]);
    const int lane = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int ec = e < C ? e : C - 1;
    const float* p = part + (long)blockIdx.y * C + ec;
    const long stride = 2L * C;
    float s0 = 0.f, s1 = 0.f;
    int k = kl;
    for (; k + KL < nparts; k += 2 * KL) { s0 += p[(long)k * stride]; s1 += p[(long)(k + KL) * stride]; }
    if (k < nparts) s0 += p[(long)k * stride];
    red[kl][lane] = s0 + s1;
    __syncthreads();
    if (kl == 0 && e < C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < KL; ++q) t += red[q][lane];
        o[e] = t;
    }
}

constexpr int LN_BWD_GRID = 1024;     // workspace rows (upper bound of the persistent grid)
static const int ln_bwd_grid = tdr_tune_env("TDR_LN_BWD_GRID") ? atoi(tdr_tune_env("TDR_LN_BWD_GRID")) : 256;
constexpr int LN_GEN_SPLITS = 16;

// ===========================================================================
// SCA 1x1 on pooled vectors and the tiny parameter-gradient chains
// ===========================================================================
__global__ __launch_bounds__(256) void sca_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ wsca,
                                                     const float* __restrict__ bsca, int C, float* __restrict__ s) {
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y, lane = threadIdx.x & 63;
    if (co >= C) return;
    float a = 0.f;
    for (int ci = lane; ci < C; ci += 64) a += wsca[(long)co * C + ci] * pooled[(long)n * C + ci];
    a = wave_sum(a);
    if (lane == 0) s[(long)n * C + co] = a + bsca[co];
}

// one block per co row: dW3 row, db3, dbeta
__global__ __launch_bounds__(256) void sca_bwd_rows_kernel(const float* __restrict__ G3, const float* __restrict__ S3,
                                                          const float* __restrict__ w3, const float* __restrict__ b3,
                                                          const float* __restrict__ beta, const float* __restrict__ s,
                                                          int N, int C, float* __restrict__ dw3, float* __restrict__ db3,
                                                          float* __restrict__ dbeta) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    const float bt = beta[co];
    float acc = 0.f;
    for (int ci = threadIdx.x; ci < C; ci += 256) {
        float sg = 0.f;
        for (int n = 0; n < N; ++n) sg += s[(long)n * C + ci] * G3[((long)n * C + co) * C + ci];
        dw3[(long)co * C + ci] = bt * sg;
        acc += w3[(long)co * C + ci] * sg;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        db3[co] = bt * S3[co];
        dbeta[co] = tot + b3[co] * S3[co];
    }
}

// ds[n][ci] = sum_co beta[co] W3[co][ci] G3[n][co][ci]; block = 64 ci x 16 co-slices
__global__ __launch_bounds__(1024) void sca_bwd_ds_kernel(const float* __restrict__ G3, const float* __restrict__ w3,
                                                         const float* __restrict__ beta, int N, int C, float* __restrict__ ds) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int ci = blockIdx.x * 64 + lane, n = blockIdx.y;
    const int cic = min(ci, C - 1);
    float a = 0.f;
    for (int co = sl; co < C; co += 16) a += beta[co] * w3[(long)co * C + cic] * G3[((long)n * C + co) * C + cic];
    red[sl][lane] = a;
    __syncthreads();
    if (sl == 0 && ci < C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][lane];
        ds[(long)n * C + ci] = t;
    }
}

// dWsca[ci][cj] = sum_n ds[n][ci] pooled[n][cj]   (grid: (C/256, C))
__global__ void sca_bwd_dw_kernel(const float* __restrict__ ds, const float* __restrict__ pooled, int N, int C,
                                  float* __restrict__ dwsca) {
    const int cj = blockIdx.x * blockDim.x + threadIdx.x, ci = blockIdx.y;
    if (cj >= C) return;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += ds[(long)n * C + ci] * pooled[(long)n * C + cj];
    dwsca[(long)ci * C + cj] = a;
}

// dpooled[n][cj] = sum_ci Wsca[ci][cj] ds[n][ci] (block = 64 cj x 16 ci-slices); dbsca[cj] = sum_n ds[n][cj]
__global__ __launch_bounds__(1024) void sca_bwd_tail_kernel(const float* __restrict__ ds, const float* __restrict__ wsca,
                                                           int N, int C, float* __restrict__ dbsca,
                                                           float* __restrict__ dpooled) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int cj = blockIdx.x * 64 + lane, n = blockIdx.y;
    const int cjc = min(cj, C - 1);
    float a = 0.f;
    for (int ci = sl; ci < C; ci += 16) a += wsca[(long)ci * C + cjc] * ds[(long)n * C + ci];
    red[sl][lane] = a;
    __syncthreads();
    if (sl == 0 && cj < C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][lane];
        dpooled[(long)n * C + cj] = t;
        if (n == 0) {
            float sb = 0.f;
            for (int m = 0; m < N; ++m) sb += ds[(long)m * C + cj];
            dbsca[cj] = sb;
        }
    }
}

// sca_bwd_rows_kernel and sca_bwd_ds_kernel are independent readers of G3: one launch, role by block index
// (blocks [0, nds): ds role over (64-channel tile, image); then ceil(C/4) blocks with four co rows each)
__global__ __launch_bounds__(1024) void sca_bwd_a_kernel(const float* __restrict__ G3, const float* __restrict__ S3,
                                                        const float* __restrict__ w3, const float* __restrict__ b3,
                                                        const float* __restrict__ beta, const float* __restrict__ s, int N,
                                                        int C, int nds, float* __restrict__ dw3, float* __restrict__ db3,
                                                        float* __restrict__ dbeta, float* __restrict__ ds) {
    __shared__ float red[16][64];
    if ((int)blockIdx.x < nds) {
        const int tiles = (C + 63) / 64;
        const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int ci = (blockIdx.x % tiles) * 64 + lane, n = blockIdx.x / tiles;
        const int cic = min(ci, C - 1);
        float a = 0.f;
        for (int co = sl; co < C; co += 16) a += beta[co] * w3[(long)co * C + cic] * G3[((long)n * C + co) * C + cic];
        red[sl][lane] = a;
        __syncthreads();
        if (sl == 0 && ci < C) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q][lane];
            ds[(long)n * C + ci] = t;
        }
        return;
    }
    const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const int co = ((int)blockIdx.x - nds) * 4 + sub;
    const bool live = co < C;
    const int coc = live ? co : C - 1;
    const float bt = beta[coc];
    float acc = 0.f;
    if (live)
        for (int ci = tid; ci < C; ci += 256) {
            float sg = 0.f;
            for (int n = 0; n < N; ++n) sg += s[(long)n * C + ci] * G3[((long)n * C + co) * C + ci];
            dw3[(long)co * C + ci] = bt * sg;
            acc += w3[(long)co * C + ci] * sg;
        }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[sub][tid >> 6] = acc;
    __syncthreads();
    if (tid == 0 && live) {
        const float tot = (red[sub][0] + red[sub][1]) + (red[sub][2] + red[sub][3]);
        db3[co] = bt * S3[co];
        dbeta[co] = tot + b3[co] * S3[co];
    }
}

// sca_bwd_dw_kernel and sca_bwd_tail_kernel both consume ds only: one launch (blocks [0, ntail): tail role over
// (64-channel tile, image); then ceil(C/4) * ceil(C/256) blocks of four dWsca rows x 256 columns)
__global__ __launch_bounds__(1024) void sca_bwd_b_kernel(const float* __restrict__ ds, const float* __restrict__ pooled,
                                                        const float* __restrict__ wsca, int N, int C, int ntail,
                                                        float* __restrict__ dwsca, float* __restrict__ dbsca,
                                                        float* __restrict__ dpooled) {
    __shared__ float red[16][64];
    if ((int)blockIdx.x < ntail) {
        const int tiles = (C + 63) / 64;
        const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int cj = (blockIdx.x % tiles) * 64 + lane, n = blockIdx.x / tiles;
        const int cjc = min(cj, C - 1);
        float a = 0.f;
        for (int ci = sl; ci < C; ci += 16) a += wsca[(long)ci * C + cjc] * ds[(long)n * C + ci];
        red[sl][lane] = a;
        __syncthreads();
        if (sl == 0 && cj < C) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q][lane];
            dpooled[(long)n * C + cj] = t;
            if (n == 0) {
                float sb = 0.f;
                for (int m = 0; m < N; ++m) sb += ds[(long)m * C + cj];
                dbsca[cj] = sb;
            }
        }
        return;
    }
    const int colb = (C + 255) / 256;
    const int r = (int)blockIdx.x - ntail;
    const int ci = (r / colb) * 4 + (threadIdx.x >> 8), cj = (r % colb) * 256 + (threadIdx.x & 255);
    if (ci >= C || cj >= C) return;
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += ds[(long)n * C + ci] * pooled[(long)n * C + cj];
    dwsca[(long)ci * C + cj] = a;
}

__global__ __launch_bounds__(256) void scaled_conv_param_kernel(const float* __restrict__ G, const float* __restrict__ S,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               const float* __restrict__ gamma, int Cin,
                                                               float* __restrict__ dw, float* __restrict__ db,
                                                               float* __restrict__ dgamma) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    const float gm = gamma[co];
    float acc = 0.f;
    for (int ci = threadIdx.x; ci < Cin; ci += 256) {
        const float gv = G[(long)co * Cin + ci];
        dw[(long)co * Cin + ci] = gm * gv;
        acc += w[(long)co * Cin + ci] * gv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        db[co] = gm * S[co];
        dgamma[co] = (red[0] + red[1]) + (red[2] + red[3]) + b[co] * S[co];
    }
}

// many problems: table rows {G, S, w, b, gamma, dw, db, dgamma, Cout, Cin}, blockIdx.y = problem, grid.x sized for the widest one;
// scaled_conv_param_kernel's order
__global__ __launch_bounds__(256) void scaled_conv_param_multi_kernel(const long long* __restrict__ tab) {
    __shared__ float red[4];
    const long long* row = tab + 10L * blockIdx.y;
    const int Cin = (int)row[9];
    if ((int)blockIdx.x >= (int)row[8]) return;
    const float* G = reinterpret_cast<const float*>(row[0]);
    const float* S = reinterpret_cast<const float*>(row[1]);
    const float* w = reinterpret_cast<const float*>(row[2]);
    const float* b = reinterpret_cast<const float*>(row[3]);
    const float* gamma = reinterpret_cast<const float*>(row[4]);
    float* dw = reinterpret_cast<float*>(row[5]);
    float* db = reinterpret_cast<float*>(row[6]);
    float* dgamma = reinterpret_cast<float*>(row[7]);
    const int co = blockIdx.x;
    const float gm = gamma[co];
    float acc = 0.f;
    for (int ci = threadIdx.x; ci < Cin; ci += 256) {
        const float gv = G[(long)co * Cin + ci];
        dw[(long)co * Cin + ci] = gm * gv;
        acc += w[(long)co * Cin + ci] * gv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        db[co] = gm * S[co];
        dgamma[co] = (red[0] + red[1]) + (red[2] + red[3]) + b[co] * S[co];
    }
}

// ===========================================================================
// reductions / glue
// ===========================================================================
constexpr int CS_SPLITS = 32;
__global__ __launch_bounds__(256) void chansum_kernel(const float* __restrict__ x, long x_ns, int N, int C, int HW,
                                                     float* __restrict__ part) {
    __shared__ float red[4];
    const int c = blockIdx.x, split = blockIdx.y;
    const long total = (long)N * HW;
    float s = 0.f;
    for (long i = (long)split * 256 + threadIdx.x; i < total; i += (long)CS_SPLITS * 256) {
        const int n = (int)(i / HW), px = (int)(i % HW);
        s += x[(long)n * x_ns + (long)c * HW + px];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)c * CS_SPLITS + split] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void chansum_finish_kernel(const float* __restrict__ part, int C, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < CS_SPLITS; ++k) s += part[(long)c * CS_SPLITS + k];
    out[c] = s;
}

template <bool ADD>
__global__ void rows_kernel(const float* __restrict__ src, long src_ns, float* __restrict__ dst, long dst_ns, long len4) {
    const int n = blockIdx.y;
    const float4* s = reinterpret_cast<const float4*>(src + (long)n * src_ns);
    float4* d = reinterpret_cast<float4*>(dst + (long)n * dst_ns);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < len4; i += (long)gridDim.x * blockDim.x) {
        float4 v = s[i];
        if (ADD) { const float4 o = d[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        d[i] = v;
    }
}

template <bool ADD>
__global__ void rows_scalar_kernel(const float* __restrict__ src, long src_ns, float* __restrict__ dst, long dst_ns, long len) {
    const int n = blockIdx.y;
    const float* s = src + (long)n * src_ns;
    float* d = dst + (long)n * dst_ns;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < len; i += (long)gridDim.x * blockDim.x)
        d[i] = ADD ? d[i] + s[i] : s[i];
}

__global__ void pixel_unshuffle2_kernel(const float* __restrict__ in, int C, int H, int W, long total,
                                        float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int m = (int)(r % (4 * C)); const long n = r / (4 * C);
        const int c = m >> 2, a = (m >> 1) & 1, b = m & 1;
        out[i] = in[((n * C + c) * (2L * H) + 2 * y + a) * (2L * W) + 2 * x + b];
    }
}

__global__ void pad_crop_kernel(const float* __restrict__ src, int Hs, int Ws, int Hd, int Wd, long total,
                                float* __restrict__ dst) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wd); long r = i / Wd;
        const int y = (int)(r % Hd); const long nc = r / Hd;
        dst[i] = (y < Hs && x < Ws) ? src[(nc * Hs + y) * Ws + x] : 0.f;
    }
}

__global__ void relu_bwd_kernel(const float* __restrict__ go, const float* __restrict__ act, long n4,
                                float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 g = reinterpret_cast<const float4*>(go)[i], a = reinterpret_cast<const float4*>(act)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f,
                                                        a.w > 0.f ? g.w : 0.f);
    }
}

constexpr int L1_BLOCKS = 1024;
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ p, const float* __restrict__ t, long numel,
                                                float gscale, const TdrStepGuard* __restrict__ guard, float* __restrict__ dpred,
                                                double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    if (guard) gscale *= guard->scale;                 // power of two: exact
    for (long i = blockIdx.x * 256L + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
        const float d = p[i] - t[i];
        s += (double)fabsf(d);
        dpred[i] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void l1_finish_kernel(const double* __restrict__ part, int nparts, double scale, float* __restrict__ loss) {
    // one wave, fixed order: lane-strided partial sums, then the xor butterfly (deterministic)
    double s = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 64) s += part[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) loss[0] = (float)(s * scale);
}

// The other criteria of losses/losses.py (MSELoss :55-82, CharbonnierLoss :111-122, PSNRLoss :84-109): value + gradient in one pass,
// same deterministic two-level reduction as l1_kernel.  KIND 1: d^2, KIND 2: sqrt(d^2 + eps^2).
template <int KIND>
__global__ __launch_bounds__(256) void pixloss_kernel(const float* __restrict__ p, const float* __restrict__ t, long numel,
                                                     float gscale, float eps2, const TdrStepGuard* __restrict__ guard,
                                                     float* __restrict__ dpred, double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    if (guard) gscale *= guard->scale;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
        const float d = p[i] - t[i];
        if (KIND == 1) {
            s += (double)(d * d);
            dpred[i] = 2.f * d * gscale;
        } else {
            const float r = sqrtf(d * d + eps2);
            s += (double)r;
            dpred[i] = d / r * gscale;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// PSNRLoss: per-image mean of d^2 (toY: d = sum_c coef_c (p_c - t_c) / 255 on the 1-channel luma, the +16 cancels)
constexpr int PSNR_BLOCKS = 64;        // per image
__device__ __forceinline__ float psnr_diff(const float* p, const float* t, long base, long HW, long i, int toY) {
    if (!toY) return p[base + i] - t[base + i];
    const float d0 = p[base + i] - t[base + i], d1 = p[base + HW + i] - t[base + HW + i],
                d2 = p[base + 2 * HW + i] - t[base + 2 * HW + i];
    return (65.481f * d0 + 128.553f * d1 + 24.966f * d2) / 255.f;
}
__global__ __launch_bounds__(256) void psnr_sum_kernel(const float* __restrict__ p, const float* __restrict__ t, long CHW, long HW,
                                                      int toY, double* __restrict__ part) {
    __shared__ double red[4];
    const long base = (long)blockIdx.y * CHW, cnt = toY ? HW : CHW;
    double s = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < cnt; i += (long)gridDim.x * 256) {
        const float d = psnr_diff(p, t, base, HW, i, toY);
        s += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.y * PSNR_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double psnr_image_mse(const double* part, int n, long cnt) {
    double s = 0.0;                                    // fixed order: every caller gets the same bits
    for (int k = 0; k < PSNR_BLOCKS; ++k) s += part[n * PSNR_BLOCKS + k];
    return s / (double)cnt;
}
__global__ __launch_bounds__(256) void psnr_grad_kernel(const float* __restrict__ p, const float* __restrict__ t, long CHW, long HW,
                                                       int toY, int N, float wscale, const TdrStepGuard* __restrict__ guard,
                                                       const double* __restrict__ part, float* __restrict__ dpred) {
    const int n = blockIdx.y;
    const long base = (long)n * CHW, cnt = toY ? HW : CHW;
    float gscale = wscale;
    if (guard) gscale *= guard->scale;
    // d/dp of w*scale/N * log(mse_n + 1e-8), mse_n = mean d^2
    const float f = (float)((double)gscale / (double)N / (psnr_image_mse(part, n, cnt) + 1e-8) * 2.0 / (double)cnt);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < cnt; i += (long)gridDim.x * 256) {
        const float d = psnr_diff(p, t, base, HW, i, toY);
        if (!toY) {
            dpred[base + i] = f * d;
        } else {
            dpred[base + i] = f * d * (65.481f / 255.f);
            dpred[base + HW + i] = f * d * (128.553f / 255.f);
            dpred[base + 2 * HW + i] = f * d * (24.966f / 255.f);
        }
    }
}
__global__ void psnr_finish_kernel(const double* __restrict__ part, int N, long cnt, double wscale, float* __restrict__ loss) {
    double s = 0.0;
    for (int n = threadIdx.x; n < N; n += 64) s += log(psnr_image_mse(part, n, cnt) + 1e-8);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)N * wscale);
}

inline int grid1d(long total, int cap = 8192) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int tdr_layernorm2d_fwd(const float* x, int64_t x_ns, const float* w, const float* b, float eps, int center,
                                   int N, int C, int HW, float* y, float* mu, float* rstd, void* stream) {
    TDR_REQUIRE(x && w && y && mu && rstd, "tdr_layernorm2d_fwd: null pointer");
    TDR_REQUIRE(N > 0 && C > 0 && HW > 0, "tdr_layernorm2d_fwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(tdr_cdiv(HW, 64), N);
#define LN_FWD(S, P) hipLaunchKernelGGL((ln_fwd_kernel<S, P>), grid, dim3(64 * S), 0, st, x, (long)x_ns, w, b, eps, center, C, HW, y, mu, rstd)
    if (C <= 32) LN_FWD(4, 8);
    else if (C <= 64) LN_FWD(4, 16);
    else if (C <= 128) LN_FWD(4, 32);
    else if (C <= 256) LN_FWD(8, 32);
    else if (C <= 512) LN_FWD(16, 32);
    else if (C <= 64 * 24 && (long)grid.x * N < 512) {      // wide rows over few pixels: 16-pixel workgroups (see ln_fwd_narrow_kernel)
        dim3 g16(tdr_cdiv(HW, 16), N);
        if (C <= 64 * 12) hipLaunchKernelGGL(ln_fwd_narrow_kernel<12>, g16, dim3(1024), 0, st, x, (long)x_ns, w, b, eps, center, C, HW, y, mu, rstd);
        else hipLaunchKernelGGL(ln_fwd_narrow_kernel<24>, g16, dim3(1024), 0, st, x, (long)x_ns, w, b, eps, center, C, HW, y, mu, rstd);
    }
    else hipLaunchKernelGGL(ln_fwd_generic_kernel, grid, dim3(1024), 0, st, x, (long)x_ns, w, b, eps, center, C, HW, y, mu, rstd);
#undef LN_FWD
    TDR_LAUNCH_CHECK("ln_fwd");
    return TDR_OK;
}

extern "C" int64_t tdr_ln_ws_floats(int N, int C, int HW) {
    (void)N; (void)HW;
    return (int64_t)LN_BWD_GRID * 2 * C;
}

extern "C" int tdr_layernorm2d_bwd(const float* go, const float* x, int64_t x_ns, const float* mu, const float* rstd,
                                   const float* w, const float* add, int64_t add_ns, int add_C, int center, int N, int C,
                                   int HW, float* gx, float* gw, float* gb, float* ws, void* stream) {
    TDR_REQUIRE(go && x && mu && rstd && w && gx && gw && gb && ws, "tdr_layernorm2d_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int tiles = tdr_cdiv(HW, 64) * N;
    int nparts;
    if (C <= 128) {
        const int cap = ln_bwd_grid < 1 ? 1 : (ln_bwd_grid > LN_BWD_GRID ? LN_BWD_GRID : ln_bwd_grid);
        const int grid = tiles < cap ? tiles : cap;
        nparts = grid;
#define LN_BWD(S, P) hipLaunchKernelGGL((ln_bwd_kernel<S, P>), dim3(grid), dim3(64 * S), 0, st, go, x, (long)x_ns, mu, rstd, w, add, (long)add_ns, add_C, center, N, C, HW, gx, ws)
        if (C <= 32) LN_BWD(4, 8);
        else if (C <= 64) LN_BWD(4, 16);
        else LN_BWD(8, 16);
#undef LN_BWD
    } else {
        static const bool fuse = tdr_tune_env("TDR_LN_NOFUSE") == nullptr;
        if (tiles <= LN_BWD_GRID && fuse) {   // the workspace holds LN_BWD_GRID partial rows: one per 64-pixel block
            static const bool cached = tdr_tune_env("TDR_LN_NOCACHE") == nullptr;
#define LN_BWDC(P) hipLaunchKernelGGL(ln_bwd_cached_kernel<P>, dim3(tdr_cdiv(HW, 64), N), dim3(1024), 0, st, go, x, (long)x_ns, mu, rstd, w, add, (long)add_ns, add_C, center, C, HW, gx, ws)
            if (cached && C <= 256) LN_BWDC(16);
            else if (cached && C <= 512) LN_BWDC(32);
            else
                hipLaunchKernelGGL(ln_bwd_generic_kernel<true>, dim3(tdr_cdiv(HW, 64), N), dim3(1024), 0, st, go, x, (long)x_ns,
                                   mu, rstd, w, add, (long)add_ns, add_C, center, C, HW, gx, ws);
#undef LN_BWDC
            nparts = tiles;
        } else {
            hipLaunchKernelGGL(ln_bwd_generic_kernel<false>, dim3(tdr_cdiv(HW, 64), N), dim3(1024), 0, st, go, x, (long)x_ns, mu,
                               rstd, w, add, (long)add_ns, add_C, center, C, HW, gx, (float*)nullptr);
            hipLaunchKernelGGL(ln_param_grad_kernel, dim3(C, LN_GEN_SPLITS), dim3(256), 0, st, go, x, (long)x_ns, mu, rstd, center,
                               N, C, HW, ws);
            nparts = LN_GEN_SPLITS;
        }
    }
    hipLaunchKernelGGL(pair_sum_partials_kernel<16>, dim3(tdr_cdiv(C, 64), 2), dim3(1024), 0, st, ws, nparts, C, gw, gb);
    TDR_LAUNCH_CHECK("ln_bwd");
    return TDR_OK;
}

// part [nparts][2][C] -> o0[c] = sum_k part[k][0][c], o1[c] = sum_k part[k][1][c] (fixed order: deterministic)
// many partial rows (the fused NAFBlock backward kernels emit one per 64 pixels: 16384 at 512 x 512, N = 4): a first stage
// folds them 256-to-1 with a grid wide enough to fill the chip -- the single-stage kernel is only C / 64 x 2 workgroups.
// part [nparts][2][C] -> mid [S][2][C], S = ceil(nparts / 256); row s sums rows s*256 .. in order (deterministic).
// Narrow rows (C2 = 64 / 128: the C = 32 / 64 levels, where the partial rows are most numerous) would leave 3 / 4 or 1 / 2 of the block's
// threads idle with 256 serial row reads each: there the block's 256 rows are cut into SL = 256 / C2 slices, one per thread group, and the
// slice sums are combined through LDS in slice order (still a fixed order: deterministic).  32 -> ~10 us per launch at C = 32.
__global__ __launch_bounds__(256) void pair_fold_partials_kernel(const float* __restrict__ part, int nparts, int C2,
                                                                float* __restrict__ mid) {
    __shared__ float red[256];
    const int SL = C2 >= 256 ? 1 : 256 / C2;                 // (C2 is a power of two below 256 on this path, or >= 256)
    const int el = SL > 1 ? (int)threadIdx.x % C2 : (int)threadIdx.x, sl = SL > 1 ? (int)threadIdx.x / C2 : 0;
    const int e = SL > 1 ? el : blockIdx.x * 256 + el;       // element of the [2][C] row
    const bool live = e < C2 && sl < SL;
    const int per = 256 / SL;
    const int r0 = blockIdx.y * 256 + sl * per, r1 = min(r0 + per, nparts);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (live) {
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            s0 += part[(long)r * C2 + e]; s1 += part[(long)(r + 1) * C2 + e];
            s2 += part[(long)(r + 2) * C2 + e]; s3 += part[(long)(r + 3) * C2 + e];
        }
        for (; r < r1; ++r) s0 += part[(long)r * C2 + e];
    }
    const float t = (s0 + s1) + (s2 + s3);
    if (SL == 1) {
        if (live) mid[(long)blockIdx.y * C2 + e] = t;
        return;
    }
    red[threadIdx.x] = live ? t : 0.f;
    __syncthreads();
    if (sl == 0 && e < C2) {
        float a = 0.f;
        for (int q = 0; q < SL; ++q) a += red[q * C2 + el];
        mid[(long)blockIdx.y * C2 + e] = a;
    }
}

extern "C" int64_t tdr_pair_sum_mid_floats(int nparts, int C) { return nparts > 1024 ? (int64_t)tdr_cdiv(nparts, 256) * 2 * C : 0; }

// mid: scratch of tdr_pair_sum_mid_floats(nparts, C) floats (may be NULL when that is 0)
extern "C" int tdr_pair_sum_partials(const float* part, int nparts, int C, float* o0, float* o1, float* mid, void* stream) {
    TDR_REQUIRE(part && o0 && o1 && nparts > 0 && C > 0, "tdr_pair_sum_partials: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (nparts > 1024) {
        TDR_REQUIRE(mid, "tdr_pair_sum_partials: %d partial rows need the fold scratch", nparts);
        const int S = tdr_cdiv(nparts, 256);
        hipLaunchKernelGGL(pair_fold_partials_kernel, dim3(tdr_cdiv(2 * C, 256), S), dim3(256), 0, st, part, nparts, 2 * C, mid);
        part = mid;
        nparts = S;
    }
    hipLaunchKernelGGL(pair_sum_partials_kernel<16>, dim3(tdr_cdiv(C, 64), 2), dim3(1024), 0, st, part, nparts, C, o0, o1);
    TDR_LAUNCH_CHECK("pair_sum_partials");
    return TDR_OK;
}

// table [nprob][5] of 64-bit words {part, o0, o1, nparts, C} in DEVICE memory; every problem a one-stage reduction (nparts <= 1024: the caller's
// contract -- the table is not readable here), max_C = the widest problem
extern "C" int tdr_pair_sum_partials_multi(const void* table, int nprob, int max_C, void* stream) {
    TDR_REQUIRE(table && nprob > 0 && max_C > 0, "tdr_pair_sum_partials_multi: bad argument");
    hipLaunchKernelGGL(pair_sum_partials_multi_kernel<16>, dim3(tdr_cdiv(max_C, 64), 2, nprob), dim3(1024), 0, (hipStream_t)stream,
                       static_cast<const long long*>(table));
    TDR_LAUNCH_CHECK("pair_sum_partials_multi");
    return TDR_OK;
}

// table [nprob][10] of 64-bit words {G, S, w, b, gamma, dw, db, dgamma, Cout, Cin} in DEVICE memory; max_Cout = the widest problem
extern "C" int tdr_scaled_conv_param_grads_multi(const void* table, int nprob, int max_Cout, void* stream) {
    TDR_REQUIRE(table && nprob > 0 && max_Cout > 0, "tdr_scaled_conv_param_grads_multi: bad argument");
    hipLaunchKernelGGL(scaled_conv_param_multi_kernel, dim3(max_Cout, nprob), dim3(256), 0, (hipStream_t)stream, static_cast<const long long*>(table));
    TDR_LAUNCH_CHECK("scaled_conv_param_multi_kernel");
    return TDR_OK;
}

extern "C" int tdr_sca_fwd(const float* pooled, const float* wsca, const float* bsca, int N, int C, float* s,
                           void* stream) {
    TDR_REQUIRE(pooled && wsca && bsca && s, "tdr_sca_fwd: null pointer");
    hipLaunchKernelGGL(sca_fwd_kernel, dim3(tdr_cdiv(C, 4), N), dim3(256), 0, (hipStream_t)stream, pooled, wsca, bsca, C, s);
    TDR_LAUNCH_CHECK("sca_fwd");
    return TDR_OK;
}

extern "C" int tdr_sca_bwd(const float* G3, const float* S3, const float* w3, const float* b3, const float* beta,
                           const float* s, const float* pooled, const float* wsca, int N, int C, float* dw3, float* db3,
                           float* dbeta, float* dwsca, float* dbsca, float* dpooled, float* ws, void* stream) {
    TDR_REQUIRE(G3 && S3 && w3 && b3 && beta && s && pooled && wsca && dw3 && db3 && dbeta && dwsca && dbsca && dpooled && ws,
                "tdr_sca_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    float* ds = ws;          // [N][C]
    static const bool split4 = tdr_tune_env("TDR_SCA_SPLIT") != nullptr;     // the four separate launches (reference for the merged pair)
    if (split4) {
        hipLaunchKernelGGL(sca_bwd_rows_kernel, dim3(C), dim3(256), 0, st, G3, S3, w3, b3, beta, s, N, C, dw3, db3, dbeta);
        hipLaunchKernelGGL(sca_bwd_ds_kernel, dim3(tdr_cdiv(C, 64), N), dim3(1024), 0, st, G3, w3, beta, N, C, ds);
        hipLaunchKernelGGL(sca_bwd_dw_kernel, dim3(tdr_cdiv(C, 256), C), dim3(256), 0, st, ds, pooled, N, C, dwsca);
        hipLaunchKernelGGL(sca_bwd_tail_kernel, dim3(tdr_cdiv(C, 64), N), dim3(1024), 0, st, ds, wsca, N, C, dbsca, dpooled);
    } else {
        const int nt = tdr_cdiv(C, 64) * N;
        hipLaunchKernelGGL(sca_bwd_a_kernel, dim3(nt + tdr_cdiv(C, 4)), dim3(1024), 0, st, G3, S3, w3, b3, beta, s, N, C, nt, dw3,
                           db3, dbeta, ds);
        hipLaunchKernelGGL(sca_bwd_b_kernel, dim3(nt + tdr_cdiv(C, 4) * tdr_cdiv(C, 256)), dim3(1024), 0, st, ds, pooled, wsca, N, C,
                           nt, dwsca, dbsca, dpooled);
    }
    TDR_LAUNCH_CHECK("sca_bwd");
    return TDR_OK;
}

extern "C" int tdr_scaled_conv_param_grads(const float* G, const float* S, const float* w, const float* b,
                                           const float* gamma, int Cout, int Cin, float* dw, float* db, float* dgamma,
                                           void* stream) {
    TDR_REQUIRE(G && S && w && b && gamma && dw && db && dgamma, "tdr_scaled_conv_param_grads: null pointer");
    hipLaunchKernelGGL(scaled_conv_param_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, G, S, w, b, gamma, Cin, dw,
                       db, dgamma);
    TDR_LAUNCH_CHECK("scaled_conv_param_kernel");
    return TDR_OK;
}

extern "C" int64_t tdr_chansum_ws_floats(int N, int C, int HW) {
    (void)N; (void)HW;
    return (int64_t)C * CS_SPLITS;
}

extern "C" int tdr_channel_sum(const float* x, int64_t x_ns, int N, int C, int HW, float* out, float* ws, void* stream) {
    TDR_REQUIRE(x && out && ws, "tdr_channel_sum: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chansum_kernel, dim3(C, CS_SPLITS), dim3(256), 0, st, x, (long)x_ns, N, C, HW, ws);
    hipLaunchKernelGGL(chansum_finish_kernel, dim3(tdr_cdiv(C, 256)), dim3(256), 0, st, ws, C, out);
    TDR_LAUNCH_CHECK("channel_sum");
    return TDR_OK;
}

extern "C" int tdr_copy_rows(const float* src, int64_t src_ns, float* dst, int64_t dst_ns, int N, int64_t len,
                             void* stream) {
    TDR_REQUIRE(src && dst, "tdr_copy_rows: null pointer");
    if (len % 4 || src_ns % 4 || dst_ns % 4 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) {
        hipLaunchKernelGGL(rows_scalar_kernel<false>, dim3(grid1d(len, 2048), N), dim3(256), 0, (hipStream_t)stream, src,
                           (long)src_ns, dst, (long)dst_ns, (long)len);
        TDR_LAUNCH_CHECK("copy_rows");
        return TDR_OK;
    }
    hipLaunchKernelGGL(rows_kernel<false>, dim3(grid1d(len / 4, 2048), N), dim3(256), 0, (hipStream_t)stream, src,
                       (long)src_ns, dst, (long)dst_ns, (long)(len / 4));
    TDR_LAUNCH_CHECK("copy_rows");
    return TDR_OK;
}

extern "C" int tdr_add_rows(const float* src, int64_t src_ns, float* dst, int64_t dst_ns, int N, int64_t len,
                            void* stream) {
    TDR_REQUIRE(src && dst, "tdr_add_rows: null pointer");
    if (len % 4 || src_ns % 4 || dst_ns % 4 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) {
        hipLaunchKernelGGL(rows_scalar_kernel<true>, dim3(grid1d(len, 2048), N), dim3(256), 0, (hipStream_t)stream, src,
                           (long)src_ns, dst, (long)dst_ns, (long)len);
        TDR_LAUNCH_CHECK("add_rows");
        return TDR_OK;
    }
    hipLaunchKernelGGL(rows_kernel<true>, dim3(grid1d(len / 4, 2048), N), dim3(256), 0, (hipStream_t)stream, src,
                       (long)src_ns, dst, (long)dst_ns, (long)(len / 4));
    TDR_LAUNCH_CHECK("add_rows");
    return TDR_OK;
}

extern "C" int tdr_pixel_unshuffle2(const float* in, int N, int C, int H, int W, float* out, void* stream) {
    TDR_REQUIRE(in && out, "tdr_pixel_unshuffle2: null pointer");
    const long total = (long)N * 4 * C * H * W;
    hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, in, C, H, W, total, out);
    TDR_LAUNCH_CHECK("pixel_unshuffle2");
    return TDR_OK;
}

namespace {
// paired random crop + the 8 flip / rot90 modes of the reference's data_augmentation, gathered on the device
//   a = src[n][:, top:top+P, left:left+P];  out[n] = T_mode(a)      (numpy semantics of np.flipud / np.rot90 on the H, W axes)
// and, optionally, the sigma-noise synthesis of the denoising datasets: out += noise * sigma[n].
__global__ void crop_augment_kernel(const float* __restrict__ src, long src_ns, int C, int Hs, int Ws, const int* __restrict__ top,
                                    const int* __restrict__ left, const int* __restrict__ mode, const float* __restrict__ noise,
                                    const float* __restrict__ sigma, int P, long total, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % P);
        long r = i / P;
        const int y = (int)(r % P); r /= P;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        const int q = P - 1;
        int sy, sx;
        switch (mode ? mode[n] : 0) {
            case 1: sy = q - y; sx = x; break;            // flipud
            case 2: sy = x; sx = q - y; break;            // rot90 (counter-clockwise)
            case 3: sy = x; sx = y; break;                // flipud(rot90) = transpose
            case 4: sy = q - y; sx = q - x; break;        // rot180
            case 5: sy = y; sx = q - x; break;            // flipud(rot180) = fliplr
            case 6: sy = q - x; sx = y; break;            // rot270
            case 7: sy = q - x; sx = q - y; break;        // flipud(rot270) = anti-transpose
            default: sy = y; sx = x; break;
        }
        // rows / columns past the image (only when P > Hs or P > Ws): the reference pads bottom / right by reflection first
        // (utils/utils_image.py:243-259, cv2.BORDER_REFLECT: ...cba|abc...|cba...), period 2*Hs
        int yy = (top ? top[n] : 0) + sy, xx = (left ? left[n] : 0) + sx;
        if (yy >= Hs) { yy %= 2 * Hs; if (yy >= Hs) yy = 2 * Hs - 1 - yy; }
        if (xx >= Ws) { xx %= 2 * Ws; if (xx >= Ws) xx = 2 * Ws - 1 - xx; }
        float v = src[(long)n * src_ns + ((long)c * Hs + yy) * Ws + xx];
        if (noise) v += noise[i] * (sigma ? sigma[n] : 1.f);
        out[i] = v;
    }
}
}  // namespace

/* data/transforms.py:24-84 (paired_random_crop), :223-270 (data_augmentation modes 0-7), restoration_dataset.py:464-476 (noise) */
extern "C" int tdr_crop_augment(const float* src, int64_t src_ns, int N, int C, int Hs, int Ws, const int* top, const int* left,
                                const int* mode, const float* noise, const float* sigma, int P, float* out, void* stream) {
    TDR_REQUIRE(src && out && N > 0 && C > 0 && P > 0 && Hs > 0 && Ws > 0, "tdr_crop_augment: bad argument (patch %d of %d x %d)", P, Hs, Ws);
    const long total = (long)N * C * P * P;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(crop_augment_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src, (long)src_ns, C, Hs, Ws, top, left,
                       mode, noise, sigma, P, total, out);
    TDR_LAUNCH_CHECK("crop_augment");
    return TDR_OK;
}

extern "C" int tdr_pad_crop(const float* src, int N, int C, int Hs, int Ws, float* dst, int Hd, int Wd, void* stream) {
    TDR_REQUIRE(src && dst, "tdr_pad_crop: null pointer");
    const long total = (long)N * C * Hd * Wd;
    hipLaunchKernelGGL(pad_crop_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, Hd, Wd, total, dst);
    TDR_LAUNCH_CHECK("pad_crop");
    return TDR_OK;
}

extern "C" int tdr_relu_bwd(const float* go, const float* act, int64_t numel, float* out, void* stream) {
    TDR_REQUIRE(go && act && out, "tdr_relu_bwd: null pointer");
    TDR_REQUIRE(numel % 4 == 0, "tdr_relu_bwd: numel must be a multiple of 4");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid1d(numel / 4)), dim3(256), 0, (hipStream_t)stream, go, act, (long)(numel / 4), out);
    TDR_LAUNCH_CHECK("relu_bwd");
    return TDR_OK;
}

extern "C" int tdr_l1_loss(const float* pred, const float* target, int64_t numel, float loss_weight, float grad_scale,
                           float* loss, float* dpred, float* ws, void* stream) {
    TDR_REQUIRE(pred && target && loss && dpred && ws, "tdr_l1_loss: null pointer (ws needs 2*1024 floats)");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid1d(numel, L1_BLOCKS);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(l1_kernel, dim3(blocks), dim3(256), 0, st, pred, target, (long)numel,
                       loss_weight / (float)numel * grad_scale, (const TdrStepGuard*)nullptr, dpred, part);
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, st, part, blocks, (double)loss_weight / (double)numel, loss);
    TDR_LAUNCH_CHECK("l1_loss");
    return TDR_OK;
}

extern "C" int tdr_pixel_loss(int kind, const float* pred, const float* target, int N, int64_t chw, int64_t hw, float loss_weight,
                              float eps, float grad_scale, const TdrStepGuard* guard, float* loss, float* dpred, float* ws,
                              void* stream) {
    TDR_REQUIRE(pred && target && loss && dpred && ws, "tdr_pixel_loss: null pointer");
    TDR_REQUIRE(kind >= 0 && kind <= 4 && N > 0 && chw > 0 && hw > 0 && chw % hw == 0, "tdr_pixel_loss: bad kind / shape");
    TDR_REQUIRE(kind != 4 || chw == 3 * hw, "tdr_pixel_loss: PSNRLoss toY needs 3 channels");
    hipStream_t st = (hipStream_t)stream;
    const int64_t numel = (int64_t)N * chw;
    double* part = reinterpret_cast<double*>(ws);
    if (kind <= 2) {
        const int blocks = grid1d(numel, L1_BLOCKS);
        // CharbonnierLoss ignores its loss_weight (losses/losses.py:114-122)
        const float w = kind == 2 ? 1.f : loss_weight;
        const float gs = w / (float)numel * grad_scale;
        if (kind == 0) hipLaunchKernelGGL(l1_kernel, dim3(blocks), dim3(256), 0, st, pred, target, (long)numel, gs, guard, dpred, part);
        else if (kind == 1) hipLaunchKernelGGL(pixloss_kernel<1>, dim3(blocks), dim3(256), 0, st, pred, target, (long)numel, gs, 0.f, guard, dpred, part);
        else hipLaunchKernelGGL(pixloss_kernel<2>, dim3(blocks), dim3(256), 0, st, pred, target, (long)numel, gs, eps * eps, guard, dpred, part);
        hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, st, part, blocks, (double)w / (double)numel, loss);
    } else {
        TDR_REQUIRE(N <= 1024, "tdr_pixel_loss: PSNRLoss batch > 1024");
        const int toY = kind == 4;
        const double wscale = (double)loss_weight * (10.0 / log(10.0));
        dim3 grid(PSNR_BLOCKS, N);
        hipLaunchKernelGGL(psnr_sum_kernel, grid, dim3(256), 0, st, pred, target, (long)chw, (long)hw, toY, part);
        hipLaunchKernelGGL(psnr_grad_kernel, grid, dim3(256), 0, st, pred, target, (long)chw, (long)hw, toY, N,
                           (float)wscale * grad_scale, guard, (const double*)part, dpred);
        hipLaunchKernelGGL(psnr_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)part, N, (long)(toY ? hw : chw), wscale, loss);
    }
    TDR_LAUNCH_CHECK("pixel_loss");
    return TDR_OK;
}

extern "C" int tdr_l1_loss_guarded(const float* pred, const float* target, int64_t numel, float loss_weight,
                                   const TdrStepGuard* guard, float* loss, float* dpred, float* ws, void* stream) {
    TDR_REQUIRE(pred && target && loss && dpred && ws && guard, "tdr_l1_loss_guarded: null pointer (ws needs 2*1024 floats)");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid1d(numel, L1_BLOCKS);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(l1_kernel, dim3(blocks), dim3(256), 0, st, pred, target, (long)numel, loss_weight / (float)numel, guard,
                       dpred, part);
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, st, part, blocks, (double)loss_weight / (double)numel, loss);
    TDR_LAUNCH_CHECK("l1_loss_guarded");
    return TDR_OK;
}
