// Stage-A (image-to-text mapping) glue kernels, SURVEY.md 8a rows a29/a30
// (scripts/train/main_train_i2t_mapping.py:40-81 Mapper, :85-98,197-233 injected cross-attention).
// The Linears run on the 1x1 conv kernels over channel-major tokens [B][D][LD] (LD = padded token count, a
// multiple of 32: column 0 class token, 1..T patch tokens, rest padding), nn.LayerNorm on the channel LayerNorm
// kernel; what is left is HBM-bound elementwise / gather work:
//   leaky_relu fwd / bwd            nn.LeakyReLU() (slope 0.01), :55,58,61
//   gather_col                      emb[:, :1]  -> [D][32] "image" holding the B class tokens as pixels (:77)
//   mapper_combine fwd / bwd        mapping_i(cls) + mapping_patch_i(patches).mean(dim=1)  (:77) and its gradient
//   transpose                       token-major [B][T][D] <-> channel-major [B][D][LD] at the attention boundary
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

inline int grid_for(long total, int cap = 8192) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

__global__ void leaky_fwd_kernel(const float* __restrict__ x, long n, float slope, float* __restrict__ y) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v > 0.f ? v : v * slope;
    }
}

// sign(y) == sign(x) for slope > 0, so the activation output is enough
__global__ void leaky_bwd_kernel(const float* __restrict__ go, const float* __restrict__ y, long n, float slope,
                                 float* __restrict__ gx) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        gx[i] = y[i] > 0.f ? go[i] : go[i] * slope;
}

// dst[d][b] = src[b][d][col] for b < B, 0 for B <= b < 32
__global__ void gather_col_kernel(const float* __restrict__ src, int B, int D, int LD, int col, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * 32) return;
    const int d = i >> 5, b = i & 31;
    dst[i] = b < B ? src[((long)b * D + d) * LD + col] : 0.f;
}

// out[b][word][d] = cls[d][b] + mean_{t=1..T} patch[b][d][t];  one wave per (b, d)
__global__ __launch_bounds__(256) void mapper_combine_kernel(const float* __restrict__ cls, const float* __restrict__ patch,
                                                            int B, int D, int LD, int T, int words, int word,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= (long)B * D) return;
    const int b = (int)(row / D), d = (int)(row % D);
    const float* p = patch + row * LD;
    float s = 0.f;
    for (int t = 1 + lane; t <= T; t += 64) s += p[t];
    s = wave_sum(s);
    if (lane == 0) out[((long)b * words + word) * D + d] = cls[d * 32 + b] + s / (float)T;
}

// dcls[d][b] = go[b][word][d] (0 for b >= B);  dpatch[b][d][t] = go[b][word][d] / T for 1 <= t <= T, else 0
__global__ void mapper_combine_bwd_kernel(const float* __restrict__ go, int B, int D, int LD, int T, int words, int word,
                                          float* __restrict__ dcls, float* __restrict__ dpatch) {
    const long total = (long)B * D * LD;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % LD);
        const long row = i / LD;
        const int b = (int)(row / D), d = (int)(row % D);
        const float g = go[((long)b * words + word) * D + d];
        dpatch[i] = (t >= 1 && t <= T) ? g / (float)T : 0.f;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)D * 32; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i >> 5), b = (int)(i & 31);
        dcls[i] = b < B ? go[((long)b * words + word) * D + d] : 0.f;
    }
}

// dst[b][c][r] = src[b][r][c] for r < R, 0 for R <= r < LDd   (src [B][R][C] dense, dst [B][C][LDd]); 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int R, int C, int LDd,
                                                       float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const float* s = src + (long)b * R * C;
    float* d = dst + (long)b * C * LDd;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < C) ? s[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < LDd) d[(long)c * LDd + r] = tile[tx][ty + 8 * k];
    }
}

// ---- stage-A train step glue (scripts/train/main_train_i2t_mapping.py:704-760) -------------------------------------------
// inj_forward_text's embedding injection (:139-151) fused with the position embedding, written channel-major:
//   new[b][p] = tok[ids[b][p]]                    p <  idx_b
//             = inj[b][p - idx_b]                 idx_b <= p < idx_b + L          (the L mapper words replace the placeholder)
//             = tok[ids[b][p - L + 1]]            p >= idx_b + L                  (the rest of the prompt shifted by L - 1)
//   out[b][d][p] = new[b][p][d] + pos[p][d]  for p < S, 0 for S <= p < LD
__global__ void text_inject_fwd_kernel(const int* __restrict__ ids, const float* __restrict__ tok, const float* __restrict__ pos,
                                       const float* __restrict__ inj, const int* __restrict__ idx, int S, int D, int L, int LD,
                                       long total, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % LD);
        const long r = i / LD;
        const int d = (int)(r % D), b = (int)(r / D);
        float v = 0.f;
        if (p < S) {
            const int i0 = idx[b];
            if (p >= i0 && p < i0 + L) v = inj[((long)b * L + (p - i0)) * D + d];
            else v = tok[(long)ids[b * S + (p < i0 ? p : p - L + 1)] * D + d];
            v += pos[(long)p * D + d];
        }
        out[i] = v;
    }
}
// dinj[b][j][d] = dnew[b][d][idx_b + j]  (0 where idx_b + j >= S: words pushed past the end of the prompt are dropped)
__global__ void text_inject_bwd_kernel(const float* __restrict__ dnew, const int* __restrict__ idx, int S, int D, int L, int LD,
                                       long total, float* __restrict__ dinj) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const long r = i / D;
        const int j = (int)(r % L), b = (int)(r / L);
        const int p = idx[b] + j;
        dinj[i] = p < S ? dnew[((long)b * D + d) * LD + p] : 0.f;
    }
}
// DDIMScheduler.add_noise (:717): out = sqrt(ac[t_b]) x + sqrt(1 - ac[t_b]) noise
__global__ void add_noise_kernel(const float* __restrict__ x, const float* __restrict__ noise, const int* __restrict__ t,
                                 const float* __restrict__ ac, long per, long total, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float a = ac[t[i / per]];
        out[i] = sqrtf(a) * x[i] + sqrtf(1.f - a) * noise[i];
    }
}
// the stub UNet's level input: channels 0..C-1 = f x f average pool of x [B][C][H][W], channels C..C+3 = the time features
// sin(2 pi tau), cos(2 pi tau), sin(4 pi tau), cos(4 pi tau), tau = t_b / 1000, constant over the plane.  out [B][C+4][H/f][W/f]
__global__ void pool_time_kernel(const float* __restrict__ x, const int* __restrict__ t, int C, int H, int W, int f, long total,
                                 float* __restrict__ out) {
    const int h = H / f, w = W / f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % w);
        long r = i / w;
        const int oy = (int)(r % h); r /= h;
        const int c = (int)(r % (C + 4)), b = (int)(r / (C + 4));
        float v;
        if (c < C) {
            const float* p = x + (((long)b * C + c) * H + (long)oy * f) * W + (long)ox * f;
            float s = 0.f;
            for (int y = 0; y < f; ++y)
                for (int xx = 0; xx < f; ++xx) s += p[(long)y * W + xx];
            v = s / (float)(f * f);
        } else {
            const float tau = (float)t[b] * 1e-3f, w2 = 6.283185307179586f * (float)(1 + ((c - C) >> 1));
            v = ((c - C) & 1) ? cosf(w2 * tau) : sinf(w2 * tau);
        }
        out[i] = v;
    }
}
// nearest-neighbour upsample by f, accumulated: dst[b][c][y][x] (+)= src[b][c][y/f][x/f]
__global__ void upsample_add_kernel(const float* __restrict__ src, int H, int W, int f, int accumulate, long total, float* __restrict__ dst) {
    const int h = H / f, w = W / f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long r = i / W;
        const int y = (int)(r % H);
        const long plane = r / H;
        const float v = src[(plane * h + y / f) * w + x / f];
        dst[i] = accumulate ? dst[i] + v : v;
    }
}
// its adjoint: dst[b][c][oy][ox] = sum over the f x f block of src
__global__ void pool_sum_kernel(const float* __restrict__ src, int H, int W, int f, long total, float* __restrict__ dst) {
    const int h = H / f, w = W / f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % w);
        const long r = i / w;
        const int oy = (int)(r % h);
        const long plane = r / h;
        const float* p = src + (plane * H + (long)oy * f) * W + (long)ox * f;
        float s = 0.f;
        for (int y = 0; y < f; ++y)
            for (int xx = 0; xx < f; ++xx) s += p[(long)y * W + xx];
        dst[i] = s;
    }
}

// ---- grouped Mapper (the 2 x num_words MLPs of :40-81 as G-way grouped GEMMs + these fused glue kernels) ----------------------
// Every tensor is [G][C][P]: G words, C channels, P pixels (tokens of ALL images of the batch along one axis).  nn.LayerNorm over C
// (eps 1e-5) followed by nn.LeakyReLU, with PER-WORD affine parameters w, b [G][C]:
//   y = lrelu(w_g (z - mu) rstd + b_g);  mu, rstd [G][P] saved
// One workgroup = 64 pixels x 16 channel slices; a thread keeps its CPT = C/16 values of z in registers (one read of z).
template <int CPT>
__global__ __launch_bounds__(1024) void group_ln_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                               const float* __restrict__ b, float eps, float slope, int C, int P,
                                                               float* __restrict__ y, float* __restrict__ mu, float* __restrict__ rstd) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = blockIdx.x * 64 + lane, g = blockIdx.y;
    const bool pok = px < P;
    const unsigned pxc = pok ? px : P - 1;
    const float* zg = z + (long)g * C * P;
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + 16 * i;
        v[i] = c < C ? (zg + (long)c * P)[pxc] : 0.f;
        s += v[i];
    }
    red[slice][lane] = s;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    const float mean = t / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const float d = (slice + 16 * i < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
    red[slice][lane] = q;
    __syncthreads();
    t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    const float rs = 1.0f / sqrtf(t / (float)C + eps);
    if (!pok) return;
    float* yg = y + (long)g * C * P;
    const float* wg = w + (long)g * C;
    const float* bg = b + (long)g * C;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = slice + 16 * i;
        if (c < C) {
            const float zn = (v[i] - mean) * rs * wg[c] + bg[c];
            (yg + (long)c * P)[px] = zn > 0.f ? zn : zn * slope;
        }
    }
    if (slice == 0) {
        mu[(long)g * P + px] = mean;
        rstd[(long)g * P + px] = rs;
    }
}

// backward of the pair: dy (gradient w.r.t. the LeakyReLU output), y (that output: its sign is the sign of the LayerNorm
// output), z, mu, rstd ->  dzn = dy * (y > 0 ? 1 : slope);  g = dzn * w;  dz = rstd (g - yhat mean_c(g yhat) - mean_c(g));
// per-workgroup partials of the per-word parameter gradients  gw = sum_p dzn yhat,  gb = sum_p dzn, and of  gs = sum_p dz  (the
// bias gradient of the Linear that produced z: a pass over dz saved)  -> part[g][tile][3][C].
// Two passes over the workgroup's 64-pixel x C tile (the second one re-reads it from L2 / MALL): caching C/16 = 80 values
// of three tensors per thread does not fit the 128 VGPRs a 1024-thread workgroup has.
__global__ __launch_bounds__(1024) void group_ln_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               const float* __restrict__ z, const float* __restrict__ mu,
                                                               const float* __restrict__ rstd, const float* __restrict__ w,
                                                               float slope, int C, int P, float* __restrict__ dz,
                                                               float* __restrict__ part) {
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = blockIdx.x * 64 + lane, g = blockIdx.y;
    const bool pok = px < P;
    const unsigned pxc = pok ? px : P - 1;
    const long base = (long)g * C * P;
    const float m = mu[(long)g * P + pxc], rs = rstd[(long)g * P + pxc];
    const float* wg = w + (long)g * C;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int c = slice; c < C; c += 16) {
        const long row = base + (long)c * P;
        const float gy = (dy + row)[pxc], yy = (y + row)[pxc], zz = (z + row)[pxc];
        const float gg = (yy > 0.f ? gy : gy * slope) * wg[c];
        s1 += gg;
        s2 += gg * ((zz - m) * rs);
    }
    red[0][slice][lane] = s1; red[1][slice][lane] = s2;
    __syncthreads();
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { S1 += red[0][k][lane]; S2 += red[1][k][lane]; }
    const float mg = S1 / (float)C, mgy = S2 / (float)C;
    float* pb = part + (((long)g * gridDim.x + blockIdx.x) * 3) * C;
#pragma unroll 4
    for (int c = slice; c < C; c += 16) {             // c is wave-uniform
        const long row = base + (long)c * P;
        const float gy = (dy + row)[pxc], yy = (y + row)[pxc], zz = (z + row)[pxc];
        const float dn = pok ? (yy > 0.f ? gy : gy * slope) : 0.f;
        const float yh = (zz - m) * rs;
        const float dzv = pok ? rs * (dn * wg[c] - yh * mgy - mg) : 0.f;
        if (pok) (dz + row)[px] = dzv;
        const float sw = wave_sum(dn * yh), sb = wave_sum(dn), sz = wave_sum(dzv);
        if (lane == 0) { pb[c] = sw; pb[C + c] = sb; pb[2 * C + c] = sz; }
    }
}

// gw[g][c] = sum_tiles part[g][tile][0][c], gb and gs likewise, fixed order
__global__ void group_ln_finish_kernel(const float* __restrict__ part, int tiles, int C, long total, float* __restrict__ gw,
                                       float* __restrict__ gb, float* __restrict__ gs) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long g = i / C;
        const float* p = part + g * tiles * 3 * C;
        float a = 0.f, b = 0.f, z = 0.f;
        for (int t = 0; t < tiles; ++t) { a += p[(long)t * 3 * C + c]; b += p[(long)t * 3 * C + C + c]; z += p[(long)t * 3 * C + 2 * C + c]; }
        gw[i] = a;
        gb[i] = b;
        gs[i] = z;
    }
}

// out[b][g][d] = cls[g][d][b] + mean_{t=1..T} patch[g][d][b*LD + t]    (cls [G][D][32], patch [G][D][B*LD]); one wave per (g, d, b)
__global__ __launch_bounds__(256) void mapper_combine_all_kernel(const float* __restrict__ cls, const float* __restrict__ patch,
                                                                int B, int G, int D, int LD, int T, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);          // (g*D + d)*B + b
    if (row >= (long)G * D * B) return;
    const int b = (int)(row % B);
    const long gd = row / B;
    const float* p = patch + gd * ((long)B * LD) + (long)b * LD;
    float s = 0.f;
    for (int t = 1 + lane; t <= T; t += 64) s += p[t];
    s = wave_sum(s);
    if (lane == 0) {
        const int g = (int)(gd / D), d = (int)(gd % D);
        out[((long)b * G + g) * D + d] = cls[gd * 32 + b] + s / (float)T;
    }
}

// dcls[g][d][b] = go[b][g][d] (0 for b >= B); dpatch[g][d][b*LD + t] = go[b][g][d] / T for 1 <= t <= T, else 0;
// gsum[g][d] = sum_b go[b][g][d] = the pixel sum of dcls AND of dpatch: the bias gradient of both last Linears
__global__ void mapper_combine_all_bwd_kernel(const float* __restrict__ go, int B, int G, int D, int LD, int T, float* __restrict__ dcls,
                                              float* __restrict__ dpatch, float* __restrict__ gsum) {
    const long total = (long)G * D * B * LD;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % LD);
        long r = i / LD;
        const int b = (int)(r % B);
        const long gd = r / B;
        const int g = (int)(gd / D), d = (int)(gd % D);
        dpatch[i] = (t >= 1 && t <= T) ? go[((long)b * G + g) * D + d] / (float)T : 0.f;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)G * D * 32; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i & 31);
        const long gd = i >> 5;
        const int g = (int)(gd / D), d = (int)(gd % D);
        dcls[i] = b < B ? go[((long)b * G + g) * D + d] : 0.f;
        if (b == 0) {
            float t = 0.f;
            for (int k = 0; k < B; ++k) t += go[((long)k * G + g) * D + d];
            gsum[gd] = t;
        }
    }
}

// dst[d][b] = src[b*img_stride + d*ch_stride + col] for b < B, 0 for B <= b < 32  (gather_col for any token layout)
__global__ void gather_col_strided_kernel(const float* __restrict__ src, int B, int D, long img_stride, long ch_stride, int col,
                                          float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * 32) return;
    const int d = i >> 5, b = i & 31;
    dst[i] = b < B ? src[(long)b * img_stride + (long)d * ch_stride + col] : 0.f;
}

// split-K finish of a 1x1 convolution whose K chunks ran as the "images" of one launch (partial sums part[S][C][P]):
//   out[c][p] = act( (sum_s part[s][c][p] + bias[c]) * scale[c] + res[c][p] )      -- the STD epilogue order of tdr_conv_forward
// fixed summation order (s = 0, 1, ...); relu codes as TdrConvDesc.relu (0 none, 1 ReLU, 2 erf-GELU, 3 quick_gelu)
__global__ void splitk_finish_kernel(const float* __restrict__ part, int S, long CP4, int P4, const float* __restrict__ bias,
                                     const float* __restrict__ scale, const float* __restrict__ res, int relu, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < CP4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(part)[i];
        for (int s = 1; s < S; ++s) {
            const float4 q = reinterpret_cast<const float4*>(part)[(long)s * CP4 + i];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const int c = (int)(i / P4);
        if (bias) { const float b = bias[c]; v.x += b; v.y += b; v.z += b; v.w += b; }
        if (scale) { const float g = scale[c]; v.x *= g; v.y *= g; v.z *= g; v.w *= g; }
        if (res) { const float4 r = reinterpret_cast<const float4*>(res)[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (relu == 1) e[k] = fmaxf(e[k], 0.f);
            else if (relu == 2) e[k] = 0.5f * e[k] * (1.f + erff(e[k] * 0.70710678118654752f));
            else if (relu == 3) e[k] = e[k] / (1.f + expf(-1.702f * e[k]));
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

}  // namespace

extern "C" int tdr_leaky_relu_fwd(const float* x, int64_t numel, float slope, float* y, void* stream) {
    TDR_REQUIRE(x && y && slope > 0.f, "tdr_leaky_relu_fwd: bad argument (slope must be > 0)");
    hipLaunchKernelGGL(leaky_fwd_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, x, (long)numel, slope, y);
    TDR_LAUNCH_CHECK("leaky_relu_fwd");
    return TDR_OK;
}

extern "C" int tdr_leaky_relu_bwd(const float* go, const float* y, int64_t numel, float slope, float* gx, void* stream) {
    TDR_REQUIRE(go && y && gx && slope > 0.f, "tdr_leaky_relu_bwd: bad argument (slope must be > 0)");
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, go, y, (long)numel, slope, gx);
    TDR_LAUNCH_CHECK("leaky_relu_bwd");
    return TDR_OK;
}

extern "C" int tdr_gather_col(const float* src, int B, int D, int LD, int col, float* dst, void* stream) {
    TDR_REQUIRE(src && dst && B > 0 && B <= 32 && col >= 0 && col < LD, "tdr_gather_col: need 1 <= B <= 32 and 0 <= col < LD");
    hipLaunchKernelGGL(gather_col_kernel, dim3(tdr_cdiv((long)D * 32, 256)), dim3(256), 0, (hipStream_t)stream, src, B, D, LD, col, dst);
    TDR_LAUNCH_CHECK("gather_col");
    return TDR_OK;
}

extern "C" int tdr_mapper_combine(const float* cls, const float* patch, int B, int D, int LD, int T, int words, int word,
                                  float* out, void* stream) {
    TDR_REQUIRE(cls && patch && out && B > 0 && B <= 32 && T >= 1 && T < LD && word >= 0 && word < words, "tdr_mapper_combine: bad argument");
    hipLaunchKernelGGL(mapper_combine_kernel, dim3(tdr_cdiv((long)B * D, 4)), dim3(256), 0, (hipStream_t)stream, cls, patch, B, D,
                       LD, T, words, word, out);
    TDR_LAUNCH_CHECK("mapper_combine");
    return TDR_OK;
}

extern "C" int tdr_mapper_combine_bwd(const float* go, int B, int D, int LD, int T, int words, int word, float* dcls,
                                      float* dpatch, void* stream) {
    TDR_REQUIRE(go && dcls && dpatch && B > 0 && B <= 32 && T >= 1 && T < LD && word >= 0 && word < words, "tdr_mapper_combine_bwd: bad argument");
    hipLaunchKernelGGL(mapper_combine_bwd_kernel, dim3(grid_for((long)B * D * LD)), dim3(256), 0, (hipStream_t)stream, go, B, D, LD,
                       T, words, word, dcls, dpatch);
    TDR_LAUNCH_CHECK("mapper_combine_bwd");
    return TDR_OK;
}

extern "C" int tdr_transpose_pad(const float* src, int B, int R, int C, int LDd, float* dst, void* stream) {
    TDR_REQUIRE(src && dst && LDd >= R, "tdr_transpose_pad: bad argument");
    hipLaunchKernelGGL(transpose_kernel, dim3(tdr_cdiv(C, 32), tdr_cdiv(LDd, 32), B), dim3(256), 0, (hipStream_t)stream, src, R, C,
                       LDd, dst);
    TDR_LAUNCH_CHECK("transpose_pad");
    return TDR_OK;
}

extern "C" int tdr_text_inject_fwd(const int* ids, const float* tok_emb, const float* pos_emb, const float* inj, const int* idx,
                                   int B, int S, int D, int L, int LD, float* out, void* stream) {
    TDR_REQUIRE(ids && tok_emb && pos_emb && inj && idx && out && B > 0 && S > 0 && L >= 1 && LD >= S, "tdr_text_inject_fwd: bad argument");
    const long total = (long)B * D * LD;
    hipLaunchKernelGGL(text_inject_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ids, tok_emb, pos_emb, inj, idx,
                       S, D, L, LD, total, out);
    TDR_LAUNCH_CHECK("text_inject_fwd");
    return TDR_OK;
}

extern "C" int tdr_text_inject_bwd(const float* dnew, const int* idx, int B, int S, int D, int L, int LD, float* dinj, void* stream) {
    TDR_REQUIRE(dnew && idx && dinj && B > 0 && S > 0 && L >= 1 && LD >= S, "tdr_text_inject_bwd: bad argument");
    const long total = (long)B * L * D;
    hipLaunchKernelGGL(text_inject_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dnew, idx, S, D, L, LD, total, dinj);
    TDR_LAUNCH_CHECK("text_inject_bwd");
    return TDR_OK;
}

extern "C" int tdr_add_noise(const float* x, const float* noise, const int* t, const float* alphas_cumprod, int B, int64_t per,
                             float* out, void* stream) {
    TDR_REQUIRE(x && noise && t && alphas_cumprod && out && B > 0 && per > 0, "tdr_add_noise: bad argument");
    const long total = (long)B * per;
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, noise, t, alphas_cumprod, (long)per,
                       total, out);
    TDR_LAUNCH_CHECK("add_noise");
    return TDR_OK;
}

extern "C" int tdr_pool_time(const float* x, const int* t, int B, int C, int H, int W, int f, float* out, void* stream) {
    TDR_REQUIRE(x && t && out && f >= 1 && H % f == 0 && W % f == 0, "tdr_pool_time: bad argument");
    const long total = (long)B * (C + 4) * (H / f) * (W / f);
    hipLaunchKernelGGL(pool_time_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, t, C, H, W, f, total, out);
    TDR_LAUNCH_CHECK("pool_time");
    return TDR_OK;
}

extern "C" int tdr_upsample_nearest_add(const float* src, int planes, int H, int W, int f, int accumulate, float* dst, void* stream) {
    TDR_REQUIRE(src && dst && f >= 1 && H % f == 0 && W % f == 0, "tdr_upsample_nearest_add: bad argument");
    const long total = (long)planes * H * W;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, f, accumulate, total, dst);
    TDR_LAUNCH_CHECK("upsample_nearest_add");
    return TDR_OK;
}

extern "C" int tdr_pool_sum(const float* src, int planes, int H, int W, int f, float* dst, void* stream) {
    TDR_REQUIRE(src && dst && f >= 1 && H % f == 0 && W % f == 0, "tdr_pool_sum: bad argument");
    const long total = (long)planes * (H / f) * (W / f);
    hipLaunchKernelGGL(pool_sum_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, H, W, f, total, dst);
    TDR_LAUNCH_CHECK("pool_sum");
    return TDR_OK;
}

extern "C" int tdr_group_ln_act_fwd(const float* z, const float* w, const float* b, float eps, float slope, int G, int C, int P,
                                    float* y, float* mu, float* rstd, void* stream) {
    TDR_REQUIRE(z && w && b && y && mu && rstd && G > 0 && C > 0 && P > 0 && slope > 0.f, "tdr_group_ln_act_fwd: bad argument");
    TDR_REQUIRE(C <= 16 * 96, "tdr_group_ln_act_fwd: C = %d > 1536 not supported", C);
    dim3 grid(tdr_cdiv(P, 64), G);
    hipStream_t st = (hipStream_t)stream;
#define GLN_FWD(K) hipLaunchKernelGGL(group_ln_act_fwd_kernel<K>, grid, dim3(1024), 0, st, z, w, b, eps, slope, C, P, y, mu, rstd)
    if (C <= 16 * 16) GLN_FWD(16);
    else if (C <= 16 * 64) GLN_FWD(64);
    else if (C <= 16 * 80) GLN_FWD(80);
    else GLN_FWD(96);
#undef GLN_FWD
    TDR_LAUNCH_CHECK("group_ln_act_fwd");
    return TDR_OK;
}

extern "C" int64_t tdr_group_ln_ws_floats(int G, int C, int P) { return (int64_t)G * tdr_cdiv(P, 64) * 3 * C; }

extern "C" int tdr_group_ln_act_bwd(const float* dy, const float* y, const float* z, const float* mu, const float* rstd,
                                    const float* w, float slope, int G, int C, int P, float* dz, float* gw, float* gb, float* gs,
                                    float* ws, void* stream) {
    TDR_REQUIRE(dy && y && z && mu && rstd && w && dz && gw && gb && gs && ws && G > 0 && slope > 0.f, "tdr_group_ln_act_bwd: bad argument");
    const int tiles = tdr_cdiv(P, 64);
    dim3 grid(tiles, G);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(group_ln_act_bwd_kernel, grid, dim3(1024), 0, st, dy, y, z, mu, rstd, w, slope, C, P, dz, ws);
    const long total = (long)G * C;
    hipLaunchKernelGGL(group_ln_finish_kernel, dim3(grid_for(total)), dim3(256), 0, st, ws, tiles, C, total, gw, gb, gs);
    TDR_LAUNCH_CHECK("group_ln_act_bwd");
    return TDR_OK;
}

extern "C" int tdr_mapper_combine_all(const float* cls, const float* patch, int B, int G, int D, int LD, int T, float* out, void* stream) {
    TDR_REQUIRE(cls && patch && out && B > 0 && B <= 32 && T >= 1 && T < LD, "tdr_mapper_combine_all: bad argument");
    hipLaunchKernelGGL(mapper_combine_all_kernel, dim3(tdr_cdiv((long)G * D * B, 4)), dim3(256), 0, (hipStream_t)stream, cls, patch, B, G,
                       D, LD, T, out);
    TDR_LAUNCH_CHECK("mapper_combine_all");
    return TDR_OK;
}

extern "C" int tdr_mapper_combine_all_bwd(const float* go, int B, int G, int D, int LD, int T, float* dcls, float* dpatch, float* gsum,
                                          void* stream) {
    TDR_REQUIRE(go && dcls && dpatch && gsum && B > 0 && B <= 32 && T >= 1 && T < LD, "tdr_mapper_combine_all_bwd: bad argument");
    hipLaunchKernelGGL(mapper_combine_all_bwd_kernel, dim3(grid_for((long)G * D * B * LD)), dim3(256), 0, (hipStream_t)stream, go, B, G, D,
                       LD, T, dcls, dpatch, gsum);
    TDR_LAUNCH_CHECK("mapper_combine_all_bwd");
    return TDR_OK;
}

extern "C" int tdr_gather_col_strided(const float* src, int B, int D, int64_t img_stride, int64_t ch_stride, int col, float* dst,
                                      void* stream) {
    TDR_REQUIRE(src && dst && B > 0 && B <= 32 && col >= 0, "tdr_gather_col_strided: need 1 <= B <= 32");
    hipLaunchKernelGGL(gather_col_strided_kernel, dim3(tdr_cdiv((long)D * 32, 256)), dim3(256), 0, (hipStream_t)stream, src, B, D,
                       (long)img_stride, (long)ch_stride, col, dst);
    TDR_LAUNCH_CHECK("gather_col_strided");
    return TDR_OK;
}

extern "C" int tdr_splitk_finish(const float* part, int S, int C, int64_t P, const float* bias, const float* scale, const float* res,
                                 int relu, float* out, void* stream) {
    TDR_REQUIRE(part && out && S >= 1 && C > 0 && P > 0 && P % 4 == 0, "tdr_splitk_finish: bad argument (P must be a multiple of 4)");
    const long CP4 = (long)C * P / 4;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(grid_for(CP4, 4096)), dim3(256), 0, (hipStream_t)stream, part, S, CP4, (int)(P / 4), bias,
                       scale, res, relu, out);
    TDR_LAUNCH_CHECK("splitk_finish");
    return TDR_OK;
}
