// PromptIR-ref's PromptGenBlock (network_promptir_guided_arch.py:417-441) around the 3x3 convolution:
//   emb = mean_hw x;  w = softmax(Linear(emb));  prompt = sum_k w_k * prompt_param_k;  bilinear resize to (H, W);  conv3x3
// The weighted sum and the resize commute (both linear), so the resize runs once per step on the L parameter planes
// instead of once per image.  Everything here is small ([N, L] weights, L * prompt_dim planes) and HBM-trivial; all
// reductions are fixed-order.
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

// one workgroup per (channel plane, image): out[n][c] = scale * sum_hw x
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ x, long x_ns, int HW, float scale,
                                                        float* __restrict__ out, int C) {
    __shared__ float red[4];
    const int c = blockIdx.x, n = blockIdx.y;
    const float* p = x + (long)n * x_ns + (long)c * HW;
    float s0 = 0.f, s1 = 0.f;
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const f32x4 v = p4[i];
            s0 += v[0] + v[1];
            s1 += v[2] + v[3];
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) s0 += p[i];
    }
    const float s = wave_sum(s0 + s1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(long)n * C + c] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// x[n][c][:] += v[n][c] * scale
__global__ void plane_add_kernel(float* __restrict__ x, long x_ns, const float* __restrict__ v, float scale, int C, int HW,
                                 long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pl = i / HW;
        const int n = (int)(pl / C), c = (int)(pl % C);
        x[(long)n * x_ns + (long)c * HW + (i - pl * HW)] += v[pl] * scale;
    }
}

// one wave per image: logits_k = b_k + sum_c W[k][c] emb[n][c], softmax over k (L <= 64)
__global__ __launch_bounds__(64) void prompt_weights_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ W,
                                                               const float* __restrict__ b, int C, int L,
                                                               float* __restrict__ w) {
    const int n = blockIdx.x, lane = threadIdx.x;
    __shared__ float lg[64];
    for (int k = 0; k < L; ++k) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += W[(long)k * C + c] * emb[(long)n * C + c];
        s = wave_sum(s);
        if (lane == 0) lg[k] = s + (b ? b[k] : 0.f);
    }
    __syncthreads();
    if (lane == 0) {
        float m = -INFINITY;
        for (int k = 0; k < L; ++k) m = fmaxf(m, lg[k]);
        float z = 0.f;
        for (int k = 0; k < L; ++k) { lg[k] = expf(lg[k] - m); z += lg[k]; }
        for (int k = 0; k < L; ++k) w[(long)n * L + k] = lg[k] / z;
    }
}

// single workgroup: softmax backward, then dW[k][c] = sum_n dl[n][k] emb[n][c], db[k] = sum_n dl[n][k],
// demb[n][c] = sum_k dl[n][k] W[k][c]
__global__ __launch_bounds__(256) void prompt_weights_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ W,
                                                                const float* __restrict__ w, const float* __restrict__ dw,
                                                                int N, int C, int L, float* __restrict__ dW,
                                                                float* __restrict__ db, float* __restrict__ demb) {
    extern __shared__ float dl[];          // [N][L]
    for (int n = threadIdx.x; n < N; n += 256) {
        float dot = 0.f;
        for (int k = 0; k < L; ++k) dot += w[(long)n * L + k] * dw[(long)n * L + k];
        for (int k = 0; k < L; ++k) dl[n * L + k] = w[(long)n * L + k] * (dw[(long)n * L + k] - dot);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * C; i += 256) {
        const int k = i / C, c = i - k * C;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dl[n * L + k] * emb[(long)n * C + c];
        dW[i] = s;
    }
    for (int k = threadIdx.x; k < L; k += 256) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dl[n * L + k];
        if (db) db[k] = s;
    }
    for (int i = threadIdx.x; i < N * C; i += 256) {
        const int n = i / C, c = i - n * C;
        float s = 0.f;
        for (int k = 0; k < L; ++k) s += dl[n * L + k] * W[(long)k * C + c];
        demb[i] = s;
    }
}

// out[n][e] = sum_k w[n][k] P[k][e]
__global__ void prompt_mix_fwd_kernel(const float* __restrict__ w, const float* __restrict__ P, int N, int L, long E,
                                      float* __restrict__ out) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        float pk[16];
        for (int k = 0; k < L; ++k) pk[k] = P[(long)k * E + e];
        for (int n = 0; n < N; ++n) {
            float s = 0.f;
            for (int k = 0; k < L; ++k) s += w[(long)n * L + k] * pk[k];
            out[(long)n * E + e] = s;
        }
    }
}

// dP[k][e] = sum_n w[n][k] d[n][e]
__global__ void prompt_mix_bwd_p_kernel(const float* __restrict__ w, const float* __restrict__ d, int N, int L, long E,
                                        float* __restrict__ dP) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        float acc[16];
        for (int k = 0; k < L; ++k) acc[k] = 0.f;
        for (int n = 0; n < N; ++n) {
            const float v = d[(long)n * E + e];
            for (int k = 0; k < L; ++k) acc[k] += w[(long)n * L + k] * v;
        }
        for (int k = 0; k < L; ++k) dP[(long)k * E + e] = acc[k];
    }
}

constexpr int MIXW_BLOCKS = 64;
// part[n][k][b] = sum over block b's slice of d[n][e] P[k][e]        grid (MIXW_BLOCKS, L, N)
__global__ __launch_bounds__(256) void prompt_mix_bwd_w_kernel(const float* __restrict__ d, const float* __restrict__ P, int L,
                                                              long E, float* __restrict__ part) {
    __shared__ float red[4];
    const int b = blockIdx.x, k = blockIdx.y, n = blockIdx.z;
    const long per = (E + MIXW_BLOCKS - 1) / MIXW_BLOCKS, e0 = b * per, e1 = e0 + per < E ? e0 + per : E;
    float s = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) s += d[(long)n * E + e] * P[(long)k * E + e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[((long)n * L + k) * MIXW_BLOCKS + b] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void prompt_mix_bwd_w_finish_kernel(const float* __restrict__ part, int total, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float s = 0.f;
    for (int b = 0; b < MIXW_BLOCKS; ++b) s += part[(long)i * MIXW_BLOCKS + b];
    dw[i] = s;
}

// adjoint of resize_bilinear_kernel (tdr_vit.hip) as a gather: source pixel (ys, xs) collects from every destination
// pixel whose two taps per axis include it -- fixed order, no atomics
__device__ __forceinline__ void bil_tap(int d, float scale, int S, int& i0, int& i1, float& l) {
    const float f = fmaxf(((float)d + 0.5f) * scale - 0.5f, 0.f);
    i0 = min((int)f, S - 1);
    i1 = min(i0 + 1, S - 1);
    l = f - (float)i0;
}
__global__ void resize_bilinear_bwd_kernel(const float* __restrict__ dd, int Hs, int Ws, int Hd, int Wd, long planes,
                                           float* __restrict__ ds) {
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    const long total = planes * Hs * Ws;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xs = (int)(i % Ws);
        const int ys = (int)((i / Ws) % Hs);
        const long pl = i / ((long)Ws * Hs);
        // destination rows whose source coordinate lies in (ys - 1, ys + 1): a conservative integer window
        const int y_lo = max(0, (int)floorf(((float)ys - 1.f + 0.5f) / sy - 0.5f) - 1);
        const int y_hi = min(Hd - 1, (int)ceilf(((float)ys + 1.f + 0.5f) / sy - 0.5f) + 1);
        const int x_lo = max(0, (int)floorf(((float)xs - 1.f + 0.5f) / sx - 0.5f) - 1);
        const int x_hi = min(Wd - 1, (int)ceilf(((float)xs + 1.f + 0.5f) / sx - 0.5f) + 1);
        const float* p = dd + pl * Hd * Wd;
        float acc = 0.f;
        for (int y = y_lo; y <= y_hi; ++y) {
            int y0, y1; float ly;
            bil_tap(y, sy, Hs, y0, y1, ly);
            const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int x = x_lo; x <= x_hi; ++x) {
                int x0, x1; float lx;
                bil_tap(x, sx, Ws, x0, x1, lx);
                const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
                if (wx != 0.f) acc += wy * wx * p[(long)y * Wd + x];
            }
        }
        ds[i] = acc;
    }
}

inline int pgrid(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

extern "C" int tdr_plane_mean(const float* x, int64_t x_ns, int N, int C, int HW, float* out, void* stream) {
    TDR_REQUIRE(x && out && N > 0 && C > 0 && HW > 0, "tdr_plane_mean: bad argument");
    hipLaunchKernelGGL(plane_mean_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, HW, 1.0f / (float)HW, out, C);
    TDR_LAUNCH_CHECK("plane_mean");
    return TDR_OK;
}

extern "C" int tdr_plane_add(float* x, int64_t x_ns, const float* v, float scale, int N, int C, int HW, void* stream) {
    TDR_REQUIRE(x && v && N > 0 && C > 0 && HW > 0, "tdr_plane_add: bad argument");
    const long total = (long)N * C * HW;
    hipLaunchKernelGGL(plane_add_kernel, dim3(pgrid(total)), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, v, scale, C, HW, total);
    TDR_LAUNCH_CHECK("plane_add");
    return TDR_OK;
}

extern "C" int tdr_prompt_weights_fwd(const float* emb, const float* W, const float* b, int N, int C, int L, float* w, void* stream) {
    TDR_REQUIRE(emb && W && w && N > 0 && C > 0 && L > 0 && L <= 16, "tdr_prompt_weights_fwd: bad argument (prompt_len <= 16)");
    hipLaunchKernelGGL(prompt_weights_fwd_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, emb, W, b, C, L, w);
    TDR_LAUNCH_CHECK("prompt_weights_fwd");
    return TDR_OK;
}

extern "C" int tdr_prompt_weights_bwd(const float* emb, const float* W, const float* w, const float* dw, int N, int C, int L,
                                      float* dW, float* db, float* demb, void* stream) {
    TDR_REQUIRE(emb && W && w && dw && dW && demb && N > 0 && C > 0 && L > 0 && L <= 16 && (long)N * L <= 8192,
                "tdr_prompt_weights_bwd: bad argument");
    hipLaunchKernelGGL(prompt_weights_bwd_kernel, dim3(1), dim3(256), (size_t)N * L * sizeof(float), (hipStream_t)stream, emb, W, w,
                       dw, N, C, L, dW, db, demb);
    TDR_LAUNCH_CHECK("prompt_weights_bwd");
    return TDR_OK;
}

extern "C" int tdr_prompt_mix_fwd(const float* w, const float* P, int N, int L, int64_t E, float* out, void* stream) {
    TDR_REQUIRE(w && P && out && N > 0 && L > 0 && L <= 16 && E > 0, "tdr_prompt_mix_fwd: bad argument (prompt_len <= 16)");
    hipLaunchKernelGGL(prompt_mix_fwd_kernel, dim3(pgrid(E)), dim3(256), 0, (hipStream_t)stream, w, P, N, L, (long)E, out);
    TDR_LAUNCH_CHECK("prompt_mix_fwd");
    return TDR_OK;
}

extern "C" int64_t tdr_prompt_mix_bwd_ws_floats(int N, int L) { return (int64_t)N * L * MIXW_BLOCKS; }

extern "C" int tdr_prompt_mix_bwd(const float* w, const float* P, const float* d, int N, int L, int64_t E, float* dP, float* dw,
                                  float* ws, void* stream) {
    TDR_REQUIRE(w && P && d && dP && dw && ws && N > 0 && L > 0 && L <= 16 && E > 0, "tdr_prompt_mix_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prompt_mix_bwd_p_kernel, dim3(pgrid(E)), dim3(256), 0, st, w, d, N, L, (long)E, dP);
    hipLaunchKernelGGL(prompt_mix_bwd_w_kernel, dim3(MIXW_BLOCKS, L, N), dim3(256), 0, st, d, P, L, (long)E, ws);
    hipLaunchKernelGGL(prompt_mix_bwd_w_finish_kernel, dim3(tdr_cdiv(N * L, 64)), dim3(64), 0, st, ws, N * L, dw);
    TDR_LAUNCH_CHECK("prompt_mix_bwd");
    return TDR_OK;
}

extern "C" int tdr_resize_bilinear_bwd(const float* ddst, int planes, int Hs, int Ws, int Hd, int Wd, float* dsrc, void* stream) {
    TDR_REQUIRE(ddst && dsrc && planes > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "tdr_resize_bilinear_bwd: bad argument");
    const long total = (long)planes * Hs * Ws;
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(pgrid(total)), dim3(256), 0, (hipStream_t)stream, ddst, Hs, Ws, Hd, Wd,
                       (long)planes, dsrc);
    TDR_LAUNCH_CHECK("resize_bilinear_bwd");
    return TDR_OK;
}
