// Restormer-ref pieces that are not plain convolutions (network_restormer_guided_arch.py):
//   MDTA core (:246-277): attn = softmax_j( t_h * q^_i . k^_j ),  out = attn v, per image and head, where the
//   "tokens" are CHANNELS (c = C/heads <= 192 per head) and the contraction runs over all H*W pixels.
//   The two big contractions are convolution-shaped and run on the MFMA kernels of this library:
//     G = q k^T over pixels        -> tdr_conv_wgrad(per_image)  (a C x C Gram matrix per image)
//     out = attn v                 -> tdr_conv_forward, 1x1, per-image weights (wp_ns)
//   What is left is O(C*c) work per image on the c x c matrices, done here:
//     tdr_row_sumsq        |q_i|^2, |k_j|^2 (F.normalize denominators, :266-267)
//     tdr_mdta_softmax     logits from G, the norms and the temperature; softmax; emits the probabilities already in
//                          the packed layout the 1x1 conv kernel reads ([cin][Mpad], block-diagonal over heads), once
//                          transposed (for attn v) and once plain (for attn^T dout in the backward)
//     tdr_mdta_bwd         softmax / temperature / normalisation backward on the c x c blocks; emits ONE symmetric
//                          packed weight matrix W [2C x 2C] per image so that  d[q;k] = W [q;k]  is a single 1x1 conv:
//                            W[i][C+j] = W[C+j][i] = dG^_ij / (|q_i||k_j|),  W[i][i] = -rho_i/|q_i|^2,
//                            W[C+j][C+j] = -rho'_j/|k_j|^2,  rho_i = sum_j dG^_ij G^_ij,  rho'_j = sum_i dG^_ij G^_ij
//   plus the small glue of TransformerResFusionBlock (x*alpha + shortcut, :353) and PixelShuffle (:391).
// All reductions are fixed-order (deterministic).
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

constexpr float NORM_EPS = 1e-12f;      // F.normalize default eps

__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ x, long x_ns, int HW,
                                                       float* __restrict__ out, int rows) {
    __shared__ float red[4];
    const int r = blockIdx.x, n = blockIdx.y;
    const float* p = x + (long)n * x_ns + (long)r * HW;
    float s0 = 0.f, s1 = 0.f;
    if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const f32x4 v = p4[i];
            s0 += v[0] * v[0] + v[1] * v[1];
            s1 += v[2] * v[2] + v[3] * v[3];
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) s0 += p[i] * p[i];
    }
    const float s = wave_sum(s0 + s1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(long)n * rows + r] = (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (heads, N); wave per row i of the head's c x c block; lane handles columns j = lane + 64 u, u < NU (c <= 64 NU)
template <int NU>
__global__ __launch_bounds__(256) void mdta_softmax_kernel(const float* __restrict__ G, const float* __restrict__ ss,
                                                          const float* __restrict__ temp, int C, int c, int Cp,
                                                          float* __restrict__ A, float* __restrict__ AT) {
    const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float t = temp[h];
    const float* Gn = G + (long)n * C * C;
    const float* sq = ss + (long)n * 2 * C;
    float* An = A + (long)n * Cp * Cp;
    float* ATn = AT + (long)n * Cp * Cp;
    float ink[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        ink[u] = j < c ? 1.0f / fmaxf(sqrtf(sq[C + h * c + j]), NORM_EPS) : 0.f;
    }
    for (int i = wv; i < c; i += 4) {
        const int gi = h * c + i;
        const float inq = 1.0f / fmaxf(sqrtf(sq[gi]), NORM_EPS);
        float L[NU];
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            L[u] = j < c ? t * (Gn[(long)gi * C + h * c + j] * inq * ink[u]) : -INFINITY;
            m = fmaxf(m, L[u]);
        }
        m = wave_max(m);
        float e[NU], s = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            e[u] = (lane + 64 * u) < c ? expf(L[u] - m) : 0.f;
            s += e[u];
        }
        s = wave_sum(s);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            if (j < c) {
                const float p = e[u] / s;
                An[(long)gi * Cp + h * c + j] = p;
                ATn[(long)(h * c + j) * Cp + gi] = p;
            }
        }
    }
}

// grid (heads, N), dynamic LDS: E [c][c+1] + rowt[c]
template <int NU>
__global__ __launch_bounds__(256) void mdta_bwd_kernel(const float* __restrict__ G, const float* __restrict__ ss,
                                                      const float* __restrict__ temp, const float* __restrict__ A,
                                                      const float* __restrict__ dA, int C, int c, int Cp, int Wp,
                                                      float* __restrict__ W, float* __restrict__ dt_part) {
    extern __shared__ float lds[];
    float* E = lds;                         // [c][c+1]
    float* rowt = lds + c * (c + 1);        // [c]
    const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int heads = gridDim.x;
    const float t = temp[h];
    const float* Gn = G + (long)n * C * C;
    const float* dAn = dA + (long)n * C * C;
    const float* An = A + (long)n * Cp * Cp;
    const float* sq = ss + (long)n * 2 * C;
    float* Wn = W + (long)n * Wp * Wp;
    float ink[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        ink[u] = j < c ? 1.0f / fmaxf(sqrtf(sq[C + h * c + j]), NORM_EPS) : 0.f;
    }
    for (int i = wv; i < c; i += 4) {
        const int gi = h * c + i;
        const float nq = sqrtf(sq[gi]);
        const float inq = 1.0f / fmaxf(nq, NORM_EPS);
        float P[NU], dP[NU], Gh[NU], pd = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            const bool ok = j < c;
            P[u] = ok ? An[(long)gi * Cp + h * c + j] : 0.f;
            dP[u] = ok ? dAn[(long)gi * C + h * c + j] : 0.f;
            Gh[u] = ok ? Gn[(long)gi * C + h * c + j] * inq * ink[u] : 0.f;
            pd += P[u] * dP[u];
        }
        pd = wave_sum(pd);
        float rt = 0.f, rho = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            const float dL = P[u] * (dP[u] - pd);
            rt += dL * Gh[u];
            const float dGh = t * dL;
            const float e = dGh * Gh[u];
            rho += e;
            if (j < c) {
                E[i * (c + 1) + j] = e;
                const float w = dGh * inq * ink[u];
                Wn[(long)gi * Wp + C + h * c + j] = w;
                Wn[(long)(C + h * c + j) * Wp + gi] = w;
            }
        }
        rt = wave_sum(rt);
        rho = wave_sum(rho);
        if (lane == 0) {
            rowt[i] = rt;
            Wn[(long)gi * Wp + gi] = nq > NORM_EPS ? -rho * inq * inq : 0.f;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < c; j += 256) {
        float s = 0.f;
        for (int i = 0; i < c; ++i) s += E[i * (c + 1) + j];
        const float nk = sqrtf(sq[C + h * c + j]);
        const float inkj = 1.0f / fmaxf(nk, NORM_EPS);
        const long d = C + h * c + j;
        Wn[d * Wp + d] = nk > NORM_EPS ? -s * inkj * inkj : 0.f;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < c; ++i) s += rowt[i];
        dt_part[(long)n * heads + h] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// DRSformer-ref's Top-K Sparse Attention (network_drsformer_guided_arch*.py:260-331): the same channel-token attention, but
//   out = sum_m a_m * softmax(mask_m(L)) v,   mask_m keeps the k_m largest logits of a row (k = c/2, 2c/3, 3c/4, 4c/5)
// All four branches multiply the same v, so they collapse into ONE c x c matrix A = sum_m a_m P_m and the rest of the
// MDTA machinery (attn v and the q / k gradients as per-image 1x1 convs) is unchanged.  Top-k by rank: entry j is kept
// iff fewer than k entries of its row are larger (ties: lower index first).
// grid (heads, N); wave per row; lane holds columns lane + 64 u.  LDS: 4 rows of logits.
// ---------------------------------------------------------------------------------------------------------------
struct TksaK { int k[4]; };

template <int NU>
__device__ __forceinline__ void tksa_row(const float* __restrict__ Grow, const float* __restrict__ sq, int C, int c, int h, float t,
                                         float inq, const float (&ink)[NU], float* __restrict__ lrow, const TksaK& kk, int lane,
                                         float (&Gh)[NU], float (&Pm)[4][NU]) {
    float L[NU], m = -INFINITY;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        Gh[u] = j < c ? Grow[h * c + j] * inq * ink[u] : 0.f;
        L[u] = j < c ? t * Gh[u] : -INFINITY;
        if (j < c) lrow[j] = L[u];
        m = fmaxf(m, L[u]);
    }
    m = wave_max(m);
    __builtin_amdgcn_wave_barrier();
    int rank[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) rank[u] = 0;
    for (int jj = 0; jj < c; ++jj) {
        const float v = lrow[jj];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            rank[u] += (v > L[u] || (v == L[u] && jj < j)) ? 1 : 0;
        }
    }
    float e[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) e[u] = (lane + 64 * u) < c ? expf(L[u] - m) : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) s += rank[u] < kk.k[q] ? e[u] : 0.f;
        s = wave_sum(s);
#pragma unroll
        for (int u = 0; u < NU; ++u) Pm[q][u] = rank[u] < kk.k[q] ? e[u] / s : 0.f;
    }
    (void)sq; (void)C;
}

template <int NU>
__global__ __launch_bounds__(256) void tksa_softmax_kernel(const float* __restrict__ G, const float* __restrict__ ss,
                                                          const float* __restrict__ temp, const float* __restrict__ am, TksaK kk,
                                                          int C, int c, int Cp, float* __restrict__ A, float* __restrict__ AT) {
    __shared__ float lrow[4][64 * NU];
    const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float t = temp[h];
    const float a0 = am[0], a1 = am[1], a2 = am[2], a3 = am[3];
    const float* Gn = G + (long)n * C * C;
    const float* sq = ss + (long)n * 2 * C;
    float* An = A + (long)n * Cp * Cp;
    float* ATn = AT + (long)n * Cp * Cp;
    float ink[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        ink[u] = j < c ? 1.0f / fmaxf(sqrtf(sq[C + h * c + j]), NORM_EPS) : 0.f;
    }
    for (int i = wv; i < c; i += 4) {
        const int gi = h * c + i;
        const float inq = 1.0f / fmaxf(sqrtf(sq[gi]), NORM_EPS);
        float Gh[NU], Pm[4][NU];
        tksa_row<NU>(Gn + (long)gi * C, sq, C, c, h, t, inq, ink, lrow[wv], kk, lane, Gh, Pm);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            if (j < c) {
                const float p = ((a0 * Pm[0][u] + a1 * Pm[1][u]) + a2 * Pm[2][u]) + a3 * Pm[3][u];
                An[(long)gi * Cp + h * c + j] = p;
                ATn[(long)(h * c + j) * Cp + gi] = p;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// backward: dL = sum_m a_m P_m (dA - <P_m, dA>);  da_m partial = <P_m, dA>;  then exactly mdta_bwd_kernel's tail.
// dynamic LDS: E [c][c+1] + rowt[c] + rowa[c][4] + lrow[4][64 NU]
template <int NU>
__global__ __launch_bounds__(256) void tksa_bwd_kernel(const float* __restrict__ G, const float* __restrict__ ss,
                                                      const float* __restrict__ temp, const float* __restrict__ am, TksaK kk,
                                                      const float* __restrict__ dA, int C, int c, int Wp, float* __restrict__ W,
                                                      float* __restrict__ dt_part, float* __restrict__ da_part) {
    extern __shared__ float lds[];
    float* E = lds;                         // [c][c+1]
    float* rowt = lds + c * (c + 1);        // [c]
    float* rowa = rowt + c;                 // [c][4]
    float* lrow = rowa + 4 * c;             // [4][64 NU]
    const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int heads = gridDim.x;
    const float t = temp[h];
    const float a[4] = {am[0], am[1], am[2], am[3]};
    const float* Gn = G + (long)n * C * C;
    const float* dAn = dA + (long)n * C * C;
    const float* sq = ss + (long)n * 2 * C;
    float* Wn = W + (long)n * Wp * Wp;
    float ink[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int j = lane + 64 * u;
        ink[u] = j < c ? 1.0f / fmaxf(sqrtf(sq[C + h * c + j]), NORM_EPS) : 0.f;
    }
    for (int i = wv; i < c; i += 4) {
        const int gi = h * c + i;
        const float nq = sqrtf(sq[gi]);
        const float inq = 1.0f / fmaxf(nq, NORM_EPS);
        float Gh[NU], Pm[4][NU], dP[NU], dL[NU];
        tksa_row<NU>(Gn + (long)gi * C, sq, C, c, h, t, inq, ink, lrow + wv * 64 * NU, kk, lane, Gh, Pm);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            dP[u] = j < c ? dAn[(long)gi * C + h * c + j] : 0.f;
            dL[u] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float pd = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) pd += Pm[q][u] * dP[u];
            pd = wave_sum(pd);
            if (lane == 0) rowa[i * 4 + q] = pd;
#pragma unroll
            for (int u = 0; u < NU; ++u) dL[u] += a[q] * Pm[q][u] * (dP[u] - pd);
        }
        float rt = 0.f, rho = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int j = lane + 64 * u;
            rt += dL[u] * Gh[u];
            const float dGh = t * dL[u];
            const float e = dGh * Gh[u];
            rho += e;
            if (j < c) {
                E[i * (c + 1) + j] = e;
                const float w = dGh * inq * ink[u];
                Wn[(long)gi * Wp + C + h * c + j] = w;
                Wn[(long)(C + h * c + j) * Wp + gi] = w;
            }
        }
        rt = wave_sum(rt);
        rho = wave_sum(rho);
        if (lane == 0) {
            rowt[i] = rt;
            Wn[(long)gi * Wp + gi] = nq > NORM_EPS ? -rho * inq * inq : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int j = threadIdx.x; j < c; j += 256) {
        float s = 0.f;
        for (int i = 0; i < c; ++i) s += E[i * (c + 1) + j];
        const float nk = sqrtf(sq[C + h * c + j]);
        const float inkj = 1.0f / fmaxf(nk, NORM_EPS);
        const long d = C + h * c + j;
        Wn[d * Wp + d] = nk > NORM_EPS ? -s * inkj * inkj : 0.f;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < c; ++i) s += rowt[i];
        dt_part[(long)n * heads + h] = s;
    }
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int i = 0; i < c; ++i) s += rowa[i * 4 + threadIdx.x];
        da_part[((long)n * heads + h) * 4 + threadIdx.x] = s;
    }
}

__global__ void tksa_da_kernel(const float* __restrict__ part, int nparts, float* __restrict__ da) {
    const int q = threadIdx.x;
    if (q >= 4) return;
    float s = 0.f;
    for (int i = 0; i < nparts; ++i) s += part[(long)i * 4 + q];
    da[q] = s;
}

__global__ void mdta_dtemp_kernel(const float* __restrict__ part, int N, int heads, float* __restrict__ dtemp) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= heads) return;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += part[(long)n * heads + h];
    dtemp[h] = s;
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ alpha, const float* __restrict__ b,
                             long n, float* __restrict__ out) {
    const float al = alpha[0];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = a[i] * al + (b ? b[i] : 0.f);
}

constexpr int DOT_BLOCKS = 512;

__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                 float* __restrict__ part) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void dot_finish_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += (double)part[i];
    out[0] = (float)s;
}

// in [N][4C][H][W] -> out [N][C][2H][2W]; out[c][2y+dy][2x+dx] = in[4c + 2dy + dx][y][x]
__global__ void pixel_shuffle2_kernel(const float* __restrict__ in, int C, int H, int W, long total, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % (2 * W));
        long r = i / (2 * W);
        const int Y = (int)(r % (2 * H)); r /= (2 * H);
        const int c = (int)(r % C);
        const long n = r / C;
        out[i] = in[((n * 4 * C + 4 * c + 2 * (Y & 1) + (X & 1)) * H + (Y >> 1)) * W + (X >> 1)];
    }
}

inline int grid_for(long total, int cap = 8192) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

extern "C" int tdr_row_sumsq(const float* x, int64_t x_ns, int N, int rows, int HW, float* out, void* stream) {
    TDR_REQUIRE(x && out, "tdr_row_sumsq: null pointer");
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(rows, N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, HW, out, rows);
    TDR_LAUNCH_CHECK("row_sumsq");
    return TDR_OK;
}

extern "C" int tdr_mdta_pad(int C) { return (C + 31) / 32 * 32; }

extern "C" int tdr_mdta_softmax(const float* G, const float* ss, const float* temp, int N, int C, int heads, float* A,
                                float* AT, void* stream) {
    TDR_REQUIRE(G && ss && temp && A && AT, "tdr_mdta_softmax: null pointer");
    TDR_REQUIRE(heads > 0 && C % heads == 0 && C / heads <= 192, "tdr_mdta_softmax: need C %% heads == 0 and C/heads <= 192 (C=%d heads=%d)", C, heads);
    hipStream_t st = (hipStream_t)stream;
    const int Cp = tdr_mdta_pad(C);
    const size_t bytes = (size_t)N * Cp * Cp * sizeof(float);
    if (hipMemsetAsync(A, 0, bytes, st) != hipSuccess || hipMemsetAsync(AT, 0, bytes, st) != hipSuccess) {
        tdr_set_error("tdr_mdta_softmax: memset failed");
        return TDR_ERR_HIP;
    }
    if (C / heads <= 128)
        hipLaunchKernelGGL(mdta_softmax_kernel<2>, dim3(heads, N), dim3(256), 0, st, G, ss, temp, C, C / heads, Cp, A, AT);
    else      // PromptIR-ref's 704-channel block (4 heads of 176, network_promptir_guided_arch.py:736)
        hipLaunchKernelGGL(mdta_softmax_kernel<3>, dim3(heads, N), dim3(256), 0, st, G, ss, temp, C, C / heads, Cp, A, AT);
    TDR_LAUNCH_CHECK("mdta_softmax");
    return TDR_OK;
}

extern "C" int tdr_mdta_bwd(const float* G, const float* ss, const float* temp, const float* A, const float* dA, int N, int C,
                            int heads, float* W, float* dtemp, float* ws, void* stream) {
    TDR_REQUIRE(G && ss && temp && A && dA && W && dtemp && ws, "tdr_mdta_bwd: null pointer (ws needs N*heads floats)");
    TDR_REQUIRE(heads > 0 && C % heads == 0 && C / heads <= 192, "tdr_mdta_bwd: need C %% heads == 0 and C/heads <= 192 (C=%d heads=%d)", C, heads);
    hipStream_t st = (hipStream_t)stream;
    const int c = C / heads, Cp = tdr_mdta_pad(C), Wp = tdr_mdta_pad(2 * C);
    if (hipMemsetAsync(W, 0, (size_t)N * Wp * Wp * sizeof(float), st) != hipSuccess) {
        tdr_set_error("tdr_mdta_bwd: memset failed");
        return TDR_ERR_HIP;
    }
    const size_t lds = ((size_t)c * (c + 1) + c) * sizeof(float);
    if (c <= 128) {
        auto kern = mdta_bwd_kernel<2>;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL(kern, dim3(heads, N), dim3(256), lds, st, G, ss, temp, A, dA, C, c, Cp, Wp, W, ws);
    } else {
        auto kern = mdta_bwd_kernel<3>;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL(kern, dim3(heads, N), dim3(256), lds, st, G, ss, temp, A, dA, C, c, Cp, Wp, W, ws);
    }
    hipLaunchKernelGGL(mdta_dtemp_kernel, dim3(tdr_cdiv(heads, 64)), dim3(64), 0, st, ws, N, heads, dtemp);
    TDR_LAUNCH_CHECK("mdta_bwd");
    return TDR_OK;
}

extern "C" int tdr_tksa_softmax(const float* G, const float* ss, const float* temp, const float* am, const int* k4, int N, int C,
                                int heads, float* A, float* AT, void* stream) {
    TDR_REQUIRE(G && ss && temp && am && k4 && A && AT, "tdr_tksa_softmax: null pointer");
    TDR_REQUIRE(heads > 0 && C % heads == 0 && C / heads <= 192, "tdr_tksa_softmax: need C %% heads == 0 and C/heads <= 192 (C=%d heads=%d)", C, heads);
    hipStream_t st = (hipStream_t)stream;
    const int Cp = tdr_mdta_pad(C);
    const size_t bytes = (size_t)N * Cp * Cp * sizeof(float);
    if (hipMemsetAsync(A, 0, bytes, st) != hipSuccess || hipMemsetAsync(AT, 0, bytes, st) != hipSuccess) {
        tdr_set_error("tdr_tksa_softmax: memset failed");
        return TDR_ERR_HIP;
    }
    TksaK kk{{k4[0], k4[1], k4[2], k4[3]}};
    if (C / heads <= 128)
        hipLaunchKernelGGL(tksa_softmax_kernel<2>, dim3(heads, N), dim3(256), 0, st, G, ss, temp, am, kk, C, C / heads, Cp, A, AT);
    else
        hipLaunchKernelGGL(tksa_softmax_kernel<3>, dim3(heads, N), dim3(256), 0, st, G, ss, temp, am, kk, C, C / heads, Cp, A, AT);
    TDR_LAUNCH_CHECK("tksa_softmax");
    return TDR_OK;
}

extern "C" int tdr_tksa_bwd(const float* G, const float* ss, const float* temp, const float* am, const int* k4, const float* dA,
                            int N, int C, int heads, float* W, float* dtemp, float* dam, float* ws, void* stream) {
    TDR_REQUIRE(G && ss && temp && am && k4 && dA && W && dtemp && dam && ws, "tdr_tksa_bwd: null pointer (ws needs 5*N*heads floats)");
    TDR_REQUIRE(heads > 0 && C % heads == 0 && C / heads <= 192, "tdr_tksa_bwd: need C %% heads == 0 and C/heads <= 192 (C=%d heads=%d)", C, heads);
    hipStream_t st = (hipStream_t)stream;
    const int c = C / heads, Wp = tdr_mdta_pad(2 * C);
    if (hipMemsetAsync(W, 0, (size_t)N * Wp * Wp * sizeof(float), st) != hipSuccess) {
        tdr_set_error("tdr_tksa_bwd: memset failed");
        return TDR_ERR_HIP;
    }
    TksaK kk{{k4[0], k4[1], k4[2], k4[3]}};
    const int NU = c <= 128 ? 2 : 3;
    const size_t lds = ((size_t)c * (c + 1) + c + 4 * c + 4 * 64 * NU) * sizeof(float);
    float* da_part = ws + (long)N * heads;
    if (NU == 2) {
        auto kern = tksa_bwd_kernel<2>;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL(kern, dim3(heads, N), dim3(256), lds, st, G, ss, temp, am, kk, dA, C, c, Wp, W, ws, da_part);
    } else {
        auto kern = tksa_bwd_kernel<3>;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL(kern, dim3(heads, N), dim3(256), lds, st, G, ss, temp, am, kk, dA, C, c, Wp, W, ws, da_part);
    }
    hipLaunchKernelGGL(mdta_dtemp_kernel, dim3(tdr_cdiv(heads, 64)), dim3(64), 0, st, ws, N, heads, dtemp);
    hipLaunchKernelGGL(tksa_da_kernel, dim3(1), dim3(64), 0, st, da_part, N * heads, dam);
    TDR_LAUNCH_CHECK("tksa_bwd");
    return TDR_OK;
}

extern "C" int tdr_axpby_dev(const float* a, const float* alpha, const float* b, int64_t numel, float* out, void* stream) {
    TDR_REQUIRE(a && alpha && out, "tdr_axpby_dev: null pointer");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(numel)), dim3(256), 0, (hipStream_t)stream, a, alpha, b, (long)numel, out);
    TDR_LAUNCH_CHECK("axpby");
    return TDR_OK;
}

extern "C" int tdr_dot(const float* a, const float* b, int64_t numel, float* out, float* ws, void* stream) {
    TDR_REQUIRE(a && b && out && ws, "tdr_dot: null pointer (ws needs 512 floats)");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = grid_for(numel, DOT_BLOCKS);
    hipLaunchKernelGGL(dot_kernel, dim3(blocks), dim3(256), 0, st, a, b, (long)numel, ws);
    hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(64), 0, st, ws, blocks, out);
    TDR_LAUNCH_CHECK("dot");
    return TDR_OK;
}

extern "C" int tdr_pixel_shuffle2(const float* in, int N, int C, int H, int W, float* out, void* stream) {
    TDR_REQUIRE(in && out, "tdr_pixel_shuffle2: null pointer");
    const long total = (long)N * 4 * C * H * W;
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, C, H, W, total, out);
    TDR_LAUNCH_CHECK("pixel_shuffle2");
    return TDR_OK;
}
