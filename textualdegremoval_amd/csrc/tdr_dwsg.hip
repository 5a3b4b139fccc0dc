// depthwise 3x3 (+bias) + SimpleGate (+ SCA pool partials), forward and backward
//   network_nafnet_guided_arch.py:185-187 (conv2, groups = 2c), :170-175 (SimpleGate), :192-196 (pool)
//
// HBM-bound stencils.  A thread owns a 4-column strip and walks RPT rows with a 3-row register window: each input
// row is fetched once as one float4 per plane (consecutive lanes -> consecutive 16-byte pieces of the row), the
// x-1 / x+4 neighbours come from the adjacent lanes by DPP shuffles (an extra dword load only at wave / row-block
// edges), the rows above/below are re-used from registers.  No LDS tiles, no barriers in the main loop.
//   MODE_FWD : g = (dw1(t1)+b1) * (dw2(t2)+b2), pool partial sums
//   MODE_DU  : u1,u2 recomputed; du1 = dg*u2, du2 = dg*u1 written to scratch; dW / db partial sums (20 per block)
//   MODE_DT  : dt1 = dw1^T(du1), dt2 = dw2^T(du2)   (the same stencil with flipped taps)
// The backward is two passes over a 2c-plane scratch tensor (du) instead of one LDS-tiled kernel with halo
// recomputation: 9c instead of 5c plane passes, but both run at streaming speed and du stays in L2/MALL.
//
// The same stencil serves Restormer-ref (network_restormer_guided_arch.py):
//   GATE_GELU : GDFN gate  g = gelu(dw1(t1)) * dw2(t2)  (:236-239, erf GELU), no pooling
//   GATE_NONE : the plain depthwise conv of MDTA's qkv_dwconv (:254,260): planes handled in pairs (c, c+C),
//               FWD writes both filtered planes, DU only accumulates dW/db from dout (nothing to recompute)
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

enum { MODE_FWD = 0, MODE_DU = 1, MODE_DT = 2 };
enum { GATE_MUL = 0, GATE_GELU = 1, GATE_NONE = 2 };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

struct DwArgs {
    const float* a;      // FWD/DU: t [N][2C][H][W]; DT: du [N][2C][H][W]
    const float* dg;     // DU: [N][C][H][W]
    const float* w;      // [2C][9]
    const float* b;      // [2C]
    float* out;          // FWD: g [N][C]; DU: du [N][2C]; DT: dt [N][2C]
    float* part;         // FWD: [N*C][nb]; DU: [N*C][nb][20]
    int C, H, W, tprw_log2, rpt, ncb;
    float pscale;        // FWD: factor on the pool partial (1, or 1/(H*W) when one block covers the plane and writes `pooled` itself)
    float* pdw;          // DT: dw [2C][9] / db [2C] (db may be NULL) finished from the DU pass's `part` by block (0, c, 0) --
    float* pdb;          //     the parameter-gradient finish rides on the second pass instead of its own launch
    const float* dgb;    // DU (GATE_MUL): optional [N][C] plane constant added to dg, times dgb_mul
    float dgb_mul;
};

struct Row6 { float v[6]; };

// one row of the 4-column strip with its two horizontal neighbours; zero outside the image
__device__ __forceinline__ Row6 fetch_row(const float* __restrict__ plane, int y, int x0, int H, int W, bool active,
                                          bool left_lane, bool right_lane) {
    const bool rok = active && y >= 0 && y < H;
    const float* row = plane + (long)min(max(y, 0), H - 1) * W;
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
    if (rok) m = *reinterpret_cast<const f32x4*>(row + x0);
    float l = __shfl_up(m[3], 1, 64), r = __shfl_down(m[0], 1, 64);
    if (!left_lane) l = (rok && x0 > 0) ? row[x0 - 1] : 0.f;
    if (!right_lane) r = (rok && x0 + 4 < W) ? row[x0 + 4] : 0.f;
    Row6 o;
    o.v[0] = l; o.v[1] = m[0]; o.v[2] = m[1]; o.v[3] = m[2]; o.v[4] = m[3]; o.v[5] = r;
    return o;
}

template <int MODE, int GATE>
__global__ __launch_bounds__(256) void dwsg_stencil_kernel(DwArgs a) {
    __shared__ float red[4][20];
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = blockIdx.y, n = blockIdx.z, C = a.C, H = a.H, W = a.W;
    const int TPRW = 1 << a.tprw_log2;
    const int cg = tid & (TPRW - 1), strip = tid >> a.tprw_log2;
    const int bx = blockIdx.x % a.ncb, by = blockIdx.x / a.ncb;
    const int x0 = (bx * TPRW + cg) * 4;
    const int ybeg = (by * (256 >> a.tprw_log2) + strip) * a.rpt;
    const bool active = x0 < W && ybeg < H;
    // the neighbour lane holds the adjacent strip unless this is the first/last lane of the wave or of the row block
    const bool left_lane = lane != 0 && cg != 0;
    const bool right_lane = lane != 63 && cg != TPRW - 1;
    const long HW = (long)H * W;
    const float* p1 = a.a + ((long)n * 2 * C + c) * HW;
    const float* p2 = p1 + (long)C * HW;
    float w1[9], w2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int k = MODE == MODE_DT ? 8 - i : i;           // transposed conv = correlation with flipped taps
        w1[i] = a.w[c * 9 + k];
        w2[i] = a.w[(c + C) * 9 + k];
    }
    const float b1 = (MODE == MODE_DT || !a.b) ? 0.f : a.b[c], b2 = (MODE == MODE_DT || !a.b) ? 0.f : a.b[c + C];
    float acc[MODE == MODE_DU ? 20 : 1];
#pragma unroll
    for (int i = 0; i < (MODE == MODE_DU ? 20 : 1); ++i) acc[i] = 0.f;

    Row6 r1[3], r2[3];
    r1[0] = fetch_row(p1, ybeg - 1, x0, H, W, active, left_lane, right_lane);
    r2[0] = fetch_row(p2, ybeg - 1, x0, H, W, active, left_lane, right_lane);
    r1[1] = fetch_row(p1, ybeg, x0, H, W, active, left_lane, right_lane);
    r2[1] = fetch_row(p2, ybeg, x0, H, W, active, left_lane, right_lane);
    for (int i = 0; i < a.rpt; ++i) {
        const int y = ybeg + i;
        r1[2] = fetch_row(p1, y + 1, x0, H, W, active, left_lane, right_lane);     // uniform trip count: shuffles stay converged
        r2[2] = fetch_row(p2, y + 1, x0, H, W, active, left_lane, right_lane);
        const bool live = active && y < H;
        float o1[4] = {b1, b1, b1, b1}, o2[4] = {b2, b2, b2, b2};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o1[e] += w1[ky * 3 + kx] * r1[ky].v[e + kx];
                    o2[e] += w2[ky * 3 + kx] * r2[ky].v[e + kx];
                }
        if (MODE == MODE_FWD && GATE == GATE_NONE) {
            if (live) {
                const f32x4 q1 = {o1[0], o1[1], o1[2], o1[3]}, q2 = {o2[0], o2[1], o2[2], o2[3]};
                *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + c) * HW + (long)y * W + x0) = q1;
                *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + c + C) * HW + (long)y * W + x0) = q2;
            }
        } else if (MODE == MODE_FWD) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (GATE == GATE_GELU ? gelu_erf(o1[e]) : o1[e]) * o2[e];
            if (live) {
                *reinterpret_cast<f32x4*>(a.out + ((long)n * C + c) * HW + (long)y * W + x0) = o;
                acc[0] += (o[0] + o[1]) + (o[2] + o[3]);
            }
        } else if (MODE == MODE_DU) {
            f32x4 d1 = {0.f, 0.f, 0.f, 0.f}, d2 = {0.f, 0.f, 0.f, 0.f};
            if (GATE == GATE_NONE) {
                if (live) {
                    d1 = *reinterpret_cast<const f32x4*>(a.dg + ((long)n * 2 * C + c) * HW + (long)y * W + x0);
                    d2 = *reinterpret_cast<const f32x4*>(a.dg + ((long)n * 2 * C + c + C) * HW + (long)y * W + x0);
                }
            } else {
                f32x4 gv = {0.f, 0.f, 0.f, 0.f};
                if (live) {
                    gv = *reinterpret_cast<const f32x4*>(a.dg + ((long)n * C + c) * HW + (long)y * W + x0);
                    if (a.dgb) gv += a.dgb[(long)n * C + c] * a.dgb_mul;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (GATE == GATE_GELU) {
                        d1[e] = gv[e] * o2[e] * gelu_erf_grad(o1[e]);
                        d2[e] = gv[e] * gelu_erf(o1[e]);
                    } else {
                        d1[e] = gv[e] * o2[e];
                        d2[e] = gv[e] * o1[e];
                    }
                }
                if (live) {
                    *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + c) * HW + (long)y * W + x0) = d1;
                    *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + c + C) * HW + (long)y * W + x0) = d2;
                }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[ky * 3 + kx] += d1[e] * r1[ky].v[e + kx];
                        acc[10 + ky * 3 + kx] += d2[e] * r2[ky].v[e + kx];
                    }
            acc[9] += (d1[0] + d1[1]) + (d1[2] + d1[3]);
            acc[19] += (d2[0] + d2[1]) + (d2[2] + d2[3]);
        } else {
            if (live) {
                const f32x4 q1 = {o1[0], o1[1], o1[2], o1[3]}, q2 = {o2[0], o2[1], o2[2], o2[3]};
                *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + c) * HW + (long)y * W + x0) = q1;
                *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + c + C) * HW + (long)y * W + x0) = q2;
            }
        }
        r1[0] = r1[1]; r1[1] = r1[2];
        r2[0] = r2[1]; r2[1] = r2[2];
    }
    if (MODE == MODE_FWD && GATE == GATE_MUL) {
        const float s = wave_sum(acc[0]);
        if (lane == 0) red[tid >> 6][0] = s;
        __syncthreads();
        if (tid == 0) a.part[((long)n * C + c) * gridDim.x + blockIdx.x] = ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) * a.pscale;
    } else if (MODE == MODE_DU) {
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            const float s = wave_sum(acc[i]);
            if (lane == 0) red[tid >> 6][i] = s;
        }
        __syncthreads();
        if (tid < 20)
            a.part[(((long)n * C + c) * gridDim.x + blockIdx.x) * 20 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    } else if (MODE == MODE_DT) {
        // dw[ch][9], db[ch] from part[N][C][nb][20] of the previous (DU) launch (ch < C: first half, ch >= C: second half);
        // same summation order as dw_param_finish_kernel
        if (a.pdw && blockIdx.x == 0 && blockIdx.z == 0 && tid < 20) {
            const int nb = gridDim.x, N = gridDim.z;
            float sacc = 0.f;
            for (int m = 0; m < N; ++m)
                for (int g = 0; g < nb; ++g) sacc += a.part[(((long)m * C + c) * nb + g) * 20 + tid];
            const int ch = tid < 10 ? c : c + C, kk = tid % 10;
            if (kk < 9) a.pdw[ch * 9 + kk] = sacc;
            else if (a.pdb) a.pdb[ch] = sacc;
        }
    }
}

__global__ void dw_pool_finish_kernel(const float* __restrict__ part, int NC, int nb, float inv_hw, float* __restrict__ pooled) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    float s = 0.f;
    for (int k = 0; k < nb; ++k) s += part[(long)i * nb + k];
    pooled[i] = s * inv_hw;
}

struct DwGeom { int tprw_log2, rpt, ncb, nby, nb; };

DwGeom dw_geom(int H, int W) {
    DwGeom g;
    int groups = W / 4, lg = 0;
    while ((1 << lg) < groups && lg < 8) ++lg;          // threads per row block: next power of two, at most 256
    g.tprw_log2 = lg;
    const int spb = 256 >> lg;                           // strips per block
    g.ncb = tdr_cdiv(groups, 1 << lg);
    int rpt = tdr_cdiv(H, spb);                          // rows per thread: up to 8, fewer on small maps (more blocks)
    if (rpt > 8) rpt = 8;
    if (rpt < 1) rpt = 1;
    g.rpt = rpt;
    g.nby = tdr_cdiv(H, spb * rpt);
    g.nb = g.ncb * g.nby;
    return g;
}

}  // namespace

extern "C" int64_t tdr_dwsg_ws_floats(int N, int C, int H, int W) {
    const DwGeom g = dw_geom(H, W);
    return (int64_t)N * C * g.nb * 20 + 4 + (int64_t)N * 2 * C * H * W;  // partial sums + (aligned) du scratch of the backward
}

extern "C" int tdr_dwsg_fwd(const float* t, const float* w, const float* b, int N, int C, int H, int W, float* g, float* pooled,
                            float* ws, void* stream) {
    TDR_REQUIRE(t && w && b && g && pooled && ws, "tdr_dwsg_fwd: null pointer");
    TDR_REQUIRE(W % 4 == 0, "tdr_dwsg_fwd: W must be a multiple of 4 (got %d)", W);
    hipStream_t st = (hipStream_t)stream;
    const DwGeom q = dw_geom(H, W);
    const float inv_hw = 1.0f / (float)((long)H * W);
    if (q.nb == 1) {     // one block per plane (the 64 x 64 level and below): the partial IS the pool sum -- no finish launch
        DwArgs a{t, nullptr, w, b, g, pooled, C, H, W, q.tprw_log2, q.rpt, q.ncb, inv_hw};
        hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_FWD, GATE_MUL>), dim3(q.nb, C, N), dim3(256), 0, st, a);
        TDR_LAUNCH_CHECK("dwsg_fwd");
        return TDR_OK;
    }
    DwArgs a{t, nullptr, w, b, g, ws, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_FWD, GATE_MUL>), dim3(q.nb, C, N), dim3(256), 0, st, a);
    hipLaunchKernelGGL(dw_pool_finish_kernel, dim3(tdr_cdiv(N * C, 256)), dim3(256), 0, st, ws, N * C, q.nb, inv_hw, pooled);
    TDR_LAUNCH_CHECK("dwsg_fwd");
    return TDR_OK;
}

extern "C" int tdr_dwsg_bwd(const float* dg, const float* t, const float* w, const float* b, int N, int C, int H, int W,
                            float* dt, float* dw, float* db, float* ws, void* stream) {
    return tdr_dwsg_bwd_biased(dg, nullptr, 0.f, t, w, b, N, C, H, W, dt, dw, db, ws, stream);
}

extern "C" int tdr_dwsg_bwd_biased(const float* dg, const float* dg_bias, float dg_bias_mul, const float* t, const float* w,
                                   const float* b, int N, int C, int H, int W, float* dt, float* dw, float* db, float* ws,
                                   void* stream) {
    TDR_REQUIRE(dg && t && w && b && dt && dw && db && ws, "tdr_dwsg_bwd: null pointer");
    TDR_REQUIRE(W % 4 == 0, "tdr_dwsg_bwd: W must be a multiple of 4 (got %d)", W);
    hipStream_t st = (hipStream_t)stream;
    const DwGeom q = dw_geom(H, W);
    float* part = ws;
    float* du = ws + (int64_t)N * C * q.nb * 20;
    du += (4 - (reinterpret_cast<uintptr_t>(du) / 4) % 4) % 4;   // 16-byte aligned scratch planes (ws holds one spare vector)
    DwArgs a1{t, dg, w, b, du, part, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, nullptr, nullptr, dg_bias, dg_bias_mul};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DU, GATE_MUL>), dim3(q.nb, C, N), dim3(256), 0, st, a1);
    DwArgs a2{du, nullptr, w, b, dt, part, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, dw, db};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DT, GATE_MUL>), dim3(q.nb, C, N), dim3(256), 0, st, a2);
    TDR_LAUNCH_CHECK("dwsg_bwd");
    return TDR_OK;
}

// ---- Restormer-ref: GDFN gate (gelu(dw1 t1) * dw2 t2) and the plain depthwise conv of MDTA; b / db may be NULL (bias=False)
extern "C" int tdr_dwgelu_fwd(const float* t, const float* w, const float* b, int N, int C, int H, int W, float* g,
                              void* stream) {
    TDR_REQUIRE(t && w && g, "tdr_dwgelu_fwd: null pointer");
    TDR_REQUIRE(W % 4 == 0, "tdr_dwgelu_fwd: W must be a multiple of 4 (got %d)", W);
    const DwGeom q = dw_geom(H, W);
    DwArgs a{t, nullptr, w, b, g, nullptr, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_FWD, GATE_GELU>), dim3(q.nb, C, N), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("dwgelu_fwd");
    return TDR_OK;
}

extern "C" int tdr_dwgelu_bwd(const float* dg, const float* t, const float* w, const float* b, int N, int C, int H, int W,
                              float* dt, float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(dg && t && w && dt && dw && ws, "tdr_dwgelu_bwd: null pointer");
    TDR_REQUIRE(W % 4 == 0, "tdr_dwgelu_bwd: W must be a multiple of 4 (got %d)", W);
    hipStream_t st = (hipStream_t)stream;
    const DwGeom q = dw_geom(H, W);
    float* part = ws;
    float* du = ws + (int64_t)N * C * q.nb * 20;
    du += (4 - (reinterpret_cast<uintptr_t>(du) / 4) % 4) % 4;
    DwArgs a1{t, dg, w, b, du, part, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DU, GATE_GELU>), dim3(q.nb, C, N), dim3(256), 0, st, a1);
    DwArgs a2{du, nullptr, w, nullptr, dt, part, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, dw, db};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DT, GATE_MUL>), dim3(q.nb, C, N), dim3(256), 0, st, a2);
    TDR_LAUNCH_CHECK("dwgelu_bwd");
    return TDR_OK;
}

// planes = 2*C (even); out[n][p] = dw_p * t[n][p] (+ b[p])
extern "C" int tdr_dwconv_fwd(const float* t, const float* w, const float* b, int N, int planes, int H, int W, float* out,
                              void* stream) {
    TDR_REQUIRE(t && w && out, "tdr_dwconv_fwd: null pointer");
    TDR_REQUIRE(W % 4 == 0 && planes % 2 == 0, "tdr_dwconv_fwd: W %% 4 and planes %% 2 must be 0 (got %d, %d)", W, planes);
    const DwGeom q = dw_geom(H, W);
    DwArgs a{t, nullptr, w, b, out, nullptr, planes / 2, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_FWD, GATE_NONE>), dim3(q.nb, planes / 2, N), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("dwconv_fwd");
    return TDR_OK;
}

// dt = dw^T(dout); dw[p][9] = sum dout[p] * shifted t[p]; db[p] = sum dout[p] (db may be NULL).  ws >= tdr_dwsg_ws_floats(N, planes/2, H, W)
extern "C" int tdr_dwconv_bwd(const float* dout, const float* t, const float* w, int N, int planes, int H, int W, float* dt,
                              float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(dout && t && w && dt && dw && ws, "tdr_dwconv_bwd: null pointer");
    TDR_REQUIRE(W % 4 == 0 && planes % 2 == 0, "tdr_dwconv_bwd: W %% 4 and planes %% 2 must be 0 (got %d, %d)", W, planes);
    hipStream_t st = (hipStream_t)stream;
    const int C = planes / 2;
    const DwGeom q = dw_geom(H, W);
    DwArgs a1{t, dout, w, nullptr, nullptr, ws, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DU, GATE_NONE>), dim3(q.nb, C, N), dim3(256), 0, st, a1);
    DwArgs a2{dout, nullptr, w, nullptr, dt, ws, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, dw, db};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DT, GATE_MUL>), dim3(q.nb, C, N), dim3(256), 0, st, a2);
    TDR_LAUNCH_CHECK("dwconv_bwd");
    return TDR_OK;
}
