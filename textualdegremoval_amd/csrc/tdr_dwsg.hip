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
//   GATE_SUM  : grouped conv with two inputs per output (DRSformer-ref MSFN dwconv3x3_1, network_drsformer_guided_arch.py:
//               231-232): out[c] = relu?(dw(t[2c]) + dw(t[2c+1]) + b[c]) -- the pair is (2c, 2c+1) instead of (c, c+C); forward
//               and one-pass backward only
#include <stdlib.h>
#include "tdr_common.h"
#include "tdr_erf.h"
#include "../../include/tdr.h"

namespace {

enum { MODE_FWD = 0, MODE_DU = 1, MODE_DT = 2 };
enum { GATE_MUL = 0, GATE_GELU = 1, GATE_NONE = 2, GATE_SUM = 3 };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_1ulp(x * 0.70710678118654752f)); }
// value and derivative from one erf
__device__ __forceinline__ void gelu_erf_both(float x, float& g, float& dg) {
    const float h = 0.5f * (1.0f + erf_1ulp(x * 0.70710678118654752f));
    g = x * h;
    dg = h + x * 0.3989422804014327f * __expf(-0.5f * x * x);
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
    const float* act;    // GATE_NONE / GATE_SUM: ReLU after the conv -- FWD: any non-null value; one-pass backward: the saved output
    long sum_ns[3];      // GATE_SUM: per-image strides (floats) of out (FWD) | dg, act (backward); the C-plane tensors may be channel
                         // slices of wider ones (MSFN writes z1 / z2 straight into the concatenated buffer)
    // GATE_NONE, "split halves": the first C planes and the last C planes of out (FWD) | dg, act (one-pass backward) live in two
    // tensors (base pointers X / XB, common per-image stride) -- MSFN's cross-concatenation [a3[:h] | a5[:h]], [a3[h:] | a5[h:]]
    // is then never copied.  XB == NULL: the dense [N][2C] layout.
    float* outB; long out_hns;
    const float* dgB; long dg_hns;
    const float* actB; long act_hns;
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

// fetch_row in two halves, so that a row can be requested one loop iteration before its neighbours are exchanged
struct RawRow { f32x4 m; float le, re; };
__device__ __forceinline__ RawRow load_raw(const float* __restrict__ plane, int y, int x0, int H, int W, bool active,
                                           bool left_lane, bool right_lane) {
    RawRow r;
    r.m = f32x4{0.f, 0.f, 0.f, 0.f};
    r.le = 0.f; r.re = 0.f;
    if (active && y >= 0 && y < H) {
        const float* row = plane + (long)y * W;
        r.m = *reinterpret_cast<const f32x4*>(row + x0);
        if (!left_lane && x0 > 0) r.le = row[x0 - 1];
        if (!right_lane && x0 + 4 < W) r.re = row[x0 + 4];
    }
    return r;
}
__device__ __forceinline__ Row6 finish_row(const RawRow& r, bool left_lane, bool right_lane) {
    float l = __shfl_up(r.m[3], 1, 64), rr = __shfl_down(r.m[0], 1, 64);
    if (!left_lane) l = r.le;
    if (!right_lane) rr = r.re;
    Row6 o;
    o.v[0] = l; o.v[1] = r.m[0]; o.v[2] = r.m[1]; o.v[3] = r.m[2]; o.v[4] = r.m[3]; o.v[5] = rr;
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
    const int pl1 = GATE == GATE_SUM ? 2 * c : c, pl2 = GATE == GATE_SUM ? 2 * c + 1 : c + C;     // the two planes of this block
    const float* p1 = a.a + ((long)n * 2 * C + pl1) * HW;
    const float* p2 = a.a + ((long)n * 2 * C + pl2) * HW;
    float w1[9], w2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int k = MODE == MODE_DT ? 8 - i : i;           // transposed conv = correlation with flipped taps
        w1[i] = a.w[pl1 * 9 + k];
        w2[i] = a.w[pl2 * 9 + k];
    }
    const float b1 = (MODE == MODE_DT || !a.b) ? 0.f : a.b[c];
    const float b2 = (MODE == MODE_DT || !a.b || GATE == GATE_SUM) ? 0.f : a.b[c + C];
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
            if (a.act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o1[e] = fmaxf(o1[e], 0.f); o2[e] = fmaxf(o2[e], 0.f); }
            }
            if (live) {
                const f32x4 q1 = {o1[0], o1[1], o1[2], o1[3]}, q2 = {o2[0], o2[1], o2[2], o2[3]};
                float* d1 = a.outB ? a.out + (long)n * a.out_hns + (long)c * HW : a.out + ((long)n * 2 * C + c) * HW;
                float* d2 = a.outB ? a.outB + (long)n * a.out_hns + (long)c * HW : a.out + ((long)n * 2 * C + c + C) * HW;
                *reinterpret_cast<f32x4*>(d1 + (long)y * W + x0) = q1;
                *reinterpret_cast<f32x4*>(d2 + (long)y * W + x0) = q2;
            }
        } else if (MODE == MODE_FWD && GATE == GATE_SUM) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = a.act ? fmaxf(o1[e] + o2[e], 0.f) : o1[e] + o2[e];
            if (live) *reinterpret_cast<f32x4*>(a.out + (long)n * a.sum_ns[0] + (long)c * HW + (long)y * W + x0) = o;
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
                        float gl, gd;
                        gelu_erf_both(o1[e], gl, gd);
                        d1[e] = gv[e] * o2[e] * gd;
                        d2[e] = gv[e] * gl;
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

// Backward in ONE pass (one column block, W <= 1024: a strip's horizontal neighbour is in its wave, in the next wave of the
// same workgroup, or outside the image).
// A thread still owns a 4-column strip and walks rows, now with two register windows: three rows of t (to recompute u and to
// accumulate dW) and three rows of du = d(loss)/d(u).  Row y of du is produced from the t window and the dg row, exchanged with
// the neighbouring lanes by DPP shuffles, and row y-1 of dt = dw^T(du) leaves from the du window -- du never exists in memory
// (the two-pass version writes and re-reads 2c planes).  A strip recomputes the du row above and below its rpt rows.
// GATE_NONE (plain depthwise conv): du IS dout, so its window is fetched like t and the kernel serves any width.
template <int GATE>
__global__ __launch_bounds__(256) void dwsg_bwd_fused_kernel(DwArgs a) {
    __shared__ float red[4][20];
    __shared__ float edge[2][4][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = blockIdx.y, n = blockIdx.z, C = a.C, H = a.H, W = a.W;
    const int TPRW = 1 << a.tprw_log2;
    const int cg = tid & (TPRW - 1), strip = tid >> a.tprw_log2;
    const int bx = blockIdx.x % a.ncb, by = blockIdx.x / a.ncb;
    const int x0 = (bx * TPRW + cg) * 4;
    const int ybeg = (by * (256 >> a.tprw_log2) + strip) * a.rpt;
    const bool active = x0 < W && ybeg < H;
    const bool left_lane = lane != 0 && cg != 0;
    const bool right_lane = lane != 63 && cg != TPRW - 1;
    const long HW = (long)H * W;
    constexpr bool DIRECT = GATE == GATE_NONE || GATE == GATE_SUM;       // du is (masked) dout: nothing to recompute
    const int pl1 = GATE == GATE_SUM ? 2 * c : c, pl2 = GATE == GATE_SUM ? 2 * c + 1 : c + C;
    const float* p1 = a.a + ((long)n * 2 * C + pl1) * HW;
    const float* p2 = a.a + ((long)n * 2 * C + pl2) * HW;
    const float* q1 = GATE == GATE_NONE ? (a.dgB ? a.dg + (long)n * a.dg_hns + (long)c * HW : a.dg + ((long)n * 2 * C + c) * HW)
                      : (GATE == GATE_SUM ? a.dg + (long)n * a.sum_ns[1] + (long)c * HW : a.dg + ((long)n * C + c) * HW);
    const float* q2 = (GATE == GATE_NONE && a.dgB) ? a.dgB + (long)n * a.dg_hns + (long)c * HW : q1 + (long)C * HW;   // GATE_NONE only
    float w1[9], w2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        w1[i] = a.w[pl1 * 9 + i];
        w2[i] = a.w[pl2 * 9 + i];
    }
    const float b1 = a.b ? a.b[c] : 0.f, b2 = (a.b && !DIRECT) ? a.b[c + C] : 0.f;
    float dgb = 0.f;
    if (GATE == GATE_MUL && a.dgb) dgb = a.dgb[(long)n * C + c] * a.dgb_mul;
    float acc[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) acc[i] = 0.f;

    Row6 t1[3], t2[3], e1[3], e2[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) { e1[0].v[k] = e1[1].v[k] = e2[0].v[k] = e2[1].v[k] = 0.f; }
    t1[0] = fetch_row(p1, ybeg - 2, x0, H, W, active, left_lane, right_lane);
    t2[0] = fetch_row(p2, ybeg - 2, x0, H, W, active, left_lane, right_lane);
    t1[1] = fetch_row(p1, ybeg - 1, x0, H, W, active, left_lane, right_lane);
    t2[1] = fetch_row(p2, ybeg - 1, x0, H, W, active, left_lane, right_lane);
    // software pipeline: the rows of iteration i+1 are requested before iteration i computes (3 waves / SIMD do not hide
    // the latency of a load that is consumed in the iteration that issues it)
    constexpr bool PF = true;
    RawRow n1, n2, m1, m2;
    const float* r1p = (DIRECT && a.act) ? (GATE == GATE_SUM ? a.act + (long)n * a.sum_ns[2] + (long)c * HW
                                            : (a.actB ? a.act + (long)n * a.act_hns + (long)c * HW
                                                      : a.act + ((long)n * 2 * C + c) * HW)) : nullptr;          // ReLU outputs
    const float* r2p = !r1p ? nullptr : (a.actB ? a.actB + (long)n * a.act_hns + (long)c * HW : r1p + (long)C * HW);   // (GATE_NONE)
    auto relu_mask = [&](RawRow& d, const float* plane, int y, bool on) {                              // d *= (act > 0)
        const RawRow k = load_raw(plane, y, x0, H, W, on, left_lane, right_lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) d.m[e] = k.m[e] > 0.f ? d.m[e] : 0.f;
        d.le = k.le > 0.f ? d.le : 0.f;
        d.re = k.re > 0.f ? d.re : 0.f;
    };
    f32x4 gn = {0.f, 0.f, 0.f, 0.f};
    if (PF) {
        n1 = load_raw(p1, ybeg, x0, H, W, active, left_lane, right_lane);
        n2 = load_raw(p2, ybeg, x0, H, W, active, left_lane, right_lane);
        if (DIRECT) {
            m1 = load_raw(q1, ybeg - 1, x0, H, W, active, left_lane, right_lane);
            if (r1p) relu_mask(m1, r1p, ybeg - 1, active);
            if (GATE == GATE_NONE) {
                m2 = load_raw(q2, ybeg - 1, x0, H, W, active, left_lane, right_lane);
                if (r1p) relu_mask(m2, r2p, ybeg - 1, active);
            }
        } else if (active && ybeg - 1 >= 0) {
            gn = *reinterpret_cast<const f32x4*>(q1 + (long)(ybeg - 1) * W + x0);
        }
    }
    for (int i = 0; i < a.rpt + 2; ++i) {
        const int y = ybeg - 1 + i;                              // the du row of this iteration
        f32x4 gv = gn;
        if (PF) {
            t1[2] = finish_row(n1, left_lane, right_lane);
            t2[2] = finish_row(n2, left_lane, right_lane);
            if (DIRECT) {
                e1[2] = finish_row(m1, left_lane, right_lane);
                e2[2] = GATE == GATE_NONE ? finish_row(m2, left_lane, right_lane) : e1[2];
            }
            const bool more = active && i <= a.rpt;             // nothing is consumed after the last iteration
            n1 = load_raw(p1, y + 2, x0, H, W, more, left_lane, right_lane);
            n2 = load_raw(p2, y + 2, x0, H, W, more, left_lane, right_lane);
            if (DIRECT) {
                m1 = load_raw(q1, y + 1, x0, H, W, more, left_lane, right_lane);
                if (r1p) relu_mask(m1, r1p, y + 1, more);
                if (GATE == GATE_NONE) {
                    m2 = load_raw(q2, y + 1, x0, H, W, more, left_lane, right_lane);
                    if (r1p) relu_mask(m2, r2p, y + 1, more);
                }
            } else {
                gn = f32x4{0.f, 0.f, 0.f, 0.f};
                if (more && y + 1 >= 0 && y + 1 < H) gn = *reinterpret_cast<const f32x4*>(q1 + (long)(y + 1) * W + x0);
            }
        } else {
            t1[2] = fetch_row(p1, y + 1, x0, H, W, active, left_lane, right_lane);
            t2[2] = fetch_row(p2, y + 1, x0, H, W, active, left_lane, right_lane);
            gv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (active && y >= 0 && y < H) gv = *reinterpret_cast<const f32x4*>(q1 + (long)y * W + x0);
        }
        const bool in_img = active && y >= 0 && y < H;
        if (!DIRECT) {
            float o1[4] = {b1, b1, b1, b1}, o2[4] = {b2, b2, b2, b2};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o1[e] += w1[ky * 3 + kx] * t1[ky].v[e + kx];
                        o2[e] += w2[ky * 3 + kx] * t2[ky].v[e + kx];
                    }
            if (GATE == GATE_MUL && in_img) gv += dgb;
            float d1[4], d2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (GATE == GATE_GELU) {
                    float gl, gd;
                    gelu_erf_both(o1[e], gl, gd);
                    d1[e] = gv[e] * o2[e] * gd;
                    d2[e] = gv[e] * gl;
                } else {
                    d1[e] = gv[e] * o2[e];
                    d2[e] = gv[e] * o1[e];
                }
                if (!in_img) { d1[e] = 0.f; d2[e] = 0.f; }         // (gv = 0 already; keeps NaN/Inf of the recomputation out)
            }
            float l1 = __shfl_up(d1[3], 1, 64), r1 = __shfl_down(d1[0], 1, 64);
            float l2 = __shfl_up(d2[3], 1, 64), r2 = __shfl_down(d2[0], 1, 64);
            if (!left_lane) { l1 = 0.f; l2 = 0.f; }                 // no lane to the left: image border, or ...
            if (!right_lane) { r1 = 0.f; r2 = 0.f; }
            if (TPRW > 64) {                                        // ... a row spans 2 / 4 waves: their edge values go through LDS
                const int wv = tid >> 6, par = i & 1;               // (one barrier per row; parity double-buffers the slots)
                if (lane == 0) { edge[par][wv][0] = d1[0]; edge[par][wv][1] = d2[0]; }
                if (lane == 63) { edge[par][wv][2] = d1[3]; edge[par][wv][3] = d2[3]; }
                __syncthreads();
                if (lane == 0 && cg != 0) { l1 = edge[par][wv - 1][2]; l2 = edge[par][wv - 1][3]; }
                if (lane == 63 && cg != TPRW - 1) { r1 = edge[par][wv + 1][0]; r2 = edge[par][wv + 1][1]; }
            }
            e1[2].v[0] = l1; e1[2].v[5] = r1; e2[2].v[0] = l2; e2[2].v[5] = r2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { e1[2].v[e + 1] = d1[e]; e2[2].v[e + 1] = d2[e]; }
        }
        if (i >= 1 && i <= a.rpt && in_img) {                       // parameter gradients: the strip's own rows only
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[ky * 3 + kx] += e1[2].v[e + 1] * t1[ky].v[e + kx];
                        acc[10 + ky * 3 + kx] += e2[2].v[e + 1] * t2[ky].v[e + kx];
                    }
            acc[9] += (e1[2].v[1] + e1[2].v[2]) + (e1[2].v[3] + e1[2].v[4]);
            acc[19] += (e2[2].v[1] + e2[2].v[2]) + (e2[2].v[3] + e2[2].v[4]);
        }
        if (i >= 2) {                                               // dt row y-1 = transposed conv of the du window
            const int yo = y - 1;
            float o1[4] = {0.f, 0.f, 0.f, 0.f}, o2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o1[e] += w1[8 - (ky * 3 + kx)] * e1[ky].v[e + kx];
                        o2[e] += w2[8 - (ky * 3 + kx)] * e2[ky].v[e + kx];
                    }
            if (active && yo < H) {
                const f32x4 v1 = {o1[0], o1[1], o1[2], o1[3]}, v2 = {o2[0], o2[1], o2[2], o2[3]};
                *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + pl1) * HW + (long)yo * W + x0) = v1;
                *reinterpret_cast<f32x4*>(a.out + ((long)n * 2 * C + pl2) * HW + (long)yo * W + x0) = v2;
            }
        }
        t1[0] = t1[1]; t1[1] = t1[2];
        t2[0] = t2[1]; t2[1] = t2[2];
        e1[0] = e1[1]; e1[1] = e1[2];
        e2[0] = e2[1]; e2[1] = e2[2];
    }
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[tid >> 6][i] = s;
    }
    __syncthreads();
    if (tid < 20)
        a.part[(((long)n * C + c) * gridDim.x + blockIdx.x) * 20 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// dw[ch][9], db[ch] from part[N][C][nb][20] (ch < C: first half of a pair, ch >= C: second half); fixed summation order
__global__ void dw_param_finish_kernel(const float* __restrict__ part, int N, int C, int nb, float* __restrict__ dw,
                                       float* __restrict__ db, int sum_pairs) {
    const int c = blockIdx.x, tid = threadIdx.x;
    if (tid >= 20) return;
    float sacc = 0.f;
    for (int m = 0; m < N; ++m)
        for (int g = 0; g < nb; ++g) sacc += part[(((long)m * C + c) * nb + g) * 20 + tid];
    const int kk = tid % 10;
    if (sum_pairs) {                                     // GATE_SUM: planes (2c, 2c+1) share the output channel c and its bias
        if (kk < 9) dw[(2 * c + (tid >= 10)) * 9 + kk] = sacc;
        else if (db && tid == 9) db[c] = sacc;
        return;
    }
    const int ch = tid < 10 ? c : c + C;
    if (kk < 9) dw[ch * 9 + kk] = sacc;
    else if (db) db[ch] = sacc;
}

// many problems: table rows {part, dw, db, N, C, nb}, blockIdx.y = problem, grid.x sized for the widest one; dw_param_finish_kernel's order
__global__ void dw_param_finish_multi_kernel(const long long* __restrict__ tab) {
    const long long* row = tab + 6L * blockIdx.y;
    const int N = (int)row[3], C = (int)row[4], nb = (int)row[5];
    if ((int)blockIdx.x >= C) return;
    const float* part = reinterpret_cast<const float*>(row[0]);
    float* dw = reinterpret_cast<float*>(row[1]);
    float* db = reinterpret_cast<float*>(row[2]);
    const int c = blockIdx.x, tid = threadIdx.x;
    if (tid >= 20) return;
    float sacc = 0.f;
    for (int m = 0; m < N; ++m)
        for (int g = 0; g < nb; ++g) sacc += part[(((long)m * C + c) * nb + g) * 20 + tid];
    const int kk = tid % 10;
    const int ch = tid < 10 ? c : c + C;
    if (kk < 9) dw[ch * 9 + kk] = sacc;
    else db[ch] = sacc;
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

// one-pass backward: longer strips (the two recomputed du rows amortise over rpt), one column block
DwGeom dw_geom_fused(int H, int W) {
    DwGeom g = dw_geom(H, W);
    const int spb = 256 >> g.tprw_log2;
    static const int cap = tdr_tune_env("TDR_DWF_RPT") ? atoi(tdr_tune_env("TDR_DWF_RPT")) : 16;     // tuning aid
    int rpt = tdr_cdiv(H, spb);
    if (rpt > cap) rpt = cap;
    if (rpt < 1) rpt = 1;
    g.rpt = rpt;
    g.nby = tdr_cdiv(H, spb * rpt);
    g.nb = g.ncb * g.nby;
    return g;
}

bool dw_two_pass() {                                     // TDR_DWSG_TWO_PASS=1: the round-1 two-pass backward (A/B tests)
    const char* e = getenv("TDR_DWSG_TWO_PASS");
    return e && e[0] == '1';
}

template <int GATE>
int dw_bwd_fused(const float* t, const float* dg, const float* dg_bias, float dg_bias_mul, const float* w, const float* b, int N,
                 int C, int H, int W, float* dt, float* dw, float* db, float* ws, hipStream_t st, const float* act = nullptr,
                 long dg_ns = 0, long act_ns = 0, const float* dgB = nullptr, const float* actB = nullptr) {
    const DwGeom q = dw_geom_fused(H, W);
    DwArgs a{t, dg, w, b, dt, ws, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, nullptr, nullptr, dg_bias, dg_bias_mul, act,
             {0, dg_ns, act_ns}, nullptr, 0, dgB, dg_ns, actB, act_ns};
    hipLaunchKernelGGL((dwsg_bwd_fused_kernel<GATE>), dim3(q.nb, C, N), dim3(256), 0, st, a);
    if (dw)        // dw = NULL: the partials stay in ws, the caller finishes them (tdr_dw_param_finish)
        hipLaunchKernelGGL(dw_param_finish_kernel, dim3(C), dim3(32), 0, st, ws, N, C, q.nb, dw, db, GATE == GATE_SUM ? 1 : 0);
    return 0;
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

extern "C" int tdr_dwsg_bwd_parts_supported(int W) { return (W <= 1024 && !dw_two_pass()) ? 1 : 0; }

extern "C" int tdr_dw_param_finish(const float* ws, int N, int C, int H, int W, float* dw, float* db, void* stream) {
    TDR_REQUIRE(ws && dw && db && N > 0 && C > 0, "tdr_dw_param_finish: bad argument");
    const DwGeom q = dw_geom_fused(H, W);
    hipLaunchKernelGGL(dw_param_finish_kernel, dim3(C), dim3(32), 0, (hipStream_t)stream, const_cast<float*>(ws), N, C, q.nb, dw, db, 0);
    TDR_LAUNCH_CHECK("dw_param_finish_kernel");
    return TDR_OK;
}

// partial-row blocks per plane of a (H, W) problem: the `nb` word of its table row
extern "C" int tdr_dw_param_finish_nb(int H, int W) { return dw_geom_fused(H, W).nb; }

// table [nprob][6] of 64-bit words {ws, dw, db, N, C, nb = tdr_dw_param_finish_nb(H, W)} in DEVICE memory; max_C = the widest problem
extern "C" int tdr_dw_param_finish_multi(const void* table, int nprob, int max_C, void* stream) {
    TDR_REQUIRE(table && nprob > 0 && max_C > 0, "tdr_dw_param_finish_multi: bad argument");
    hipLaunchKernelGGL(dw_param_finish_multi_kernel, dim3(max_C, nprob), dim3(32), 0, (hipStream_t)stream, static_cast<const long long*>(table));
    TDR_LAUNCH_CHECK("dw_param_finish_multi_kernel");
    return TDR_OK;
}

extern "C" int tdr_dwsg_bwd(const float* dg, const float* t, const float* w, const float* b, int N, int C, int H, int W,
                            float* dt, float* dw, float* db, float* ws, void* stream) {
    return tdr_dwsg_bwd_biased(dg, nullptr, 0.f, t, w, b, N, C, H, W, dt, dw, db, ws, stream);
}

extern "C" int tdr_dwsg_bwd_biased(const float* dg, const float* dg_bias, float dg_bias_mul, const float* t, const float* w,
                                   const float* b, int N, int C, int H, int W, float* dt, float* dw, float* db, float* ws,
                                   void* stream) {
    TDR_REQUIRE(dg && t && w && b && dt && ws, "tdr_dwsg_bwd: null pointer");
    TDR_REQUIRE((dw != nullptr) == (db != nullptr), "tdr_dwsg_bwd: dw and db are given together or not at all");
    TDR_REQUIRE(dw || tdr_dwsg_bwd_parts_supported(W), "tdr_dwsg_bwd: dw = db = NULL (partials left to the caller) needs the one-pass backward");
    TDR_REQUIRE(W % 4 == 0, "tdr_dwsg_bwd: W must be a multiple of 4 (got %d)", W);
    hipStream_t st = (hipStream_t)stream;
    if (W <= 1024 && !dw_two_pass()) {
        dw_bwd_fused<GATE_MUL>(t, dg, dg_bias, dg_bias_mul, w, b, N, C, H, W, dt, dw, db, ws, st);
        TDR_LAUNCH_CHECK("dwsg_bwd_fused");
        return TDR_OK;
    }
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
    if (W <= 1024 && !dw_two_pass()) {
        dw_bwd_fused<GATE_GELU>(t, dg, nullptr, 0.f, w, b, N, C, H, W, dt, dw, db, ws, st);
        TDR_LAUNCH_CHECK("dwgelu_bwd_fused");
        return TDR_OK;
    }
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

// planes = 2*C (even); out[n][p] = dw_p * t[n][p] (+ b[p]), then ReLU if relu != 0
extern "C" int tdr_dwconv_fwd(const float* t, const float* w, const float* b, int N, int planes, int H, int W, float* out,
                              void* stream) {
    return tdr_dwconv_act_fwd(t, w, b, N, planes, H, W, 0, out, stream);
}

extern "C" int tdr_dwconv_act_fwd(const float* t, const float* w, const float* b, int N, int planes, int H, int W, int relu,
                                  float* out, void* stream) {
    TDR_REQUIRE(t && w && out, "tdr_dwconv_fwd: null pointer");
    TDR_REQUIRE(W % 4 == 0 && planes % 2 == 0, "tdr_dwconv_fwd: W %% 4 and planes %% 2 must be 0 (got %d, %d)", W, planes);
    const DwGeom q = dw_geom(H, W);
    DwArgs a{t, nullptr, w, b, out, nullptr, planes / 2, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, nullptr, nullptr, nullptr, 0.f,
             relu ? out : nullptr};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_FWD, GATE_NONE>), dim3(q.nb, planes / 2, N), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("dwconv_fwd");
    return TDR_OK;
}

// dt = dw^T(dout); dw[p][9] = sum dout[p] * shifted t[p]; db[p] = sum dout[p] (db may be NULL).  ws >= tdr_dwsg_ws_floats(N, planes/2, H, W)
extern "C" int tdr_dwconv_bwd(const float* dout, const float* t, const float* w, int N, int planes, int H, int W, float* dt,
                              float* dw, float* db, float* ws, void* stream) {
    return tdr_dwconv_act_bwd(dout, nullptr, t, w, N, planes, H, W, dt, dw, db, ws, stream);
}

// act != NULL: the forward ended in a ReLU and act is its output -- dout is masked by act > 0 as it is read (one-pass kernel only)
extern "C" int tdr_dwconv_act_bwd(const float* dout, const float* act, const float* t, const float* w, int N, int planes, int H,
                                  int W, float* dt, float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(dout && t && w && dt && dw && ws, "tdr_dwconv_bwd: null pointer");
    TDR_REQUIRE(!act || !dw_two_pass(), "tdr_dwconv_act_bwd: the ReLU mask needs the one-pass kernel (unset TDR_DWSG_TWO_PASS)");
    TDR_REQUIRE(W % 4 == 0 && planes % 2 == 0, "tdr_dwconv_bwd: W %% 4 and planes %% 2 must be 0 (got %d, %d)", W, planes);
    hipStream_t st = (hipStream_t)stream;
    const int C = planes / 2;
    if (!dw_two_pass()) {
        dw_bwd_fused<GATE_NONE>(t, dout, nullptr, 0.f, w, nullptr, N, C, H, W, dt, dw, db, ws, st, act);
        TDR_LAUNCH_CHECK("dwconv_bwd_fused");
        return TDR_OK;
    }
    const DwGeom q = dw_geom(H, W);
    DwArgs a1{t, dout, w, nullptr, nullptr, ws, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DU, GATE_NONE>), dim3(q.nb, C, N), dim3(256), 0, st, a1);
    DwArgs a2{dout, nullptr, w, nullptr, dt, ws, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, dw, db};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_DT, GATE_MUL>), dim3(q.nb, C, N), dim3(256), 0, st, a2);
    TDR_LAUNCH_CHECK("dwconv_bwd");
    return TDR_OK;
}

// ---- DRSformer-ref MSFN second stage: grouped 3x3 with two inputs per output.  t [N][2C][H][W], w [C][2][3][3], b [C] | NULL,
// out [N][C][H][W] = relu?(sum of the two filtered planes + b).  Backward (one pass, W <= 1024): act = saved output | NULL.
extern "C" int tdr_dwpair_fwd(const float* t, const float* w, const float* b, int N, int C, int H, int W, int relu, float* out,
                              int64_t out_ns, void* stream) {
    TDR_REQUIRE(t && w && out, "tdr_dwpair_fwd: null pointer");
    TDR_REQUIRE(W % 4 == 0 && out_ns % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                "tdr_dwpair_fwd: W and out_ns must be multiples of 4, out 16-byte aligned (got %d)", W);
    const DwGeom q = dw_geom(H, W);
    DwArgs a{t, nullptr, w, b, out, nullptr, C, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, nullptr, nullptr, nullptr, 0.f,
             relu ? out : nullptr, {(long)out_ns, 0, 0}};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_FWD, GATE_SUM>), dim3(q.nb, C, N), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("dwpair_fwd");
    return TDR_OK;
}

extern "C" int tdr_dwpair_bwd(const float* dout, int64_t dout_ns, const float* act, int64_t act_ns, const float* t, const float* w,
                              int N, int C, int H, int W, float* dt, float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(dout && t && w && dt && dw && ws, "tdr_dwpair_bwd: null pointer");
    TDR_REQUIRE(W % 4 == 0 && W <= 1024, "tdr_dwpair_bwd: W must be a multiple of 4, at most 1024 (got %d)", W);
    TDR_REQUIRE(((dout_ns | act_ns) & 3) == 0 && ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(act)) & 15) == 0,
                "tdr_dwpair_bwd: dout / act strides must be multiples of 4 floats, pointers 16-byte aligned");
    dw_bwd_fused<GATE_SUM>(t, dout, nullptr, 0.f, w, nullptr, N, C, H, W, dt, dw, db, ws, (hipStream_t)stream, act, (long)dout_ns,
                           (long)act_ns);
    TDR_LAUNCH_CHECK("dwpair_bwd");
    return TDR_OK;
}

// ---- the plain depthwise 3x3 (+ ReLU) with its 2C output planes split over two tensors: planes [0, C) at outA, planes [C, 2C) at
// outB, both [N][C][H][W] with per-image stride out_ns (channel slices of wider buffers).  The backward reads dout and act the same way.
extern "C" int tdr_dwconv_halves_fwd(const float* t, const float* w, const float* b, int N, int planes, int H, int W, int relu,
                                     float* outA, float* outB, int64_t out_ns, void* stream) {
    TDR_REQUIRE(t && w && outA && outB, "tdr_dwconv_halves_fwd: null pointer");
    TDR_REQUIRE(W % 4 == 0 && planes % 2 == 0 && out_ns % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(outA) | reinterpret_cast<uintptr_t>(outB)) & 15) == 0,
                "tdr_dwconv_halves_fwd: W %% 4, planes %% 2, out_ns %% 4 must be 0 and the outputs 16-byte aligned");
    const DwGeom q = dw_geom(H, W);
    DwArgs a{t, nullptr, w, b, outA, nullptr, planes / 2, H, W, q.tprw_log2, q.rpt, q.ncb, 1.0f, nullptr, nullptr, nullptr, 0.f,
             relu ? outA : nullptr, {0, 0, 0}, outB, (long)out_ns, nullptr, 0, nullptr, 0};
    hipLaunchKernelGGL((dwsg_stencil_kernel<MODE_FWD, GATE_NONE>), dim3(q.nb, planes / 2, N), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("dwconv_halves_fwd");
    return TDR_OK;
}

extern "C" int tdr_dwconv_halves_bwd(const float* doutA, const float* doutB, int64_t dout_ns, const float* actA, const float* actB,
                                     int64_t act_ns, const float* t, const float* w, int N, int planes, int H, int W, float* dt,
                                     float* dw, float* db, float* ws, void* stream) {
    TDR_REQUIRE(doutA && doutB && t && w && dt && dw && ws && (!actA == !actB), "tdr_dwconv_halves_bwd: null pointer");
    TDR_REQUIRE(W % 4 == 0 && planes % 2 == 0 && ((dout_ns | act_ns) & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(doutA) | reinterpret_cast<uintptr_t>(doutB) | reinterpret_cast<uintptr_t>(actA) |
                      reinterpret_cast<uintptr_t>(actB)) & 15) == 0,
                "tdr_dwconv_halves_bwd: W %% 4, planes %% 2, strides %% 4 must be 0 and the inputs 16-byte aligned");
    dw_bwd_fused<GATE_NONE>(t, doutA, nullptr, 0.f, w, nullptr, N, planes / 2, H, W, dt, dw, db, ws, (hipStream_t)stream, actA,
                            (long)dout_ns, (long)act_ns, doutB, actB);
    TDR_LAUNCH_CHECK("dwconv_halves_bwd");
    return TDR_OK;
}
