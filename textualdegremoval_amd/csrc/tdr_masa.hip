// MASA match-and-transfer kernels (network_nafnet_guided_arch.py:495-707).
// The two correlation searches run on the MFMA conv kernel (tdr_conv_forward with
// per-block packed patches as filters); this file holds the norms, arg-max/box
// arithmetic, the fused multi-scale transfer (gather + overlap average + soft
// attention, no unfold/fold materialisation) and the backward kernels.
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

__device__ __forceinline__ int wrap_idx(int v, int n) { return v < 0 ? v + n : v; }   // python negative index

// ---------------------------------------------------------------------------
// replicate-pad(1) + block cut  (:627-629) and its adjoint
// ---------------------------------------------------------------------------
__global__ void lr_blocks_fwd_kernel(const float* __restrict__ feat, int C, int H, int W, int py, int px, int ky, int kx,
                                     long total, float* __restrict__ blk) {
    const int BH = ky + 2, BW = kx + 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int u = (int)(i % BW); long r = i / BW;
        const int v = (int)(r % BH); r /= BH;
        const int c = (int)(r % C); r /= C;
        const int p = (int)(r % (py * px)); const long n = r / (py * px);
        const int by = p / px, bx = p % px;
        const int y = min(max(by * ky + v - 1, 0), H - 1), x = min(max(bx * kx + u - 1, 0), W - 1);
        blk[i] = feat[((n * C + c) * H + y) * W + x];
    }
}

__global__ void lr_blocks_bwd_kernel(const float* __restrict__ dblk, int C, int H, int W, int py, int px, int ky, int kx,
                                     long total, float* __restrict__ dfeat) {
    const int BH = ky + 2, BW = kx + 2, P = py * px;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C); const long n = r / C;
        float s = 0.f;
        // padded coordinates that replicate onto (y,x)
        for (int Y = (y == 0 ? 0 : y + 1); Y <= (y == H - 1 ? H + 1 : y + 1); ++Y) {
            for (int X = (x == 0 ? 0 : x + 1); X <= (x == W - 1 ? W + 1 : x + 1); ++X) {
                for (int by = max(0, (Y - 2) / ky - 1); by < py && by * ky <= Y; ++by) {
                    const int v = Y - by * ky;
                    if (v < 0 || v >= BH) continue;
                    for (int bx = max(0, (X - 2) / kx - 1); bx < px && bx * kx <= X; ++bx) {
                        const int u = X - bx * kx;
                        if (u < 0 || u >= BW) continue;
                        s += dblk[(((n * P + by * px + bx) * C + c) * BH + v) * BW + u];
                    }
                }
            }
        }
        dfeat[i] = s;
    }
}

// ---------------------------------------------------------------------------
// inverse L2 norm of 3x3 (dilated) patches over all channels; zero outside the map
// inv[b][oy][ox] = 1 / max(sqrt(sum_{c,ky,kx} x[b,c,oy*step+off+ky*dil, ox*step+off+kx*dil]^2), 1e-12)
// a block = 64 consecutive output positions (lanes: coalesced rows) x 16 channel slices (waves), summed through LDS in a
// fixed order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void patch_inv_norm_kernel(const float* __restrict__ x, int C, int H, int W, int OH,
                                                             int OW, int dil, int step, int off, long npos,
                                                             float* __restrict__ inv) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long pos = blockIdx.x * 64L + lane;
    const bool live = pos < npos;
    const long pc = live ? pos : npos - 1;
    const int ox = (int)(pc % OW); const long r = pc / OW;
    const int oy = (int)(r % OH); const long b = r / OH;
    int offs[9];                      // tap offsets inside a channel plane, -1 outside the map
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = oy * step + off + (k / 3) * dil, xx = ox * step + off + (k % 3) * dil;
        offs[k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? yy * W + xx : -1;
    }
    float s = 0.f;
    const long HW = (long)H * W;
    for (int c = sl; c < C; c += 16) {
        const float* xc = x + (b * C + c) * HW;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float v = offs[k] >= 0 ? xc[offs[k]] : 0.f;
            s += v * v;
        }
    }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][lane];
        inv[pos] = 1.0f / fmaxf(sqrtf(t), 1e-12f);
    }
}

// few positions (the LR centre patches: N*P of them): one wave per position, lanes over channels
__global__ __launch_bounds__(256) void patch_inv_norm_wave_kernel(const float* __restrict__ x, int C, int H, int W, int OH,
                                                                 int OW, int dil, int step, int off, long npos,
                                                                 float* __restrict__ inv) {
    const long pos = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (pos >= npos) return;
    const int lane = threadIdx.x & 63;
    const int ox = (int)(pos % OW); long r = pos / OW;
    const int oy = (int)(r % OH); const long b = r / OH;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float* xc = x + (b * C + c) * (long)H * W;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = oy * step + off + (k / 3) * dil, xx = ox * step + off + (k % 3) * dil;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float v = xc[(long)yy * W + xx];
                s += v * v;
            }
        }
    }
    s = wave_sum(s);
    if (lane == 0) inv[pos] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
}

// ---------------------------------------------------------------------------
// coarse arg-max over all ref positions + box arithmetic (:534, :635-657)
// one block per (n,p)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int box_start(int idx, int size, int diameter) {
    int lo = idx - diameter / 2 - 1, hi = idx + diameter / 2 + 1;
    if (lo < 0) { lo = 0; hi = diameter + 1; }
    if (hi > size - 1) { hi = size - 1; lo = hi - (diameter + 1); }
    return lo;
}

__global__ __launch_bounds__(256) void coarse_argmax_box_kernel(const float* __restrict__ dots,
                                                               const float* __restrict__ invq,
                                                               const float* __restrict__ invk, int ND, int N, int P,
                                                               int Hr, int Wr, int diameter, int* __restrict__ index,
                                                               int* __restrict__ y1, int* __restrict__ x1) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int np = blockIdx.x, n = np / P;
    const int R = Hr * Wr;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int r = threadIdx.x; r < R; r += 256) {
        float s = 0.f;
        for (int d = 0; d < ND; ++d)
            s += dots[((long)d * N * P + np) * R + r] * invq[(long)d * N * P + np] * invk[((long)d * N + n) * R + r];
        if (s > best) { best = s; bi = r; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v2 = sv[threadIdx.x + o]; const int i2 = si[threadIdx.x + o];
            if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v2; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int idx = si[0];
        index[np] = idx;
        x1[np] = box_start(idx % Wr, Wr, diameter);
        y1[np] = box_start(idx / Wr, Hr, diameter);
    }
}

// gather ref block with python negative wrap (:672-678)
__global__ void gather_ref_block_kernel(const float* __restrict__ feat, int C, int H, int W, const int* __restrict__ y1,
                                        const int* __restrict__ x1, int P, int side, int s, long total,
                                        float* __restrict__ out) {
    const int SS = side * s;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int u = (int)(i % SS); long r = i / SS;
        const int v = (int)(r % SS); r /= SS;
        const int c = (int)(r % C); const long b = r / C;
        const long n = b / P;
        const int y = wrap_idx(y1[b] * s + v, H), x = wrap_idx(x1[b] * s + u, W);
        out[i] = feat[((n * C + c) * H + y) * W + x];
    }
}

// adjoint of gather_ref_block as a GATHER over the feature map (deterministic: no atomics).  A feature pixel (y, x) is
// read by block b at the block-local rows v with wrap(y1[b] + v) == y: v = y - y1[b], and v = y - H - y1[b] (the python
// negative-index wrap, :672-678) -- each valid when inside [0, side).  Blocks, then rows, then columns in fixed order.
__global__ void scatter_ref_block_kernel(const float* __restrict__ dblk, int C, int H, int W, const int* __restrict__ y1,
                                         const int* __restrict__ x1, int P, int side, long total,
                                         float* __restrict__ dfeat) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C); const long n = r / C;
        float acc = 0.f;
        for (int p = 0; p < P; ++p) {
            const long b = n * P + p;
            const int by = y1[b], bx = x1[b];
#pragma unroll
            for (int wy = 0; wy < 2; ++wy) {
                const int v = y - wy * H - by;
                if (v < 0 || v >= side || (wy == 1) != (by + v < 0)) continue;
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const int u = x - wx * W - bx;
                    if (u < 0 || u >= side || (wx == 1) != (bx + u < 0)) continue;
                    acc += dblk[((b * C + c) * side + v) * side + u];
                }
            }
        }
        dfeat[i] += acc;
    }
}

// fine arg-max: one wave per (b,p) row of R candidates
__global__ __launch_bounds__(256) void fine_argmax_kernel(const float* __restrict__ dots, const float* __restrict__ invq,
                                                         const float* __restrict__ invk, long rows, int P, int R,
                                                         int* __restrict__ index_all, float* __restrict__ soft_att) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const long b = row / P;
    const float iq = invq[row];
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int r = lane; r < R; r += 64) {
        const float v = dots[row * R + r] * iq * invk[b * R + r];
        if (v > best) { best = v; bi = r; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(bi, o, 64);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (lane == 0) { index_all[row] = bi; soft_att[row] = best; }
}

// ---------------------------------------------------------------------------
// fine-search backward (gradient of the selected cosine similarity)
//  dq = datt*invq*(k*invk - att*q*invq),  dk = datt*invk*(q*invq - att*k*invk)
// gather form (deterministic): one block per (b, c).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fine_search_bwd_kernel(const float* __restrict__ datt,
                                                             const float* __restrict__ soft_att,
                                                             const int* __restrict__ index_all,
                                                             const float* __restrict__ lrb, const float* __restrict__ refb,
                                                             const float* __restrict__ invq, const float* __restrict__ invk,
                                                             int C, int K, int D, float* __restrict__ dlrb,
                                                             float* __restrict__ drefb) {
    extern __shared__ float sm[];
    const int b = blockIdx.y, c = blockIdx.x;
    const int P = K * K, BS = K + 2, R1 = D - 2;
    float* s_da = sm;               // [P] datt
    float* s_at = s_da + P;         // [P] att
    float* s_iq = s_at + P;         // [P]
    float* s_ik = s_iq + P;         // [P] invk of the selected ref patch
    int* s_ix = reinterpret_cast<int*>(s_ik + P);   // [P]
    float* s_q = reinterpret_cast<float*>(s_ix + P);  // [BS*BS]
    float* s_k = s_q + BS * BS;                        // [D*D]
    int* s_ry = reinterpret_cast<int*>(s_k + D * D);   // [P] row / col of the selected ref patch, q offset of patch p:
    int* s_rx = s_ry + P;                              //     the integer divisions leave the P x D*D inner loop
    int* s_qo = s_rx + P;
    for (int p = threadIdx.x; p < P; p += 256) {
        const int ix = index_all[(long)b * P + p];
        s_da[p] = datt[(long)b * P + p];
        s_at[p] = soft_att[(long)b * P + p];
        s_iq[p] = invq[(long)b * P + p];
        s_ik[p] = invk[(long)b * R1 * R1 + ix];
        s_ix[p] = ix;
        s_ry[p] = ix / R1;
        s_rx[p] = ix % R1;
        s_qo[p] = (p / K) * BS + (p % K);
    }
    for (int i = threadIdx.x; i < BS * BS; i += 256) s_q[i] = lrb[((long)b * C + c) * BS * BS + i];
    for (int i = threadIdx.x; i < D * D; i += 256) s_k[i] = refb[((long)b * C + c) * D * D + i];
    __syncthreads();
    // dlrb[y][x] = sum over patches p=(y-ky, x-kx)
    for (int i = threadIdx.x; i < BS * BS; i += 256) {
        const int y = i / BS, x = i % BS;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int pyy = y - k / 3, pxx = x - k % 3;
            if (pyy < 0 || pyy >= K || pxx < 0 || pxx >= K) continue;
            const int p = pyy * K + pxx;
            const int ry = s_ry[p], rx = s_rx[p];
            const float kv = s_k[(ry + k / 3) * D + rx + k % 3];
            s += s_da[p] * s_iq[p] * (kv * s_ik[p] - s_at[p] * s_q[i] * s_iq[p]);
        }
        dlrb[((long)b * C + c) * BS * BS + i] = s;
    }
    // drefb[v][u] = sum over p whose selected patch covers (v,u)
    for (int i = threadIdx.x; i < D * D; i += 256) {
        const int v = i / D, u = i % D;
        float s = 0.f;
        for (int p = 0; p < P; ++p) {
            const int kyy = v - s_ry[p], kxx = u - s_rx[p];
            if (kyy < 0 || kyy > 2 || kxx < 0 || kxx > 2) continue;
            const float qv = s_q[s_qo[p] + kyy * BS + kxx];
            s += s_da[p] * s_ik[p] * (qv * s_iq[p] - s_at[p] * s_k[i] * s_ik[p]);
        }
        drefb[((long)b * C + c) * D * D + i] = s;
    }
}

// ---------------------------------------------------------------------------
// fused transfer.  Output pixel (n, Y, X) of the re-tiled warped map; its LR block
// b=(n,by,bx), local (Yl,Xl) in [0,K*s)^2.  Up to 3x3 LR patches (i,j) cover it.
// ---------------------------------------------------------------------------
struct TrGeom {
    int src[9];     // source offsets (y*W + x) in the ref feature map, -1 if not covering
    float wgt;      // att_up / cnt
    float inv_cnt;
    // bilinear corners of soft_att (indices into [K*K]) and weights
    int a00, a01, a10, a11; float w00, w01, w10, w11;
};

__device__ __forceinline__ void tr_geometry(int Yl, int Xl, int b, const int* __restrict__ y1, const int* __restrict__ x1,
                                            const int* __restrict__ index_all, const float* __restrict__ soft_att, int K,
                                            int side, int s, int H, int W, TrGeom& g) {
    const int R1 = side - 2;
    const int i0 = Yl / s, j0 = Xl / s;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i = i0 + k / 3 - 1, j = j0 + k % 3 - 1;
        g.src[k] = -1;
        if (i < 0 || i >= K || j < 0 || j >= K) continue;
        const int idx = index_all[(long)b * K * K + i * K + j];
        const int ry = idx / R1, rx = idx % R1;
        const int sy = ry * s + Yl - i * s + s, sx = rx * s + Xl - j * s + s;
        const int gy = wrap_idx(y1[b] * s + sy, H), gx = wrap_idx(x1[b] * s + sx, W);
        g.src[k] = gy * W + gx;
        ++cnt;
    }
    g.inv_cnt = 1.0f / (float)cnt;
    // bilinear (align_corners=False) upsample of soft_att by s
    const float fy = fmaxf(((float)Yl + 0.5f) / (float)s - 0.5f, 0.f), fx = fmaxf(((float)Xl + 0.5f) / (float)s - 0.5f, 0.f);
    const int yy0 = min((int)fy, K - 1), xx0 = min((int)fx, K - 1);
    const int yy1 = min(yy0 + 1, K - 1), xx1 = min(xx0 + 1, K - 1);
    const float ly = fy - (float)yy0, lx = fx - (float)xx0;
    g.a00 = yy0 * K + xx0; g.a01 = yy0 * K + xx1; g.a10 = yy1 * K + xx0; g.a11 = yy1 * K + xx1;
    g.w00 = (1.f - ly) * (1.f - lx); g.w01 = (1.f - ly) * lx; g.w10 = ly * (1.f - lx); g.w11 = ly * lx;
    const float* at = soft_att + (long)b * K * K;
    // same association as the oracle: (top*(1-ly) + bot*ly) with top=(a00*(1-lx)+a01*lx)
    const float top = at[g.a00] * (1.f - lx) + at[g.a01] * lx;
    const float bot = at[g.a10] * (1.f - lx) + at[g.a11] * lx;
    g.wgt = (top * (1.f - ly) + bot * ly);
}

constexpr int TR_CG = 8;   // channels per thread-iteration group

__global__ __launch_bounds__(256) void transfer_fwd_kernel(const float* __restrict__ feat, int C, int H, int W,
                                                          const int* __restrict__ y1, const int* __restrict__ x1,
                                                          const int* __restrict__ index_all,
                                                          const float* __restrict__ soft_att, int py, int px, int K,
                                                          int side, int s, float* __restrict__ out, long out_ns) {
    const int OW = px * K * s, OH = py * K * s;
    const int pix = blockIdx.x * 256 + threadIdx.x, n = blockIdx.z;
    if (pix >= OH * OW) return;
    const int Y = pix / OW, X = pix % OW;
    const int by = Y / (K * s), bx = X / (K * s);
    const int b = (n * py + by) * px + bx;
    TrGeom g;
    tr_geometry(Y - by * K * s, X - bx * K * s, b, y1, x1, index_all, soft_att, K, side, s, H, W, g);
    const long HWf = (long)H * W, HWo = (long)OH * OW;
    // the <= 9 covering patches usually point at the same source pixel (coherent matches): one gather per distinct
    // source, weighted by its multiplicity (as in transfer_bwd_kernel)
    float mult[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int m = 0;
        bool first = g.src[k] >= 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const bool same = g.src[q] == g.src[k];
            m += same ? 1 : 0;
            if (q < k && same) first = false;
        }
        mult[k] = first ? (float)m : 0.f;
    }
    const int c0 = blockIdx.y * TR_CG;
    float accs[TR_CG];
#pragma unroll
    for (int i = 0; i < TR_CG; ++i) {              // all channels' gathers in flight before the first store
        const float* f = feat + ((long)n * C + min(c0 + i, C - 1)) * HWf;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (mult[k] != 0.f) acc += mult[k] * f[g.src[k]];
        accs[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < TR_CG; ++i)
        if (c0 + i < C) out[(long)n * out_ns + (long)(c0 + i) * HWo + pix] = accs[i] * g.inv_cnt * g.wgt;
}

// order-independent accumulation: every contribution is rounded to a multiple of 2^-shift (shift chosen from the
// largest |dout| of the launch so that 2^16 contributions of that size still fit) and added as a 64-bit integer -- integer
// addition is associative, so the sum does not depend on the order in which the workgroups arrive (float atomics did).
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ x, long x_ns, long per4, int N,
                                                         unsigned* __restrict__ out) {
    float m = 0.f;
    const f32x4* p = reinterpret_cast<const f32x4*>(x + (long)blockIdx.y * x_ns);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per4; i += (long)gridDim.x * 256) {
        const f32x4 v = p[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
}
__device__ __forceinline__ int fixed_shift(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xff) - 127;                   // floor(log2 absmax); -127 for 0 / subnormal
    return 46 - (e + 1);                                                   // |value| * 2^shift < 2^46; 2^16 of them < 2^62
}
__global__ void fixed_to_float_kernel(const long long* __restrict__ acc, long n, const unsigned* __restrict__ amax_bits,
                                      float* __restrict__ out) {
    const double inv = ldexp(1.0, -fixed_shift(amax_bits[0]));
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] += (float)((double)acc[i] * inv);
}

// backward pass 1 (output-pixel parallel): dfeat scatter (64-bit fixed-point atomics: deterministic) + per-pixel
// dA = sum_c dout * acc/cnt.
// grid = (pixel blocks, channel chunks, N): the coarse levels have few pixels and many channels, so channels are
// split across blocks (dA partial per chunk, summed by pass 2).  The <= 9 patches covering an output pixel usually
// point at the SAME source pixel (coherent matches): duplicates are merged once per pixel.
template <bool FIXED>
__global__ __launch_bounds__(256) void transfer_bwd_kernel(const float* __restrict__ dout, long dout_ns,
                                                          const float* __restrict__ feat, int C, int H, int W,
                                                          const int* __restrict__ y1, const int* __restrict__ x1,
                                                          const int* __restrict__ index_all,
                                                          const float* __restrict__ soft_att, int py, int px, int K,
                                                          int side, int s, int cpb /*channels per block*/, int N,
                                                          const unsigned* __restrict__ amax_bits,
                                                          unsigned long long* __restrict__ dfacc /*[N][C][H*W] fixed point*/,
                                                          float* __restrict__ dfeat /*[N][C][H*W], !FIXED*/,
                                                          float* __restrict__ dA /*[chunks][N][OH*OW]*/) {
    const int OW = px * K * s, OH = py * K * s;
    const int pix = blockIdx.x * 256 + threadIdx.x, n = blockIdx.z;
    if (pix >= OH * OW) return;
    const int Y = pix / OW, X = pix % OW;
    const int by = Y / (K * s), bx = X / (K * s);
    const int b = (n * py + by) * px + bx;
    TrGeom g;
    tr_geometry(Y - by * K * s, X - bx * K * s, b, y1, x1, index_all, soft_att, K, side, s, H, W, g);
    const long HWf = (long)H * W, HWo = (long)OH * OW;
    const float coef = g.inv_cnt * g.wgt;
    float mult[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int m = 0;
        bool first = g.src[k] >= 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const bool same = g.src[q] == g.src[k];
            m += same ? 1 : 0;
            if (q < k && same) first = false;
        }
        mult[k] = first ? (float)m : 0.f;
    }
    float da = 0.f;
    const int c0 = blockIdx.y * cpb, c1 = min(c0 + cpb, C);
    const float fscale = FIXED ? ldexpf(1.f, fixed_shift(amax_bits[0])) : 1.f;
    for (int c = c0; c < c1; ++c) {
        const float* f = feat + ((long)n * C + c) * HWf;
        const float go = dout[(long)n * dout_ns + (long)c * HWo + pix];
        float acc = 0.f;
        const float gv = go * coef;
#pragma unroll
        for (int k = 0; k < 9; ++k)
            if (mult[k] > 0.f) {
                acc += mult[k] * f[g.src[k]];
                if (FIXED) atomicAdd(&dfacc[((long)n * C + c) * HWf + g.src[k]], (unsigned long long)__float2ll_rn(gv * mult[k] * fscale));   // two's complement
                else atomicAdd(&dfeat[((long)n * C + c) * HWf + g.src[k]], gv * mult[k]);
            }
        da += go * acc * g.inv_cnt;
    }
    dA[((long)blockIdx.y * N + n) * HWo + pix] = da;
}

// backward pass 2 (deterministic gather): datt[b][i][j] += sum over local pixels whose bilinear
// footprint touches (i,j)
__global__ __launch_bounds__(64) void transfer_datt_kernel(const float* __restrict__ dA, int py, int px, int K, int s,
                                                          int chunks, long chunk_stride, float* __restrict__ datt) {
    const int b = blockIdx.x, e = blockIdx.y;           // e = i*K + j
    const int P = py * px, n = b / P, by = (b % P) / px, bx = b % px;
    const int i = e / K, j = e % K;
    const int OW = px * K * s, OH = py * K * s, KS = K * s;
    const int ylo = max((i - 1) * s, 0), yhi = min((i + 2) * s, KS);   // generous footprint
    const int xlo = max((j - 1) * s, 0), xhi = min((j + 2) * s, KS);
    const int nw = xhi - xlo, tot = (yhi - ylo) * nw;
    float acc = 0.f;
    for (int t = threadIdx.x; t < tot; t += 64) {
        const int Yl = ylo + t / nw, Xl = xlo + t % nw;
        const float fy = fmaxf(((float)Yl + 0.5f) / (float)s - 0.5f, 0.f), fx = fmaxf(((float)Xl + 0.5f) / (float)s - 0.5f, 0.f);
        const int yy0 = min((int)fy, K - 1), xx0 = min((int)fx, K - 1);
        const int yy1 = min(yy0 + 1, K - 1), xx1 = min(xx0 + 1, K - 1);
        const float ly = fy - (float)yy0, lx = fx - (float)xx0;
        float wgt = 0.f;
        if (yy0 == i && xx0 == j) wgt += (1.f - ly) * (1.f - lx);
        if (yy0 == i && xx1 == j) wgt += (1.f - ly) * lx;
        if (yy1 == i && xx0 == j) wgt += ly * (1.f - lx);
        if (yy1 == i && xx1 == j) wgt += ly * lx;
        if (wgt != 0.f) {
            const long o = (long)n * OH * OW + (long)(by * KS + Yl) * OW + bx * KS + Xl;
            float v = 0.f;
            for (int q = 0; q < chunks; ++q) v += dA[q * chunk_stride + o];
            acc += wgt * v;
        }
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) datt[(long)b * K * K + e] += acc;
}

inline int grid1d(long total, int cap = 8192) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

extern "C" int tdr_lr_blocks_fwd(const float* feat, int N, int C, int H, int W, int py, int px, int ky, int kx, float* blk,
                                 void* stream) {
    TDR_REQUIRE(feat && blk, "tdr_lr_blocks_fwd: null pointer");
    const long total = (long)N * py * px * C * (ky + 2) * (kx + 2);
    hipLaunchKernelGGL(lr_blocks_fwd_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, feat, C, H, W, py, px, ky,
                       kx, total, blk);
    TDR_LAUNCH_CHECK("lr_blocks_fwd");
    return TDR_OK;
}

extern "C" int tdr_lr_blocks_bwd(const float* dblk, int N, int C, int H, int W, int py, int px, int ky, int kx,
                                 float* dfeat, void* stream) {
    TDR_REQUIRE(dblk && dfeat, "tdr_lr_blocks_bwd: null pointer");
    const long total = (long)N * C * H * W;
    hipLaunchKernelGGL(lr_blocks_bwd_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, dblk, C, H, W, py, px, ky,
                       kx, total, dfeat);
    TDR_LAUNCH_CHECK("lr_blocks_bwd");
    return TDR_OK;
}

extern "C" int tdr_patch_inv_norm(const float* x, int B, int C, int H, int W, int OH, int OW, int dil, int pad, int step,
                                  int off, float* inv, void* stream) {
    TDR_REQUIRE(x && inv, "tdr_patch_inv_norm: null pointer");
    const long npos = (long)B * OH * OW;
    if (npos >= 1024)
        hipLaunchKernelGGL(patch_inv_norm_kernel, dim3((unsigned)((npos + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, x, C, H,
                           W, OH, OW, dil, step, off - pad, npos, inv);
    else
        hipLaunchKernelGGL(patch_inv_norm_wave_kernel, dim3((unsigned)((npos + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, C,
                           H, W, OH, OW, dil, step, off - pad, npos, inv);
    TDR_LAUNCH_CHECK("patch_inv_norm");
    return TDR_OK;
}

extern "C" int tdr_coarse_argmax_box(const float* dots, const float* invq, const float* invk, int ND, int N, int P, int Hr,
                                     int Wr, int diameter, int* index, int* y1, int* x1, void* stream) {
    TDR_REQUIRE(dots && invq && invk && index && y1 && x1, "tdr_coarse_argmax_box: null pointer");
    hipLaunchKernelGGL(coarse_argmax_box_kernel, dim3(N * P), dim3(256), 0, (hipStream_t)stream, dots, invq, invk, ND, N, P, Hr,
                       Wr, diameter, index, y1, x1);
    TDR_LAUNCH_CHECK("coarse_argmax_box");
    return TDR_OK;
}

extern "C" int tdr_gather_ref_block(const float* feat, int N, int C, int H, int W, const int* y1, const int* x1, int P,
                                    int side, int s, float* out, void* stream) {
    TDR_REQUIRE(feat && y1 && x1 && out, "tdr_gather_ref_block: null pointer");
    const long total = (long)N * P * C * side * s * side * s;
    hipLaunchKernelGGL(gather_ref_block_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, feat, C, H, W, y1, x1,
                       P, side, s, total, out);
    TDR_LAUNCH_CHECK("gather_ref_block");
    return TDR_OK;
}

extern "C" int tdr_scatter_ref_block(const float* dblk, int N, int C, int H, int W, const int* y1, const int* x1, int P,
                                     int side, float* dfeat, void* stream) {
    TDR_REQUIRE(dblk && y1 && x1 && dfeat, "tdr_scatter_ref_block: null pointer");
    const long total = (long)N * C * H * W;
    hipLaunchKernelGGL(scatter_ref_block_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, dblk, C, H, W, y1, x1,
                       P, side, total, dfeat);
    TDR_LAUNCH_CHECK("scatter_ref_block");
    return TDR_OK;
}

extern "C" int tdr_fine_argmax(const float* dots, const float* invq, const float* invk, int B, int P, int R, int* index_all,
                               float* soft_att, void* stream) {
    TDR_REQUIRE(dots && invq && invk && index_all && soft_att, "tdr_fine_argmax: null pointer");
    const long rows = (long)B * P;
    hipLaunchKernelGGL(fine_argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dots, invq, invk,
                       rows, P, R, index_all, soft_att);
    TDR_LAUNCH_CHECK("fine_argmax");
    return TDR_OK;
}

extern "C" int tdr_fine_search_bwd(const float* datt, const float* soft_att, const int* index_all, const float* lrb,
                                   const float* refb, const float* invq, const float* invk, int B, int C, int K, int D,
                                   float* dlrb, float* drefb, void* stream) {
    TDR_REQUIRE(datt && soft_att && index_all && lrb && refb && invq && invk && dlrb && drefb, "tdr_fine_search_bwd: null pointer");
    const size_t lds = (size_t)(8 * K * K + (K + 2) * (K + 2) + D * D) * sizeof(float);
    hipLaunchKernelGGL(fine_search_bwd_kernel, dim3(C, B), dim3(256), lds, (hipStream_t)stream, datt, soft_att, index_all, lrb,
                       refb, invq, invk, C, K, D, dlrb, drefb);
    TDR_LAUNCH_CHECK("fine_search_bwd");
    return TDR_OK;
}

extern "C" int tdr_transfer_fwd(const float* feat, int N, int C, int H, int W, const int* y1, const int* x1,
                                const int* index_all, const float* soft_att, int py, int px, int K, int side, int s,
                                float* out, int64_t out_ns, void* stream) {
    TDR_REQUIRE(feat && y1 && x1 && index_all && soft_att && out, "tdr_transfer_fwd: null pointer");
    const long opix = (long)py * K * s * px * K * s;
    hipLaunchKernelGGL(transfer_fwd_kernel, dim3((unsigned)((opix + 255) / 256), tdr_cdiv(C, TR_CG), N), dim3(256), 0,
                       (hipStream_t)stream, feat, C, H, W, y1, x1, index_all, soft_att, py, px, K, side, s, out, (long)out_ns);
    TDR_LAUNCH_CHECK("transfer_fwd");
    return TDR_OK;
}

// channel chunks of transfer_bwd: enough blocks to fill the chip (>= ~1024), at least 8 channels per block
static int tdr_transfer_chunks(int N, int C, int py, int px, int K, int s) {
    const long opix = (long)py * K * s * px * K * s;
    const long pix_blocks = ((opix + 255) / 256) * N;
    long chunks = (1024 + pix_blocks - 1) / pix_blocks;
    if (chunks > C / 8) chunks = C / 8;
    if (chunks < 1) chunks = 1;
    return (int)chunks;
}

// workspace (4-byte words): dA [chunks][N][opix] | absmax word (+ padding to 8 bytes) | 64-bit accumulators [N][C][H*W]
extern "C" int64_t tdr_transfer_ws_floats(int N, int C, int H, int W, int py, int px, int K, int s) {
    const int64_t opix = (int64_t)py * K * s * px * K * s;
    return (int64_t)tdr_transfer_chunks(N, C, py, px, K, s) * N * opix + 4 + 2 * (int64_t)N * C * H * W;
}

extern "C" int tdr_transfer_bwd(const float* dout, int64_t dout_ns, const float* feat, int N, int C, int H, int W,
                                const int* y1, const int* x1, const int* index_all, const float* soft_att, int py, int px,
                                int K, int side, int s, int deterministic, float* dfeat, float* datt, float* ws, void* stream) {
    TDR_REQUIRE(dout && feat && y1 && x1 && index_all && soft_att && dfeat && datt && ws, "tdr_transfer_bwd: null pointer");
    const long opix = (long)py * K * s * px * K * s;
    hipStream_t st = (hipStream_t)stream;
    const int chunks = tdr_transfer_chunks(N, C, py, px, K, s);
    const int cpb = tdr_cdiv(C, chunks);
    float* dA = ws;
    const dim3 grid((unsigned)((opix + 255) / 256), tdr_cdiv(C, cpb), N);
    if (!deterministic) {
        hipLaunchKernelGGL(transfer_bwd_kernel<false>, grid, dim3(256), 0, st, dout, (long)dout_ns, feat, C, H, W, y1, x1, index_all,
                           soft_att, py, px, K, side, s, cpb, N, (const unsigned*)nullptr, (unsigned long long*)nullptr, dfeat, dA);
    } else {
        TDR_REQUIRE(opix % 4 == 0 && dout_ns % 4 == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0, "tdr_transfer_bwd: dout must be 16-byte aligned");
        char* tail = reinterpret_cast<char*>(ws + (long)chunks * N * opix);
        tail += (8 - reinterpret_cast<uintptr_t>(tail) % 8) % 8;
        unsigned* amax = reinterpret_cast<unsigned*>(tail);
        unsigned long long* acc = reinterpret_cast<unsigned long long*>(tail + 8);
        const long nacc = (long)N * C * H * W;
        if (hipMemsetAsync(tail, 0, 8 + (size_t)nacc * 8, st) != hipSuccess) {
            tdr_set_error("tdr_transfer_bwd: hipMemsetAsync failed");
            return TDR_ERR_HIP;
        }
        const long per4 = (long)C * opix / 4;
        hipLaunchKernelGGL(absmax_bits_kernel, dim3(grid1d(per4, 512), N), dim3(256), 0, st, dout, (long)dout_ns, per4, N, amax);
        hipLaunchKernelGGL(transfer_bwd_kernel<true>, grid, dim3(256), 0, st, dout, (long)dout_ns, feat, C, H, W, y1, x1, index_all,
                           soft_att, py, px, K, side, s, cpb, N, amax, acc, (float*)nullptr, dA);
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3(grid1d(nacc, 4096)), dim3(256), 0, st, reinterpret_cast<const long long*>(acc),
                           nacc, amax, dfeat);
    }
    hipLaunchKernelGGL(transfer_datt_kernel, dim3(N * py * px, K * K), dim3(64), 0, st, ws, py, px, K, s, tdr_cdiv(C, cpb),
                       (long)N * opix, datt);
    TDR_LAUNCH_CHECK("transfer_bwd");
    return TDR_OK;
}

// bits of max |x| over x[n][0..per) (n < N, image stride x_ns), atomically max-ed into *slot (order-independent: the
// bit patterns of non-negative floats order like the values).  The caller zeroes the slot.  NaN / Inf inputs give a
// pattern >= 0x7f800000.  Serves the fp16-range survey of the train step (kernels.RangeSurvey).
__global__ __launch_bounds__(256) void absmax_bits_any_kernel(const float* __restrict__ x, long x_ns, long per, unsigned* __restrict__ slot) {
    const float* p = x + (long)blockIdx.y * x_ns;
    float m = 0.f;
    unsigned bad = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const float v = fabsf(p[i]);
        if (!(v <= 3.4028235e38f)) bad = 0x7fc00000u;        // NaN or Inf
        m = fmaxf(m, v);
    }
    unsigned b = __float_as_uint(m) | bad;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(slot, b);
}

extern "C" int tdr_absmax_bits(const float* x, int64_t x_ns, int N, int64_t per, unsigned* slot, void* stream) {
    TDR_REQUIRE(x && slot && N > 0 && per > 0, "tdr_absmax_bits: bad argument");
    hipLaunchKernelGGL(absmax_bits_any_kernel, dim3(grid1d(per, 256), N), dim3(256), 0, (hipStream_t)stream, x, (long)x_ns, (long)per, slot);
    TDR_LAUNCH_CHECK("absmax_bits");
    return TDR_OK;
}
