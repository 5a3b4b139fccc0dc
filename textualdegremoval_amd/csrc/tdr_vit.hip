// Frozen ViT window matcher (DINOv2 ViT-B/14) -- forward-only pieces that are not convolutions.
//   models/image_restoration_ref_model.py:215-247   window unfold, bilinear resize, token cosine, top-1 gather
//   models/dino/patch_embed.py:26-80                14x14 stride-14 projection  (patch gather here + 1x1 conv kernel)
//   models/dino/vision_transformers.py:209-236      cls token + position embedding
//   models/dino/attention.py:36-71                  softmax(q k^T * d^-1/2) v, all heads
// Activations are channel-major [B][D][T] so every Linear of the ViT is a 1x1 convolution on the matrix-core
// kernels and the token LayerNorm is the channel LayerNorm kernel (eps 1e-6 in both networks).
#include <stdlib.h>
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

// F.interpolate(mode='bilinear', align_corners=False): src = (dst + 0.5) * (in/out) - 0.5, clamped at 0
__global__ void resize_bilinear_kernel(const float* __restrict__ src, int Hs, int Ws, float* __restrict__ dst, int Hd, int Wd,
                                       long planes) {
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    const long total = planes * Hd * Wd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wd);
        const int y = (int)((i / Wd) % Hd);
        const long pl = i / ((long)Wd * Hd);
        const float fy = fmaxf(((float)y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)x + 0.5f) * sx - 0.5f, 0.f);
        const int y0 = min((int)fy, Hs - 1), x0 = min((int)fx, Ws - 1);
        const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* p = src + pl * Hs * Ws;
        const float top = p[y0 * Ws + x0] * (1.f - lx) + p[y0 * Ws + x1] * lx;
        const float bot = p[y1 * Ws + x0] * (1.f - lx) + p[y1 * Ws + x1] * lx;
        dst[i] = top * (1.f - ly) + bot * ly;
    }
}

// F.unfold(ref, (h,h), stride) -> windows [B*N][C][h][h], n = wy*nx + wx
__global__ void unfold_windows_kernel(const float* __restrict__ ref, int C, int Hr, int Wr, int h, int stride, int nx, int N,
                                      long total, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % h);
        long r = i / h;
        const int y = (int)(r % h); r /= h;
        const int c = (int)(r % C); r /= C;
        const int n = (int)(r % N);
        const long b = r / N;
        const int wy = n / nx, wx = n % nx;
        out[i] = ref[((b * C + c) * Hr + wy * stride + y) * Wr + wx * stride + x];
    }
}

// Token tensors are stored with a padded row length LD (a multiple of 32, >= 1+T) so that the 1x1-conv kernels can
// treat [B][D][LD] as a [LD/32] x 32 image; column 0 is the class token, columns 1..T the patches, the rest padding.
// x [B][Ci][H][W] -> out [B][Ci*p*p][LD]: row k = c*p*p + ky*p + kx, patch t = ty*cols + tx at column 1+t, zeros elsewhere
// flat: out [Ci*p*p][B*LD] instead (image b at columns b*LD .. b*LD + LD - 1), the batch-flattened token layout
__global__ void patchify_kernel(const float* __restrict__ x, int Ci, int H, int W, int p, int cols, int T, int LD, long total,
                                int B, int flat, float* __restrict__ out) {
    const int K = Ci * p * p;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % LD);
        long r = i / LD;
        const int k = (int)(r % K);
        const long b = r / K;
        float v = 0.f;
        if (col >= 1 && col <= T) {
            const int t = col - 1;
            const int c = k / (p * p), ky = (k / p) % p, kx = k % p;
            const int ty = t / cols, tx = t % cols;
            v = x[((b * Ci + c) * H + ty * p + ky) * W + tx * p + kx];
        }
        out[flat ? ((long)k * B + b) * LD + col : i] = v;
    }
}

// in place on tok [B][D][LD]: column 0 = cls + pos[:,0]; columns 1..T += pos; padding columns = 0   (pos is [D][1+T])
// (flat layout [D][B*LD]: every LD-column group is one image's tokens, the channel is i / (B*LD))
__global__ void vit_assemble_kernel(float* __restrict__ tok, const float* __restrict__ cls, const float* __restrict__ pos,
                                    int D, int T, int LD, long total, long ch_span) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % LD);
        const int d = (int)((i / ch_span) % D);
        float v = 0.f;
        if (col == 0) v = cls[d] + pos[(long)d * (T + 1)];
        else if (col <= T) v = tok[i] + pos[(long)d * (T + 1) + col];
        tok[i] = v;
    }
}

// softmax(q k^T * scale) v.  Workgroup = 4 waves = 4 x 32 queries of one (image, head); K/V tiles of 32 keys are
// staged once per workgroup.  The score tile is computed TRANSPOSED on the exact fp32 MFMA (S^T = K Q^T): a lane then
// owns one query column with 16 of the 32 keys in its registers, so the online-softmax max / sum are register
// reductions plus ONE cross-half exchange, and P^T is already in B-operand layout for O^T += V^T P^T when the
// contraction walks the keys in the accumulator's own row order (key(s, half) = (s&3) + 8(s>>2) + 4 half) -- no LDS
// round trip, no transposes.  Output columns (queries) are contiguous across lanes: coalesced stores.
// qkv [B][3C][LD] channel-major: q rows h*HD.., k rows C + h*HD.., v rows 2C + h*HD..;  out [B][C][LD].
// Separate q / k / v tensors (cross-attention: Tq queries, Tk keys; lse optional = m + log(l) per query, for the backward).
// Addressing is (batch offset, channel stride): element (b, c, t) of q lives at q + b*q_bs + c*qcs + t.  The per-image layout
// [B][C][LD] has q_bs = C*LD, qcs = LD; the batch-flattened layout [C][B*LD] of the CLIP encoder (all images' tokens along one
// pixel axis, so the Linears see B*LD-pixel GEMMs) has q_bs = LD, qcs = B*LD.
struct AttnArgs {
    const float* q; long q_bs; int LDq, Tq;           // [B][C][LDq]
    const float* k; const float* v; long kv_bs; int LDk, Tk;
    int C; float scale;
    float* out;                                        // element (b, c, t) at out + b*out_bs + c*qcs + t
    float* lse;                                        // [B][heads][LDq] or null
    long qcs, kcs;                                     // channel strides of q / out and of k / v
    long out_bs;
};

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    constexpr int NDT = (HD + 31) / 32;
    __shared__ float sK[HD][33], sV[HD][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32, h = blockIdx.y, b = blockIdx.z;
    const int C = a.C, T = a.Tk, LD = a.LDk, LDq = a.LDq;
    const float scale = a.scale;
    float* out = a.out;
    const float* Q = a.q + (long)b * a.q_bs + (long)h * HD * a.qcs;
    const float* Kp = a.k + (long)b * a.kv_bs + (long)h * HD * a.kcs;
    const float* Vp = a.v + (long)b * a.kv_bs + (long)h * HD * a.kcs;
    const bool qok = q0 + j < a.Tq;
    float qb[HD / 2];                                      // B operand of S^T: Q[q = j][d = 2s + kk] * scale
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) qb[s] = qok ? Q[(long)(2 * s + kk) * a.qcs + q0 + j] * scale : 0.f;
    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -1e30f, l = 0.f;

    for (int key0 = 0; key0 < T; key0 += 32) {
        __syncthreads();                                   // everyone is done with the previous K/V tile
        for (int e = tid; e < HD * 32; e += 256) {
            const int d = e >> 5, kx = e & 31;
            const bool kok = key0 + kx < T;
            const int kc = kok ? key0 + kx : T - 1;
            const float kv = Kp[(long)d * a.kcs + kc], vv = Vp[(long)d * a.kcs + kc];
            sK[d][kx] = kok ? kv : 0.f;
            sV[d][kx] = kok ? vv : 0.f;
        }
        __syncthreads();
        f32x16 st;                                         // S^T tile: rows = keys, column j = this lane's query
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[2 * s + kk][j], qb[s], st, 0, 0, 0);
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            st[r] = key < T ? st[r] : -1e30f;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));            // the other half of the wave holds the other 16 keys
        const float mnew = fmaxf(m, mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            st[r] = key < T ? __expf(st[r] - mnew) : 0.f;
            sum += st[r];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float alpha = __expf(m - mnew);
        l = l * alpha + sum;
        m = mnew;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            const int d = dt * 32 + j;
#pragma unroll
            for (int s = 0; s < 16; ++s) {                 // O^T[d][q] += V^T[d][key] P^T[key][q], key = row order of st
                const int kx = (s & 3) + 8 * (s >> 2) + 4 * kk;
                const float av = d < HD ? sV[d < HD ? d : 0][kx] : 0.f;
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, st[s], o[dt], 0, 0, 0);
            }
        }
    }
    const float inv = 1.f / l;
    if (q0 + j < LDq) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (d < HD) out[(long)b * a.out_bs + (long)(h * HD + d) * a.qcs + q0 + j] = qok ? o[dt][r] * inv : 0.f;
            }
        if (a.lse && kk == 0) a.lse[((long)b * gridDim.y + h) * LDq + q0 + j] = qok ? m + __logf(l) : 0.f;
    }
}

// The same attention on the 2-way fp16 split (TDR_MATH=hx2: every fp32 product as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16, fp32 accumulate and fp32 softmax) for the frozen, no-grad ViTs (DINOv2 matcher, CLIP image encoder):
// 12 + 12 f16 MFMAs of 32 cycles per 32-key tile instead of 32 + 32 fp32 MFMAs of 64 cycles.  Same skeleton: S^T = K Q^T with a
// lane owning one query column; P^T is split in registers and is already the B operand of O^T += V^T P^T, whose contraction
// walks the keys in the accumulator's row order -- slot (s, kk, i) of a 16-key step is key 16 s + 4 kk + (i & 3) + 8 (i >> 2), so
// the A operand is two 8-byte LDS reads.  K / V tiles are converted to hi / lo halves once per workgroup while they are staged.
typedef _Float16 vh8 __attribute__((ext_vector_type(8)));
typedef _Float16 vh4 __attribute__((ext_vector_type(4)));

// H1 = true: ONE product per operand pair (the hi planes only; lo planes are neither built nor multiplied) -- plain fp16 MFMA with fp32
// accumulation and softmax, a third of the matrix work.  Reduced precision (2^-11 per operand): only for the DINOv2 window matcher,
// whose sole output is an arg-max that tests/test_hip_dino.py pins bit-exactly against the split arithmetic (math code 3).
template <int HD, int KT, bool H1 = false>
__global__ __launch_bounds__(256) void attn_fwd_hx2_kernel(AttnArgs a) {
    constexpr int NDT = (HD + 31) / 32, KS = HD / 16, NOCT = HD / 8, NKB = KT / 32;
    constexpr int VP = KT + 8;                              // halves per d row of the V tile
    constexpr int KIT = (NOCT * KT + 255) / 256;            // K slots (8 d x 1 key, 16 bytes per plane) per thread and tile
    constexpr int VIT = HD * KT / 256;                      // V elements per thread and tile
    static_assert(HD % 16 == 0 && (KT == 32 || KT == 64), "head dim must be a multiple of 16, key tile 32 or 64");
    __shared__ __attribute__((aligned(16))) vh8 sK[2][NOCT][KT];          // [hi | lo][d octet][key]: one fragment per slot
    __shared__ __attribute__((aligned(16))) _Float16 sV[2][NDT * 32][VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32, h = blockIdx.y, b = blockIdx.z;
    const int C = a.C, T = a.Tk, LD = a.LDk, LDq = a.LDq;
    float* out = a.out;
    const float* Q = a.q + (long)b * a.q_bs + (long)h * HD * a.qcs;
    const float* Kp = a.k + (long)b * a.kv_bs + (long)h * HD * a.kcs;
    const float* Vp = a.v + (long)b * a.kv_bs + (long)h * HD * a.kcs;
    const bool qok = q0 + j < a.Tq;
    vh8 qh[KS], ql[KS];                                     // B operand of S^T: Q[q = j][d = 16 s + 8 kk + i] * scale, hi / lo
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = qok ? Q[(long)(16 * s + 8 * kk + i) * a.qcs + q0 + j] * a.scale : 0.f;
            asm volatile("" : "+v"(v));                     // one fp32 value for head and residual
            const _Float16 hh = (_Float16)v;
            qh[s][i] = hh;
            ql[s][i] = (_Float16)(v - (float)hh);
        }
    for (int e = tid; e < 2 * (NDT * 32 - HD) * VP; e += 256)   // d rows beyond the head dim (HD = 16, 80): zero once
        (&sV[0][0][0])[(e / ((NDT * 32 - HD) * VP)) * (NDT * 32 * VP) + HD * VP + e % ((NDT * 32 - HD) * VP)] = (_Float16)0.f;
    // K / V of the next tile are requested while the current one is multiplied (registers), converted and stored after it
    float rk[KIT][8], rv[VIT];
    auto load_tile = [&](int key0) {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int slot = tid + 256 * it, oc = slot / KT, kx = slot % KT;
            const bool ok = oc < NOCT && key0 + kx < T;
            const int kc = key0 + kx < T ? key0 + kx : T - 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) rk[it][i] = ok ? Kp[(long)(8 * (oc < NOCT ? oc : 0) + i) * a.kcs + kc] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e = tid + 256 * it, d = e / KT, kx = e % KT;
            const int kc = key0 + kx < T ? key0 + kx : T - 1;
            const float vv = Vp[(long)d * a.kcs + kc];
            rv[it] = key0 + kx < T ? vv : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int slot = tid + 256 * it, oc = slot / KT, kx = slot % KT;
            vh8 hi, lo;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = rk[it][i];
                asm volatile("" : "+v"(v));
                const _Float16 hh = (_Float16)v;
                hi[i] = hh;
                lo[i] = (_Float16)(v - (float)hh);
            }
            if (oc < NOCT) { sK[0][oc][kx] = hi; if constexpr (!H1) sK[1][oc][kx] = lo; }
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e = tid + 256 * it, d = e / KT, kx = e % KT;
            float v = rv[it];
            asm volatile("" : "+v"(v));
            const _Float16 hh = (_Float16)v;
            sV[0][d][kx] = hh;
            if constexpr (!H1) sV[1][d][kx] = (_Float16)(v - (float)hh);
        }
    };
    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -1e30f, l = 0.f;

    load_tile(0);
    for (int key0 = 0; key0 < T; key0 += KT) {
        __syncthreads();                                   // everyone is done with the previous K/V tile
        store_tile();
        __syncthreads();
        if (key0 + KT < T) load_tile(key0 + KT);
        f32x16 st[NKB];
        float mx = -1e30f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const vh8 kh = sK[0][2 * s + kk][kb * 32 + j];
                if constexpr (!H1) {
                    const vh8 kl = sK[1][2 * s + kk][kb * 32 + j];
                    st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], st[kb], 0, 0, 0);       // small cross terms first
                    st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], st[kb], 0, 0, 0);
                }
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], st[kb], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                st[kb][r] = key < T ? st[kb][r] : -1e30f;
                mx = fmaxf(mx, st[kb][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m, mx);
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                st[kb][r] = key < T ? __expf(st[kb][r] - mnew) : 0.f;
                sum += st[kb][r];
            }
        sum += __shfl_xor(sum, 32, 64);
        const float alpha = __expf(m - mnew);
        l = l * alpha + sum;
        m = mnew;
        vh8 ph[2 * NKB], pl[2 * NKB];                       // P^T in 16-key steps, in the accumulator's row order
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float pv = st[kb][8 * s + i];
                    asm volatile("" : "+v"(pv));
                    const _Float16 hh = (_Float16)pv;
                    ph[2 * kb + s][i] = hh;
                    pl[2 * kb + s][i] = (_Float16)(pv - (float)hh);
                }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            const int d = dt * 32 + j;
#pragma unroll
            for (int s = 0; s < 2 * NKB; ++s) {
                const int kb = (s >> 1) * 32 + 16 * (s & 1) + 4 * kk;
                vh8 vh;
                const vh4 a0 = *reinterpret_cast<const vh4*>(&sV[0][d][kb]), a1 = *reinterpret_cast<const vh4*>(&sV[0][d][kb + 8]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { vh[i] = a0[i]; vh[4 + i] = a1[i]; }
                if constexpr (!H1) {
                    vh8 vl;
                    const vh4 b0 = *reinterpret_cast<const vh4*>(&sV[1][d][kb]), b1 = *reinterpret_cast<const vh4*>(&sV[1][d][kb + 8]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { vl[i] = b0[i]; vl[4 + i] = b1[i]; }
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s], o[dt], 0, 0, 0);
                }
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s], o[dt], 0, 0, 0);
            }
        }
    }
    const float inv = 1.f / l;
    if (q0 + j < LDq) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (d < HD) out[(long)b * a.out_bs + (long)(h * HD + d) * a.qcs + q0 + j] = qok ? o[dt][r] * inv : 0.f;
            }
        if (a.lse && kk == 0) a.lse[((long)b * gridDim.y + h) * LDq + q0 + j] = qok ? m + __logf(l) : 0.f;
    }
}

// ---- the same skeleton in the DEFAULT arithmetic (TDR_MATH=bx3; round 6): q, k, v and P as three bf16 planes (h + m + l = the fp32 value
// to 24+ bits on fp32's exponent range), six v_mfma_f32_32x32x16_bf16 products per operand pair (lh hl mm mh hm hh, small cross terms first),
// fp32 accumulation and softmax -- the frozen ViTs' attention (DINOv2 window matcher, CLIP image encoder) ran on the exact fp32 MFMA kernel
// in this mode (157 TF peak: 1.37 ms per matcher launch at 84 TF); the 6-product ceiling is 417 TF.  Planes are packed pairwise by
// tdr_split3_bf16 (two values per dword); an A / B fragment is four dwords = 8 bf16.
typedef __bf16 vb8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ vb8 as_vb8(uint4 u) { return __builtin_bit_cast(vb8, u); }

template <int HD, int KT>
__global__ __launch_bounds__(256) void attn_fwd_bx3_kernel(AttnArgs a) {
    constexpr int NDT = (HD + 31) / 32, KS = HD / 16, NOCT = HD / 8, NKB = KT / 32;
    constexpr int VP = KT + 8;                              // 16-bit elements per d row of the V tile
    constexpr int KIT = (NOCT * KT + 255) / 256;            // K slots (8 d x 1 key, 16 bytes per plane) per thread and tile
    constexpr int VIT = HD * KT / 256;                      // V elements per thread and tile (even)
    static_assert(HD % 16 == 0 && (KT == 32 || KT == 64) && VIT % 2 == 0, "head dim a multiple of 16, key tile 32 or 64");
    __shared__ __attribute__((aligned(16))) uint4 sK[3][NOCT][KT];                 // [plane][d octet][key]: one fragment per slot
    __shared__ __attribute__((aligned(16))) unsigned short sV[3][NDT * 32][VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32, h = blockIdx.y, b = blockIdx.z;
    const int T = a.Tk, LDq = a.LDq;
    float* out = a.out;
    const float* Q = a.q + (long)b * a.q_bs + (long)h * HD * a.qcs;
    const float* Kp = a.k + (long)b * a.kv_bs + (long)h * HD * a.kcs;
    const float* Vp = a.v + (long)b * a.kv_bs + (long)h * HD * a.kcs;
    const bool qok = q0 + j < a.Tq;
    constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};          // (A plane, B plane): lh hl mm mh hm hh
    uint4 qp[3][KS];                                        // B operand of S^T: Q[q = j][d = 16 s + 8 kk + i] * scale, three planes
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        unsigned pl[3][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v0 = qok ? Q[(long)(16 * s + 8 * kk + 2 * i) * a.qcs + q0 + j] * a.scale : 0.f;
            const float v1 = qok ? Q[(long)(16 * s + 8 * kk + 2 * i + 1) * a.qcs + q0 + j] * a.scale : 0.f;
            tdr_split3_bf16(v0, v1, pl[0][i], pl[1][i], pl[2][i]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) qp[p][s] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    }
    for (int e = tid; e < 3 * (NDT * 32 - HD) * VP; e += 256)   // d rows beyond the head dim (HD = 16, 80): zero once
        (&sV[0][0][0])[(e / ((NDT * 32 - HD) * VP)) * (NDT * 32 * VP) + HD * VP + e % ((NDT * 32 - HD) * VP)] = 0;
    float rk[KIT][8], rv[VIT];
    auto load_tile = [&](int key0) {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int slot = tid + 256 * it, oc = slot / KT, kx = slot % KT;
            const bool ok = oc < NOCT && key0 + kx < T;
            const int kc = key0 + kx < T ? key0 + kx : T - 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) rk[it][i] = ok ? Kp[(long)(8 * (oc < NOCT ? oc : 0) + i) * a.kcs + kc] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e = tid + 256 * it, d = e / KT, kx = e % KT;
            const int kc = key0 + kx < T ? key0 + kx : T - 1;
            const float vv = Vp[(long)d * a.kcs + kc];
            rv[it] = key0 + kx < T ? vv : 0.f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int slot = tid + 256 * it, oc = slot / KT, kx = slot % KT;
            unsigned pl[3][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) tdr_split3_bf16(rk[it][2 * i], rk[it][2 * i + 1], pl[0][i], pl[1][i], pl[2][i]);
            if (oc < NOCT) {
#pragma unroll
                for (int p = 0; p < 3; ++p) sK[p][oc][kx] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
            }
        }
#pragma unroll
        for (int it = 0; it < VIT; it += 2) {
            const int e0 = tid + 256 * it, e1 = tid + 256 * (it + 1);
            unsigned pl[3];
            tdr_split3_bf16(rv[it], rv[it + 1], pl[0], pl[1], pl[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                sV[p][e0 / KT][e0 % KT] = (unsigned short)(pl[p] & 0xffffu);
                sV[p][e1 / KT][e1 % KT] = (unsigned short)(pl[p] >> 16);
            }
        }
    };
    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -1e30f, l = 0.f;

    load_tile(0);
    for (int key0 = 0; key0 < T; key0 += KT) {
        __syncthreads();                                   // everyone is done with the previous K/V tile
        store_tile();
        __syncthreads();
        if (key0 + KT < T) load_tile(key0 + KT);
        f32x16 st[NKB];
        float mx = -1e30f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                uint4 kf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) kf[p] = sK[p][2 * s + kk][kb * 32 + j];
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_vb8(kf[SA[q]]), as_vb8(qp[SB[q]][s]), st[kb], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                st[kb][r] = key < T ? st[kb][r] : -1e30f;
                mx = fmaxf(mx, st[kb][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m, mx);
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                st[kb][r] = key < T ? __expf(st[kb][r] - mnew) : 0.f;
                sum += st[kb][r];
            }
        sum += __shfl_xor(sum, 32, 64);
        const float alpha = __expf(m - mnew);
        l = l * alpha + sum;
        m = mnew;
        uint4 pp[3][2 * NKB];                               // P^T in 16-key steps, in the accumulator's row order, three planes
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned pl[3][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) tdr_split3_bf16(st[kb][8 * s + 2 * i], st[kb][8 * s + 2 * i + 1], pl[0][i], pl[1][i], pl[2][i]);
#pragma unroll
                for (int p = 0; p < 3; ++p) pp[p][2 * kb + s] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
            }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            const int d = dt * 32 + j;
#pragma unroll
            for (int s = 0; s < 2 * NKB; ++s) {
                const int kb = (s >> 1) * 32 + 16 * (s & 1) + 4 * kk;
                uint4 vf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const uint2 a0 = *reinterpret_cast<const uint2*>(&sV[p][d][kb]), a1 = *reinterpret_cast<const uint2*>(&sV[p][d][kb + 8]);
                    vf[p] = make_uint4(a0.x, a0.y, a1.x, a1.y);
                }
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_vb8(vf[SA[q]]), as_vb8(pp[SB[q]][s]), o[dt], 0, 0, 0);
            }
        }
    }
    const float inv = 1.f / l;
    if (q0 + j < LDq) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (d < HD) out[(long)b * a.out_bs + (long)(h * HD + d) * a.qcs + q0 + j] = qok ? o[dt][r] * inv : 0.f;
            }
        if (a.lse && kk == 0) a.lse[((long)b * gridDim.y + h) * LDq + q0 + j] = qok ? m + __logf(l) : 0.f;
    }
}

// ---- backward (injected cross-attention of the stage-A trainers, main_train_i2t_mapping.py:197-233) ----------------
// Two deterministic passes instead of one with atomics:
//   attn_bwd_dq : one wave per 32 queries, walks the key tiles (same skeleton as the forward: S^T, then dP^T = V dO^T
//                 with the same operand roles, dS^T = P^T (dP^T - D), dQ^T += K^T dS^T in the accumulator's row order)
//   attn_bwd_dkv: one wave per 32 keys, walks the query tiles with the roles of q and k swapped
//                 (S = Q K^T rows = queries; dV^T += dO^T P, dK^T += Q^T dS)
// D[q] = sum_d dO[d][q] O[d][q] comes from attn_rowdot_kernel.
struct AttnBwdArgs {
    const float* q; long q_bs; int LDq, Tq;
    const float* k; const float* v; long kv_bs; int LDk, Tk;
    const float* dout;                                 // [B][C][LDq]
    const float* lse; const float* D;                  // [B][heads][LDq]
    int C; float scale;
    float* dq;                                         // [B][C][LDq]
    float* dk; float* dv; long dkv_bs;                 // [B][C][LDk]
};

__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* __restrict__ dout, const float* __restrict__ o, int C,
                                                         int hd, int LD, int T, float* __restrict__ D) {
    const int t = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    if (t >= LD) return;
    float s = 0.f;
    if (t < T) {
        const long base = ((long)b * C + (long)h * hd) * LD + t;
        for (int d = 0; d < hd; ++d) s += dout[base + (long)d * LD] * o[base + (long)d * LD];
    }
    D[((long)b * gridDim.y + h) * LD + t] = s;
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs a) {
    constexpr int NDT = (HD + 31) / 32;
    __shared__ float sK[HD][33], sV[HD][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    const int q0 = blockIdx.x * 128 + wave * 32, h = blockIdx.y, b = blockIdx.z;
    const int C = a.C, T = a.Tk, LD = a.LDk, LDq = a.LDq;
    const float* Q = a.q + (long)b * a.q_bs + (long)h * HD * LDq;
    const float* dO = a.dout + ((long)b * C + (long)h * HD) * LDq;
    const float* Kp = a.k + (long)b * a.kv_bs + (long)h * HD * LD;
    const float* Vp = a.v + (long)b * a.kv_bs + (long)h * HD * LD;
    const bool qok = q0 + j < a.Tq;
    const int qc = qok ? q0 + j : 0;
    float qb[HD / 2], dob[HD / 2];
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) {
        qb[s] = qok ? Q[(long)(2 * s + kk) * LDq + qc] * a.scale : 0.f;
        dob[s] = qok ? dO[(long)(2 * s + kk) * LDq + qc] : 0.f;
    }
    const float lse = a.lse[((long)b * gridDim.y + h) * LDq + qc], Dq = a.D[((long)b * gridDim.y + h) * LDq + qc];
    f32x16 acc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    for (int key0 = 0; key0 < T; key0 += 32) {
        __syncthreads();
        for (int e = tid; e < HD * 32; e += 256) {
            const int d = e >> 5, kx = e & 31;
            const bool kok = key0 + kx < T;
            const int kc = kok ? key0 + kx : T - 1;
            const float kv = Kp[(long)d * LD + kc], vv = Vp[(long)d * LD + kc];
            sK[d][kx] = kok ? kv : 0.f;
            sV[d][kx] = kok ? vv : 0.f;
        }
        __syncthreads();
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < HD / 2; ++s) {
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[2 * s + kk][j], qb[s], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[2 * s + kk][j], dob[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float p = (key < T && qok) ? __expf(st[r] - lse) : 0.f;
            st[r] = p * (dp[r] - Dq) * a.scale;                               // dS^T * scale
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int d = dt * 32 + j;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kx = (s & 3) + 8 * (s >> 2) + 4 * kk;
                const float av = d < HD ? sK[d < HD ? d : 0][kx] : 0.f;
                acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, st[s], acc[dt], 0, 0, 0);
            }
        }
    }
    if (q0 + j < LDq) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (d < HD) a.dq[((long)b * C + h * HD + d) * LDq + q0 + j] = qok ? acc[dt][r] : 0.f;
            }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
    constexpr int NDT = (HD + 31) / 32;
    __shared__ float sQ[HD][33], sO[HD][33], sL[32], sD[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    const int k0 = blockIdx.x * 128 + wave * 32, h = blockIdx.y, b = blockIdx.z;
    const int C = a.C, Tk = a.Tk, LDk = a.LDk, LDq = a.LDq, Tq = a.Tq;
    const float* Q = a.q + (long)b * a.q_bs + (long)h * HD * LDq;
    const float* dO = a.dout + ((long)b * C + (long)h * HD) * LDq;
    const float* Kp = a.k + (long)b * a.kv_bs + (long)h * HD * LDk;
    const float* Vp = a.v + (long)b * a.kv_bs + (long)h * HD * LDk;
    const float* lsep = a.lse + ((long)b * gridDim.y + h) * LDq;
    const float* Dp = a.D + ((long)b * gridDim.y + h) * LDq;
    const bool kok = k0 + j < Tk;
    const int kc = kok ? k0 + j : 0;
    float kb[HD / 2], vb[HD / 2];
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) {
        kb[s] = kok ? Kp[(long)(2 * s + kk) * LDk + kc] * a.scale : 0.f;
        vb[s] = kok ? Vp[(long)(2 * s + kk) * LDk + kc] : 0.f;
    }
    f32x16 ak[NDT], av_[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[dt][r] = 0.f; av_[dt][r] = 0.f; }
    for (int qt = 0; qt < Tq; qt += 32) {
        __syncthreads();
        for (int e = tid; e < HD * 32; e += 256) {
            const int d = e >> 5, qx = e & 31;
            const bool ok = qt + qx < Tq;
            const int qc = ok ? qt + qx : Tq - 1;
            const float qv = Q[(long)d * LDq + qc], ov = dO[(long)d * LDq + qc];
            sQ[d][qx] = ok ? qv : 0.f;
            sO[d][qx] = ok ? ov : 0.f;
        }
        if (tid < 32) {
            const bool ok = qt + tid < Tq;
            sL[tid] = ok ? lsep[qt + tid] : 0.f;
            sD[tid] = ok ? Dp[qt + tid] : 0.f;
        }
        __syncthreads();
        f32x16 st, dp;                                     // S tile: rows = queries, column j = this lane's key
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < HD / 2; ++s) {
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[2 * s + kk][j], kb[s], st, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sO[2 * s + kk][j], vb[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qx = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float p = (qt + qx < Tq && kok) ? __expf(st[r] - sL[qx]) : 0.f;
            st[r] = p;                                     // P
            dp[r] = p * (dp[r] - sD[qx]) * a.scale;        // dS * scale
        }
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int d = dt * 32 + j;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int qx = (s & 3) + 8 * (s >> 2) + 4 * kk;
                const float ao = d < HD ? sO[d < HD ? d : 0][qx] : 0.f;
                const float aq = d < HD ? sQ[d < HD ? d : 0][qx] : 0.f;
                av_[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, st[s], av_[dt], 0, 0, 0);     // dV^T += dO^T P
                ak[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq, dp[s], ak[dt], 0, 0, 0);       // dK^T += Q^T dS
            }
        }
    }
    if (k0 + j < LDk) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (d < HD) {
                    const long o = (long)b * a.dkv_bs + ((long)h * HD + d) * LDk + k0 + j;
                    a.dk[o] = kok ? ak[dt][r] : 0.f;
                    a.dv[o] = kok ? av_[dt][r] : 0.f;
                }
            }
    }
}

// corr[b][n] = <L_b, R_{b,n}> / (max(|L_b|,1e-12) max(|R_{b,n}|,1e-12)) over the patch tokens (columns 1..T of [D][1+T])
// Two stages: CORR_SPLIT workgroups per (image, window), each over a band of the D rows, write (dot, |L|^2, |R|^2) partials; the
// finish sums them in a fixed order.  (One workgroup per pair was 16 workgroups streaming 8.4 MB each: 1.9 ms per step.)
constexpr int CORR_SPLIT = 48;
__global__ __launch_bounds__(256) void token_corr_kernel(const float* __restrict__ fl, const float* __restrict__ fr, int D, int T1,
                                                        int LD, int N, float* __restrict__ part) {
    __shared__ float red[3][4];
    const int n = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    const int rows = (D + CORR_SPLIT - 1) / CORR_SPLIT, d0 = sp * rows, d1 = min(D, d0 + rows);
    const float* L = fl + (long)b * D * LD;
    const float* R = fr + ((long)b * N + n) * D * LD;
    float dot = 0.f, nl = 0.f, nr = 0.f;
    for (long i = (long)d0 * LD + threadIdx.x; i < (long)d1 * LD; i += 256) {
        const int col = (int)(i % LD);
        if (col == 0 || col >= T1) continue;         // class token column, padding
        const float a = L[i], c = R[i];
        dot += a * c; nl += a * a; nr += c * c;
    }
    dot = wave_sum(dot); nl = wave_sum(nl); nr = wave_sum(nr);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = dot; red[1][threadIdx.x >> 6] = nl; red[2][threadIdx.x >> 6] = nr; }
    __syncthreads();
    if (threadIdx.x < 3)
        part[(((long)b * N + n) * CORR_SPLIT + sp) * 3 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ void token_corr_finish_kernel(const float* __restrict__ part, int pairs, float* __restrict__ corr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs) return;
    float d = 0.f, a = 0.f, c = 0.f;
    for (int sp = 0; sp < CORR_SPLIT; ++sp) {
        d += part[((long)i * CORR_SPLIT + sp) * 3];
        a += part[((long)i * CORR_SPLIT + sp) * 3 + 1];
        c += part[((long)i * CORR_SPLIT + sp) * 3 + 2];
    }
    corr[i] = d / (fmaxf(sqrtf(a), 1e-12f) * fmaxf(sqrtf(c), 1e-12f));
}

// index[b] = argmax_n corr[b][n] (first maximum, like torch.topk); out[b] = windows[b][index[b]]
__global__ __launch_bounds__(256) void select_window_kernel(const float* __restrict__ corr, const float* __restrict__ win, int N,
                                                           long per, int* __restrict__ index, float* __restrict__ out) {
    __shared__ int sbest;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        int best = 0;
        float bv = corr[(long)b * N];
        for (int n = 1; n < N; ++n) {
            const float v = corr[(long)b * N + n];
            if (v > bv) { bv = v; best = n; }
        }
        sbest = best;
        if (blockIdx.x == 0) index[b] = best;
    }
    __syncthreads();
    const float* src = win + ((long)b * N + sbest) * per;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += (long)gridDim.x * 256) out[(long)b * per + i] = src[i];
}

inline int vgrid(long total, int cap = 16384) {
    long b = (total + 255) / 256;
    return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

extern "C" int tdr_resize_bilinear(const float* src, int planes, int Hs, int Ws, float* dst, int Hd, int Wd, void* stream) {
    TDR_REQUIRE(src && dst && planes > 0, "tdr_resize_bilinear: bad argument");
    const long total = (long)planes * Hd * Wd;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(vgrid(total)), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, dst, Hd, Wd,
                       (long)planes);
    TDR_LAUNCH_CHECK("resize_bilinear");
    return TDR_OK;
}

extern "C" int tdr_unfold_windows(const float* ref, int B, int C, int Hr, int Wr, int h, int stride, float* out, void* stream) {
    TDR_REQUIRE(ref && out && h <= Hr && h <= Wr && stride > 0, "tdr_unfold_windows: bad argument");
    const int ny = (Hr - h) / stride + 1, nx = (Wr - h) / stride + 1;
    const long total = (long)B * ny * nx * C * h * h;
    hipLaunchKernelGGL(unfold_windows_kernel, dim3(vgrid(total)), dim3(256), 0, (hipStream_t)stream, ref, C, Hr, Wr, h, stride, nx,
                       ny * nx, total, out);
    TDR_LAUNCH_CHECK("unfold_windows");
    return TDR_OK;
}

extern "C" int tdr_patchify(const float* x, int B, int Ci, int H, int W, int p, int LD, int flat, float* out, void* stream) {
    TDR_REQUIRE(x && out && H % p == 0 && W % p == 0, "tdr_patchify: H, W must be multiples of the patch size");
    const int cols = W / p, T = (H / p) * cols;
    TDR_REQUIRE(LD >= T + 1, "tdr_patchify: LD %d < 1 + T", LD);
    const long total = (long)B * Ci * p * p * LD;
    hipLaunchKernelGGL(patchify_kernel, dim3(vgrid(total)), dim3(256), 0, (hipStream_t)stream, x, Ci, H, W, p, cols, T, LD, total, B, flat, out);
    TDR_LAUNCH_CHECK("patchify");
    return TDR_OK;
}

extern "C" int tdr_vit_assemble(float* tok, const float* cls, const float* pos, int B, int D, int T, int LD, int flat, void* stream) {
    TDR_REQUIRE(tok && cls && pos && LD >= T + 1, "tdr_vit_assemble: bad argument");
    const long total = (long)B * D * LD;
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(vgrid(total)), dim3(256), 0, (hipStream_t)stream, tok, cls, pos, D, T, LD, total,
                       flat ? (long)B * LD : (long)LD);
    TDR_LAUNCH_CHECK("vit_assemble");
    return TDR_OK;
}

static int attn_fwd_launch(const AttnArgs& a, int B, int heads, hipStream_t st, int math = 0) {
    const int hd = a.C / heads;
    dim3 grid(tdr_cdiv(a.LDq, 128), heads, B);
    if (math == 3 && (hd == 64 || hd == 32 || hd == 16)) {          // plain fp16 (DINOv2 matcher only)
        if (hd == 64) hipLaunchKernelGGL((attn_fwd_hx2_kernel<64, 32, true>), grid, dim3(256), 0, st, a);
        else if (hd == 32) hipLaunchKernelGGL((attn_fwd_hx2_kernel<32, 32, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_hx2_kernel<16, 32, true>), grid, dim3(256), 0, st, a);
        TDR_LAUNCH_CHECK("attention_fwd_h1");
        return TDR_OK;
    }
    if (math == 1 && (hd == 80 || hd == 64 || hd == 32 || hd == 16)) {      // 3-way bf16 split: the default arithmetic (TDR_MATH=bx3)
        if (hd == 80) hipLaunchKernelGGL((attn_fwd_bx3_kernel<80, 32>), grid, dim3(256), 0, st, a);
        else if (hd == 64) hipLaunchKernelGGL((attn_fwd_bx3_kernel<64, 32>), grid, dim3(256), 0, st, a);     // (64-key tiles: 919 vs 740 us at the matcher's shape)
        else if (hd == 32) hipLaunchKernelGGL((attn_fwd_bx3_kernel<32, 32>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_bx3_kernel<16, 32>), grid, dim3(256), 0, st, a);
        TDR_LAUNCH_CHECK("attention_fwd_bx3");
        return TDR_OK;
    }
    if ((math == 2 || math == 3) && (hd == 80 || hd == 64 || hd == 32 || hd == 16)) {
        static const int kt = tdr_tune_env("TDR_ATTN_KT") ? atoi(tdr_tune_env("TDR_ATTN_KT")) : 32;       // key tile (tuning aid)
        if (hd == 80) hipLaunchKernelGGL((attn_fwd_hx2_kernel<80, 32>), grid, dim3(256), 0, st, a);
        else if (hd == 64 && kt == 64) hipLaunchKernelGGL((attn_fwd_hx2_kernel<64, 64>), grid, dim3(256), 0, st, a);
        else if (hd == 64) hipLaunchKernelGGL((attn_fwd_hx2_kernel<64, 32>), grid, dim3(256), 0, st, a);
        else if (hd == 32) hipLaunchKernelGGL((attn_fwd_hx2_kernel<32, 32>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_hx2_kernel<16, 32>), grid, dim3(256), 0, st, a);
        TDR_LAUNCH_CHECK("attention_fwd_hx2");
        return TDR_OK;
    }
    if (hd == 80) hipLaunchKernelGGL(attn_fwd_kernel<80>, grid, dim3(256), 0, st, a);
    else if (hd == 64) hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), 0, st, a);
    else if (hd == 32) hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(256), 0, st, a);
    else if (hd == 16) hipLaunchKernelGGL(attn_fwd_kernel<16>, grid, dim3(256), 0, st, a);
    else { tdr_set_error("attention: head dim %d not supported (16, 32, 64, 80)", hd); return TDR_ERR_UNSUPPORTED; }
    TDR_LAUNCH_CHECK("attention_fwd");
    return TDR_OK;
}

extern "C" int tdr_attention_fwd(const float* qkv, int B, int C, int heads, int T, int LD, float scale, float* out, void* stream) {
    return tdr_attention_fwd_math(qkv, B, C, heads, T, LD, scale, 0, 0, out, stream);
}

// math 0: exact fp32 MFMA; 1: 3-way bf16 split (24-bit operands, fp32 range: the default arithmetic); 2: 2-way fp16 split (operands within
// the fp16 range: LayerNorm-ed ViT activations);
// 3: plain fp16, one product (reduced precision: the DINOv2 matcher, whose only output is an arg-max pinned by its tests)
extern "C" int tdr_attention_fwd_math(const float* qkv, int B, int C, int heads, int T, int LD, float scale, int math, int flat,
                                      float* out, void* stream) {
    TDR_REQUIRE(qkv && out && heads > 0 && C % heads == 0 && LD >= T && (math >= 0 && math <= 3), "tdr_attention_fwd: bad argument");
    AttnArgs a{qkv, 3L * C * LD, LD, T, qkv + (long)C * LD, qkv + 2L * C * LD, 3L * C * LD, LD, T, C, scale, out, nullptr, LD, LD, (long)C * LD};
    if (flat) {      // [3C][B*LD]: image b at column offset b*LD, channel stride B*LD; out [C][B*LD]
        const long cs = (long)B * LD;
        a = AttnArgs{qkv, LD, LD, T, qkv + (long)C * cs, qkv + 2L * C * cs, LD, LD, T, C, scale, out, nullptr, cs, cs, LD};
    }
    return attn_fwd_launch(a, B, heads, (hipStream_t)stream, math);
}

extern "C" int tdr_cross_attention_fwd(const float* q, const float* k, const float* v, int B, int C, int heads, int Tq, int LDq,
                                       int Tk, int LDk, float scale, float* out, float* lse, void* stream) {
    TDR_REQUIRE(q && k && v && out && heads > 0 && C % heads == 0 && LDq >= Tq && LDk >= Tk && Tq > 0 && Tk > 0,
                "tdr_cross_attention_fwd: bad argument");
    AttnArgs a{q, (long)C * LDq, LDq, Tq, k, v, (long)C * LDk, LDk, Tk, C, scale, out, lse, LDq, LDk, (long)C * LDq};
    return attn_fwd_launch(a, B, heads, (hipStream_t)stream);
}

extern "C" int tdr_cross_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                       const float* lse, int B, int C, int heads, int Tq, int LDq, int Tk, int LDk, float scale,
                                       float* dq, float* dk, float* dv, float* ws, void* stream) {
    TDR_REQUIRE(q && k && v && out && dout && lse && dk && dv && ws, "tdr_cross_attention_bwd: null pointer (ws: B*heads*LDq floats)");
    TDR_REQUIRE(heads > 0 && C % heads == 0 && LDq >= Tq && LDk >= Tk && Tq > 0 && Tk > 0, "tdr_cross_attention_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int hd = C / heads;
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3(tdr_cdiv(LDq, 256), heads, B), dim3(256), 0, st, dout, out, C, hd, LDq, Tq, ws);
    AttnBwdArgs a{q, (long)C * LDq, LDq, Tq, k, v, (long)C * LDk, LDk, Tk, dout, lse, ws, C, scale, dq, dk, dv, (long)C * LDk};
    dim3 gq(tdr_cdiv(LDq, 128), heads, B), gk(tdr_cdiv(LDk, 128), heads, B);
#define ATTN_BWD(H)                                                                            \
    do {                                                                                       \
        if (dq) hipLaunchKernelGGL(attn_bwd_dq_kernel<H>, gq, dim3(256), 0, st, a);            \
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<H>, gk, dim3(256), 0, st, a);                   \
    } while (0)
    if (hd == 80) ATTN_BWD(80);
    else if (hd == 64) ATTN_BWD(64);
    else if (hd == 32) ATTN_BWD(32);
    else if (hd == 16) ATTN_BWD(16);
    else { tdr_set_error("tdr_cross_attention_bwd: head dim %d not supported (16, 32, 64, 80)", hd); return TDR_ERR_UNSUPPORTED; }
#undef ATTN_BWD
    TDR_LAUNCH_CHECK("cross_attention_bwd");
    return TDR_OK;
}

extern "C" int tdr_token_match(const float* fl, const float* fr, const float* windows, int B, int N, int D, int T1, int LD, int64_t per,
                               float* corr, int* index, float* ref_in, void* stream) {
    TDR_REQUIRE(fl && fr && windows && corr && index && ref_in && N > 0, "tdr_token_match: bad argument");
    hipStream_t st = (hipStream_t)stream;
    // the partials live at the start of ref_in until select_window_kernel overwrites it (stream order): B*N*48*3 floats << B*per
    TDR_REQUIRE((int64_t)B * per >= (int64_t)B * N * CORR_SPLIT * 3, "tdr_token_match: ref_in too small for the correlation partials");
    hipLaunchKernelGGL(token_corr_kernel, dim3(N, B, CORR_SPLIT), dim3(256), 0, st, fl, fr, D, T1, LD, N, ref_in);
    hipLaunchKernelGGL(token_corr_finish_kernel, dim3(tdr_cdiv(B * N, 64)), dim3(64), 0, st, ref_in, B * N, corr);
    hipLaunchKernelGGL(select_window_kernel, dim3(vgrid(per, 256), B), dim3(256), 0, st, corr, windows, N, (long)per, index, ref_in);
    TDR_LAUNCH_CHECK("token_match");
    return TDR_OK;
}
