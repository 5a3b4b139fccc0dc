// Token-major fp16 pipeline of the frozen DINOv2 ViT-B/14 window matcher (forward only)
//   models/image_restoration_ref_model.py:215-247 (get_ref_in: the matcher's only output is an arg-max over window similarities),
//   models/dino/vision_transformers.py (blocks: norm1 -> qkv -> attention -> proj * ls1 + x -> norm2 -> fc1 -> GELU -> fc2 * ls2 + x).
//
// The channel-major engines keep fp32 activations [D][tokens] and convert / transpose them into MFMA fragments inside every GEMM
// workgroup, once per output-channel tile (24 times for fc1): the staging, not the matrix pipe, bounded the matcher's Linears at
// ~275 TFLOP/s, and the attention re-read fp32 K / V from L2 once per 128 queries (3.5 TB/s of L2 traffic).  Here:
//   * the residual stream stays fp32, token-major [P][D] (P = images x padded tokens, D contiguous);
//   * every GEMM operand is fp16, token-major, rounded ONCE by its producer (LayerNorm, GEMM epilogue, attention) -- the same
//     single rounding the 'h1' arithmetic applied at each GEMM input, so the numbers agree with it up to summation order;
//   * both MFMA operands are K-contiguous in memory: 16-byte global loads -> 16-byte LDS stores -> ds_read_b128 fragments,
//     no VALU work in the main loop; fp32 accumulation; the C tile goes through LDS once for full-line stores.
// tok_gemm_kernel   : Y[P][N] = X[P][K] W[N][K]^T, 128 x 128 x 64 tiles, 4 waves x (64 x 64), double-buffered LDS, 2 workgroups / CU
//                     epilogues: +bias -> fp16 | +bias, erf-GELU -> fp16 | residual += ls * (acc + bias) (fp32, in place)
// tok_attn_kernel   : flash attention over fp16 q / k / v slices of the qkv rows; 256 queries per workgroup (64 per wave), 64-key
//                     tiles; K staged as is, V transposed on its way into LDS; softmax in fp32 on exp2 with the scale folded in
// tok_ln_kernel     : nn.LayerNorm over D of a token row (one wave per token), fp16 or fp32 output
// transpose kernel  : [R][C] <-> [C][R] fp32 (entering / leaving the channel-major world of the patch embedding and of tdr_token_match)
#include "tdr_common.h"
#include "tdr_erf.h"
#include "../../include/tdr.h"

typedef _Float16 vh8 __attribute__((ext_vector_type(8)));
typedef _Float16 vh4 __attribute__((ext_vector_type(4)));

// ---- transpose -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int R, int C, float* __restrict__ dst) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    src += (long)blockIdx.z * R * C;
    dst += (long)blockIdx.z * R * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        t[ty + 8 * i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) dst[(long)c * R + r] = t[tx][ty + 8 * i];
    }
}

extern "C" int tdr_transpose_f32(const float* src, int batch, int R, int C, float* dst, void* stream) {
    TDR_REQUIRE(src && dst && batch > 0 && R > 0 && C > 0, "tdr_transpose_f32: bad argument");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3(tdr_cdiv(C, 32), tdr_cdiv(R, 32), batch), dim3(256), 0, (hipStream_t)stream, src, R, C, dst);
    TDR_LAUNCH_CHECK("transpose_f32");
    return TDR_OK;
}

// ---- LayerNorm over the row of a token ---------------------------------------------------------------------------------
template <bool OUT16>
__global__ __launch_bounds__(256) void tok_ln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                    int D, long P, float eps, void* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= P) return;
    const int n4 = D >> 2;
    const float4* row = reinterpret_cast<const float4*>(x + tok * D);
    float4 v[4];                                            // D <= 1024
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n4 ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < n4) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        if (c >= n4) continue;
        const float4 g = reinterpret_cast<const float4*>(w)[c], h = reinterpret_cast<const float4*>(b)[c];
        const float y0 = (v[i].x - mean) * rstd * g.x + h.x, y1 = (v[i].y - mean) * rstd * g.y + h.y;
        const float y2 = (v[i].z - mean) * rstd * g.z + h.z, y3 = (v[i].w - mean) * rstd * g.w + h.w;
        if constexpr (OUT16) {
            vh4 o = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
            reinterpret_cast<vh4*>(reinterpret_cast<_Float16*>(out) + tok * D)[c] = o;
        } else {
            reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + tok * D)[c] = make_float4(y0, y1, y2, y3);
        }
    }
}

extern "C" int tdr_tok_layernorm(const float* x, const float* w, const float* b, int64_t P, int D, float eps, int out_f16, void* out,
                                 void* stream) {
    TDR_REQUIRE(x && w && b && out && P > 0, "tdr_tok_layernorm: bad argument");
    TDR_REQUIRE(D % 4 == 0 && D <= 1024, "tdr_tok_layernorm: D must be a multiple of 4, at most 1024 (got %d)", D);
    const dim3 grid(tdr_cdiv(P, 4));
    if (out_f16) hipLaunchKernelGGL(tok_ln_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, D, (long)P, eps, out);
    else hipLaunchKernelGGL(tok_ln_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, D, (long)P, eps, out);
    TDR_LAUNCH_CHECK("tok_layernorm");
    return TDR_OK;
}

// ---- fp16 GEMM ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int OS = BK + 8;                 // halves per operand row in LDS: 144 B -> the 16 lanes of a ds_read_b128 pass hit 64 distinct banks
constexpr int CS = BN + 8;                 // floats per C row in LDS
constexpr int GEMM_LDS = 2 * (BM + BN) * OS * 2;            // 73 728 B: two stages of both operands; the C tile (69 632 B) reuses it
static_assert(BM * CS * 4 <= GEMM_LDS, "C tile must fit the operand buffers");

struct TokGemmArgs {
    const _Float16* x; const _Float16* w; const float* bias; const float* ls;
    long P; int N, K, nt;
    _Float16* y; float* res;
};
}  // namespace

template <int EPI>      // 0: y = acc + bias; 1: y = gelu(acc + bias); 2: res += ls * (acc + bias)
__global__ __launch_bounds__(256, 2) void tok_gemm_kernel(TokGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tok_smem[];
    _Float16* sA = reinterpret_cast<_Float16*>(tok_smem);        // [2][BM][OS]
    _Float16* sB = sA + 2 * BM * OS;                             // [2][BN][OS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    // consecutive tiles (same token rows, neighbouring output columns) on ONE XCD: the X rows they share stay in that L2
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int tn = id % a.nt, tm = id / a.nt;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN, K = a.K;
    const int wm = wave >> 1, wn = wave & 1;
    const int srow = tid >> 3, sc8 = (tid & 7) * 8;              // staging: rows srow + 32 i, 8 halves at sc8
    const _Float16* gx[4];
    const _Float16* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0 + srow + 32 * i;
        gx[i] = a.x + (m < a.P ? m : a.P - 1) * K + sc8;
        gw[i] = a.w + (long)(n0 + srow + 32 * i) * K + sc8;
    }
    // global -> registers two stages ahead, registers -> LDS one stage ahead: with two waves per SIMD a load has ~2 stages of MFMA
    // time (> 1000 cycles) to land; one stage ahead left the matrix pipe waiting on L2 every iteration
    struct Stage { uint4 a0, a1, a2, a3, b0, b1, b2, b3; };            // named members: the sets must stay in registers
    auto gload = [&](int k0) {
        Stage g;
        g.a0 = *reinterpret_cast<const uint4*>(gx[0] + k0); g.b0 = *reinterpret_cast<const uint4*>(gw[0] + k0);
        g.a1 = *reinterpret_cast<const uint4*>(gx[1] + k0); g.b1 = *reinterpret_cast<const uint4*>(gw[1] + k0);
        g.a2 = *reinterpret_cast<const uint4*>(gx[2] + k0); g.b2 = *reinterpret_cast<const uint4*>(gw[2] + k0);
        g.a3 = *reinterpret_cast<const uint4*>(gx[3] + k0); g.b3 = *reinterpret_cast<const uint4*>(gw[3] + k0);
        return g;
    };
    auto sstore = [&](const Stage& g, int buf) {
        _Float16* pa = sA + (buf * BM + srow) * OS + sc8;
        _Float16* pb = sB + (buf * BN + srow) * OS + sc8;
        *reinterpret_cast<uint4*>(pa) = g.a0;           *reinterpret_cast<uint4*>(pb) = g.b0;
        *reinterpret_cast<uint4*>(pa + 32 * OS) = g.a1; *reinterpret_cast<uint4*>(pb + 32 * OS) = g.b1;
        *reinterpret_cast<uint4*>(pa + 64 * OS) = g.a2; *reinterpret_cast<uint4*>(pb + 64 * OS) = g.b2;
        *reinterpret_cast<uint4*>(pa + 96 * OS) = g.a3; *reinterpret_cast<uint4*>(pb + 96 * OS) = g.b3;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    auto compute = [&](int buf) {
        const _Float16* pa = sA + (buf * BM + wm * 64 + j) * OS + kk * 8;
        const _Float16* pb = sB + (buf * BN + wn * 64 + j) * OS + kk * 8;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const vh8 fa0 = *reinterpret_cast<const vh8*>(pa + ks * 16), fa1 = *reinterpret_cast<const vh8*>(pa + 32 * OS + ks * 16);
            const vh8 fb0 = *reinterpret_cast<const vh8*>(pb + ks * 16), fb1 = *reinterpret_cast<const vh8*>(pb + 32 * OS + ks * 16);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb1, acc[1][1], 0, 0, 0);
        }
    };
    const int nk = K / BK;
    // Every workgroup starts its K loop at a different stage and wraps around: the token rows of a tile lie K * 2 bytes apart
    // (1536 B / 6144 B), so workgroups marching through K in step asked 2 - 8 of an XCD's 16 L2 channels for everything at once
    // (fc2 ran at 2.2 TB/s of L2 traffic, 141 TFLOP/s).  The loads are unconditional (a wrapped prefetch past the last stage is
    // simply not used), which also lets the compiler count them: the LDS store of stage s + 1 waits for its own 8 loads only.
    int kn = (int)((tm * 5 + tn * 3) % nk) * BK;
    auto next = [&]() { const int k = kn; kn += BK; kn = kn >= K ? kn - K : kn; return k; };
    Stage g0 = gload(next());
    sstore(g0, 0);
    Stage g1 = gload(next());
    // stage s travels in register set s & 1 and lives in LDS buffer s & 1
    for (int kt = 0; kt < nk; kt += 2) {
        __syncthreads();                                    // stage kt is in LDS; every wave is done reading stage kt - 1
        g0 = gload(next());
        asm volatile("" ::: "memory");                      // the scheduler otherwise sinks these loads below the MFMA block
        compute(0);
        sstore(g1, 1);
        if (kt + 1 >= nk) break;
        __syncthreads();
        g1 = gload(next());
        asm volatile("" ::: "memory");
        compute(1);
        sstore(g0, 0);
    }
    // C tile through LDS: accumulator rows are tokens (r & 3) + 8 (r >> 2) + 4 kk, its column is output channel j
    __syncthreads();
    float* sC = reinterpret_cast<float*>(tok_smem);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[(wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * CS + wn * 64 + ni * 32 + j] = acc[mi][ni][r];
    __syncthreads();
    const int c8 = (tid & 15) * 8, er = tid >> 4;          // 8 consecutive output channels of rows er + 16 i
    float bias[8], ls[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bias[e] = a.bias ? a.bias[n0 + c8 + e] : 0.f;
        ls[e] = (EPI == 2 && a.ls) ? a.ls[n0 + c8 + e] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
        const int row = er + 16 * i;
        const long m = m0 + row;
        if (m >= a.P) continue;
        const float4 v0 = *reinterpret_cast<const float4*>(sC + row * CS + c8), v1 = *reinterpret_cast<const float4*>(sC + row * CS + c8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] += bias[e];
            if constexpr (EPI == 1) v[e] = 0.5f * v[e] * (1.0f + erf_1ulp(v[e] * 0.70710678118654752f));
        }
        if constexpr (EPI == 2) {
            float4* rp = reinterpret_cast<float4*>(a.res + m * a.N + n0 + c8);
            float4 r0 = rp[0], r1 = rp[1];
            r0.x += ls[0] * v[0]; r0.y += ls[1] * v[1]; r0.z += ls[2] * v[2]; r0.w += ls[3] * v[3];
            r1.x += ls[4] * v[4]; r1.y += ls[5] * v[5]; r1.z += ls[6] * v[6]; r1.w += ls[7] * v[7];
            rp[0] = r0;
            rp[1] = r1;
        } else {
            vh8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
            *reinterpret_cast<vh8*>(a.y + m * a.N + n0 + c8) = o;
        }
    }
}

extern "C" int tdr_tok16_gemm(const void* x16, const void* w16, const float* bias, int64_t P, int N, int K, int epi, void* y16,
                              float* res, const float* ls, void* stream) {
    TDR_REQUIRE(x16 && w16 && P > 0 && N > 0 && K > 0, "tdr_tok16_gemm: bad argument");
    TDR_REQUIRE(N % BN == 0 && K % BK == 0, "tdr_tok16_gemm: N must be a multiple of %d and K of %d (got %d, %d)", BN, BK, N, K);
    TDR_REQUIRE(epi == 2 ? res != nullptr : (y16 != nullptr && (epi == 0 || epi == 1)), "tdr_tok16_gemm: epilogue %d lacks its output", epi);
    TokGemmArgs a{(const _Float16*)x16, (const _Float16*)w16, bias, ls, (long)P, N, K, N / BN, (_Float16*)y16, res};
    const dim3 grid(tdr_cdiv(P, BM) * (N / BN));
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tok_gemm_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tok_gemm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tok_gemm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr = true;
    }
    hipStream_t st = (hipStream_t)stream;
    if (epi == 0) hipLaunchKernelGGL(tok_gemm_kernel<0>, grid, dim3(256), GEMM_LDS, st, a);
    else if (epi == 1) hipLaunchKernelGGL(tok_gemm_kernel<1>, grid, dim3(256), GEMM_LDS, st, a);
    else hipLaunchKernelGGL(tok_gemm_kernel<2>, grid, dim3(256), GEMM_LDS, st, a);
    TDR_LAUNCH_CHECK("tok16_gemm");
    return TDR_OK;
}

// ---- attention -----------------------------------------------------------------------------------------------------------
namespace {
constexpr int HD = 64, KT = 64, QB = 256;
constexpr int KSTR = HD + 8;               // halves per key row of the K tile (ds_read_b128 fragments: conflict-free)
constexpr int VSTR = KT + 4;               // halves per d row of the transposed V tile (34 dwords: the 32 ds_read_b64 of a pass hit distinct banks)
struct TokAttnArgs {
    const _Float16* qkv; _Float16* out;
    int C, heads, T, LD, nq;
    float c;                               // scale * log2(e)
};
}  // namespace

__global__ __launch_bounds__(256, 2) void tok_attn_kernel(TokAttnArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 sK[2][KT][KSTR];
    __shared__ __attribute__((aligned(16))) _Float16 sV[2][HD][VSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);         // the query blocks of one (image, head) on one XCD
    const int qb = id % a.nq, h = (id / a.nq) % a.heads, img = id / (a.nq * a.heads);
    const int T = a.T, LD = a.LD, C3 = 3 * a.C;
    const _Float16* base = a.qkv + (long)img * LD * C3 + h * HD;
    const int q0 = qb * QB + wave * 64;
    const bool active = q0 < T;                             // waves past the last token only help staging
    vh8 qf[2][4];                                           // B operand of S^T = K Q^T: Q[q = j][d = 16 s + 8 kk ..]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 32 + j;
        const _Float16* qp = base + (long)(q < LD ? q : LD - 1) * C3 + 8 * kk;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qt][s] = *reinterpret_cast<const vh8*>(qp + 16 * s);
    }
    // staging: 64 keys x 8 chunks of 8 halves, K and V: two chunks of each per thread
    const int skey = tid >> 3, sc8 = (tid & 7) * 8;
    uint4 rk[2], rv[2];
    auto gload = [&](int key0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = key0 + skey + 32 * i;
            const _Float16* p = base + (long)(key < LD ? key : LD - 1) * C3 + a.C + sc8;
            rk[i] = *reinterpret_cast<const uint4*>(p);
            rv[i] = *reinterpret_cast<const uint4*>(p + a.C);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = skey + 32 * i;
            *reinterpret_cast<uint4*>(&sK[buf][key][sc8]) = rk[i];
            const vh8 v = __builtin_bit_cast(vh8, rv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sV[buf][sc8 + e][key] = v[e];
        }
    };
    f32x16 o[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][dt][r] = 0.f;
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};
    const float c = a.c;
    gload(0);
    sstore(0);
    const int ntile = (T + KT - 1) / KT;
    for (int t = 0; t < ntile; ++t) {
        const int key0 = t * KT, buf = t & 1;
        __syncthreads();
        if (t + 1 < ntile) gload(key0 + KT);
        if (active) {
            f32x16 st[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { st[0][kb][r] = 0.f; st[1][kb][r] = 0.f; }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const vh8 kf = *reinterpret_cast<const vh8*>(&sK[buf][kb * 32 + j][16 * s + 8 * kk]);
                    st[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0][s], st[0][kb], 0, 0, 0);
                    st[1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[1][s], st[1][kb], 0, 0, 0);
                }
            }
            if (key0 + KT > T) {                            // last tile: keys past the sequence
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool ok = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk < T;
                        st[0][kb][r] = ok ? st[0][kb][r] : -1e30f;
                        st[1][kb][r] = ok ? st[1][kb][r] : -1e30f;
                    }
            }
            vh8 pf[2][4];                                   // P^T per 16-key step, keys in the accumulator's row order
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float mx = -1e30f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[qt][kb][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mnew = fmaxf(m[qt], mx), nb = -mnew * c;
                const float alpha = __builtin_amdgcn_exp2f((m[qt] - mnew) * c);
                float sum = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(st[qt][kb][r], c, nb));
                        sum += p;
                        pf[qt][2 * kb + (r >> 3)][r & 7] = (_Float16)p;
                    }
                sum += __shfl_xor(sum, 32, 64);
                l[qt] = l[qt] * alpha + sum;
                m[qt] = mnew;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const _Float16* vp = &sV[buf][dt * 32 + j][(s >> 1) * 32 + 16 * (s & 1) + 4 * kk];
                    const vh4 v0 = *reinterpret_cast<const vh4*>(vp), v1 = *reinterpret_cast<const vh4*>(vp + 8);
                    const vh8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[0][s], o[0][dt], 0, 0, 0);
                    o[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[1][s], o[1][dt], 0, 0, 0);
                }
        }
        if (t + 1 < ntile) sstore(buf ^ 1);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 32 + j;
        if (q >= LD) continue;
        const float inv = (active && q < T) ? 1.f / l[qt] : 0.f;
        _Float16* op = a.out + ((long)img * LD + q) * a.C + h * HD + 4 * kk;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vh4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (_Float16)(inv == 0.f ? 0.f : o[qt][dt][4 * g + e] * inv);
                *reinterpret_cast<vh4*>(op + dt * 32 + 8 * g) = w;
            }
    }
}

extern "C" int tdr_tok16_attention(const void* qkv16, int B, int C, int heads, int T, int LD, float scale, void* out16, void* stream) {
    TDR_REQUIRE(qkv16 && out16 && B > 0 && heads > 0 && T > 0 && LD >= T, "tdr_tok16_attention: bad argument");
    TDR_REQUIRE(C == heads * HD, "tdr_tok16_attention: head dim must be %d (C %d, heads %d)", HD, C, heads);
    const int nq = tdr_cdiv(LD, QB);
    TokAttnArgs a{(const _Float16*)qkv16, (_Float16*)out16, C, heads, T, LD, nq, scale * 1.4426950408889634f};
    hipLaunchKernelGGL(tok_attn_kernel, dim3(nq * heads * B), dim3(256), 0, (hipStream_t)stream, a);
    TDR_LAUNCH_CHECK("tok16_attention");
    return TDR_OK;
}
