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
#include <utility>
#include "tdr_common.h"
#include "tdr_erf.h"
#include "../../include/tdr.h"

typedef _Float16 vh8 __attribute__((ext_vector_type(8)));
typedef _Float16 vh4 __attribute__((ext_vector_type(4)));
typedef __bf16 vb8 __attribute__((ext_vector_type(8)));

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
template <int OUT>         // 0: fp32 | 1: fp16 | 2: hi | lo fp16 planes (2-way split) | 3: h | m | l bf16 planes (3-way split)
__global__ __launch_bounds__(256) void tok_ln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                    int D, long P, float eps, void* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= P) return;
    const int n4 = D >> 2;
    const float4* row = reinterpret_cast<const float4*>(x + tok * D);
    float4 v[5];                                            // D <= 1280
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n4 ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
        if (lane + 64 * i < n4) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = lane + 64 * i;
        if (c >= n4) continue;
        const float4 g = reinterpret_cast<const float4*>(w)[c], h = reinterpret_cast<const float4*>(b)[c];
        const float y0 = (v[i].x - mean) * rstd * g.x + h.x, y1 = (v[i].y - mean) * rstd * g.y + h.y;
        const float y2 = (v[i].z - mean) * rstd * g.z + h.z, y3 = (v[i].w - mean) * rstd * g.w + h.w;
        if constexpr (OUT == 3) {
            unsigned h0, m0, l0, h1, m1, l1;
            tdr_split3_bf16(y0, y1, h0, m0, l0);
            tdr_split3_bf16(y2, y3, h1, m1, l1);
            uint2* o = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + tok * D) + c;
            const long ps = P * D / 4;                          // plane stride in 8-byte pieces
            o[0] = make_uint2(h0, h1);
            o[ps] = make_uint2(m0, m1);
            o[2 * ps] = make_uint2(l0, l1);
        } else if constexpr (OUT >= 1) {
            const vh4 o = {(_Float16)y0, (_Float16)y1, (_Float16)y2, (_Float16)y3};
            reinterpret_cast<vh4*>(reinterpret_cast<_Float16*>(out) + tok * D)[c] = o;
            if constexpr (OUT == 2) {
                const vh4 l = {(_Float16)(y0 - (float)o[0]), (_Float16)(y1 - (float)o[1]), (_Float16)(y2 - (float)o[2]), (_Float16)(y3 - (float)o[3])};
                reinterpret_cast<vh4*>(reinterpret_cast<_Float16*>(out) + (P + tok) * D)[c] = l;
            }
        } else {
            reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + tok * D)[c] = make_float4(y0, y1, y2, y3);
        }
    }
}

extern "C" int tdr_tok_layernorm(const float* x, const float* w, const float* b, int64_t P, int D, float eps, int out_f16, void* out,
                                 void* stream) {
    TDR_REQUIRE(x && w && b && out && P > 0, "tdr_tok_layernorm: bad argument");
    TDR_REQUIRE(D % 4 == 0 && D <= 1280, "tdr_tok_layernorm: D must be a multiple of 4, at most 1280 (got %d)", D);
    TDR_REQUIRE(out_f16 >= 0 && out_f16 <= 3, "tdr_tok_layernorm: output mode %d", out_f16);
    const dim3 grid(tdr_cdiv(P, 4));
    if (out_f16 == 3) hipLaunchKernelGGL(tok_ln_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, D, (long)P, eps, out);
    else if (out_f16 == 2) hipLaunchKernelGGL(tok_ln_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, D, (long)P, eps, out);
    else if (out_f16) hipLaunchKernelGGL(tok_ln_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, D, (long)P, eps, out);
    else hipLaunchKernelGGL(tok_ln_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, D, (long)P, eps, out);
    TDR_LAUNCH_CHECK("tok_layernorm");
    return TDR_OK;
}

// ---- fp16 GEMM ---------------------------------------------------------------------------------------------------------
// Three operand formats of one kernel:
//   <BM 128, BK 64, NPL 1>     plain fp16 operands (the DINOv2 matcher: 'h1' arithmetic), 27 000 token rows: 128 x 128 tiles
//   <BM  64, BK 32, NPL 2>     2-way split operands (hi / lo fp16 planes; the three products lo*hi + hi*lo + hi*hi per fragment
//                              pair, fp32-faithful -- the frozen CLIP encoder of the stage-A trainers feeds a trained path):
//                              ~1 150 token rows, so 64 x 128 tiles (180 - 720 workgroups) and 32-deep stages (61 KB of LDS,
//                              two workgroups per CU)
//   <BM 128, BK 16, NPL 3>     3-way split operands (h / m / l bf16 planes, six products per fragment pair: the default 'bx3'
//                              arithmetic -- 24-bit operands on the whole fp32 exponent range; the matcher at its default)
namespace {
constexpr int BN = 128;

// N 16-byte registers with compile-time indexing
template <int N>
struct RegSet {
    uint4 head;
    RegSet<N - 1> tail;
    template <int I>
    __device__ __forceinline__ uint4& at() {
        if constexpr (I == 0) return head;
        else return tail.template at<I - 1>();
    }
};
template <>
struct RegSet<0> {};
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct TokGemmArgs {
    const _Float16* x; const _Float16* w; const float* bias; const float* ls;     // X2: the lo planes follow at x + P*K, w + N*K
    long P; int N, K, nt;
    _Float16* y; float* res;                                                      // y: fp16 [P][N] (X2 + EPI_PLANES: hi | lo); res: fp32
};
enum { EPI_F16 = 0, EPI_GELU_F16 = 1, EPI_RES = 2, EPI_CM_F32 = 3, EPI_ACT_PLANES = 4 };
}  // namespace

// EPI_F16: y = acc + bias | EPI_GELU_F16: y = erf-GELU(acc + bias) | EPI_RES: res += ls * (acc + bias)
// EPI_CM_F32: res[n][P] = acc + bias, fp32 channel-major (the input layout of tdr_attention_fwd_math)
// EPI_ACT_PLANES: y = split(ACT(acc + bias)) as hi | lo planes; ACT 0 none, 2 erf-GELU, 3 quick_gelu
template <int EPI, int BM, int BK, int NPL, int ACT, int OCC = 2, bool PF2 = false>
__global__ __launch_bounds__(256, OCC) void tok_gemm_kernel(TokGemmArgs a) {
    constexpr bool X2 = NPL == 2, X3 = NPL == 3;   // fp16 hi | lo planes (3 products) / bf16 h | m | l planes (6 products)
    constexpr int OS = BK + 8;                 // halves per operand row in LDS (144 B / 80 B): ds_read_b128 passes without bank conflicts
    constexpr int CS = EPI == EPI_CM_F32 ? BN + 5 : BN + 8;                      // floats per C row in LDS
    constexpr int NP = NPL;                    // operand planes
    constexpr int CPR = BK / 8, RPP = 256 / CPR, NA = BM / RPP, NB = BN / RPP;   // 16-byte chunks per row, rows per staging pass
    constexpr int MI = BM / 64;                // 32-row accumulator tiles per wave (waves 2 x 2, wave tile BM/2 x 64)
    // X3 with 32-deep stages at two workgroups per CU: ONE LDS buffer (61 KB), the next stage waits in registers
    // (BM 256: 128 x 64 wave tiles -- 18 instead of 24 fragment reads per 48 MFMAs -- always in one buffer)
    constexpr bool SB = X3 && ((BK == 32 && OCC == 2) || BM == 256 || OCC == 3);
    constexpr int NBUF = SB ? 1 : 2;
    constexpr int LDSB = NBUF * (BM + BN) * OS * 2 * NP;
    constexpr int EP = BM * CS * 4 <= LDSB ? 1 : ((BM / 2) * CS * 4 <= LDSB ? 2 : 4);   // epilogue passes (the C tile reuses the operand buffers)
    constexpr int ER = BM / EP;                                                  // token rows per pass
    static_assert(ER * CS * 4 <= LDSB && (EP == 1 || BM >= 128) && ER % 32 == 0, "C tile must fit the operand buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char tok_smem[];
    _Float16* sA = reinterpret_cast<_Float16*>(tok_smem);        // [2 stages][NP][BM][OS]
    _Float16* sB = sA + NBUF * NP * BM * OS;                     // [2 stages][NP][BN][OS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
    // consecutive tiles (same token rows, neighbouring output columns) on ONE XCD: the X rows they share stay in that L2
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    else if constexpr (X3) {                                     // any grid size: XCD x takes a contiguous run of q (+ 1 for x < r) tiles
        const int q = total >> 3, r = total & 7, xcd = id & 7;
        id = xcd * q + (xcd < r ? xcd : r) + (id >> 3);
    }
    const int tn = id % a.nt, tm = id / a.nt;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN, K = a.K;
    const int wm = wave >> 1, wn = wave & 1;
    const int srow = tid / CPR, sc8 = (tid % CPR) * 8;           // staging: rows srow + RPP i, 8 halves at sc8
    const long xlo = a.P * K, wlo = (long)a.N * K;               // plane strides
    const _Float16* gx[NA];
    const _Float16* gw[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const long m = m0 + srow + RPP * i;
        gx[i] = a.x + (m < a.P ? m : a.P - 1) * K + sc8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) gw[i] = a.w + (long)(n0 + srow + RPP * i) * K + sc8;
    // the two register sets of the prefetch: named members walked by compile-time indices (as arrays -- local, in a struct, or
    // handed to a lambda by reference -- the compiler demoted them to scratch memory)
    constexpr int NR = NP * (NA + NB);
    auto gload = [&](int k0) {
        RegSet<NR> g;
        static_for<NR>([&](auto q) {
            constexpr int Q = decltype(q)::value, pl = Q / (NA + NB), i = Q % (NA + NB);
            if constexpr (i < NA) g.template at<Q>() = *reinterpret_cast<const uint4*>(gx[i] + pl * xlo + k0);
            else g.template at<Q>() = *reinterpret_cast<const uint4*>(gw[i - NA] + pl * wlo + k0);
        });
        return g;
    };
    auto sstore = [&](RegSet<NR> g, int buf) {
        static_for<NR>([&](auto q) {
            constexpr int Q = decltype(q)::value, pl = Q / (NA + NB), i = Q % (NA + NB);
            if constexpr (i < NA) *reinterpret_cast<uint4*>(sA + ((buf * NP + pl) * BM + srow + RPP * i) * OS + sc8) = g.template at<Q>();
            else *reinterpret_cast<uint4*>(sB + ((buf * NP + pl) * BN + srow + RPP * (i - NA)) * OS + sc8) = g.template at<Q>();
        });
    };
    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    auto compute = [&](int buf) {
        const _Float16* pa = sA + (buf * NP * BM + wm * (BM / 2) + j) * OS + kk * 8;
        const _Float16* pb = sB + (buf * NP * BN + wn * 64 + j) * OS + kk * 8;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            vh8 fa[NP][MI], fb[NP][2];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[pl][mi] = *reinterpret_cast<const vh8*>(pa + (pl * BM + mi * 32) * OS + ks * 16);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) fb[pl][ni] = *reinterpret_cast<const vh8*>(pb + (pl * BN + ni * 32) * OS + ks * 16);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (X3) {                     // l h, h l, m m, m h, h m, h h: the library's 6-product order, small terms first
                        constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                        for (int q = 0; q < 6; ++q)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vb8, fa[SA[q]][mi]),
                                                                                  __builtin_bit_cast(vb8, fb[SB[q]][ni]), acc[mi][ni], 0, 0, 0);
                    } else {
                        if constexpr (X2) {                 // small cross terms first
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][mi], fb[0][ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][mi], fb[1][ni], acc[mi][ni], 0, 0, 0);
                        }
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][mi], fb[0][ni], acc[mi][ni], 0, 0, 0);
                    }
                }
        }
    };
    const int nk = K / BK;
    // Global loads run two stages ahead of the MFMAs (registers), LDS stores one stage ahead.  The loads are unconditional (the
    // prefetch past the last stage wraps around and is simply not used), which lets the compiler count them: the LDS store of stage
    // s + 1 waits for its own loads only (vmcnt(8)), not for those of stage s + 2.  (With conditional loads it waited for vmcnt(0),
    // and without the scheduling barriers below it sank the loads under the MFMA block: 141 - 175 TFLOP/s instead of 450 - 580.)
    int kn = 0;
    auto next = [&]() { const int k = kn; kn += BK; kn = kn >= K ? kn - K : kn; return k; };
    RegSet<NR> g0 = gload(next());
    sstore(g0, 0);
    RegSet<NR> g1 = gload(next());
    if constexpr (SB && PF2) {                              // two register stages in flight behind the one in LDS
        g0 = gload(next());                                 // (stage 1 is in g1, stage 2 in g0)
        for (int kt = 0; kt < nk; kt += 2) {
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __syncthreads();
            sstore(g1, 0);
            g1 = gload(next());
            if (kt + 1 >= nk) break;
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __syncthreads();
            sstore(g0, 0);
            g0 = gload(next());
        }
    } else if constexpr (SB) {
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();                                // stage kt is in LDS
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __syncthreads();                                // every wave is done reading it
            sstore(g1, 0);
            g1 = gload(next());
        }
    } else
    // stage s travels in register set s & 1 and lives in LDS buffer s & 1
    for (int kt = 0; kt < nk; kt += 2) {
        __syncthreads();                                    // stage kt is in LDS; every wave is done reading stage kt - 1
        g0 = gload(next());
        __builtin_amdgcn_sched_barrier(0);                  // the scheduler otherwise sinks these loads below the MFMA block
        compute(0);
        sstore(g1, 1);
        if (kt + 1 >= nk) break;
        __syncthreads();
        g1 = gload(next());
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        sstore(g0, 0);
    }
    // C tile through LDS: accumulator rows are tokens (r & 3) + 8 (r >> 2) + 4 kk, its column is output channel j
    float* sC = reinterpret_cast<float*>(tok_smem);
#pragma unroll
    for (int ep = 0; ep < EP; ++ep) {
        __syncthreads();
        if constexpr (BM == 256) {                          // pass ep takes the 32-row tiles whose rows fall into [ep ER, (ep + 1) ER)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row0 = wm * (BM / 2) + mi * 32;
                if (row0 / ER != ep) continue;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sC[(row0 - ep * ER + (r & 3) + 8 * (r >> 2) + 4 * kk) * CS + wn * 64 + ni * 32 + j] = acc[mi][ni][r];
            }
        } else if (EP == 1 || wm == ep) {
            const int rbase = EP == 1 ? wm * (BM / 2) : 0;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sC[(rbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * CS + wn * 64 + ni * 32 + j] = acc[mi][ni][r];
        }
        __syncthreads();
        const long mb = m0 + ep * ER;
        if constexpr (EPI == EPI_CM_F32) {
            // 8 consecutive tokens of one output channel per thread; lanes walk the token groups first: 32-byte pieces of one output
            // row side by side, LDS reads without bank conflicts (CS = 133)
            constexpr int TG = ER / 8, CPP = 256 / TG;
            const int tg = tid % TG, c0 = tid / TG;
            const long m = mb + 8 * tg;
#pragma unroll
            for (int i = 0; i < BN / CPP; ++i) {
                const int c = c0 + CPP * i;
                if (m >= a.P) continue;
                const float bv = a.bias ? a.bias[n0 + c] : 0.f;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sC[(8 * tg + e) * CS + c] + bv;
                float4* op = reinterpret_cast<float4*>(a.res + (long)(n0 + c) * a.P + m);
                op[0] = make_float4(v[0], v[1], v[2], v[3]);
                op[1] = make_float4(v[4], v[5], v[6], v[7]);
            }
        } else {
            const int c8 = (tid & 15) * 8, er = tid >> 4;          // 8 consecutive output channels of rows er + 16 i
            float bias[8], ls[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bias[e] = a.bias ? a.bias[n0 + c8 + e] : 0.f;
                ls[e] = (EPI == EPI_RES && a.ls) ? a.ls[n0 + c8 + e] : 1.f;
            }
#pragma unroll
            for (int i = 0; i < ER / 16; ++i) {
                const int row = er + 16 * i;
                const long m = mb + row;
                if (m >= a.P) continue;
                const float4 v0 = *reinterpret_cast<const float4*>(sC + row * CS + c8), v1 = *reinterpret_cast<const float4*>(sC + row * CS + c8 + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] += bias[e];
                    if constexpr (EPI == EPI_GELU_F16 || (EPI == EPI_ACT_PLANES && ACT == 2)) v[e] = 0.5f * v[e] * (1.0f + erf_1ulp(v[e] * 0.70710678118654752f));
                    if constexpr (EPI == EPI_ACT_PLANES && ACT == 3) v[e] = v[e] / (1.f + expf(-1.702f * v[e]));
                }
                if constexpr (EPI == EPI_RES) {
                    float4* rp = reinterpret_cast<float4*>(a.res + m * a.N + n0 + c8);
                    float4 r0 = rp[0], r1 = rp[1];
                    r0.x += ls[0] * v[0]; r0.y += ls[1] * v[1]; r0.z += ls[2] * v[2]; r0.w += ls[3] * v[3];
                    r1.x += ls[4] * v[4]; r1.y += ls[5] * v[5]; r1.z += ls[6] * v[6]; r1.w += ls[7] * v[7];
                    rp[0] = r0;
                    rp[1] = r1;
                } else if constexpr (X3) {
                    static_assert(!X3 || EPI == EPI_ACT_PLANES, "bf16 triple planes: EPI_RES / EPI_CM_F32 / EPI_ACT_PLANES only");
                    unsigned ph[4], pm[4], pl[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) tdr_split3_bf16(v[2 * e], v[2 * e + 1], ph[e], pm[e], pl[e]);
                    unsigned short* yb = reinterpret_cast<unsigned short*>(a.y);
                    *reinterpret_cast<uint4*>(yb + m * a.N + n0 + c8) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                    *reinterpret_cast<uint4*>(yb + (a.P + m) * a.N + n0 + c8) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
                    *reinterpret_cast<uint4*>(yb + (2 * a.P + m) * a.N + n0 + c8) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                } else {
                    vh8 o, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        o[e] = (_Float16)v[e];
                        lo[e] = (_Float16)(v[e] - (float)o[e]);
                    }
                    *reinterpret_cast<vh8*>(a.y + m * a.N + n0 + c8) = o;
                    if constexpr (EPI == EPI_ACT_PLANES) *reinterpret_cast<vh8*>(a.y + (a.P + m) * a.N + n0 + c8) = lo;
                }
            }
        }
    }
}

namespace {
template <int EPI, int BM, int BK, int NPL, int ACT, int OCC = 2, bool PF2 = false>
int launch_tok_gemm(const TokGemmArgs& a, hipStream_t st) {
    constexpr int lds = ((NPL == 3 && ((BK == 32 && OCC == 2) || BM == 256 || OCC == 3)) ? 1 : 2) * (BM + BN) * (BK + 8) * 2 * NPL;
    auto kern = tok_gemm_kernel<EPI, BM, BK, NPL, ACT, OCC, PF2>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(tdr_cdiv(a.P, BM) * a.nt), dim3(256), lds, st, a);
    TDR_LAUNCH_CHECK("tok16_gemm");
    return TDR_OK;
}
}  // namespace

extern "C" int tdr_tok16_gemm(const void* x16, const void* w16, const float* bias, int64_t P, int N, int K, int epi, void* y16,
                              float* res, const float* ls, void* stream) {
    TDR_REQUIRE(x16 && w16 && P > 0 && N > 0 && K > 0, "tdr_tok16_gemm: bad argument");
    TDR_REQUIRE(N % BN == 0 && K % 64 == 0, "tdr_tok16_gemm: N must be a multiple of %d and K of 64 (got %d, %d)", BN, N, K);
    TDR_REQUIRE(epi == 2 ? res != nullptr : (y16 != nullptr && (epi == 0 || epi == 1)), "tdr_tok16_gemm: epilogue %d lacks its output", epi);
    TokGemmArgs a{(const _Float16*)x16, (const _Float16*)w16, bias, ls, (long)P, N, K, N / BN, (_Float16*)y16, res};
    hipStream_t st = (hipStream_t)stream;
    if (epi == 0) return launch_tok_gemm<EPI_F16, 128, 64, 1, 0>(a, st);
    if (epi == 1) return launch_tok_gemm<EPI_GELU_F16, 128, 64, 1, 0>(a, st);
    return launch_tok_gemm<EPI_RES, 128, 64, 1, 0>(a, st);
}

extern "C" int tdr_tok16x2_gemm(const void* x16x2, const void* w16x2, const float* bias, int64_t P, int N, int K, int epi, int act,
                                void* y16x2, float* out32, void* stream) {
    TDR_REQUIRE(x16x2 && w16x2 && P > 0 && N > 0 && K > 0, "tdr_tok16x2_gemm: bad argument");
    TDR_REQUIRE(N % BN == 0 && K % 32 == 0 && P % 8 == 0, "tdr_tok16x2_gemm: N %% %d, K %% 32, P %% 8 must be 0 (got %d, %d, %lld)", BN, N, K,
                (long long)P);
    TDR_REQUIRE(epi == 4 ? (y16x2 != nullptr && (act == 0 || act == 2 || act == 3)) : ((epi == 2 || epi == 3) && out32 != nullptr),
                "tdr_tok16x2_gemm: epilogue %d (act %d) lacks its output", epi, act);
    TokGemmArgs a{(const _Float16*)x16x2, (const _Float16*)w16x2, bias, nullptr, (long)P, N, K, N / BN, (_Float16*)y16x2, out32};
    hipStream_t st = (hipStream_t)stream;
    // ~1 150 token rows: (a) the wide Linears (q/k/v, fc1: > 512 tiles of 64 rows) take 128-row tiles -- one round of 270 - 360
    // workgroups, two per CU (2 x 80 KB of LDS), and 2/3 of the L2 traffic of 64-row tiles, which ran them at 7 TB/s of operand
    // re-reads; (b) the narrow ones (out-projection, fc2: 180 tiles) are a single partial round whose time is the serial chain of K
    // stages behind cold weights (~0.7 us per stage at two stages of prefetch): 64-deep stages halve the chain.
    const bool wide = (long)tdr_cdiv(P, 64) * (N / BN) > 512;
    const int cfg = wide ? 0 : (K % 64 == 0 ? 1 : 2);
#define TOK_X2(EPI_, ACT_)                                                               \
    (cfg == 0 ? launch_tok_gemm<EPI_, 128, 32, 2, ACT_>(a, st)                        \
              : cfg == 1 ? launch_tok_gemm<EPI_, 64, 64, 2, ACT_>(a, st) : launch_tok_gemm<EPI_, 64, 32, 2, ACT_>(a, st))
    if (epi == 2) return TOK_X2(EPI_RES, 0);
    if (epi == 3) return TOK_X2(EPI_CM_F32, 0);
    if (act == 2) return TOK_X2(EPI_ACT_PLANES, 2);
    if (act == 3) return TOK_X2(EPI_ACT_PLANES, 3);
    return TOK_X2(EPI_ACT_PLANES, 0);
#undef TOK_X2
}

// 3-way bf16 split planes (h | m | l, x = h + m + l exactly: the library's default 'bx3' arithmetic, 6 products per fragment pair):
// x [3][P][K], w [3][N][K] bf16; epi 2: out32 [P][N] += acc + bias | 3: out32 [N][P] = acc + bias | 4: y [3][P][N] = split(act(acc + bias)).
// Stage forms (TDR_TOK3_STAGE, measured at the matcher's shapes, profiles/r5/probe_tok16x3_v2.log / _v3.log):
//   2 (default) 128 x 128 x 32 stages in ONE LDS buffer (61 KB, two workgroups per CU), the next stage waiting in registers  131 - 154 TF
//   1           128 x 128 x 32, double-buffered: 120 KB, one workgroup per CU                                                113 - 136 TF
//   0           128 x 128 x 16, double-buffered: 72 KB, two per CU (32-byte row pieces: half of every 64-byte request)       107 - 126 TF
//   3 / 4       256 x 128 x 16 / x 32 in one buffer (128 x 64 wave tiles: 25 % fewer LDS fragment reads), two / one per CU   125 - 140 TF
//   5           128 x 128 x 16 in one buffer, three per CU                                                                     96 - 114 TF
//   6           as 2 with TWO register stages in flight behind the one in LDS (226 VGPRs, no spill)                          flat (_v5.log)
// Counters (profiles/r5/pmc_tok16x3.txt): matrix pipe 42 % busy, LDS 34 %, waves 64 % of their cycles in s_waitcnt vmcnt -- the operand
// stream (64-byte L1 -> L2 requests, 6.3 GB per fc1 launch at ~7 TB/s, 27 % of them past the 4 MB L2) bounds every form; a second register
// stage in flight changes nothing, so it is the request throughput of that stream, not its latency: fewer operand bytes per product (256 x 256
// block tiles on 8 waves) or full-line pieces are what is left.
extern "C" int tdr_tok16x3_gemm(const void* x16x3, const void* w16x3, const float* bias, int64_t P, int N, int K, int epi, int act,
                                void* y16x3, float* out32, const float* ls, void* stream) {
    TDR_REQUIRE(x16x3 && w16x3 && P > 0 && N > 0 && K > 0, "tdr_tok16x3_gemm: bad argument");
    TDR_REQUIRE(N % BN == 0 && K % 32 == 0 && P % 8 == 0, "tdr_tok16x3_gemm: N %% %d, K %% 32, P %% 8 must be 0 (got %d, %d, %lld)", BN, N, K,
                (long long)P);
    TDR_REQUIRE(epi == 4 ? (y16x3 != nullptr && (act == 0 || act == 2 || act == 3)) : ((epi == 2 || epi == 3) && out32 != nullptr),
                "tdr_tok16x3_gemm: epilogue %d (act %d) lacks its output", epi, act);
    TokGemmArgs a{(const _Float16*)x16x3, (const _Float16*)w16x3, bias, ls, (long)P, N, K, N / BN, (_Float16*)y16x3, out32};
    hipStream_t st = (hipStream_t)stream;
    static const int deep = tdr_tune_env("TDR_TOK3_STAGE") ? atoi(tdr_tune_env("TDR_TOK3_STAGE")) : 2;
    const bool wide = (long)tdr_cdiv(P, 64) * (N / BN) > 512;
    const int cfg = wide ? (deep == 2 ? 3 : (deep >= 3 ? deep + 1 : (deep ? 1 : 0))) : 2;
#define TOK_X3(EPI_, ACT_)                                                               \
    (cfg == 7 ? launch_tok_gemm<EPI_, 128, 32, 3, ACT_, 2, true>(a, st)                  \
     : cfg == 6 ? launch_tok_gemm<EPI_, 128, 16, 3, ACT_, 3>(a, st)                      \
     : cfg == 4 ? launch_tok_gemm<EPI_, 256, 16, 3, ACT_, 2>(a, st)                      \
     : cfg == 5 ? launch_tok_gemm<EPI_, 256, 32, 3, ACT_, 1>(a, st)                      \
     : cfg == 0 ? launch_tok_gemm<EPI_, 128, 16, 3, ACT_, 2>(a, st)                      \
              : cfg == 1 ? launch_tok_gemm<EPI_, 128, 32, 3, ACT_, 1>(a, st)             \
                         : cfg == 3 ? launch_tok_gemm<EPI_, 128, 32, 3, ACT_, 2>(a, st) : launch_tok_gemm<EPI_, 64, 32, 3, ACT_, 1>(a, st))
    if (epi == 2) return TOK_X3(EPI_RES, 0);
    if (epi == 3) return TOK_X3(EPI_CM_F32, 0);
    if (act == 2) return TOK_X3(EPI_ACT_PLANES, 2);
    if (act == 3) return TOK_X3(EPI_ACT_PLANES, 3);
    return TOK_X3(EPI_ACT_PLANES, 0);
#undef TOK_X3
}

// ---- 2-way split planes: producers ----------------------------------------------------------------------------------------
// fp32 channel-major [C][P] (the attention output) -> token-major hi | lo fp16 planes [2][P][C]
template <int NPL>
__global__ __launch_bounds__(256) void cm_to_planes_kernel(const float* __restrict__ src, int C, long P, _Float16* __restrict__ dst) {
    __shared__ float t[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i;
        const long pp = p0 + tx;
        t[ty + 8 * i][tx] = (c < C && pp < P) ? src[(long)c * P + pp] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long pp = p0 + ty + 8 * i;
        const int c = c0 + tx;
        if (pp < P && c < C) {
            const float v = t[tx][ty + 8 * i];
            if constexpr (NPL == 3) {                       // h | m | l bf16 (the low halves of a tdr_split3_bf16 pair)
                unsigned h, m, l;
                tdr_split3_bf16(v, 0.f, h, m, l);
                unsigned short* d = reinterpret_cast<unsigned short*>(dst);
                d[pp * C + c] = (unsigned short)h;
                d[(P + pp) * C + c] = (unsigned short)m;
                d[(2 * P + pp) * C + c] = (unsigned short)l;
            } else {
                const _Float16 h = (_Float16)v;
                dst[pp * C + c] = h;
                dst[(P + pp) * C + c] = (_Float16)(v - (float)h);
            }
        }
    }
}

extern "C" int tdr_cm_to_tok16x2(const float* src, int C, int64_t P, void* dst16x2, void* stream) {
    TDR_REQUIRE(src && dst16x2 && C > 0 && P > 0, "tdr_cm_to_tok16x2: bad argument");
    hipLaunchKernelGGL(cm_to_planes_kernel<2>, dim3(tdr_cdiv(P, 32), tdr_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, src, C, (long)P,
                       (_Float16*)dst16x2);
    TDR_LAUNCH_CHECK("cm_to_tok16x2");
    return TDR_OK;
}

extern "C" int tdr_cm_to_tok16x3(const float* src, int C, int64_t P, void* dst16x3, void* stream) {
    TDR_REQUIRE(src && dst16x3 && C > 0 && P > 0, "tdr_cm_to_tok16x3: bad argument");
    hipLaunchKernelGGL(cm_to_planes_kernel<3>, dim3(tdr_cdiv(P, 32), tdr_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, src, C, (long)P,
                       (_Float16*)dst16x3);
    TDR_LAUNCH_CHECK("cm_to_tok16x3");
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
    struct KV { uint4 k0, k1, v0, v1; };                    // (named members, returned by value: arrays went to scratch memory)
    auto gload = [&](int key0) {
        KV g;
        const int ka = key0 + skey, kb = ka + 32;
        const _Float16* pa = base + (long)(ka < LD ? ka : LD - 1) * C3 + a.C + sc8;
        const _Float16* pb = base + (long)(kb < LD ? kb : LD - 1) * C3 + a.C + sc8;
        g.k0 = *reinterpret_cast<const uint4*>(pa); g.v0 = *reinterpret_cast<const uint4*>(pa + a.C);
        g.k1 = *reinterpret_cast<const uint4*>(pb); g.v1 = *reinterpret_cast<const uint4*>(pb + a.C);
        return g;
    };
    auto sstore = [&](KV g, int buf) {
        *reinterpret_cast<uint4*>(&sK[buf][skey][sc8]) = g.k0;
        *reinterpret_cast<uint4*>(&sK[buf][skey + 32][sc8]) = g.k1;
        const vh8 v0 = __builtin_bit_cast(vh8, g.v0), v1 = __builtin_bit_cast(vh8, g.v1);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sV[buf][sc8 + e][skey] = v0[e]; sV[buf][sc8 + e][skey + 32] = v1[e]; }
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
    KV g = gload(0);
    sstore(g, 0);
    const int ntile = (T + KT - 1) / KT;
    for (int t = 0; t < ntile; ++t) {
        const int key0 = t * KT, buf = t & 1;
        __syncthreads();
        if (t + 1 < ntile) g = gload(key0 + KT);
        if (active) {
            f32x16 st[2][2];
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const vh8 kf = *reinterpret_cast<const vh8*>(&sK[buf][kb * 32 + j][16 * s + 8 * kk]);
                    // (the first product takes the constant 0 as its C operand: no 64 v_mov per tile to clear the score tiles)
                    st[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0][s], s == 0 ? zero : st[0][kb], 0, 0, 0);
                    st[1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[1][s], s == 0 ? zero : st[1][kb], 0, 0, 0);
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
                // the running maximum of a query stops moving after a few tiles: the rescale (alpha == 1 exactly) is skipped for the wave
                if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
                }
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
        if (t + 1 < ntile) sstore(g, buf ^ 1);
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
