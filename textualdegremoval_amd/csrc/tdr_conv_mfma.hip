// Implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
//   out[n][m][pix] = epi( sum_k A[m][k] * B[k][pix] ),  k = (channel chunk, tap, channel-in-chunk)
//
// One workgroup = 4 waves (one per SIMD).  The block owns BM = 32*TM*WM output
// channels x 32*TN*WN output pixels; pixels are grouped in 32-pixel sub-tiles of
// (32/TW rows x TW cols), TW in {8,16,32} chosen at run time so deep (small) maps
// still fill the 32-wide MFMA N dimension.  Per K-chunk (CK channels x all taps)
// the block stages
//   s_w [KC][BM]          packed weights (M contiguous -> conflict-free A reads)
//   s_in[CK][LH][LW]      input halo tile (consecutive pixels -> conflict-free B reads)
// through registers (global loads for chunk i+1 are in flight while chunk i is
// on the matrix pipe), then issues v_mfma_f32_32x32x2_f32: exact fp32 (bitwise a
// k-ordered fmaf chain), 64 cycles each, so the two ds_read_b32 per MFMA and the
// address VALU are free.  fp32 MFMA keeps the 1e-4 max-abs parity bar of the
// reference's fp32 CPU path; see DESIGN.md for the roofline of this choice.
#include "tdr_common.h"
#include "tdr_conv_epi.h"
#include "tdr_pack.h"
#include "../../include/tdr.h"

namespace {

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int plane_elems(int NT, int TW, int KH, int S, int D) {
    return (((NT * 32 / TW) - 1) * S + (KH - 1) * D + 1) * ((TW - 1) * S + (KH - 1) * D + 1);
}
constexpr int max_plane(int NT, int KH, int S, int D) {
    return cmax(plane_elems(NT, 8, KH, S, D), cmax(plane_elems(NT, 16, KH, S, D), plane_elems(NT, 32, KH, S, D)));
}

template <int KH, int S, int D, int CK, int WM, int TM, int TN, int EPI, bool GATE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = TN * WN;
    constexpr int TAPS = KH * KH;
    constexpr int KC = CK * TAPS;
    constexpr int MAXPIT = (max_plane(NT, KH, S, D) + 63) / 64;
    constexpr int CPW = CK / 4;                       // channels staged per wave
    constexpr int W4 = KC * BM / 4;                   // float4 count of the weight tile
    constexpr int WIT = (W4 + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w = smem;
    float* s_in = smem + KC * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, kk = lane >> 5;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2, TH = NT * SR;
    const int LH = (TH - 1) * S + (KH - 1) * D + 1;
    const int LW = (TW - 1) * S + (KH - 1) * D + 1;
    const int plane = LH * LW;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int m0 = blockIdx.y * BM;
    const int n = blockIdx.z;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const long HWin = (long)a.H * a.W;

    // ---- per-lane staging geometry (identical for every chunk) ----
    // Loads are UNCONDITIONAL (addresses clamped into the tensor) and kept raw in registers;
    // validity masks, kscale and the SimpleGate product are applied when the registers are
    // written to LDS.  (Conditional loads make hipcc emit a branch + s_waitcnt vmcnt(0) per
    // load, which serialises the whole staging phase.)
    int gsafe[MAXPIT];
    unsigned okmask = 0, inplane = 0;
#pragma unroll
    for (int it = 0; it < MAXPIT; ++it) {
        const int p = lane + 64 * it;
        const int r = p / LW, x = p - r * LW;
        const int gy = iy0 + r, gx = ix0 + x;
        const bool inp = p < plane;
        const bool ok = inp && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        gsafe[it] = ok ? gy * a.W + gx : 0;
        okmask |= (ok ? 1u : 0u) << it;
        inplane |= (inp ? 1u : 0u) << it;
    }
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* wp_n = a.wp + (long)n * a.wp_ns;
    const float* ks_n = a.kscale ? a.kscale + (long)n * a.kscale_ns : nullptr;

    float rin[CPW][MAXPIT];
    float rin2[GATE ? CPW : 1][GATE ? MAXPIT : 1];
    float rks[CPW];
    float4 rw[WIT];
    const int nchunks = (a.Cin + CK - 1) / CK;
    const long wrow_max = (long)nchunks * KC - 1;

    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int ic = 0; ic < CPW; ++ic) {
            const int ci = min(ch * CK + wave + 4 * ic, a.Cin - 1);
            const float* base = in_n + (long)ci * HWin;
            rks[ic] = ks_n ? ks_n[ci] : 1.f;
#pragma unroll
            for (int it = 0; it < MAXPIT; ++it) {
                rin[ic][it] = base[gsafe[it]];
                if (GATE) rin2[ic][it] = base[gsafe[it] + a.gate_off];
            }
        }
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int idx = min(tid + 256 * i, W4 - 1);
            const int k = idx / (BM / 4), c4 = idx % (BM / 4);
            const int m = min(m0 + c4 * 4, a.Mpad - 4);
            rw[i] = *reinterpret_cast<const float4*>(wp_n + min((long)ch * KC + k, wrow_max) * a.Mpad + m);
        }
    };
    auto store_chunk = [&](int ch) {
#pragma unroll
        for (int ic = 0; ic < CPW; ++ic) {
            const int c = wave + 4 * ic;
            const bool cok = ch * CK + c < a.Cin;
#pragma unroll
            for (int it = 0; it < MAXPIT; ++it) {
                float v = rin[ic][it];
                if (GATE) v *= rin2[ic][it];
                v *= rks[ic];
                v = (cok && ((okmask >> it) & 1u)) ? v : 0.f;
                if ((inplane >> it) & 1u) s_in[c * plane + lane + 64 * it] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < WIT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < W4) {
                const int c4 = idx % (BM / 4);
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(s_w + idx * 4) = (m0 + c4 * 4 < a.Mpad) ? rw[i] : z;
            }
        }
    };

    // ---- fragment base addresses ----
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int t = wn * TN + tn;
        const int py = t * SR + (j >> a.tw_log2), px = j & (TW - 1);
        bbase[tn] = kk * plane + py * S * LW + px * S;
    }
    const int abase = kk * BM + wm * TM * 32 + j;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        {
            __syncthreads();
            store_chunk(ch);
            __syncthreads();
        }
        if (ch + 1 < nchunks) load_chunk(ch + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int tapoff = (tap / KH) * D * LW + (tap % KH) * D;
#pragma unroll
            for (int c2 = 0; c2 < CK / 2; ++c2) {
                float af[TM], bf[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    af[tm] = s_w[abase + (tap * CK + 2 * c2) * BM + tm * 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    bf[tn] = s_in[bbase[tn] + 2 * c2 * plane + tapoff];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }

    conv_epilogue<TM, TN, EPI>(a, acc, n, m0, wm, wn, oy0, ox0, j, kk);
}

template <int KH, int S, int D, int CK, int WM, int TM, int TN, int EPI, bool GATE>
int launch_cfg(const ConvArgs& a, int N, hipStream_t st) {
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = TN * WN;
    constexpr int KC = CK * KH * KH;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2, TH = NT * SR;
    const int LH = (TH - 1) * S + (KH - 1) * D + 1, LW = (TW - 1) * S + (KH - 1) * D + 1;
    const size_t lds = (size_t)(KC * BM + CK * LH * LW) * sizeof(float);
    ConvArgs b = a;
    b.tiles_x = tdr_cdiv(a.OW, TW);
    const int tiles_y = tdr_cdiv(a.OH, TH);
    dim3 grid(b.tiles_x * tiles_y, tdr_cdiv(a.Cout, BM), N);
    auto kern = conv_mfma_kernel<KH, S, D, CK, WM, TM, TN, EPI, GATE>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b);
    TDR_LAUNCH_CHECK("conv_mfma_kernel");
    return TDR_OK;
}

// tile configuration choice: largest tile that still yields >= ~2 blocks per CU
template <int KH, int S, int D, int CK, int EPI, bool GATE>
int launch_shape(const ConvArgs& a, int N, hipStream_t st) {
    const long pix = (long)a.OH * a.OW;
    auto blocks = [&](int bm, int bn) { return (long)tdr_cdiv(a.Cout, bm) * tdr_cdiv(pix, bn) * N; };
    if (a.Cout <= 32) {
        // few pixels (the MASA coarse search: 16 filters over a 32 x 32 map, K = 4608): 128-pixel tiles double the workgroups
        if (blocks(32, 256) < 256) return launch_cfg<KH, S, D, CK, 1, 1, 1, EPI, GATE>(a, N, st);   // 32 x 128
        return launch_cfg<KH, S, D, CK, 1, 1, 2, EPI, GATE>(a, N, st);                               // 32 x 256
    }
    if (a.Cout > 64 && blocks(128, 128) >= 512) return launch_cfg<KH, S, D, CK, 2, 2, 2, EPI, GATE>(a, N, st);
    if (blocks(64, 128) >= 512) return launch_cfg<KH, S, D, CK, 2, 1, 2, EPI, GATE>(a, N, st);  // 64 x 128
    return launch_cfg<KH, S, D, CK, 2, 1, 1, EPI, GATE>(a, N, st);                              // 64 x 64
}

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int mode, int CK,
                                    int M, int Kch, int KHe, int Mpad, long total, float* __restrict__ wp) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        wp[i] = tdr_pack_f32_elem(w, Cin, KH, mode, CK, M, Kch, KHe, Mpad, i);
}

__global__ void pack_patches_kernel(const float* __restrict__ blk, int G, int C, int BH, int BW, int PH, int PW, int pstep,
                                    int dil, int off, int CK, int Mpad, long per_b, long total, float* __restrict__ wp) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / per_b;
        long r = i % per_b;
        const int m = (int)(r % Mpad); r /= Mpad;
        const int ck = (int)(r % CK); r /= CK;
        const int tap = (int)(r % 9);
        const int chunk = (int)(r / 9);
        const int c = chunk * CK + ck;
        float v = 0.f;
        if (m < G * PH * PW && c < C) {
            const int g = m / (PH * PW), q = m % (PH * PW);
            const int py = q / PW, px = q % PW;
            const int y = py * pstep + (tap / 3) * dil + off, x = px * pstep + (tap % 3) * dil + off;
            v = blk[(((b * G + g) * C + c) * BH + y) * BW + x];
        }
        wp[i] = v;
    }
}

}  // namespace

extern "C" int tdr_conv_ck(int KH_eff) { return KH_eff == 1 ? 32 : (KH_eff == 2 ? 16 : 8); }

extern "C" int64_t tdr_packed_weight_floats(int M, int Kch, int KH_eff) {
    const int CK = tdr_conv_ck(KH_eff);
    const long Mpad = (M + 31) / 32 * 32;
    return (long)((Kch + CK - 1) / CK) * KH_eff * KH_eff * CK * Mpad;
}

extern "C" int tdr_pack_weights(const float* w, int Cout, int Cin, int KH, int mode, float* wp, void* stream) {
    TDR_REQUIRE(w && wp, "tdr_pack_weights: null pointer");
    TDR_REQUIRE(mode >= 0 && mode <= 3, "tdr_pack_weights: bad mode %d", mode);
    int M, Kch, KHe;
    if (mode == 0) { M = Cout; Kch = Cin; KHe = KH; }
    else if (mode == 1) { M = Cin; Kch = Cout; KHe = KH; }
    else if (mode == 2) { TDR_REQUIRE(KH == 2, "mode 2 needs a 2x2 kernel"); M = 4 * Cin; Kch = Cout; KHe = 1; }
    else { TDR_REQUIRE(KH == 3, "mode 3 needs a 3x3 kernel"); M = 4 * Cin; Kch = Cout; KHe = 2; }
    const int CK = tdr_conv_ck(KHe);
    const int Mpad = (M + 31) / 32 * 32;
    const long total = tdr_packed_weight_floats(M, Kch, KHe);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, KH, mode, CK,
                       M, Kch, KHe, Mpad, total, wp);
    TDR_LAUNCH_CHECK("pack_weights_kernel");
    return TDR_OK;
}

extern "C" int tdr_pack_patches(const float* blk, int B, int G, int C, int BH, int BW, int PH, int PW, int pstep, int dil,
                                int off, float* wp, void* stream) {
    TDR_REQUIRE(blk && wp, "tdr_pack_patches: null pointer");
    const int CK = 8;
    const int Mpad = (G * PH * PW + 31) / 32 * 32;
    const long per_b = (long)((C + CK - 1) / CK) * 9 * CK * Mpad;
    const long total = per_b * B;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_patches_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, blk, G, C, BH, BW, PH, PW,
                       pstep, dil, off, CK, Mpad, per_b, total, wp);
    TDR_LAUNCH_CHECK("pack_patches_kernel");
    return TDR_OK;
}

extern "C" int tdr_conv_forward(const TdrConvDesc* d, void* stream) {
    TDR_REQUIRE(d && d->in && d->wp && d->out, "tdr_conv_forward: null pointer");
    TDR_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->OH > 0 && d->OW > 0, "tdr_conv_forward: bad shape");
    TDR_REQUIRE(d->Mpad % 32 == 0 && d->Mpad >= d->Cout, "tdr_conv_forward: Mpad %d invalid for Cout %d", d->Mpad, d->Cout);
    TDR_REQUIRE(d->epi != EPI_GATEBWD || d->aux, "tdr_conv_forward: GATEBWD needs aux");
    if (d->wp_fmt >= 1 && d->wp_fmt <= 3) return tdr_conv_forward_bx3(d, stream);
    TDR_REQUIRE(d->wp_fmt == 0, "tdr_conv_forward: unknown wp_fmt %d", d->wp_fmt);
    ConvArgs a;
    a.in = d->in; a.in_ns = d->in_ns; a.Cin = d->Cin; a.H = d->H; a.W = d->W;
    a.wp = (const float*)d->wp; a.wp_ns = d->wp_ns; a.Mpad = d->Mpad; a.Cout = d->Cout;
    a.out = d->out; a.out_ns = d->out_ns; a.OH = d->OH; a.OW = d->OW;
    a.pad = d->pad;
    a.tw_log2 = d->OW >= 24 ? 5 : (d->OW >= 12 ? 4 : 3);
    a.tiles_x = 0;
    a.kscale = d->kscale; a.kscale_ns = d->kscale_ns;
    a.gate_off = (long)d->Cin * d->H * d->W;
    a.bias = d->bias; a.bias_ns = d->bias_ns; a.scale = d->scale; a.scale_ns = d->scale_ns;
    a.bias2 = d->bias2; a.bias2_ns = d->bias2_ns; a.bias2_mul = d->bias2_mul;
    a.res = d->res; a.res_ns = d->res_ns; a.mask = d->mask; a.mask_ns = d->mask_ns;
    a.aux = d->aux; a.aux_ns = d->aux_ns; a.relu = d->relu;
    a.vec_epi = 0; a.single_buf = 0; a.scheme = 0;
    hipStream_t st = (hipStream_t)stream;
    const int N = d->N;
    const int key = d->KH * 1000 + d->stride * 100 + d->dil * 10 + d->epi;
    const bool g = d->gate != 0;
    switch (key) {
        case 1110: return g ? launch_shape<1, 1, 1, 32, EPI_STD, true>(a, N, st)
                            : launch_shape<1, 1, 1, 32, EPI_STD, false>(a, N, st);
        case 1111: if (!g) return launch_shape<1, 1, 1, 32, EPI_GATEBWD, false>(a, N, st); break;
        case 1112: if (!g) return launch_shape<1, 1, 1, 32, EPI_PSHUF, false>(a, N, st); break;
        case 3110: if (!g) return launch_shape<3, 1, 1, 8, EPI_STD, false>(a, N, st); break;
        case 3112: if (!g) return launch_shape<3, 1, 1, 8, EPI_PSHUF, false>(a, N, st); break;
        case 3120: if (!g) return launch_shape<3, 1, 2, 8, EPI_STD, false>(a, N, st); break;
        case 3130: if (!g) return launch_shape<3, 1, 3, 8, EPI_STD, false>(a, N, st); break;
        case 3210: if (!g) return launch_shape<3, 2, 1, 8, EPI_STD, false>(a, N, st); break;
        case 2210: if (!g) return launch_shape<2, 2, 1, 16, EPI_STD, false>(a, N, st); break;
        case 2112: if (!g) return launch_shape<2, 1, 1, 16, EPI_PSHUF, false>(a, N, st); break;
        default: break;
    }
    tdr_set_error("tdr_conv_forward: unsupported (KH=%d stride=%d dil=%d epi=%d gate=%d)", d->KH, d->stride, d->dil,
                  d->epi, d->gate);
    return TDR_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------
// multi-tensor packing: every weight of the network for every use (forward / data-gradient layout,
// fp32 rows or split-bf16 fragments) in ONE launch per step instead of ~480 small ones.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void pack_multi_kernel(const TdrPackJob* __restrict__ jobs, int n_jobs) {
    // binary search: last job whose first_block <= blockIdx.x
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TdrPackJob jb = jobs[lo];
    const long i = ((long)blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (i >= jb.total) return;
    if (jb.fmt == 0)
        reinterpret_cast<float*>(jb.wp)[i] = tdr_pack_f32_elem(jb.w, jb.Cin, jb.KH, jb.mode, jb.CK, jb.M, jb.Kch, jb.KHe, jb.Mx, i);
    else if (jb.fmt == 1 && jb.KHe == 3 && jb.mode == 0)   // 3x3: one thread per (group, m-tile, lane), all nine taps (tdr_pack_job_init sizes `total`)
        tdr_pack_bx3_alltaps9<0>(jb.w, jb.Cin, jb.M, jb.Kch, jb.Mx, i, reinterpret_cast<uint4*>(jb.wp));
    else if (jb.fmt == 1 && jb.KHe == 3 && jb.mode == 1)
        tdr_pack_bx3_alltaps9<1>(jb.w, jb.Cin, jb.M, jb.Kch, jb.Mx, i, reinterpret_cast<uint4*>(jb.wp));
    else if (jb.fmt == 1)
        tdr_pack_bx3_frag(jb.w, jb.Cin, jb.KH, jb.mode, jb.M, jb.Kch, jb.KHe, jb.Mx, i, reinterpret_cast<uint4*>(jb.wp));
    else if (jb.mode == 0 && jb.KHe == 1)            // forward layout: one thread per (group, m-tile, lane), all taps (tdr_pack_job_init sizes `total`)
        tdr_pack_hx2_fwd_alltaps<1>(jb.w, jb.Cin, jb.M, jb.Kch, jb.Mx, i, reinterpret_cast<uint4*>(jb.wp));
    else if (jb.mode == 0 && jb.KHe == 2)
        tdr_pack_hx2_fwd_alltaps<4>(jb.w, jb.Cin, jb.M, jb.Kch, jb.Mx, i, reinterpret_cast<uint4*>(jb.wp));
    else if (jb.mode == 0 && jb.KHe == 3)
        tdr_pack_hx2_fwd_alltaps<9>(jb.w, jb.Cin, jb.M, jb.Kch, jb.Mx, i, reinterpret_cast<uint4*>(jb.wp));
    else
        tdr_pack_hx2_frag(jb.w, jb.Cin, jb.KH, jb.mode, jb.M, jb.Kch, jb.KHe, jb.Mx, i, reinterpret_cast<uint4*>(jb.wp));
}
}  // namespace

extern "C" int tdr_pack_job_init(TdrPackJob* job, const float* w, int Cout, int Cin, int KH, int mode, int fmt, void* wp) {
    TDR_REQUIRE(job && w && wp, "tdr_pack_job_init: null pointer");
    TDR_REQUIRE(mode >= 0 && mode <= 3 && fmt >= 0 && fmt <= 2, "tdr_pack_job_init: bad mode/fmt");
    int M, Kch, KHe;
    if (mode == 0) { M = Cout; Kch = Cin; KHe = KH; }
    else if (mode == 1) { M = Cin; Kch = Cout; KHe = KH; }
    else if (mode == 2) { TDR_REQUIRE(KH == 2, "mode 2 needs a 2x2 kernel"); M = 4 * Cin; Kch = Cout; KHe = 1; }
    else { TDR_REQUIRE(KH == 3, "mode 3 needs a 3x3 kernel"); M = 4 * Cin; Kch = Cout; KHe = 2; }
    job->w = w; job->wp = wp; job->Cout = Cout; job->Cin = Cin; job->KH = KH; job->mode = mode; job->fmt = fmt;
    job->M = M; job->Kch = Kch; job->KHe = KHe;
    if (fmt == 0) {
        job->CK = tdr_conv_ck(KHe);
        job->Mx = (M + 31) / 32 * 32;
        job->total = tdr_packed_weight_floats(M, Kch, KHe);
    } else {
        job->CK = 16;
        job->Mx = (M + 31) / 32;
        job->total = (long)((Kch + 15) / 16) * KHe * KHe * job->Mx * 64;
        // hx2 forward layout: one thread makes the fragments of all taps (tdr_pack_hx2_fwd_alltaps)
        if (fmt == 2 && mode == 0 && KHe >= 1 && KHe <= 3) job->total = (long)((Kch + 15) / 16) * job->Mx * 64;
        // bx3, 3x3 forward / stride-1 data-gradient layouts: likewise (tdr_pack_bx3_alltaps9)
        if (fmt == 1 && KHe == 3 && mode <= 1) job->total = (long)((Kch + 15) / 16) * job->Mx * 64;
    }
    job->first_block = 0;
    return TDR_OK;
}

extern "C" int tdr_pack_weights_multi(const TdrPackJob* jobs_dev, int n_jobs, int64_t total_blocks, void* stream) {
    TDR_REQUIRE(jobs_dev && n_jobs > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "tdr_pack_weights_multi: bad argument");
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
    TDR_LAUNCH_CHECK("pack_multi_kernel");
    return TDR_OK;
}
