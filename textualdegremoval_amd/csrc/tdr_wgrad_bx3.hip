// Weight-gradient GEMM (K = output pixels) on the gfx950 bf16 matrix cores with fp32-equivalent
// arithmetic: the 3-way bf16 split of tdr_conv_bx3.hip (x = h + m + l, six cross products per
// fp32 product, fp32 accumulation on v_mfma_f32_32x32x16_bf16).
//
//   G[co][ci][tap] = sum_{n,oy,ox} dout[n][co][oy][ox] * in[n][ci][oy+ky-pad][ox+kx-pad]      (stride 1)
//
// MFMA view: A[i=co][k=pixel] (dout), B[k=pixel][j=ci] (input, shifted per tap); a k-step is 16
// pixels, each lane half holding 8 consecutive pixels of one tile row.  One 32x32 accumulator tile
// per (co-tile, ci-tile, tap).  A workgroup (4 waves = WMw x WNw x WKw) walks P-pixel tiles
// (P/C rows x C cols, C in {8,16,32}); per tile it loads dout and the input halo tile from HBM
// (8 consecutive pixels per lane), splits them in registers and stores
//   s_d[split][co][P]            bf16, row pitch 16 B x odd  -> conflict-free ds_read_b128 A fragments
//   s_i[split][ci][rows][C+8]    bf16, channel pitch 16 B x odd
// For 3x3 the kx = 0 / 2 fragments start one bf16 before / after a 16-byte boundary: the wave reads
// the two aligned 16-byte pieces around the fragment once per (ky, split) and builds the three kx
// fragments with v_alignbit_b32 (8 VALU per 18 MFMAs).  Split-K partials go to the workspace and
// are reduced in a fixed order by the caller (tdr_wgrad_mfma.hip), so results are deterministic.
#include "tdr_common.h"
#include "tdr_wgrad_common.h"
#include "../../include/tdr.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

union WFrag {
    uint4 u;
    unsigned d[4];
    bf16x8 v;
};

__device__ __forceinline__ void wsplit3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

__device__ __forceinline__ void split8(const float (&v)[8], WFrag& h, WFrag& m, WFrag& l) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 a0, a1, a2;
        wsplit3(v[i], a0, a1, a2);
        h.v[i] = a0; m.v[i] = a1; l.v[i] = a2;
    }
}

template <int KH, int P, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE>
__global__ __launch_bounds__(256, 2) void wgrad_bx3_kernel(WgArgs a) {
    static_assert(WMw * WNw * WKw == 4, "4 waves");
    static_assert(KH == 1 || (TMW == 1 && TNW == 1), "3x3: one 32x32 tile pair (9 accumulators) per wave");
    constexpr int TAPS = KH * KH;
    constexpr int BMc = 32 * TMW * WMw, BNc = 32 * TNW * WNw;
    constexpr int HALO = KH == 3 ? 1 : 0;
    constexpr int DCH = P / 8;                       // 8-pixel chunks per dout row
    constexpr int NITD = (BMc * DCH + 255) / 256;
    constexpr int KSTEPS = P / 16;
    constexpr int KPW = KSTEPS / WKw;
    static_assert(KPW >= 1, "tile too small for the K split");
    constexpr int DPITCH = P + 8;                    // bf16 elements; (2P+16)/16 is odd for P = 32, 64, 128

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WKw, wn = (wave / WKw) % WNw, wm = wave / (WKw * WNw);
    const int j = lane & 31, kg = lane >> 5;
    const int cl = a.tw_log2, C = 1 << cl, R = P >> cl;
    const int CP = KH == 3 ? C + 8 : C;              // LDS row pitch of the input tile (elements)
    const int LR = R + 2 * HALO;
    const int NCH = (LR * CP) >> 3;                  // 8-element chunks per input channel
    const int IPITCH = (NCH | 1) << 3;               // channel pitch (elements): 16 B x odd
    __bf16* s_d = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* s_i = s_d + 3 * BMc * DPITCH;

    const int split = blockIdx.x;
    const int n = split / a.spi;
    const int t_begin = (split % a.spi) * a.tps;
    const int t_end = min(t_begin + a.tps, a.tpi);
    const int co0 = blockIdx.y * BMc, ci0 = blockIdx.z * BNc;
    const long HWin = (long)a.H * a.W, HWo = (long)a.OH * a.OW;
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* do_n = a.dout + (long)n * a.dout_ns;

    float dsum[NITD];
#pragma unroll
    for (int i = 0; i < NITD; ++i) dsum[i] = 0.f;

    const int n_iitems = BNc * NCH;

    auto stage = [&](int t) {
        const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
        const int oy0 = ty * R, ox0 = tx * C;
        // ---- dout rows
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            const int id = tid + 256 * it;
            const int col = id / DCH, ch = id - col * DCH;
            if (BMc * DCH % 256 != 0 && id >= BMc * DCH) break;
            const int row = (ch * 8) >> cl, xo = (ch * 8) & (C - 1);
            const int oy = oy0 + row, ox = ox0 + xo;
            const int co = min(co0 + col, a.Cout - 1);
            const bool rok = oy < a.OH && co0 + col < a.Cout;
            const float* src = do_n + (long)co * HWo + (long)min(oy, a.OH - 1) * a.OW;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[min(ox + i, a.OW - 1)];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = (rok && ox + i < a.OW) ? v[i] : 0.f;
                s += v[i];
            }
            dsum[it] += s;
            WFrag h, m, l;
            split8(v, h, m, l);
            uint4* dst = reinterpret_cast<uint4*>(s_d + (long)col * DPITCH + ch * 8);
            dst[0] = h.u;
            dst[(BMc * DPITCH) >> 3] = m.u;
            dst[(2 * BMc * DPITCH) >> 3] = l.u;
        }
        // ---- input halo tile
        for (int id = tid; id < n_iitems; id += 256) {
            const int cil = id / NCH, ch = id - cil * NCH;
            const int lrow = (ch * 8) / CP, c0 = ch * 8 - lrow * CP;
            const int gy = oy0 - a.pad + lrow;
            const int gx0 = KH == 3 ? ox0 + c0 - 1 - a.pad : ox0 + c0;
            const int ci = min(ci0 + cil, a.Cin - 1);
            const bool rok = gy >= 0 && gy < a.H && ci0 + cil < a.Cin;
            const float* src = in_n + (long)ci * HWin + (long)min(max(gy, 0), a.H - 1) * a.W;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gx = min(max(gx0 + i, 0), a.W - 1);
                v[i] = src[gx];
                if (GATE) v[i] *= src[gx + a.gate_off];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (rok && gx0 + i >= 0 && gx0 + i < a.W) ? v[i] : 0.f;
            WFrag h, m, l;
            split8(v, h, m, l);
            uint4* dst = reinterpret_cast<uint4*>(s_i + (long)cil * IPITCH + ch * 8);
            dst[0] = h.u;
            dst[(BNc * IPITCH) >> 3] = m.u;
            dst[(2 * BNc * IPITCH) >> 3] = l.u;
        }
    };

    f32x16 acc[TMW][TNW][TAPS];
#pragma unroll
    for (int x = 0; x < TMW; ++x)
#pragma unroll
        for (int y = 0; y < TNW; ++y)
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][t][r] = 0.f;

    constexpr int SA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int SB[6] = {0, 2, 1, 0, 1, 0};

    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        stage(t);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KPW; ++q) {
            const int u = 2 * (wk * KPW + q) + kg;                       // this lane half's 8-pixel chunk
            const int row = (u * 8) >> cl, xo = (u * 8) & (C - 1);
            WFrag af[TMW][3];
#pragma unroll
            for (int x = 0; x < TMW; ++x)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    af[x][s].u = *reinterpret_cast<const uint4*>(s_d + (long)(s * BMc + (wm * TMW + x) * 32 + j) * DPITCH + u * 8);
            if constexpr (KH == 1) {
                WFrag bf[TNW][3];
#pragma unroll
                for (int y = 0; y < TNW; ++y)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        bf[y][s].u = *reinterpret_cast<const uint4*>(s_i + (long)(s * BNc + (wn * TNW + y) * 32 + j) * IPITCH + u * 8);
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int x = 0; x < TMW; ++x)
#pragma unroll
                        for (int y = 0; y < TNW; ++y)
                            acc[x][y][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[x][SA[p]].v, bf[y][SB[p]].v, acc[x][y][0], 0, 0, 0);
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    WFrag bf[3][3];                                      // [kx][split]
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const uint4* src = reinterpret_cast<const uint4*>(s_i + (long)(s * BNc + wn * 32 + j) * IPITCH + (row + ky) * CP + xo);
                        WFrag lo, hi;
                        lo.u = src[0];
                        hi.u = src[1];
                        // LDS col c holds input x = ox0 + c - 1 - pad; tap kx reads cols xo + kx + 1 ... + 8
                        const unsigned D[6] = {lo.d[0], lo.d[1], lo.d[2], lo.d[3], hi.d[0], hi.d[1]};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            bf[0][s].d[i] = __builtin_amdgcn_alignbit(D[i + 1], D[i], 16);
                            bf[1][s].d[i] = D[i + 1];
                            bf[2][s].d[i] = __builtin_amdgcn_alignbit(D[i + 2], D[i + 1], 16);
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 6; ++p)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            acc[0][0][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][SA[p]].v, bf[kx][SB[p]].v, acc[0][0][ky * 3 + kx], 0, 0, 0);
                }
            }
        }
    }

    // ---- bias gradient partial: deterministic in-block reduction of the per-thread dout sums
    if (a.dbpart && blockIdx.z == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            const int id = tid + 256 * it;
            if (id < BMc * DCH) red[id] = dsum[it];
        }
        __syncthreads();
        if (tid < BMc && co0 + tid < a.Cout) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DCH; ++c) s += red[tid * DCH + c];
            a.dbpart[(long)split * a.Cout + co0 + tid] = s;
        }
    }
    // ---- waves that split K inside the block are summed through LDS in a fixed order (wk = 1, 2, ..)
    if constexpr (WKw > 1) {
        float* red = reinterpret_cast<float*>(smem_raw) + (wave / WKw) * (TMW * TNW * TAPS * 16 * 64);
#pragma unroll
        for (int w = 1; w < WKw; ++w) {
            __syncthreads();
            if (wk == w) {
#pragma unroll
                for (int x = 0; x < TMW; ++x)
#pragma unroll
                    for (int y = 0; y < TNW; ++y)
#pragma unroll
                        for (int t = 0; t < TAPS; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) red[(((x * TNW + y) * TAPS + t) * 16 + r) * 64 + lane] = acc[x][y][t][r];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int x = 0; x < TMW; ++x)
#pragma unroll
                    for (int y = 0; y < TNW; ++y)
#pragma unroll
                        for (int t = 0; t < TAPS; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[x][y][t][r] += red[(((x * TNW + y) * TAPS + t) * 16 + r) * 64 + lane];
            }
        }
        if (wk != 0) return;
    }
    // partial[split][co][ci][tap]
    float* part = a.part + (long)split * a.Cout * a.Cin * TAPS;
#pragma unroll
    for (int x = 0; x < TMW; ++x)
#pragma unroll
        for (int y = 0; y < TNW; ++y) {
            const int ci = ci0 + (wn * TNW + y) * 32 + j;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * TMW + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (co >= a.Cout) continue;
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) part[((long)co * a.Cin + ci) * TAPS + tap] = acc[x][y][tap][r];
            }
        }
}

template <int KH, int P, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE>
int launch_wgb(const WgArgs& a, const WgPlan& p, int N, hipStream_t st) {
    constexpr int BMc = 32 * TMW * WMw, BNc = 32 * TNW * WNw;
    const int C = 1 << a.tw_log2, R = P / C;
    const int CP = KH == 3 ? C + 8 : C, LR = KH == 3 ? R + 2 : R;
    const int ipitch = (((LR * CP) >> 3) | 1) << 3;
    const size_t lds = (size_t)(3 * BMc * (P + 8) + 3 * BNc * ipitch) * 2;
    dim3 grid(N * p.spi, tdr_cdiv(a.Cout, BMc), tdr_cdiv(a.Cin, BNc));
    auto kern = wgrad_bx3_kernel<KH, P, WMw, WNw, WKw, TMW, TNW, GATE>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad_bx3_kernel");
    return TDR_OK;
}

}  // namespace

// cfg (bx3): 0 = 1x1, waves 2x2x1, wave tile 2x2 (128 co x 128 ci), P = 32
//            1 = 1x1, waves 2x2x1, wave tile 1x1 (64 x 64), P = 64
//            2 = 1x1, waves 1x1x4 (32 x 32, K split), P = 128
//            3 = 3x3, waves 2x2x1 (64 x 64), P = 32
//            4 = 3x3, waves 1x1x4 (32 x 32, K split), P = 128
//            5 = 1x1, waves 2x1x2 (64 co x 32 ci, K split 2), P = 64
bool tdr_wgrad_bx3_supported(const TdrWgradDesc* d) {
    return d->stride == 1 && (d->KH == 1 || (d->KH == 3 && !d->gate)) && d->OW >= 8 && d->Cin >= 8;
}

WgPlan tdr_wgrad_bx3_plan(const TdrWgradDesc* d) {
    WgPlan p;
    p.tw_log2 = d->OW >= 24 ? 5 : (d->OW >= 12 ? 4 : 3);
    if (d->KH == 1) {
        if (d->Cout > 64 && d->Cin > 64) p.cfg = 0;
        else if (d->Cin <= 32 && d->Cout <= 32) p.cfg = 2;
        else if (d->Cin <= 32) p.cfg = 5;
        else p.cfg = 1;
    } else {
        p.cfg = (d->Cin <= 32 && d->Cout <= 32) ? 4 : 3;
    }
    static const int bm[6] = {128, 64, 32, 64, 32, 64}, bn[6] = {128, 64, 32, 64, 32, 32};
    static const int pp[6] = {32, 64, 128, 32, 128, 64};
    p.BMc = bm[p.cfg]; p.BNc = bn[p.cfg];
    p.WKw = 1;                                        // K-split waves are reduced inside the block
    const int P = pp[p.cfg];
    const int C = 1 << p.tw_log2, R = P / C;
    p.tiles_x = tdr_cdiv(d->OW, C);
    p.tiles_y = tdr_cdiv(d->OH, R);
    p.tpi = p.tiles_x * p.tiles_y;
    const long out_tiles = (long)tdr_cdiv(d->Cout, p.BMc) * tdr_cdiv(d->Cin, p.BNc);
    long want = 768 / out_tiles;                      // 2 blocks per CU resident, ~1.5 rounds of blocks
    if (want < 1) want = 1;
    long spi = (want + d->N - 1) / d->N;              // splits per image
    if (spi > p.tpi / 4) spi = p.tpi / 4;             // at least 4 pixel tiles per block
    if (spi < 1) spi = 1;
    p.tps = tdr_cdiv(p.tpi, spi);
    p.spi = tdr_cdiv(p.tpi, p.tps);
    return p;
}

int tdr_wgrad_bx3_launch(const WgArgs& a, const WgPlan& p, const TdrWgradDesc* d, hipStream_t st) {
    const bool g = d->gate != 0;
    switch (p.cfg) {
        case 0: return g ? launch_wgb<1, 32, 2, 2, 1, 2, 2, true>(a, p, d->N, st) : launch_wgb<1, 32, 2, 2, 1, 2, 2, false>(a, p, d->N, st);
        case 1: return g ? launch_wgb<1, 64, 2, 2, 1, 1, 1, true>(a, p, d->N, st) : launch_wgb<1, 64, 2, 2, 1, 1, 1, false>(a, p, d->N, st);
        case 2: return g ? launch_wgb<1, 128, 1, 1, 4, 1, 1, true>(a, p, d->N, st) : launch_wgb<1, 128, 1, 1, 4, 1, 1, false>(a, p, d->N, st);
        case 3: return launch_wgb<3, 32, 2, 2, 1, 1, 1, false>(a, p, d->N, st);
        case 4: return launch_wgb<3, 128, 1, 1, 4, 1, 1, false>(a, p, d->N, st);
        default: return g ? launch_wgb<1, 64, 2, 1, 2, 1, 1, true>(a, p, d->N, st) : launch_wgb<1, 64, 2, 1, 2, 1, 1, false>(a, p, d->N, st);
    }
}
