// Weight-gradient GEMM on the gfx950 fp32 matrix cores (K = output pixels).
//
//   G[co][ci][tap] = sum_{n,oy,ox} dout[n][co][oy][ox] * B[n][ci][oy*S+ky-pad][ox*S+kx-pad]
//
// MFMA view: A[i=co][k=pixel] (dout), B[k=pixel][j=ci] (input, shifted per tap),
// one 32x32 accumulator tile per (co-tile, ci-tile, tap).  A workgroup (4 waves)
// walks 64-pixel tiles (64/TW rows x TW cols); per tile it stages
//   s_d[BMc][65]        dout rows (pitch 65 -> conflict-free A reads across co)
//   s_i[BNc][plane|1]   input halo tiles (odd pitch -> conflict-free B reads across ci)
// through registers (next tile's global loads in flight under the MFMAs).
// Waves are arranged WMw x WNw x WKw: for narrow layers (32 channels) the four
// waves split the K (pixel) range instead of the output tile.  Split-K partials
// go to a workspace and are reduced in a fixed order (deterministic).
#include <stdio.h>
#include <stdlib.h>
#include "tdr_common.h"
#include "tdr_wgrad_common.h"
#include "../../include/tdr.h"

namespace {

constexpr int wg_plane(int TW, int KH, int S) { return (((64 / TW) - 1) * S + KH) * ((TW - 1) * S + KH); }
constexpr int wg_cmax(int a, int b) { return a > b ? a : b; }
constexpr int wg_max_plane(int KH, int S) {
    return wg_cmax(wg_plane(8, KH, S), wg_cmax(wg_plane(16, KH, S), wg_plane(32, KH, S)));
}

template <int KH, int S, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgArgs a) {
    static_assert(WMw * WNw * WKw == 4, "4 waves");
    constexpr int TAPS = KH * KH;
    constexpr int BMc = 32 * TMW * WMw, BNc = 32 * TNW * WNw;
    constexpr int MAXPIT = (wg_max_plane(KH, S) + 63) / 64;
    constexpr int DPT = BMc / 4;          // dout rows per wave
    constexpr int IPT = BNc / 4;          // input channels per wave
    constexpr int KPW = 32 / WKw;         // k-pairs per wave per tile

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WKw, wn = (wave / WKw) % WNw, wm = wave / (WKw * WNw);
    const int j = lane & 31, kk = lane >> 5;
    const int TW = 1 << a.tw_log2;
    const int LH = ((64 >> a.tw_log2) - 1) * S + KH, LW = (TW - 1) * S + KH;
    const int plane = LH * LW, planeP = plane | 1;
    float* s_d = smem;
    float* s_i = smem + BMc * 65;

    const int split = blockIdx.x;
    const int n = split / a.spi;
    const int t_begin = (split % a.spi) * a.tps;
    const int t_end = min(t_begin + a.tps, a.tpi);
    const int co0 = blockIdx.y * BMc, ci0 = blockIdx.z * BNc;
    const long HWin = (long)a.H * a.W, HWo = (long)a.OH * a.OW;
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* do_n = a.dout + (long)n * a.dout_ns;

    // lane's pixel inside the 64-pixel tile (dout staging) and halo positions (input staging)
    const int dpy = lane >> a.tw_log2, dpx = lane & (TW - 1);
    int pr[MAXPIT], pc[MAXPIT];
#pragma unroll
    for (int it = 0; it < MAXPIT; ++it) {
        const int p = lane + 64 * it;
        pr[it] = p < plane ? p / LW : -1;
        pc[it] = p - (p / LW) * LW;
    }

    // unconditional (clamped) loads kept raw in registers; masks and the gate product are applied
    // at LDS-store time (conditional loads would be serialised by per-load vmcnt(0) waits).
    float rd[DPT];
    float rdsum[DPT];
#pragma unroll
    for (int i = 0; i < DPT; ++i) rdsum[i] = 0.f;
    float ri[IPT][MAXPIT];
    float ri2[GATE ? IPT : 1][GATE ? MAXPIT : 1];
    bool dok_r = false;
    unsigned iok_r = 0;
    auto load_tile = [&](int t) {
        const int ty = t / a.tiles_x, tx = t % a.tiles_x;
        const int oy0 = ty * (64 >> a.tw_log2), ox0 = tx * TW;
        const int oy = oy0 + dpy, ox = ox0 + dpx;
        dok_r = oy < a.OH && ox < a.OW;
        const long doff = dok_r ? (long)oy * a.OW + ox : 0;
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const int co = min(co0 + wave + 4 * i, a.Cout - 1);
            rd[i] = do_n[(long)co * HWo + doff];
        }
        const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
        iok_r = 0;
#pragma unroll
        for (int it = 0; it < MAXPIT; ++it) {
            const int gy = iy0 + pr[it], gx = ix0 + pc[it];
            const bool ok = pr[it] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            iok_r |= (ok ? 1u : 0u) << it;
            const long off = ok ? (long)gy * a.W + gx : 0;
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int ci = min(ci0 + wave + 4 * i, a.Cin - 1);
                ri[i][it] = in_n[(long)ci * HWin + off];
                if (GATE) ri2[i][it] = in_n[(long)ci * HWin + off + a.gate_off];
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const float dv = (dok_r && co0 + wave + 4 * i < a.Cout) ? rd[i] : 0.f;
            s_d[(wave + 4 * i) * 65 + lane] = dv;
            rdsum[i] += dv;
        }
#pragma unroll
        for (int it = 0; it < MAXPIT; ++it)
            if (pr[it] >= 0) {
#pragma unroll
                for (int i = 0; i < IPT; ++i) {
                    float v = ri[i][it];
                    if (GATE) v *= ri2[i][it];
                    s_i[(wave + 4 * i) * planeP + lane + 64 * it] =
                        (((iok_r >> it) & 1u) && ci0 + wave + 4 * i < a.Cin) ? v : 0.f;
                }
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

    const int abase = (wm * TMW * 32 + j) * 65 + kk;
    const int bbase = (wn * TNW * 32 + j) * planeP + kk * S;

    if (t_begin < t_end) load_tile(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t + 1 < t_end) load_tile(t + 1);
#pragma unroll
        for (int q = 0; q < KPW; ++q) {
            const int k2 = wk * KPW + q;
            const int p0 = 2 * k2;
            const int py0 = p0 >> a.tw_log2, px0 = p0 & (TW - 1);
            const int pixoff = py0 * S * LW + px0 * S;
            float af[TMW];
#pragma unroll
            for (int x = 0; x < TMW; ++x) af[x] = s_d[abase + x * 32 * 65 + p0];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int toff = pixoff + (tap / KH) * LW + (tap % KH);
#pragma unroll
                for (int y = 0; y < TNW; ++y) {
                    const float bf = s_i[bbase + y * 32 * planeP + toff];
#pragma unroll
                    for (int x = 0; x < TMW; ++x)
                        acc[x][y][tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[x], bf, acc[x][y][tap], 0, 0, 0);
                }
            }
        }
    }

    if (a.dbpart && blockIdx.z == 0) {
#pragma unroll
        for (int i = 0; i < DPT; ++i) {
            const float sv = wave_sum(rdsum[i]);
            const int co = co0 + wave + 4 * i;
            if (lane == 0 && co < a.Cout) a.dbpart[(long)split * a.Cout + co] = sv;
        }
    }
    // partial[(split*WKw + wk)][co][ci][tap]
    float* part = a.part + ((long)split * WKw + wk) * a.Cout * a.Cin * TAPS;
#pragma unroll
    for (int x = 0; x < TMW; ++x)
#pragma unroll
        for (int y = 0; y < TNW; ++y) {
            const int ci = ci0 + (wn * TNW + y) * 32 + j;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * TMW + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (co >= a.Cout) continue;
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) part[((long)co * a.Cin + ci) * TAPS + tap] = acc[x][y][tap][r];
            }
        }
}

// out[g][e] = sum_{s < per_group} part[(g*per_group + s)][e]; block = 64 elements x KL partial-lanes.
// Blocks with blockIdx.x >= nb_main run the second (bias-gradient) job in the same launch.
template <int KL>
__global__ __launch_bounds__(64 * KL) void wgrad_reduce_kernel(const float* __restrict__ part, long elems, int per_group,
                                                               float* __restrict__ out, int nb_main,
                                                               const float* __restrict__ part2, long elems2, int per_group2,
                                                               float* __restrict__ out2) {
    __shared__ float red[KL][64];
    const int lane = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const bool second = (int)blockIdx.x >= nb_main;
    if (second && blockIdx.y != 0) return;
    const float* pbase = second ? part2 : part;
    const long ne = second ? elems2 : elems;
    const int pg = second ? per_group2 : per_group;
    float* o = second ? out2 : out;
    const long e = ((int)blockIdx.x - (second ? nb_main : 0)) * 64L + lane;
    const long ec = e < ne ? e : ne - 1;
    const float* p = pbase + (second ? 0L : (long)blockIdx.y * pg * ne) + ec;
    // eight partials in flight per thread: the two-at-a-time loop was a chain of dependent ~1 us loads (9.4 us per launch whatever
    // the size, 224 launches per step); fixed order of additions, so still deterministic
    float s0 = 0.f;
    for (int k = kl; k < pg; k += 8 * KL) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k + u * KL < pg ? p[(long)(k + u * KL) * ne] : 0.f;
        s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    red[kl][lane] = s0;
    __syncthreads();
    if (kl == 0 && e < ne) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < KL; ++q) t += red[q][lane];
        o[(second ? 0L : (long)blockIdx.y * ne) + e] = t;
    }
}

// cfg: 0 = waves 2x2x1 tile 2x2 (128x128) | 1 = waves 1x1x4 tile 2x1 (64x32) | 2 = 2x2x1 tile 1x1 (64x64)
//      3 = waves 1x1x4 tile 1x1 (32x32)
WgPlan make_plan(const TdrWgradDesc* d) {
    if (tdr_wgrad_1x1_supported(d)) return tdr_wgrad_1x1_plan(d);
    if (d->math >= 1 && tdr_wgrad_bx3_supported(d)) return tdr_wgrad_bx3_plan(d);
    if (tdr_wgrad_s2_supported(d)) return tdr_wgrad_s2_plan(d);
    WgPlan p;
    p.tw_log2 = d->OW >= 24 ? 5 : (d->OW >= 12 ? 4 : 3);
    const int TW = 1 << p.tw_log2, TH = 64 >> p.tw_log2;
    p.tiles_x = tdr_cdiv(d->OW, TW);
    p.tiles_y = tdr_cdiv(d->OH, TH);
    p.tpi = p.tiles_x * p.tiles_y;
    if (d->KH == 1) {
        if (d->Cout >= 128 && d->Cin >= 128) p.cfg = 0;
        else if (d->Cin <= 32) p.cfg = d->Cout <= 32 ? 3 : 1;
        else p.cfg = 2;
    } else if (d->KH == 3 && d->stride == 1) {
        p.cfg = (d->Cin <= 32 && d->Cout <= 32) ? 3 : 2;
    } else {
        p.cfg = 2;
    }
    static const int bm[4] = {128, 64, 64, 32}, bn[4] = {128, 32, 64, 32}, wk[4] = {1, 4, 1, 4};
    p.BMc = bm[p.cfg]; p.BNc = bn[p.cfg]; p.WKw = wk[p.cfg];
    const long out_tiles = (long)tdr_cdiv(d->Cout, p.BMc) * tdr_cdiv(d->Cin, p.BNc);
    long want = (p.WKw > 1 ? 256 : 512) / out_tiles;  // ~1-2 blocks per CU in total; K-split waves write WKw partials each
    if (want < 1) want = 1;
    long spi = (want + d->N - 1) / d->N;              // splits per image
    if (spi > p.tpi / 4) spi = p.tpi / 4;             // at least 4 pixel tiles per block
    if (spi < 1) spi = 1;
    p.tps = tdr_cdiv(p.tpi, spi);
    p.spi = tdr_cdiv(p.tpi, p.tps);
    return p;
}

template <int KH, int S, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE>
int launch_wg(const WgArgs& a, const WgPlan& p, int N, hipStream_t st) {
    constexpr int BMc = 32 * TMW * WMw, BNc = 32 * TNW * WNw;
    const int TW = 1 << a.tw_log2;
    const int LH = ((64 >> a.tw_log2) - 1) * S + KH, LW = (TW - 1) * S + KH;
    const size_t lds = (size_t)(BMc * 65 + BNc * ((LH * LW) | 1)) * sizeof(float);
    dim3 grid(N * p.spi, tdr_cdiv(a.Cout, BMc), tdr_cdiv(a.Cin, BNc));
    auto kern = wgrad_mfma_kernel<KH, S, WMw, WNw, WKw, TMW, TNW, GATE>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad_mfma_kernel");
    return TDR_OK;
}

}  // namespace

extern "C" int64_t tdr_wgrad_ws_floats(const TdrWgradDesc* d) {
    const WgPlan p = make_plan(d);
    return (int64_t)d->N * p.spi * p.WKw * d->Cout * d->Cin * d->KH * d->KH + (int64_t)d->N * p.spi * d->Cout;
}

extern "C" int tdr_conv_wgrad(const TdrWgradDesc* d, void* stream) {
    TDR_REQUIRE(d && d->in && d->dout && d->g && d->ws, "tdr_conv_wgrad: null pointer");
    const WgPlan p = make_plan(d);
    const int64_t need = tdr_wgrad_ws_floats(d);
    TDR_REQUIRE(d->ws_floats >= need, "tdr_conv_wgrad: workspace too small (%lld < %lld)", (long long)d->ws_floats,
                (long long)need);
    WgArgs a;
    a.in = d->in; a.in_ns = d->in_ns; a.Cin = d->Cin; a.H = d->H; a.W = d->W;
    a.gate_off = (long)d->Cin * d->H * d->W;
    a.dout = d->dout; a.dout_ns = d->dout_ns; a.Cout = d->Cout; a.OH = d->OH; a.OW = d->OW;
    a.grp_tab = nullptr; a.grp_bpp = 0; a.grp_pairs = 0;
    a.pad = d->pad; a.tw_log2 = p.tw_log2; a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.tpi = p.tpi; a.tps = p.tps; a.spi = p.spi;
    // a single partial per output group (per-image Grams / grouped GEMMs with one split, or one image with one split) IS the
    // result: the kernel writes it straight to g and the fixed-order reduction (here a 131 MB copy for the 20 x 1280 x 1280
    // weight gradients of a grouped Mapper layer) is skipped
    const bool direct = !d->db && (d->per_image ? p.spi : d->N * p.spi) * p.WKw == 1;
    a.part = direct ? d->g : d->ws;
    a.dbpart = d->db ? d->ws + (int64_t)d->N * p.spi * p.WKw * d->Cout * d->Cin * d->KH * d->KH : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const bool g = d->gate != 0;
    int rc = TDR_ERR_UNSUPPORTED;
    const int key = d->KH * 10 + d->stride;
    a.scheme = d->math == 3 ? 2 : (d->math == 2 ? 1 : 0);   // 3: plain fp16 (TDR_MATH=h1)
    static const bool dbg = tdr_tune_env("TDR_WG_DEBUG") != nullptr;   // which shapes miss the split kernel
    if (dbg && !(d->math >= 1 && tdr_wgrad_bx3_supported(d)) && !tdr_wgrad_s2_supported(d))
        fprintf(stderr, "[tdr] exact wgrad: math %d N %d %d->%d @%dx%d k%d s%d pad %d gate %d per_image %d in_ns %ld dout_ns %ld\n", d->math,
                d->N, d->Cin, d->Cout, d->H, d->W, d->KH, d->stride, d->pad, d->gate, d->per_image, (long)d->in_ns, (long)d->dout_ns);
    if (tdr_wgrad_1x1_supported(d)) {
        rc = tdr_wgrad_1x1_launch(a, p, d, st);
    } else if (d->math >= 1 && tdr_wgrad_bx3_supported(d)) {
        rc = tdr_wgrad_bx3_launch(a, p, d, st);
    } else if (tdr_wgrad_s2_supported(d)) {
        rc = tdr_wgrad_s2_launch(a, p, d, st);
    } else if (key == 11) {
        switch (p.cfg) {
            case 0: rc = g ? launch_wg<1, 1, 2, 2, 1, 2, 2, true>(a, p, d->N, st) : launch_wg<1, 1, 2, 2, 1, 2, 2, false>(a, p, d->N, st); break;
            case 1: rc = g ? launch_wg<1, 1, 1, 1, 4, 2, 1, true>(a, p, d->N, st) : launch_wg<1, 1, 1, 1, 4, 2, 1, false>(a, p, d->N, st); break;
            case 2: rc = g ? launch_wg<1, 1, 2, 2, 1, 1, 1, true>(a, p, d->N, st) : launch_wg<1, 1, 2, 2, 1, 1, 1, false>(a, p, d->N, st); break;
            default: rc = g ? launch_wg<1, 1, 1, 1, 4, 1, 1, true>(a, p, d->N, st) : launch_wg<1, 1, 1, 1, 4, 1, 1, false>(a, p, d->N, st); break;
        }
    } else if (key == 31 && !g) {
        rc = p.cfg == 3 ? launch_wg<3, 1, 1, 1, 4, 1, 1, false>(a, p, d->N, st) : launch_wg<3, 1, 2, 2, 1, 1, 1, false>(a, p, d->N, st);
    } else if (key == 32 && !g) {
        rc = launch_wg<3, 2, 2, 2, 1, 1, 1, false>(a, p, d->N, st);
    } else if (key == 22 && !g) {
        rc = launch_wg<2, 2, 2, 2, 1, 1, 1, false>(a, p, d->N, st);
    } else {
        tdr_set_error("tdr_conv_wgrad: unsupported (KH=%d stride=%d gate=%d)", d->KH, d->stride, d->gate);
        return TDR_ERR_UNSUPPORTED;
    }
    if (rc != TDR_OK || direct) return rc;
    const long elems = (long)d->Cout * d->Cin * d->KH * d->KH;
    const int groups = d->per_image ? d->N : 1;
    const int per_group = (d->per_image ? p.spi : d->N * p.spi) * p.WKw;
    const int nb_main = tdr_cdiv(elems, 64);
    const int nb2 = d->db ? tdr_cdiv(d->Cout, 64) : 0;
    const int pg2 = d->N * p.spi;
    const dim3 rgrid(nb_main + nb2, groups);
    if (per_group <= 8 && pg2 <= 8)
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, rgrid, dim3(64), 0, st, d->ws, elems, per_group, d->g, nb_main, a.dbpart,
                           (long)d->Cout, pg2, d->db);
    else if (per_group <= 64 && pg2 <= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, rgrid, dim3(256), 0, st, d->ws, elems, per_group, d->g, nb_main, a.dbpart,
                           (long)d->Cout, pg2, d->db);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, rgrid, dim3(1024), 0, st, d->ws, elems, per_group, d->g, nb_main, a.dbpart,
                           (long)d->Cout, pg2, d->db);
    TDR_LAUNCH_CHECK("wgrad_reduce_kernel");
    return TDR_OK;
}
