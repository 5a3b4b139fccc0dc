// EXPERIMENT (round 2, not part of libtdr_hip.so): producer / consumer ("warp-specialised") variant of the 3x3 implicit GEMM.
// Result on MI355X (profiles/probe_conv3x3_ws.py; bit-identical outputs to conv_bx3_kernel on all seven shapes):
//   128->128 @128x128 N=8: 188 us vs 127 us (conv_bx3_kernel); 256->256 @64x64: 172 vs 128; 512->512 @32x32: 190 vs 140.
// clock64() timeline of one workgroup (1.9 GHz): first tap of a launch 20 k cycles (cold weight fragments, all CUs at once),
// steady-state tap 1.25 - 1.8 k cycles against 768 of MFMA issue (one MFMA wave per SIMD and a 256-VGPR budget: the compiler
// re-uses B-fragment registers right after the MFMA that read them, so LDS latency is exposed several times per tap),
// epilogue of a 128 x 256 tile 36 k cycles with every CU in the same phase (33 MB burst at 1.8 TB/s), and after it the
// weight-fragment waits of the next tile queue behind the tile's stores (stores and loads share the in-order vmcnt).
// What the kernel needs to win: two de-phased tiles per CU (one storing while the other owns the matrix pipe) and two
// MFMA waves per SIMD -- 12 waves x 168 VGPRs with 64-register accumulators.  Kept here as the starting point for that.
// 3x3 / stride-1 implicit GEMM on the 2-way fp16 split with PRODUCER / CONSUMER waves (gfx950).
//
// conv_bx3_kernel (tdr_conv_bx3.hip) lets every wave do everything: fetch the fp32 halo tile of the next 16-channel
// group from HBM, stream the packed weight fragments from L2, split, write LDS, run the MFMAs.  Vector-memory results
// return in order (one vmcnt per wave), so a wave that has the long-latency operand loads of group g+1 in flight
// cannot wait for the short-latency weight fragments of the next tap without also waiting for HBM: once per group the
// matrix pipe of that SIMD idles for an HBM round trip, and the C >= 128 pyramid levels run at 40 - 55 % of the
// MFMA-only rate of the same instruction stream.  Here the two streams live in different waves:
//
//   waves 4-7 (producers): global loads of group k+2 -> registers, split + ds_write of group k+1 -> LDS buffer (k+1)&1
//   waves 0-3 (consumers): weight-fragment ring (3 taps ahead, L2 latency only) + ds_read_b128 + 3 MFMAs per product
//
// one s_barrier per group couples them.  One workgroup (8 waves, 256 VGPRs each) per CU; a workgroup walks its tiles
// persistently, so the producers already hold the first two groups of the next tile while the consumers write a
// tile's epilogue.  Same LDS operand layout, packed weights, accumulator layout and epilogues as conv_bx3_kernel --
// the arithmetic (products, their order, the fp32 accumulation order over groups and taps) is identical, results are
// bit-identical to that kernel.
#include <stdlib.h>
#include "tdr_common.h"
#include "tdr_conv_epi.h"

#ifndef WS_AD
#define WS_AD 2
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union FragW {
    uint4 u;
    f16x8 hv;
};

constexpr int ws_cmax(int a, int b) { return a > b ? a : b; }
constexpr int ws_plane(int NT, int TW) { return ((NT * 32 / TW) + 2) * (TW + 2); }
constexpr int ws_max_plane(int NT) { return ws_cmax(ws_plane(NT, 8), ws_cmax(ws_plane(NT, 16), ws_plane(NT, 32))); }

// tile t of the launch: m-tile major, then image, then pixel tile; XCD k (workgroup id % 8) walks a contiguous range
// of that sequence, so an XCD's L2 holds the weights of one or two m-tiles and neighbouring pixel tiles share halo lines
struct TileGeo {
    int n, m0, oy0, ox0;
};

template <int WM, int TM, int TN, int TWL>
__global__ __launch_bounds__(512, 1) void conv3x3_ws_kernel(ConvArgs a, int N, int ptiles, int T) {
    constexpr int NS = 2, TAPS = 9, AD = WS_AD;
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = TN * WN;
    constexpr int TW = 1 << TWL, SR = 32 >> TWL, TH = NT * SR;
    constexpr int LH = TH + 2, LW = TW + 2;
    constexpr int plane = LH * LW;
    constexpr int NIT = (plane + 127) / 128;

    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int w4 = wave & 3;
    const int ngroups = (a.Cin + 15) >> 4;
    const int G = gridDim.x;
    const int nloc = (T - (int)blockIdx.x + G - 1) / G;          // tiles of this workgroup
    const int total = nloc * ngroups;                            // (tile, group) steps

    auto tile_geo = [&](int i) {
        const int vb = blockIdx.x + i * G;                       // G % 8 == 0 or G == T: the XCD of a virtual id is the real one
        const int q = T >> 3, r = T & 7, xcd = vb & 7, slot = vb >> 3;
        int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        logical = min(logical, T - 1);
        TileGeo t;
        const int mtile = logical / (N * ptiles), rem = logical - mtile * (N * ptiles);
        t.n = rem / ptiles;
        const int ptile = rem - t.n * ptiles;
        const int ty = ptile / a.tiles_x, tx = ptile - ty * a.tiles_x;
        t.m0 = mtile * BM;
        t.oy0 = ty * TH;
        t.ox0 = tx * TW;
        return t;
    };

    if (producer) {
        // ---- staging: waves {4,5} stage channels 0-7 of the group, waves {6,7} channels 8-15
        const int sg = w4 >> 1;
        const int sp0 = (w4 & 1) * 64 + lane;
        unsigned inplane = 0;
        int prow[NIT], pcol[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int p = sp0 + 128 * it;
            prow[it] = p / LW;
            pcol[it] = p - prow[it] * LW;
            inplane |= (p < plane ? 1u : 0u) << it;
        }
        float rin[2][NIT][8];
        unsigned okm[2] = {0, 0};
        int gsafe[NIT];
        const float* in_n = a.in;
        int geo_tile = -1;
        unsigned okcur = 0;
        auto load_step = [&](int seq, int set) {
            const int i = seq / ngroups, g = seq - i * ngroups;
            if (i != geo_tile) {
                geo_tile = i;
                const TileGeo t = tile_geo(i);
                in_n = a.in + (long)t.n * a.in_ns;
                okcur = 0;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int gy = t.oy0 - 1 + prow[it], gx = t.ox0 - 1 + pcol[it];
                    const bool ok = ((inplane >> it) & 1u) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                    gsafe[it] = ok ? gy * a.W + gx : 0;
                    okcur |= (ok ? 1u : 0u) << it;
                }
            }
            okm[set] = okcur;
            const long HWin = (long)a.H * a.W;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int ci = min(g * 16 + sg * 8 + c, a.Cin - 1);
                const float* base = in_n + (long)ci * HWin;
#pragma unroll
                for (int it = 0; it < NIT; ++it) rin[set][it][c] = base[gsafe[it]];
            }
        };
        auto store_step = [&](int seq, int set) {
            const int g = seq % ngroups;
            uint4* sb = smem4 + ((seq & 1) * (2 * NS) + sg) * plane;
            const int cbase = g * 16 + sg * 8;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                FragW h, m;
                const bool ok = (okm[set] >> it) & 1u;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float v = (ok && cbase + c < a.Cin) ? rin[set][it][c] : 0.f;
                    asm volatile("" : "+v"(v));                  // one fp32 value for head and residual (tdr_conv_bx3.hip)
                    const _Float16 hh = (_Float16)v;
                    h.hv[c] = hh;
                    m.hv[c] = (_Float16)(v - (float)hh);
                }
                if ((inplane >> it) & 1u) {
                    const int p = sp0 + 128 * it;
                    sb[p] = h.u;
                    sb[2 * plane + p] = m.u;
                }
            }
        };
        load_step(0, 0);
        if (total > 1) load_step(1, 1);
        store_step(0, 0);
        __syncthreads();
        for (int k0 = 0; k0 < total; k0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = k0 + u;
                if (k < total) {
                    if (k + 2 < total) load_step(k + 2, u);          // set u held step k, already in LDS
                    if (k + 1 < total) store_step(k + 1, u ^ 1);
                    __syncthreads();
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------- consumers
    const int wm = w4 / WN, wn = w4 % WN;
    const int j = lane & 31, kk = lane >> 5;
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int t = wn * TN + tn;
        const int py = t * SR + (j >> TWL), px = j & (TW - 1);
        bbase[tn] = kk * plane + py * LW + px;
    }
    const int MT = a.Mpad >> 5;
    const long wstep = (long)MT * (NS * 64);                     // 16-byte units per (group, tap)
    const int GT = ngroups * TAPS;
    // weight fragments: wave-uniform 16-byte index (scalar registers) + lane
    const uint4* const wpl = reinterpret_cast<const uint4*>(a.wp) + lane;
    auto wbase = [&](const TileGeo& t, int tm) {
        const int mt = min((t.m0 >> 5) + wm * TM + tm, MT - 1);
        return (long)t.n * (a.wp_ns >> 2) + (long)mt * (NS * 64);
    };
    float* const sw = reinterpret_cast<float*>(smem4 + 2 * (2 * NS) * plane) + w4 * (32 * 36);   // vector-epilogue patch of this wave

    f32x16 acc[TM][TN];
    FragW aq[AD][TM][NS];
    TileGeo cur = tile_geo(0);
    long wf[TM], wfn[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) wf[tm] = wbase(cur, tm);
#pragma unroll
    for (int d = 0; d < AD; ++d)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int s = 0; s < NS; ++s) aq[d][tm][s].u = wpl[wf[tm] + (long)min(d, GT - 1) * wstep + s * 64];
    __syncthreads();

    for (int i = 0; i < nloc; ++i) {
        const TileGeo nxt = tile_geo(min(i + 1, nloc - 1));
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) wfn[tm] = wbase(nxt, tm);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

        for (int g = 0; g < ngroups; ++g) {
            const uint4* sb = smem4 + ((i * ngroups + g) & 1) * (2 * NS) * plane;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int tapoff = (tap / 3) * LW + (tap % 3);
                FragW bf[TN][NS];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int s = 0; s < NS; ++s) bf[tn][s].u = sb[s * 2 * plane + bbase[tn] + tapoff];
                constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};          // mh hm hh: small cross terms first
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[tap % AD][tm][HA[q]].hv, bf[tn][HB[q]].hv, acc[tm][tn], 0, 0, 0);
                // the slot is consumed: request the fragments AD steps ahead into it (the first steps of the next tile at the end)
                {
                    const int nx = g * TAPS + tap + AD;
                    const bool wrap = nx >= GT;
                    const long idx = (long)(wrap ? min(nx - GT, GT - 1) : nx) * wstep;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        const long p = (wrap ? wfn[tm] : wf[tm]) + idx;
#pragma unroll
                        for (int s = 0; s < NS; ++s) aq[tap % AD][tm][s].u = wpl[p + s * 64];
                    }
                }
            }
            if (g == ngroups - 1) {
                conv_epilogue_vec<TM, TN, EPI_STD>(a, acc, cur.n, cur.m0, wm, wn, cur.oy0, cur.ox0, lane, sw);
            }
            __syncthreads();
        }
        cur = nxt;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) wf[tm] = wfn[tm];
    }
}

template <int WM, int TM, int TN, int TWL>
int launch_ws_t(const ConvArgs& a0, int N, hipStream_t st) {
    constexpr int WN = 4 / WM, BM = 32 * TM * WM, NT = TN * WN;
    ConvArgs a = a0;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2, TH = NT * SR;
    const int plane = (TH + 2) * (TW + 2);
    a.tiles_x = tdr_cdiv(a.OW, TW);
    const int ptiles = a.tiles_x * tdr_cdiv(a.OH, TH);
    a.mtiles = tdr_cdiv(a.Cout, BM);
    const int T = a.mtiles * N * ptiles;
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 8 ? n & ~7 : 8;
    }();
    // every workgroup the same number of tiles where possible (one CU = one workgroup; a ragged last round idles CUs)
    const int rounds = tdr_cdiv(T, cus);
    int G = tdr_cdiv(T, rounds);
    G = (G + 7) & ~7;                                   // the XCD of a virtual workgroup id must be the real one
    if (G >= T) G = T;
    const size_t lds = (size_t)2 * 4 * plane * 16 + 4 * 32 * 36 * sizeof(float);
    auto kern = conv3x3_ws_kernel<WM, TM, TN, TWL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, st, a, N, ptiles, T);
    TDR_LAUNCH_CHECK("conv3x3_ws_kernel");
    return TDR_OK;
}

template <int WM, int TM, int TN>
int launch_ws(const ConvArgs& a, int N, hipStream_t st) {
    return a.tw_log2 == 5 ? launch_ws_t<WM, TM, TN, 5>(a, N, st) : launch_ws_t<WM, TM, TN, 4>(a, N, st);
}

}  // namespace

// conv_args: the ConvArgs block of tdr_conv_forward_bx3 (same layout in both translation units).  Returns
// TDR_ERR_UNSUPPORTED without touching anything when the shape is not one this kernel is built for.
int tdr_conv3x3_ws_launch(const void* conv_args, int N, void* stream) {
    const ConvArgs& a = *static_cast<const ConvArgs*>(conv_args);
    static const int off = getenv("TDR_CONV_WS") ? !atoi(getenv("TDR_CONV_WS")) : 0;
    if (off || a.scheme != 1 || !a.vec_epi || a.kscale || a.pad != 1 || a.H != a.OH || a.W != a.OW || a.Cin % 16 || a.Cout % 128 || a.Cin < 64 || a.tw_log2 < 4) return TDR_ERR_UNSUPPORTED;
    const long pix = (long)a.OH * a.OW;
    hipStream_t st = (hipStream_t)stream;
    const long t256 = (long)(a.Cout / 128) * N * tdr_cdiv(pix, 256);
    if (t256 >= 256) return launch_ws<2, 2, 4>(a, N, st);      // 128 x 256
    return launch_ws<2, 2, 2>(a, N, st);                        // 128 x 128
}
