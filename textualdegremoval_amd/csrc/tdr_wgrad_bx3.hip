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
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));

namespace {

// operand schemes, as in tdr_conv_bx3.hip: 0 = 3-way bf16 split (6 products), 1 = 2-way fp16 split (3 products; both
// operands inside the fp16 range: activations, and gradients of a loss-scaled backward pass)
// WSCH_H1: head plane only, one fp16 product (plain fp16 MFMA with fp32 accumulation; TDR_MATH=h1, reduced precision)
enum { WSCH_BX3 = 0, WSCH_HX2 = 1, WSCH_H1 = 2 };

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte fragment (8 bf16) as dwords

__device__ __forceinline__ void wsplit3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

#define WSPLIT_ONE(x, i)                      \
    {                                         \
        __bf16 a0, a1, a2;                    \
        float xp__ = (x);                     \
        asm volatile("" : "+v"(xp__));        \
        wsplit3(xp__, a0, a1, a2);            \
        hv[i] = a0; mv[i] = a1; lv[i] = a2;   \
    }

constexpr int wb_max_nch(int KH, int P) {
    // 8-element chunks per input channel, maximum over the tile widths C = 32, 16, 8
    return KH == 3 ? ((P / 32 + 2) * 5 > (P / 8 + 2) * 2 ? ((P / 32 + 2) * 5 > (P / 16 + 2) * 3 ? (P / 32 + 2) * 5 : (P / 16 + 2) * 3)
                                                         : ((P / 8 + 2) * 2 > (P / 16 + 2) * 3 ? (P / 8 + 2) * 2 : (P / 16 + 2) * 3))
                   : P / 8;
}

// PIN: the operand is a product (gated input) -- see tdr_split2_f16
template <int SCH, bool PIN = true>
__device__ __forceinline__ void split8v(const f32x4& a, const f32x4& b, u32x4& h, u32x4& m, u32x4& l) {
    if constexpr (SCH == WSCH_H1) {
        wf16x8 hv;
#pragma unroll
        for (int i = 0; i < 8; ++i) hv[i] = (_Float16)(i < 4 ? a[i & 3] : b[i & 3]);
        h = __builtin_bit_cast(u32x4, hv);
        m = h;
        l = h;
    } else if constexpr (SCH == WSCH_HX2) {
        // head and residual from the same fp32 value (gated operands are products): tdr_split2_f16 pins its inputs
        unsigned h0, h1, h2, h3, m0, m1, m2, m3;
        tdr_split2_f16<PIN>(a[0], a[1], h0, m0);
        tdr_split2_f16<PIN>(a[2], a[3], h1, m1);
        tdr_split2_f16<PIN>(b[0], b[1], h2, m2);
        tdr_split2_f16<PIN>(b[2], b[3], h3, m3);
        h = (u32x4){h0, h1, h2, h3};
        m = (u32x4){m0, m1, m2, m3};
        l = m;
    } else {
        bf16x8 hv, mv, lv;
        WSPLIT_ONE(a[0], 0) WSPLIT_ONE(a[1], 1) WSPLIT_ONE(a[2], 2) WSPLIT_ONE(a[3], 3)
        WSPLIT_ONE(b[0], 4) WSPLIT_ONE(b[1], 5) WSPLIT_ONE(b[2], 6) WSPLIT_ONE(b[3], 7)
        h = __builtin_bit_cast(u32x4, hv);
        m = __builtin_bit_cast(u32x4, mv);
        l = __builtin_bit_cast(u32x4, lv);
    }
}

// PRE: the global loads of tile t+1 are issued before the MFMAs of tile t (register double buffering)
// DMA (3x3 only): the raw fp32 rows of tile t+1 are fetched by global_load_lds (LDS-DMA, no VGPRs) into a staging
// patch while tile t is on the matrix pipe, and split from there after the barrier -- HBM latency hidden without
// the register budget a prefetch would need.  Tiles that start a column are staged synchronously.
// CL = log2 of the tile width C (a template constant: the staging index arithmetic divides by C-derived sizes for
// every item of every tile; with run-time divisors that integer math out-weighed the MFMAs)
template <int KH, int P, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE, bool PRE, bool DMA, int CL, int SCH>
__global__ __launch_bounds__(256, 2) void wgrad_bx3_kernel(WgArgs a) {
    constexpr int NS = SCH == WSCH_BX3 ? 3 : (SCH == WSCH_HX2 ? 2 : 1);      // operand planes
    constexpr int NP = SCH == WSCH_BX3 ? 6 : (SCH == WSCH_HX2 ? 3 : 1);      // matrix products per fp32 product
    static_assert(!DMA || (KH == 3 && !PRE && !GATE), "DMA staging: 3x3 only");
    static_assert(WMw * WNw * WKw == 4, "4 waves");
    static_assert(KH == 1 || (TMW == 1 && TNW == 1), "3x3: one 32x32 tile pair (9 accumulators) per wave");
    constexpr int TAPS = KH * KH;
    constexpr int BMc = 32 * TMW * WMw, BNc = 32 * TNW * WNw;
    constexpr int HALO = KH == 3 ? 1 : 0;
    constexpr int DCH = P / 8;                       // 8-pixel chunks per dout row
    constexpr int NITD = (BMc * DCH + 255) / 256;
    constexpr int NITI = (BNc * (((P >> CL) + (KH == 3 ? 2 : 0)) * ((KH == 3 ? (1 << CL) + 8 : (1 << CL)) >> 3)) + 255) / 256;
    constexpr int KSTEPS = P / 16;
    constexpr int KPW = KSTEPS / WKw;
    static_assert(KPW >= 1, "tile too small for the K split");
    constexpr int DPITCH = P + 8;                    // bf16 elements; (2P+16)/16 is odd for P = 32, 64, 128

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wk = wave % WKw, wn = (wave / WKw) % WNw, wm = wave / (WKw * WNw);
    const int j = lane & 31, kg = lane >> 5;
    constexpr int cl = CL, C = 1 << cl, R = P >> cl;
    static_assert(R >= 1, "tile narrower than its width");
    constexpr int CP = KH == 3 ? C + 8 : C;          // LDS row pitch of the input tile (elements)
    constexpr int LR = R + 2 * HALO;
    constexpr int NCH = (LR * CP) >> 3;              // 8-element chunks per input channel
    constexpr int IPITCH = (NCH | 1) << 3;           // channel pitch (elements): 16 B x odd
    __bf16* s_d = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* s_i = s_d + NS * BMc * DPITCH;           // (2-byte elements: bf16 or f16 planes)
    float* raw_d = reinterpret_cast<float*>(s_i + NS * BNc * IPITCH);    // [BMc][P] fp32 (DMA only)
    float* raw_i = raw_d + BMc * P;                                       // [BNc][R][CP] fp32: the R new halo rows

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

    // ---- staging.  dout item: (co row, 8-pixel chunk); input item: (ci row, 8-column chunk of the halo tile).
    // LDS column c of the halo tile holds input x = ox0 + c - 4 (3x3, pad 1), so every 8-column chunk is two
    // 16-byte-aligned float4 loads that are either fully inside the image row or fully outside
    // (W % 4 == 0 is required by tdr_wgrad_bx3_supported).  All loads are unconditional (clamped address),
    // validity is applied when the registers are converted.
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // tile order: row-major for 1x1; column-major (walk down a 32-column strip) for 3x3, so consecutive tiles of a
    // block share input rows and only the R new halo rows are staged per tile (the LDS tile is a ring of rows)
    auto tile_origin = [&](int t, int& oy0, int& ox0, int& ty) {
        int tx;
        if (KH == 3) { tx = t / a.tiles_y; ty = t - tx * a.tiles_y; }
        else { ty = t / a.tiles_x; tx = t - ty * a.tiles_x; }
        oy0 = ty * R;
        ox0 = tx * C;
    };
    auto d_load = [&](int t, int it, f32x4& v0, f32x4& v1, int& ldsoff, bool& live, bool from_raw = false) {
        int oy0, ox0, ty;
        tile_origin(t, oy0, ox0, ty);
        const int id = tid + 256 * it;
        live = (BMc * DCH % 256 == 0) || id < BMc * DCH;
        const int idc = min(id, BMc * DCH - 1);
        const int col = idc / DCH, ch = idc - col * DCH;
        const int row = (ch * 8) >> cl, xo = (ch * 8) & (C - 1);
        const int oy = oy0 + row, ox = ox0 + xo;
        const int co = min(co0 + col, a.Cout - 1);
        const bool rok = oy < a.OH && co0 + col < a.Cout;
        const float* src = do_n + (long)co * HWo + (long)min(oy, a.OH - 1) * a.OW;
        const bool ok0 = rok && ox < a.OW, ok1 = rok && ox + 4 < a.OW;
        if (DMA && from_raw) {
            v0 = *reinterpret_cast<const f32x4*>(raw_d + idc * 8);
            v1 = *reinterpret_cast<const f32x4*>(raw_d + idc * 8 + 4);
        } else {
            v0 = *reinterpret_cast<const f32x4*>(src + (ok0 ? ox : 0));
            v1 = *reinterpret_cast<const f32x4*>(src + (ok1 ? ox + 4 : 0));
        }
        // the out-of-range masks ride in the two top bits of the LDS offset and are applied by d_store: NOTHING here consumes
        // the loaded values, so a prefetch stays in flight under the MFMAs of the current tile (with the selects here the
        // compiler waited for every load right behind its issue: vmcnt(7..0) in front of the MFMA block)
        ldsoff = (col * DPITCH + ch * 8) | (ok0 ? 1 << 30 : 0) | (ok1 ? (int)0x80000000u : 0);
    };
    auto d_store = [&](int it, const f32x4& r0, const f32x4& r1, int ldsoff_, bool live) {
        if (!live) return;
        const f32x4 v0 = (ldsoff_ >> 30) & 1 ? r0 : z4, v1 = ((unsigned)ldsoff_ >> 31) ? r1 : z4;
        const int ldsoff = ldsoff_ & 0x3fffffff;
        dsum[it] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
        u32x4 h, m, l;
        split8v<SCH, false>(v0, v1, h, m, l);          // dout: straight from the load
        u32x4* dst = reinterpret_cast<u32x4*>(s_d + ldsoff);
        dst[0] = h;
        if constexpr (NS >= 2) dst[(BMc * DPITCH) >> 3] = m;
        if constexpr (NS == 3) dst[(2 * BMc * DPITCH) >> 3] = l;
    };
    auto i_load = [&](int t, int it, f32x4& v0, f32x4& v1, f32x4& g0, f32x4& g1, int& ldsoff, bool& live, bool from_raw = false) {
        int oy0, ox0, ty;
        tile_origin(t, oy0, ox0, ty);
        // 3x3: rows 0,1 of the halo tile are the previous tile's rows R, R+1 unless this is the first tile of the
        // block or of a column
        const int lr0 = (KH == 3 && t != t_begin && ty != 0) ? 2 : 0;
        constexpr int CPR = CP >> 3;
        const int nchr = (LR - lr0) * CPR;
        const int id = tid + 256 * it;
        live = id < BNc * nchr;
        const int idc = min(id, BNc * nchr - 1);
        const int cil = idc / nchr, rem = idc - cil * nchr;
        const int lrow = lr0 + rem / CPR, c0 = (rem % CPR) * 8;
        const int slot = KH == 3 ? (oy0 + lrow) % LR : lrow;
        const int gy = oy0 - a.pad + lrow;
        const int gx0 = KH == 3 ? ox0 + c0 - 4 : ox0 + c0;
        const int ci = min(ci0 + cil, a.Cin - 1);
        const bool rok = gy >= 0 && gy < a.H && ci0 + cil < a.Cin;
        const float* src = in_n + (long)ci * HWin + (long)min(max(gy, 0), a.H - 1) * a.W;
        const bool ok0 = rok && gx0 >= 0 && gx0 < a.W, ok1 = rok && gx0 + 4 >= 0 && gx0 + 4 < a.W;
        const int o0 = ok0 ? gx0 : 0, o1 = ok1 ? gx0 + 4 : 0;
        if (DMA && from_raw) {                       // steady-state tile: items enumerate exactly the raw patch
            v0 = *reinterpret_cast<const f32x4*>(raw_i + idc * 8);
            v1 = *reinterpret_cast<const f32x4*>(raw_i + idc * 8 + 4);
        } else {
            v0 = *reinterpret_cast<const f32x4*>(src + o0);
            v1 = *reinterpret_cast<const f32x4*>(src + o1);
        }
        if (GATE) {                                  // (the gate product is formed by i_store, like the masks)
            g0 = *reinterpret_cast<const f32x4*>(src + o0 + a.gate_off);
            g1 = *reinterpret_cast<const f32x4*>(src + o1 + a.gate_off);
        }
        ldsoff = (cil * IPITCH + slot * CP + c0) | (ok0 ? 1 << 30 : 0) | (ok1 ? (int)0x80000000u : 0);     // masks applied by i_store (see d_load)
    };
    auto i_store = [&](const f32x4& r0, const f32x4& r1, const f32x4& g0, const f32x4& g1, int ldsoff_, bool live) {
        if (!live) return;
        const f32x4 v0 = (ldsoff_ >> 30) & 1 ? (GATE ? r0 * g0 : r0) : z4, v1 = ((unsigned)ldsoff_ >> 31) ? (GATE ? r1 * g1 : r1) : z4;
        const int ldsoff = ldsoff_ & 0x3fffffff;
        u32x4 h, m, l;
        split8v<SCH, GATE>(v0, v1, h, m, l);           // gated input = a product: pinned
        u32x4* dst = reinterpret_cast<u32x4*>(s_i + ldsoff);
        dst[0] = h;
        if constexpr (NS >= 2) dst[(BNc * IPITCH) >> 3] = m;
        if constexpr (NS == 3) dst[(2 * BNc * IPITCH) >> 3] = l;
    };
    // prefetch registers (PRE only): every index below is a compile-time constant after unrolling
    f32x4 pd0[PRE ? NITD : 1], pd1[PRE ? NITD : 1], pi0[PRE ? NITI : 1], pi1[PRE ? NITI : 1];
    f32x4 pg0[PRE && GATE ? NITI : 1], pg1[PRE && GATE ? NITI : 1];
    int pdo[PRE ? NITD : 1], pio[PRE ? NITI : 1];
    bool pdl[PRE ? NITD : 1], pil[PRE ? NITI : 1];
    auto prefetch = [&](int t) {
        if constexpr (PRE) {
#pragma unroll
            for (int it = 0; it < NITD; ++it) d_load(t, it, pd0[it], pd1[it], pdo[it], pdl[it]);
#pragma unroll
            for (int it = 0; it < NITI; ++it) i_load(t, it, pi0[it], pi1[it], pg0[GATE ? it : 0], pg1[GATE ? it : 0], pio[it], pil[it]);
        }
    };
    auto commit = [&]() {
        if constexpr (PRE) {
#pragma unroll
            for (int it = 0; it < NITD; ++it) d_store(it, pd0[it], pd1[it], pdo[it], pdl[it]);
#pragma unroll
            for (int it = 0; it < NITI; ++it) i_store(pi0[it], pi1[it], pg0[GATE ? it : 0], pg1[GATE ? it : 0], pio[it], pil[it]);
        }
    };
    auto stage_sync = [&](int t, bool from_raw = false) {   // load + convert + store, item by item
#pragma unroll
        for (int it = 0; it < NITD; ++it) {
            f32x4 v0, v1; int o; bool live;
            d_load(t, it, v0, v1, o, live, from_raw);
            d_store(it, v0, v1, o, live);
        }
#pragma unroll
        for (int it = 0; it < NITI; ++it) {
            f32x4 v0, v1, g0 = z4, g1 = z4; int o; bool live;
            i_load(t, it, v0, v1, g0, g1, o, live, from_raw);
            i_store(v0, v1, g0, g1, o, live);
        }
    };
    // a tile whose rows 0,1 are already in the LDS ring (not the first of the block / of a column)
    auto steady = [&](int t) {
        int oy0, ox0, ty;
        tile_origin(t, oy0, ox0, ty);
        return KH == 3 && t != t_begin && ty != 0;
    };
    // LDS-DMA of the raw fp32 operands of a steady tile: 16-byte pieces, lane-linear destination
    auto dma_issue = [&](int t) {
        if constexpr (DMA) {
            int oy0, ox0, ty;
            tile_origin(t, oy0, ox0, ty);
            constexpr int DP = BMc * P / 4;                       // dout pieces
#pragma unroll
            for (int k = 0; k < (DP + 255) / 256; ++k) {
                const int id = tid + 256 * k;
                if (DP % 256 == 0 || id < DP) {
                    const int col = id / (P / 4), c4 = id - col * (P / 4);
                    const int row = (c4 * 4) >> cl, xo = (c4 * 4) & (C - 1);
                    const int oy = min(oy0 + row, a.OH - 1), ox = ox0 + xo;
                    const float* g = do_n + (long)min(co0 + col, a.Cout - 1) * HWo + (long)oy * a.OW + (ox < a.OW ? ox : 0);
                    float* l = raw_d + __builtin_amdgcn_readfirstlane((wave * 64 + 256 * k) * 4);
                    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)g,
                                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
                }
            }
            constexpr int CPQ = CP >> 2, per_c = R * CPQ, IP = BNc * per_c;   // input pieces: [ci][R new rows][CP/4]
            for (int k = 0; k * 256 < IP; ++k) {
                const int id = tid + 256 * k;
                if (id < IP) {
                    const int cil = id / per_c, rem = id - cil * per_c;
                    const int lrow = 2 + rem / CPQ, q = rem % CPQ;
                    const int gy = min(max(oy0 - a.pad + lrow, 0), a.H - 1), gx = ox0 - 4 + 4 * q;
                    const float* g = in_n + (long)min(ci0 + cil, a.Cin - 1) * HWin + (long)gy * a.W + ((gx >= 0 && gx < a.W) ? gx : 0);
                    float* l = raw_i + __builtin_amdgcn_readfirstlane((wave * 64 + 256 * k) * 4);
                    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)g,
                                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
                }
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

    // bx3: lh hl mm mh hm hh; hx2: mh hm hh (small cross terms first)
    constexpr int SA[6] = {SCH == WSCH_HX2 ? 1 : (SCH == WSCH_H1 ? 0 : 2), 0, SCH == WSCH_HX2 ? 0 : 1, 1, 0, 0};
    constexpr int SB[6] = {0, SCH == WSCH_HX2 ? 1 : (SCH == WSCH_H1 ? 0 : 2), SCH == WSCH_HX2 ? 0 : 1, 0, 1, 0};
    auto mma = [](const u32x4& x, const u32x4& y, const f32x16& c) {
        if constexpr (SCH != WSCH_BX3)
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8, x), __builtin_bit_cast(wf16x8, y), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    };

    if (PRE && t_begin < t_end) prefetch(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA pieces have landed
        __syncthreads();
        if constexpr (PRE) commit();
        else if constexpr (DMA) stage_sync(t, steady(t));
        else stage_sync(t);
        __syncthreads();
        if (PRE && t + 1 < t_end) prefetch(t + 1);
        if (DMA && t + 1 < t_end && steady(t + 1)) dma_issue(t + 1);
        int ring0 = 0;
        if (KH == 3) { int ox0_, ty_; tile_origin(t, ring0, ox0_, ty_); }
#pragma unroll(KH == 3 ? 1 : KPW)
        for (int q = 0; q < KPW; ++q) {
            const int u = 2 * (wk * KPW + q) + kg;                       // this lane half's 8-pixel chunk
            const int row = (u * 8) >> cl, xo = (u * 8) & (C - 1);
            u32x4 af[TMW][NS];
#pragma unroll
            for (int x = 0; x < TMW; ++x)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    af[x][s] = *reinterpret_cast<const u32x4*>(s_d + (long)(s * BMc + (wm * TMW + x) * 32 + j) * DPITCH + u * 8);
            if constexpr (KH == 1) {
                u32x4 bf[TNW][NS];
#pragma unroll
                for (int y = 0; y < TNW; ++y)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        bf[y][s] = *reinterpret_cast<const u32x4*>(s_i + (long)(s * BNc + (wn * TNW + y) * 32 + j) * IPITCH + u * 8);
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int x = 0; x < TMW; ++x)
#pragma unroll
                        for (int y = 0; y < TNW; ++y) {
                            acc[x][y][0] = mma(af[x][SA[p]], bf[y][SB[p]], acc[x][y][0]);
                        }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    u32x4 bf[3][NS];                                     // [kx][split]
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const u32x4* src = reinterpret_cast<const u32x4*>(s_i + (long)(s * BNc + wn * 32 + j) * IPITCH + ((ring0 + row + ky) % LR) * CP + xo);
                        const u32x4 lo = src[0], hi = src[1];
                        // LDS col c holds input x = ox0 + c - 4; tap kx reads cols xo + kx + 3 ... + 10 (pad = 1):
                        // dwords D0..D6 = lo[0..3], hi[0..2]
                        bf[1][s] = (u32x4){lo[2], lo[3], hi[0], hi[1]};
                        bf[0][s] = (u32x4){__builtin_amdgcn_alignbit(lo[2], lo[1], 16), __builtin_amdgcn_alignbit(lo[3], lo[2], 16),
                                           __builtin_amdgcn_alignbit(hi[0], lo[3], 16), __builtin_amdgcn_alignbit(hi[1], hi[0], 16)};
                        bf[2][s] = (u32x4){__builtin_amdgcn_alignbit(lo[3], lo[2], 16), __builtin_amdgcn_alignbit(hi[0], lo[3], 16),
                                           __builtin_amdgcn_alignbit(hi[1], hi[0], 16), __builtin_amdgcn_alignbit(hi[2], hi[1], 16)};
                    }
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[0][0][ky * 3 + kx] = mma(af[0][SA[p]], bf[kx][SB[p]], acc[0][0][ky * 3 + kx]);
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

template <int KH, int P, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE, bool PRE, bool DMA, int CL, int SCH>
int launch_wgb_cl_s(const WgArgs& a, const WgPlan& p, int N, hipStream_t st) {
    constexpr int NS = SCH == WSCH_BX3 ? 3 : (SCH == WSCH_HX2 ? 2 : 1);
    constexpr int BMc = 32 * TMW * WMw, BNc = 32 * TNW * WNw;
    constexpr int C = 1 << CL, R = P / C;
    constexpr int CP = KH == 3 ? C + 8 : C, LR = KH == 3 ? R + 2 : R;
    constexpr int ipitch = (((LR * CP) >> 3) | 1) << 3;
    size_t lds = (size_t)(NS * BMc * (P + 8) + NS * BNc * ipitch) * 2 + (DMA ? (size_t)(BMc * P + BNc * R * CP) * 4 : 0);
    const size_t red = (size_t)(WKw > 1 ? (4 / WKw) * TMW * TNW * KH * KH * 16 * 64 : BMc * (P / 8)) * 4;   // in-block reductions reuse the tile memory
    if (lds < red) lds = red;
    dim3 grid(N * p.spi, tdr_cdiv(a.Cout, BMc), tdr_cdiv(a.Cin, BNc));
    auto kern = wgrad_bx3_kernel<KH, P, WMw, WNw, WKw, TMW, TNW, GATE, PRE, DMA, CL, SCH>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad_bx3_kernel");
    return TDR_OK;
}

template <int KH, int P, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE, bool PRE, bool DMA, int CL>
int launch_wgb_cl(const WgArgs& a, const WgPlan& p, int N, hipStream_t st) {
    if (a.scheme == WSCH_HX2) return launch_wgb_cl_s<KH, P, WMw, WNw, WKw, TMW, TNW, GATE, PRE, DMA, CL, WSCH_HX2>(a, p, N, st);
    if (a.scheme == WSCH_H1) return launch_wgb_cl_s<KH, P, WMw, WNw, WKw, TMW, TNW, GATE, PRE, DMA, CL, WSCH_H1>(a, p, N, st);
    return launch_wgb_cl_s<KH, P, WMw, WNw, WKw, TMW, TNW, GATE, PRE, DMA, CL, WSCH_BX3>(a, p, N, st);
}

template <int KH, int P, int WMw, int WNw, int WKw, int TMW, int TNW, bool GATE, bool PRE, bool DMA = false>
int launch_wgb(const WgArgs& a, const WgPlan& p, int N, hipStream_t st) {
    if (a.tw_log2 == 5) return launch_wgb_cl<KH, P, WMw, WNw, WKw, TMW, TNW, GATE, PRE, DMA, 5>(a, p, N, st);
    if (a.tw_log2 == 4) return launch_wgb_cl<KH, P, WMw, WNw, WKw, TMW, TNW, GATE, PRE, DMA, 4>(a, p, N, st);
    return launch_wgb_cl<KH, P, WMw, WNw, WKw, TMW, TNW, GATE, PRE, DMA, 3>(a, p, N, st);
}

}  // namespace

// cfg (bx3): 0 = 1x1, waves 2x2x1, wave tile 2x2 (128 co x 128 ci), P = 32
//            1 = 1x1, waves 2x2x1, wave tile 1x1 (64 x 64), P = 64
//            2 = 1x1, waves 1x1x4 (32 x 32, K split), P = 128
//            3 = 3x3, waves 2x2x1 (64 x 64), P = 32
//            4 = 3x3, waves 1x1x4 (32 x 32, K split), P = 128
//            5 = 1x1, waves 2x1x2 (64 co x 32 ci, K split 2), P = 64
bool tdr_wgrad_bx3_supported(const TdrWgradDesc* d) {
    // float4 staging: rows must be 16-byte aligned pieces (W % 4 == 0), 3x3 only with pad 1
    if (d->stride != 1 || d->OW < 8 || d->W % 4 != 0 || d->OW % 4 != 0) return false;
    if (d->in_ns % 4 != 0 || d->dout_ns % 4 != 0) return false;
    if (d->KH == 1) return d->pad == 0;
    return d->KH == 3 && d->pad == 1 && !d->gate;
}

WgPlan tdr_wgrad_bx3_plan(const TdrWgradDesc* d) {
    WgPlan p;
    p.tw_log2 = d->OW >= 24 ? 5 : (d->OW >= 12 ? 4 : 3);
    if (d->KH == 1) {
        static const int force1 = tdr_tune_env("TDR_WGB_CFG1X1") ? atoi(tdr_tune_env("TDR_WGB_CFG1X1")) : -1;   // tuning aid: 0 | 1 | 5
        if (force1 >= 0 && d->Cout >= 64 && d->Cin >= 64) p.cfg = force1;
        else if (d->Cout > 64 && d->Cin > 64) p.cfg = 0;
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
    // split-K so that ONE round of blocks fills the chip (2 resident workgroups x 256 CUs): 768 (1.5 rounds) leaves a
    // half-empty tail round on every launch -- 94.7 vs 90.4 ms per cfg2 step; TDR_WG_WANT overrides (tuning aid)
    // (3-way bf16 split, round 5: 256 -- a block's matrix phase is twice as long there, half the split-K partials win: 68.8 -> 68.2 ms same box)
    static const long want_env = tdr_tune_env("TDR_WG_WANT") ? atol(tdr_tune_env("TDR_WG_WANT")) : 0;
    const long want_total = want_env > 0 ? want_env : (d->math == 1 ? 256 : 512);
    long want = want_total / out_tiles;
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
        case 0: return g ? launch_wgb<1, 32, 2, 2, 1, 2, 2, true, true>(a, p, d->N, st) : launch_wgb<1, 32, 2, 2, 1, 2, 2, false, true>(a, p, d->N, st);
        case 1: return g ? launch_wgb<1, 64, 2, 2, 1, 1, 1, true, true>(a, p, d->N, st) : launch_wgb<1, 64, 2, 2, 1, 1, 1, false, true>(a, p, d->N, st);
        case 2: return g ? launch_wgb<1, 128, 1, 1, 4, 1, 1, true, true>(a, p, d->N, st) : launch_wgb<1, 128, 1, 1, 4, 1, 1, false, true>(a, p, d->N, st);
        case 3: return launch_wgb<3, 32, 2, 2, 1, 1, 1, false, false, true>(a, p, d->N, st);
        case 4: return launch_wgb<3, 128, 1, 1, 4, 1, 1, false, false>(a, p, d->N, st);
        default: return g ? launch_wgb<1, 64, 2, 1, 2, 1, 1, true, true>(a, p, d->N, st) : launch_wgb<1, 64, 2, 1, 2, 1, 1, false, true>(a, p, d->N, st);
    }
}
