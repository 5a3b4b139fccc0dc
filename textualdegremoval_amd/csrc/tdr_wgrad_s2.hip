// Weight gradient of the STRIDE-2 convolutions (H = 2 OH, W = 2 OW) on the split-operand matrix cores (3-way bf16: the default
// arithmetic, round 5; 2-way fp16: the fast mode it was written for in round 4): 3x3 pad 1 -- the four level
// transitions conv_L2 .. conv_L5 of the MASA encoder (reference models/archs/network_nafnet_guided_arch.py:122-128) -- and
// 2x2 pad 0 -- the `downs` of the U-Net (:434-437).  They were the last dense weight gradients of the step on the exact-fp32
// kernel (tdr_wgrad_mfma.hip: 4 + 4 launches, 1.4 + 0.35 ms).  Described for 3x3; the 2x2 variant has no halo and no shifted tap.
//
//   G[co][ci][ky][kx] = sum_{n,oy,ox} dout[n][co][oy][ox] * in[n][ci][2 oy + ky - 1][2 ox + kx - 1]
//
// MFMA view as in tdr_wgrad_bx3.hip: A[i = co][k = output pixel] (dout), B[k][j = ci] (input, per tap), 32x32x16 f16, three
// products per fp32 product (mh, hm, hh), fp32 accumulate.  What stride 2 changes is the B operand: 8 consecutive output
// pixels of a tap are 8 input columns two apart.  The staging therefore DE-INTERLEAVES every input row by column parity
// while it splits it:
//     E[e] = in[2 (ox0 + e)]         (kx = 1)
//     O[o] = in[2 (ox0 + o) + 1]     (kx = 2 reads O[ox], kx = 0 reads O[ox - 1])
// so the kx = 1 / 2 fragments are aligned 16-byte LDS reads and kx = 0 is the kx = 2 piece shifted by one element
// (v_alignbit with the dword in front of it).  A tile is one output row x 32 columns (two k-steps); a block walks DOWN a
// 32-column strip, so of the three input rows of a tile only two are new: the LDS tile is a ring of five input rows --
// the two rows of tile t+1 are converted and written while tile t is on the matrix pipe (they never alias the three rows
// tile t reads), one barrier per tile; the global loads of tile t+2 are in flight in registers meanwhile.
//
// Workgroup = 2 (co halves) x WN (ci halves) x 3 (ky) waves: a wave owns one 32 x 32 (co, ci) tile of ONE kernel row (three
// accumulators, 48 VGPRs) -- 12 waves per CU instead of the 4 x 144-register waves of the stride-1 kernel, so the
// conversion VALU of some waves runs under the MFMAs of the others.  Split-K partials go to the workspace in the layout of
// the other weight-gradient kernels ([split][co][ci][tap]) and are reduced in a fixed order by tdr_conv_wgrad.
#include "tdr_common.h"
#include "tdr_wgrad_common.h"
#include "../../include/tdr.h"
#include <stdlib.h>

typedef _Float16 s2f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s2bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned s2u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned s2u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int S2_BM = 64;        // co rows per block
constexpr int S2_C = 32;         // output columns per tile
constexpr int S2_PL = 40;        // elements per parity plane of one input row: idx 8 + e, e in [-4, 32)
constexpr int S2_RP = 2 * S2_PL; // one input row: E plane | O plane
// ring of input rows: 5 (3x3: rows 2 oy - 1 .. 2 oy + 3 live at once) / 4 (2x2: 2 oy .. 2 oy + 3); channel pitch (elements) = slots * 80
// rounded up to an odd number of 16-byte pieces: 408 = 51 x 8 / 328 = 41 x 8
constexpr int s2_slots(int KH) { return KH == 3 ? 5 : 4; }
constexpr int s2_ip(int KH) { return KH == 3 ? 408 : 328; }
constexpr int S2_DP = 40;        // dout row pitch (elements): 5 x 16 bytes

// Operand schemes (WgArgs.scheme): S2_BX3 = 0: x = h + m + l, bf16 each, 6 products (24-bit operands, fp32 range: TDR_MATH=bx3, the
// default arithmetic); S2_HX2 = 1: x = h + m, fp16 each, 3 products (fp16 window: loss-scaled backward); S2_H1 = 2: one fp16 plane.
enum { S2_BX3 = 0, S2_HX2 = 1, S2_H1 = 2 };
constexpr int s2_ns(int SCH) { return SCH == S2_BX3 ? 3 : (SCH == S2_HX2 ? 2 : 1); }
// four fp32 values -> their NS packed planes (tdr_common.h: tdr_split2_f16 / tdr_split3_bf16; operands straight from buffer loads: no pin)
template <int SCH>
__device__ __forceinline__ void s2_split4(float x0, float x1, float x2, float x3, s2u32x2 (&p)[s2_ns(SCH)]) {
    if constexpr (SCH == S2_BX3) {
        unsigned h0, m0, l0, h1, m1, l1;
        tdr_split3_bf16<false>(x0, x1, h0, m0, l0);
        tdr_split3_bf16<false>(x2, x3, h1, m1, l1);
        p[0] = (s2u32x2){h0, h1}; p[1] = (s2u32x2){m0, m1}; p[2] = (s2u32x2){l0, l1};
    } else {
        unsigned h0, m0, h1, m1;
        tdr_split2_f16<false>(x0, x1, h0, m0);
        tdr_split2_f16<false>(x2, x3, h1, m1);
        p[0] = (s2u32x2){h0, h1};
        if constexpr (SCH == S2_HX2) p[1] = (s2u32x2){m0, m1};
    }
}

// KH = 3 (pad 1) or 2 (pad 0: the 2x2 stride-2 `downs` of the U-Net, :434-437 -- rows 2 oy, 2 oy + 1, columns 2 ox (E) / 2 ox + 1 (O), no halo)
template <int KH, int WN, int SCH>
__global__ __launch_bounds__(128 * KH * WN) void wgrad_s2_kernel(WgArgs a) {
    constexpr bool H1 = SCH == S2_H1;
    constexpr int NT = 128 * KH * WN;
    constexpr int PAD = KH == 3 ? 1 : 0;
    constexpr int NCH = KH == 3 ? 9 : 8;                   // 8-column chunks per staged input row
    constexpr int S2_SLOTS = s2_slots(KH), S2_IP = s2_ip(KH);
    constexpr int NS = s2_ns(SCH);
    constexpr int BN = 32 * WN;
    constexpr int NITI = (BN * KH * NCH + NT - 1) / NT;    // input items per thread, non-steady tile (KH rows x NCH chunks per channel)
    constexpr int DBUF = NS * S2_BM * S2_DP;               // elements of one dout buffer

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* s_d = reinterpret_cast<_Float16*>(smem_raw);                 // [2][NS][BM][DP]
    _Float16* s_i = s_d + 2 * DBUF;                                        // [NS][BN][IP]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kyw = wave % KH, wn = (wave / KH) % WN, wm = wave / (KH * WN);
    const int j = lane & 31, kg = lane >> 5;

    const int split = blockIdx.x;
    const int n = split / a.spi;
    const int t_begin = (split % a.spi) * a.tps;
    const int t_end = min(t_begin + a.tps, a.tpi);
    const int co0 = blockIdx.y * S2_BM, ci0 = blockIdx.z * BN;
    const long HWin = (long)a.H * a.W, HWo = (long)a.OH * a.OW;
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* do_n = a.dout + (long)n * a.dout_ns;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // raw buffer descriptors of the image's two operands: a masked piece (outside the image, channel beyond the layer) is a load
    // at an offset beyond num_records, which returns zeros -- no select on the 16 loaded values, 32-bit offsets
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_n), 0, (int)(a.Cin * HWin * 4), 0x00020000);
    const auto rs_do = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(do_n), 0, (int)(a.Cout * HWo * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto bload = [](decltype(rs_in) rs, unsigned off) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    };

    // tiles in column-major order: t -> (tx, ty = output row)

    // ---- dout: item = (co row, 8-pixel chunk), threads 0..255
    const int dcol = (tid >> 2) & 63, dch = tid & 3;
    const bool dthread = tid < 256;
    float dsum = 0.f;
    const bool d_rok = co0 + dcol < a.Cout;
    const unsigned d_base = (unsigned)(((co0 + dcol) * (int)HWo + dch * 8) * 4);      // per-thread part of the dout offset
    auto d_load = [&](int tx, int oy, f32x4& v0, f32x4& v1) {
        const int ox = tx * S2_C + dch * 8;
        const bool rok = d_rok;
        const unsigned base = d_base + (unsigned)((oy * a.OW + tx * S2_C) * 4);
        v0 = bload(rs_do, rok && ox < a.OW ? base : OOB);
        v1 = bload(rs_do, rok && ox + 4 < a.OW ? base + 16 : OOB);
    };
    auto d_store = [&](int buf, const f32x4& v0, const f32x4& v1) {
        dsum += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
        s2u32x2 p0[NS], p1[NS];
        s2_split4<SCH>(v0[0], v0[1], v0[2], v0[3], p0);
        s2_split4<SCH>(v1[0], v1[1], v1[2], v1[3], p1);
        _Float16* dst = s_d + buf * DBUF + dcol * S2_DP + dch * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<s2u32x4*>(dst + s * S2_BM * S2_DP) = (s2u32x4){p0[s][0], p0[s][1], p1[s][0], p1[s][1]};
    };

    // ---- input: item = (ci row, input row r of the tile's three, 8-column chunk q of nine); chunk q holds input columns
    // 2 ox0 - 8 + 8 q .. + 7, i.e. plane elements e = o = -4 + 4 q .. + 3 (LDS idx 4 + 4 q ..)
    auto i_load = [&](int tx, int oy, bool steady, int it, f32x4& v0, f32x4& v1, int& ldsoff) {
        const int lr0 = (KH == 3 && steady) ? 1 : 0;       // 3x3: row 2 oy - 1 of a steady tile is the previous tile's row 2 oy + 1
        const int per = (KH - lr0) * NCH;
        const int nitems = BN * per;
        if (NT * it >= nitems) { ldsoff = 0; return; }      // (block-uniform) no item of this round is live
        const int id = tid + NT * it;
        const bool live = id < nitems;
        const int idc = min(id, nitems - 1);
        const int cil = lr0 ? idc / ((KH - 1) * NCH) : idc / (KH * NCH);
        const int rem = idc - cil * per;
        const int r = lr0 + rem / NCH, q = rem % NCH;
        const int gy = 2 * oy - PAD + r;
        const int gx0 = 2 * tx * S2_C - (KH == 3 ? 8 : 0) + 8 * q;
        const int slot = (gy + S2_SLOTS) % S2_SLOTS;
        const bool rok = gy >= 0 && gy < a.H && ci0 + cil < a.Cin;
        const unsigned base = (unsigned)(((ci0 + cil) * (int)HWin + gy * a.W + gx0) * 4);
        v0 = bload(rs_in, rok && gx0 >= 0 && gx0 < a.W ? base : OOB);
        v1 = bload(rs_in, rok && gx0 + 4 >= 0 && gx0 + 4 < a.W ? base + 16 : OOB);
        // liveness rides in a top bit of the offset: nothing here consumes the loaded values
        ldsoff = (cil * S2_IP + slot * S2_RP + (KH == 3 ? 4 : 8) + 4 * q) | (live ? 1 << 30 : 0);
    };
    auto i_store = [&](const f32x4& v0, const f32x4& v1, int ldsoff_) {
        if (!((ldsoff_ >> 30) & 1)) return;
        const int off = ldsoff_ & 0x0fffffff;
        s2u32x2 pe[NS], po[NS];
        s2_split4<SCH>(v0[0], v0[2], v1[0], v1[2], pe);
        s2_split4<SCH>(v0[1], v0[3], v1[1], v1[3], po);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            *reinterpret_cast<s2u32x2*>(s_i + s * BN * S2_IP + off) = pe[s];
            *reinterpret_cast<s2u32x2*>(s_i + s * BN * S2_IP + off + S2_PL) = po[s];
        }
    };

    // ---- steady tiles (every tile but the first of a block / of a column strip): which (channel, row, chunk) a thread stages does not
    // change from tile to tile, only the row index does -- the item descriptors are computed once per column strip and a steady load is
    // a handful of VALU (the generic path above re-derives everything per item: ~100 instructions, more than the split itself)
    constexpr int LR0S = KH == 3 ? 1 : 0;
    constexpr int PERS = (KH - LR0S) * NCH;                // items per channel of a steady tile (two new rows)
    constexpr int NITS = (BN * PERS + NT - 1) / NT;
    unsigned sd_gofs[NITS];
    int sd_lds[NITS], sd_fl[NITS];                          // flags: 1 = first half inside, 2 = second half inside, 4 = live, 8 = second row
    auto make_desc = [&](int tx) {
#pragma unroll
        for (int it = 0; it < NITS; ++it) {
            const int id = tid + NT * it;
            const bool live = id < BN * PERS;
            const int idc = min(id, BN * PERS - 1);
            const int cil = idc / PERS, rem = idc - cil * PERS;
            const int r2 = rem / NCH, q = rem - r2 * NCH;
            const int gx0 = 2 * tx * S2_C - (KH == 3 ? 8 : 0) + 8 * q;
            const bool chok = ci0 + cil < a.Cin;
            sd_gofs[it] = (unsigned)(((ci0 + cil) * (int)HWin + gx0) * 4);
            sd_lds[it] = cil * S2_IP + (KH == 3 ? 4 : 8) + 4 * q;
            sd_fl[it] = ((chok && gx0 >= 0 && gx0 < a.W) ? 1 : 0) | ((chok && gx0 + 4 >= 0 && gx0 + 4 < a.W) ? 2 : 0) | (live ? 4 : 0) | (r2 ? 8 : 0);
        }
    };
    auto i_load_steady = [&](int oy, int it, f32x4& v0, f32x4& v1, int& ldsoff) {
        if (NT * it >= BN * PERS) { ldsoff = 0; return; }
        const int gyA = 2 * oy - PAD + LR0S;                // the two new rows gyA, gyA + 1 (>= 0 in a steady tile); uniform
        const int slotA = (gyA + S2_SLOTS) % S2_SLOTS, slotB = (gyA + 1 + S2_SLOTS) % S2_SLOTS;
        const unsigned rowA = (unsigned)(gyA * a.W * 4), rowB = rowA + (unsigned)(a.W * 4);
        const bool okA = gyA < a.H, okB = gyA + 1 < a.H;
        const int fl = sd_fl[it];
        const bool second = (fl & 8) != 0;
        const bool rok = second ? okB : okA;
        const unsigned base = sd_gofs[it] + (second ? rowB : rowA);
        v0 = bload(rs_in, (rok && (fl & 1)) ? base : OOB);
        v1 = bload(rs_in, (rok && (fl & 2)) ? base + 16 : OOB);
        ldsoff = (sd_lds[it] + (second ? slotB : slotA) * S2_RP) | ((fl & 4) ? 1 << 30 : 0);
    };

    // prefetch registers of one tile
    f32x4 pd0 = z4, pd1 = z4, pi0[NITI], pi1[NITI];
    int pio[NITI];
    auto prefetch = [&](int tx, int oy) {                  // steady <=> not the first tile of a column strip (callers stage the block's first tile by prefetch0)
#if defined(S2_ABL) && S2_ABL == 3
        return;
#endif
        if (dthread) d_load(tx, oy, pd0, pd1);
        if (oy != 0) {
#pragma unroll
            for (int it = 0; it < NITI; ++it) {
                if (it < NITS) i_load_steady(oy, it, pi0[it], pi1[it], pio[it]);
                else pio[it] = 0;
            }
        } else {
#pragma unroll
            for (int it = 0; it < NITI; ++it) i_load(tx, oy, false, it, pi0[it], pi1[it], pio[it]);
            make_desc(tx);                                  // a new column strip: descriptors of its steady tiles
        }
    };
    auto prefetch0 = [&](int tx, int oy) {                 // first tile of the block: all rows, wherever in the strip it is
        if (dthread) d_load(tx, oy, pd0, pd1);
#pragma unroll
        for (int it = 0; it < NITI; ++it) i_load(tx, oy, false, it, pi0[it], pi1[it], pio[it]);
        make_desc(tx);
    };
    auto commit = [&](int buf) {
#if defined(S2_ABL) && S2_ABL == 2
        return;
#endif
        if (dthread) d_store(buf, pd0, pd1);
#pragma unroll
        for (int it = 0; it < NITI; ++it) i_store(pi0[it], pi1[it], pio[it]);
    };

    f32x16 acc[KH];
#pragma unroll
    for (int kx = 0; kx < KH; ++kx)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kx][r] = 0.f;

    auto mma = [](const s2u32x4& x, const s2u32x4& y, const f32x16& c) {
        if constexpr (SCH == S2_BX3)
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s2bf16x8, x), __builtin_bit_cast(s2bf16x8, y), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s2f16x8, x), __builtin_bit_cast(s2f16x8, y), c, 0, 0, 0);
    };
    auto mfma_tile = [&](int oy, int buf) {
#if defined(S2_ABL) && S2_ABL == 1
        return;
#endif
        const int slot = (2 * oy - PAD + kyw + S2_SLOTS) % S2_SLOTS;
        const _Float16* sd = s_d + buf * DBUF + (wm * 32 + j) * S2_DP;
        const _Float16* si = s_i + (wn * 32 + j) * S2_IP + slot * S2_RP;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = 2 * q + kg;                       // this lane half's 8-pixel chunk
            s2u32x4 af[NS], bf[KH][NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                af[s] = *reinterpret_cast<const s2u32x4*>(sd + s * S2_BM * S2_DP + 8 * u);
                const _Float16* b = si + s * BN * S2_IP;
                const s2u32x4 E = *reinterpret_cast<const s2u32x4*>(b + 8 + 8 * u);
                const s2u32x4 O = *reinterpret_cast<const s2u32x4*>(b + S2_PL + 8 + 8 * u);
                if constexpr (KH == 3) {
                    const unsigned pv = *reinterpret_cast<const unsigned*>(b + S2_PL + 8 * u + 6);   // O[8u - 2], O[8u - 1]
                    bf[1][s] = E;
                    bf[2][s] = O;
                    bf[0][s] = (s2u32x4){__builtin_amdgcn_alignbit(O[0], pv, 16), __builtin_amdgcn_alignbit(O[1], O[0], 16),
                                         __builtin_amdgcn_alignbit(O[2], O[1], 16), __builtin_amdgcn_alignbit(O[3], O[2], 16)};
                } else {
                    bf[0][s] = E;                           // kx = 0: column 2 ox; kx = 1: column 2 ox + 1
                    bf[1][s] = O;
                }
            }
            if constexpr (H1) {
#pragma unroll
                for (int kx = 0; kx < KH; ++kx) acc[kx] = mma(af[0], bf[kx][0], acc[kx]);
            } else if constexpr (SCH == S2_BX3) {
                // lh hl mm mh hm hh (small cross terms first), as every 3-way bf16 kernel of the library
                constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int kx = 0; kx < KH; ++kx) acc[kx] = mma(af[SA[pr]], bf[kx][SB[pr]], acc[kx]);
            } else {
                // small cross terms first: m x h, h x m, h x h
#pragma unroll
                for (int kx = 0; kx < KH; ++kx) acc[kx] = mma(af[NS - 1], bf[kx][0], acc[kx]);
#pragma unroll
                for (int kx = 0; kx < KH; ++kx) acc[kx] = mma(af[0], bf[kx][NS - 1], acc[kx]);
#pragma unroll
                for (int kx = 0; kx < KH; ++kx) acc[kx] = mma(af[0], bf[kx][0], acc[kx]);
            }
        }
    };

    if (t_begin < t_end) {
        // (tx, ty) of tiles t, t + 1, t + 2, advanced incrementally (column-major order)
        auto adv = [&](int& tx, int& ty) { if (++ty == a.tiles_y) { ty = 0; ++tx; } };
        int tx0 = t_begin / a.tiles_y, ty0 = t_begin - tx0 * a.tiles_y;
        int tx1 = tx0, ty1 = ty0; adv(tx1, ty1);
        int tx2 = tx1, ty2 = ty1; adv(tx2, ty2);
        prefetch0(tx0, ty0);                               // first tile of the block: all three rows, synchronously
        commit(0);
        __syncthreads();
        if (t_begin + 1 < t_end) prefetch(tx1, ty1);
        for (int t = t_begin; t < t_end; ++t) {
            const int b = (t - t_begin) & 1;
            const bool has_next = t + 1 < t_end;
            const bool next_steady = has_next && ty1 != 0;
            if (next_steady) {
                commit(b ^ 1);                             // rows 2 oy + 2, 2 oy + 3: not among the three rows tile t reads
                if (t + 2 < t_end) prefetch(tx2, ty2);
            }
            mfma_tile(ty0, b);
            __syncthreads();
            if (has_next && !next_steady) {                // first tile of a column: its rows alias the ring -- after the barrier
                commit(b ^ 1);
                __syncthreads();
                if (t + 2 < t_end) prefetch(tx2, ty2);
            }
            tx0 = tx1; ty0 = ty1; tx1 = tx2; ty1 = ty2; adv(tx2, ty2);
        }
    }

    // ---- bias gradient partial: fixed-order in-block reduction of the per-thread dout sums
    if (a.dbpart && blockIdx.z == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);
        if (dthread) red[tid] = dsum;
        __syncthreads();
        if (tid < S2_BM && co0 + tid < a.Cout)
            a.dbpart[(long)split * a.Cout + co0 + tid] = (red[tid * 4] + red[tid * 4 + 1]) + (red[tid * 4 + 2] + red[tid * 4 + 3]);
    }
    // partial[split][co][ci][tap]
    float* part = a.part + (long)split * a.Cout * a.Cin * (KH * KH);
    const int ci = ci0 + wn * 32 + j;
    if (ci < a.Cin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (co >= a.Cout) continue;
#pragma unroll
            for (int kx = 0; kx < KH; ++kx) part[((long)co * a.Cin + ci) * (KH * KH) + kyw * KH + kx] = acc[kx][r];
        }
    }
}

template <int KH, int WN, int SCH>
int launch_s2(const WgArgs& a, const WgPlan& p, int N, hipStream_t st) {
    constexpr int NS = s2_ns(SCH), BN = 32 * WN;
    static_assert((size_t)(2 * NS * S2_BM * S2_DP + NS * BN * s2_ip(KH)) * 2 <= 160 * 1024, "LDS");
    const size_t lds = (size_t)(2 * NS * S2_BM * S2_DP + NS * BN * s2_ip(KH)) * 2;
    dim3 grid(N * p.spi, tdr_cdiv(a.Cout, S2_BM), tdr_cdiv(a.Cin, BN));
    auto kern = wgrad_s2_kernel<KH, WN, SCH>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(128 * KH * WN), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad_s2_kernel");
    return TDR_OK;
}

}  // namespace

bool tdr_wgrad_s2_supported(const TdrWgradDesc* d) {
    static const bool off = tdr_tune_env("TDR_WG_S2") && atoi(tdr_tune_env("TDR_WG_S2")) == 0;   // A/B aid: 0 = exact-fp32 kernel as before
    if (off || d->math < 1 || d->gate) return false;
    if (d->stride != 2 || !((d->KH == 3 && d->pad == 1) || (d->KH == 2 && d->pad == 0))) return false;
    if (d->H != 2 * d->OH || d->W != 2 * d->OW || d->OW < 8 || d->OW % 4 != 0) return false;
    if ((long)d->Cin * d->H * d->W >= (1L << 29) || (long)d->Cout * d->OH * d->OW >= (1L << 29)) return false;   // 32-bit buffer offsets
    return d->in_ns % 4 == 0 && d->dout_ns % 4 == 0;
}

WgPlan tdr_wgrad_s2_plan(const TdrWgradDesc* d) {
    WgPlan p;
    p.tw_log2 = 5;
    // three bf16 planes (math 1): the 64-channel input ring would need 187 KB of LDS -- 32 input channels per block there (109 KB)
    p.cfg = (d->Cin <= 32 || d->math == 1) ? 0 : 1;
    p.BMc = S2_BM; p.BNc = p.cfg == 0 ? 32 : 64;
    p.WKw = 1;
    p.tiles_x = tdr_cdiv(d->OW, S2_C);
    p.tiles_y = d->OH;
    p.tpi = p.tiles_x * p.tiles_y;
    const long out_tiles = (long)tdr_cdiv(d->Cout, p.BMc) * tdr_cdiv(d->Cin, p.BNc);
    // one round of blocks: one 12-wave workgroup (125 KB of LDS) per CU, or two of the 6-wave workgroups of the Cin <= 32 variant (73 KB)
    static const long want_env = tdr_tune_env("TDR_WG_S2_WANT") ? atol(tdr_tune_env("TDR_WG_S2_WANT")) : 0;
    const long want_total = want_env > 0 ? want_env : ((p.BNc == 32 && d->math != 1) ? 512 : 256);     // (three planes: 94 - 109 KB, one workgroup per CU)
    long want = want_total / out_tiles;
    if (want < 1) want = 1;
    long spi = (want + d->N - 1) / d->N;
    if (spi > p.tpi / 8) spi = p.tpi / 8;             // at least 8 tiles (one output row each) per block
    if (spi < 1) spi = 1;
    p.tps = tdr_cdiv(p.tpi, spi);
    p.spi = tdr_cdiv(p.tpi, p.tps);
    return p;
}

int tdr_wgrad_s2_launch(const WgArgs& a, const WgPlan& p, const TdrWgradDesc* d, hipStream_t st) {
    const bool h1 = a.scheme == 2;
    if (a.scheme == 0)      // 3-way bf16 split: always the 32-input-channel blocks (tdr_wgrad_s2_plan)
        return d->KH == 3 ? launch_s2<3, 1, S2_BX3>(a, p, d->N, st) : launch_s2<2, 1, S2_BX3>(a, p, d->N, st);
    if (d->KH == 3) {
        if (p.cfg == 0) return h1 ? launch_s2<3, 1, S2_H1>(a, p, d->N, st) : launch_s2<3, 1, S2_HX2>(a, p, d->N, st);
        return h1 ? launch_s2<3, 2, S2_H1>(a, p, d->N, st) : launch_s2<3, 2, S2_HX2>(a, p, d->N, st);
    }
    if (p.cfg == 0) return h1 ? launch_s2<2, 1, S2_H1>(a, p, d->N, st) : launch_s2<2, 1, S2_HX2>(a, p, d->N, st);
    return h1 ? launch_s2<2, 2, S2_H1>(a, p, d->N, st) : launch_s2<2, 2, S2_HX2>(a, p, d->N, st);
}
