// Implicit-GEMM convolution on the gfx950 bf16 matrix cores with fp32-equivalent arithmetic.
//
// Every fp32 operand x is split on the fly into three bf16 terms x = h + m + l
// (h = rn_bf16(x), m = rn_bf16(x - h), l = rn_bf16(x - h - m): 24+ significant bits), and each
// fp32 product a*b is evaluated as the six cross products
//     al*bh + ah*bl + am*bm + am*bh + ah*bm + ah*bh          (dropped terms <= 2^-25 |a*b|)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Per product this is at least as accurate as
// one fp32 multiply; the accumulation is fp32 like the exact kernel (tdr_conv_mfma.hip).  The bf16
// pipe is 16x the fp32-MFMA rate, so six products cost 0.375x of v_mfma_f32_32x32x2_f32
// (ceiling 2.5 PFLOP/s / 6 = 416 fp32-equivalent TFLOP/s vs 157).  profiles/probes/bf16x3_probe.hip
// measures the error of this scheme against the exact chain and an fp64 reference.
//
//   out[n][m][pix] = epi( sum_k A[m][k] * B[k][pix] ),  k = (16-channel group, tap, channel)
//
// One workgroup = 4 waves, one per SIMD, two workgroups per CU.  The block owns BM = 32*TM*WM
// output channels x 32*TN*WN pixels (32-pixel sub-tiles of 32/TW rows x TW cols).
//   A (weights): pre-split and pre-packed in MFMA fragment order by tdr_pack_weights_bx3, read
//                straight from L2/L1 into VGPRs (1 KiB coalesced wave loads, never through LDS);
//   B (pixels) : the input halo tile of 16 channels is loaded NCHW-coalesced, split in registers
//                and written to LDS as s_in[buf][split][kgroup][pixel] 16-byte slots (8 channels),
//                so a B fragment is one conflict-free ds_read_b128 per lane for any tap shift.
// LDS is double buffered: the global loads of group c+1 are in flight under the MFMAs of group c,
// one barrier per group.
#include <stdlib.h>
#include "tdr_common.h"
#include "tdr_conv_epi.h"
#include "tdr_pack.h"
#include "../../include/tdr.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (the ablation / timeline probe variants of round 1 -- profiles/README.md -- were built from this file at commit 7d0042b;
// the product source carries no probe code)

namespace {

constexpr int bx_cmax(int a, int b) { return a > b ? a : b; }
constexpr int bx_plane(int NT, int TW, int KH, int S) {
    return (((NT * 32 / TW) - 1) * S + KH) * ((TW - 1) * S + KH);
}
constexpr int bx_max_plane(int NT, int KH, int S) {
    return bx_cmax(bx_plane(NT, 8, KH, S), bx_cmax(bx_plane(NT, 16, KH, S), bx_plane(NT, 32, KH, S)));
}

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union Frag {
    uint4 u;
    bf16x8 v;
    f16x8 hv;
};

// Operand split schemes (fp32 tensors in, fp32 accumulation, fp32-class products):
//   SCH_BX3  x = h + m + l, bf16 each: 6 cross products (lh, hl, mm, mh, hm, hh) -- any fp32 range
//   SCH_HX2  x = h + m, fp16 each (11 + 11 mantissa bits): 3 cross products (mh, hm, hh; m*m is 2^-22 relative and
//            measurably irrelevant, profiles/r1/fp16x2_probe_mi355x.log) -- half the matrix work and two operand planes
//            instead of three, for operands inside the fp16 range (|x| < 65504, magnitudes of interest above ~2^-14):
//            forward activations and weights; NOT raw gradients (3e-7-sized values lose everything without a pre-scale).
//   SCH_H1   x ~ h, one fp16 plane, one product (plain fp16 MFMA, fp32 accumulate): reads the head plane of an hx2 weight pack.
//            Reduced precision (2^-11 per operand) -- BASELINE configs[4]'s "fp16 MFMA" arithmetic, selected by TDR_MATH=h1 only.
enum { SCH_BX3 = 0, SCH_HX2 = 1, SCH_H1 = 2 };

// Occupancy: the high-resolution 3x3 launches (C = 32 level: 32 x 256 tiles, 21 KB of LDS) are latency-bound -- SQ counters show their
// waves waiting 55 % of the resident time with 3 waves per SIMD, the occupancy 148 VGPRs allow; asking for 4 (5) resident workgroups
// caps the allocation at 128 (102) registers: 4 fits without spilling (118 VGPRs; same-box A/B 59.0 -> 58.6 ms per cfg2 step), 5 spills
// 81 registers, the 64 x 256 tiles of the C = 64 level spill 130 at 4, the 128 x 128 tiles of the 1x1 kernel 334.  TDR_CONV_OCC (compile-time -D) selects the request.
#ifndef TDR_CONV_OCC
#define TDR_CONV_OCC 4
#endif
template <int KH, int S, int WM, int TM, int TN, int EPI, bool GATE, int SCH, int AD = 0>
__global__ __launch_bounds__(256, (KH == 3 && S == 1 && WM == 1 && TM == 1 && TN == 2 && AD != 9) ? TDR_CONV_OCC : ((KH == 3 && S == 1 && WM == 2 && TM == 1 && TN == 4 && AD == 3) ? 3 : 2)) void conv_bx3_kernel(ConvArgs a) {
    constexpr int NS = SCH == SCH_BX3 ? 3 : (SCH == SCH_HX2 ? 2 : 1);   // operand planes (LDS, fragments)
    constexpr int NSW = SCH == SCH_BX3 ? 3 : 2;                         // planes of the weight pack
    constexpr int NP = SCH == SCH_BX3 ? 6 : (SCH == SCH_HX2 ? 3 : 1);   // matrix products per fp32 product
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = TN * WN;
    constexpr int TAPS = KH * KH;
    constexpr int NIT = (bx_max_plane(NT, KH, S) + 127) / 128;   // a wave pair stages 128 halo pixels per pass
    // prefetch depth in 16-channel groups: a 1x1 group is only TM*TN*6 MFMAs (0.2-0.6 us), far less than the
    // HBM latency, so its operand loads are issued two groups ahead; 9-tap groups are long enough for one.
    constexpr int PF = (KH == 1) ? 2 : 1;
    // AD > 0: weight-fragment prefetch ring of the 9-tap kernels.  A (group, tap) step is only TM*TN*NP MFMAs (0.1 - 0.4 us),
    // less than an L2 round trip, so the fragments of step i + AD are requested when step i has consumed its slot (slot =
    // tap % AD is a compile-time index: AD divides 9).  AD = 0: the next step's fragments only (afn double buffer).
    // MI355X, N = 8: 512 -> 512 @32x32 175 -> 122 us and 256 -> 256 @64x64 125 -> 114 us with AD = 9; the 72 extra VGPRs
    // cost the many-round high-resolution launches their occupancy (32 -> 32 @512x512 230 -> 241 us), and with two m-tiles
    // per wave the ring spills: launch_bx_cfg_s picks it for single-round launches of the TM = 1 kernels only.
    static_assert(AD == 0 || (KH == 3 && 9 % AD == 0), "ring depth must divide the 9 taps");

    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

    // wave index through readfirstlane: everything derived from it (channel rows, LDS bases) is then provably
    // wave-uniform and its address arithmetic runs on the scalar unit
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, kk = lane >> 5;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2, TH = NT * SR;
    const int LH = (TH - 1) * S + KH, LW = (TW - 1) * S + KH;
    const int plane = LH * LW;
    // XCD-aware block order (workgroup b runs on XCD b % 8, each XCD has its own L2): XCD k walks a contiguous
    // range of the logical (pixel tile, m-tile) sequence with the m-tiles of one pixel tile back to back, so an
    // input tile is fetched from HBM once per XCD and re-used from its L2 by the other m-tiles, and halo lines
    // are shared between neighbouring pixel tiles on the same XCD.  Bijective for any grid size; speed only.
    int logical;
    {
        const int T = gridDim.x, b = blockIdx.x;
        const int q = T >> 3, r = T & 7, xcd = b & 7, slot = b >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int mtile = logical % a.mtiles, ptile = logical / a.mtiles;
    const int tx = ptile % a.tiles_x, ty = ptile / a.tiles_x;
    const int m0 = mtile * BM;
    const int n = blockIdx.z;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const long HWin = (long)a.H * a.W;

    // ---- staging geometry: waves {0,1} stage channels 0-7 of the group, waves {2,3} channels 8-15
    const int sg = wave >> 1;
    const int sp0 = (wave & 1) * 64 + lane;
    int gsafe[NIT];
    unsigned okmask = 0, inplane = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = sp0 + 128 * it;
        const int r = p / LW, x = p - r * LW;
        const int gy = iy0 + r, gx = ix0 + x;
        const bool inp = p < plane;
        const bool ok = inp && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        gsafe[it] = ok ? gy * a.W + gx : 0;
        okmask |= (ok ? 1u : 0u) << it;
        inplane |= (inp ? 1u : 0u) << it;
    }
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* ks_n = a.kscale ? a.kscale + (long)n * a.kscale_ns : nullptr;
    const int ngroups = (a.Cin + 15) >> 4;

    float rin[PF][NIT][8];
    float rin2[GATE ? PF : 1][GATE ? NIT : 1][GATE ? 8 : 1];
    float rks[PF][8];
    auto load_group = [&](int g, int set) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ci = min(g * 16 + sg * 8 + i, a.Cin - 1);
            const float* base = in_n + (long)ci * HWin;
            rks[set][i] = ks_n ? ks_n[ci] : 1.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                rin[set][it][i] = base[gsafe[it]];
                if (GATE) rin2[set][it][i] = base[gsafe[it] + a.gate_off];
            }
        }
    };
    auto store_group = [&](int g, int set, int buf) {
        uint4* sb = smem4 + (buf * (2 * NS) + sg) * plane;
        const int cbase = g * 16 + sg * 8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            Frag h, m, l;
            const bool ok = (okmask >> it) & 1u;
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = rin[set][it][i];
                if (GATE) v *= rin2[set][it][i];
                v *= rks[set][i];
                vv[i] = (ok && cbase + i < a.Cin) ? v : 0.f;
            }
            if constexpr (SCH == SCH_HX2) {
                // head and residual must start from the SAME fp32 value (tdr_split2_f16 pins the gated / scaled operand in a VGPR:
                // left free, the compiler may round the head from the exact product and the residual from the rounded one, and at
                // an fp16 tie the pair is then off by a whole ulp of the head -- tdr_nafblock.hip, split_hm)
                unsigned hd[4], md[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) tdr_split2_f16(vv[2 * i], vv[2 * i + 1], hd[i], md[i]);
                h.u = make_uint4(hd[0], hd[1], hd[2], hd[3]);
                m.u = make_uint4(md[0], md[1], md[2], md[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = vv[i];
                    asm volatile("" : "+v"(v));            // pin: every split plane from the same fp32 value
                    if constexpr (SCH == SCH_H1) {
                        h.hv[i] = (_Float16)v;
                    } else {
                        __bf16 hh, mm, ll;
                        split3(v, hh, mm, ll);
                        h.v[i] = hh; m.v[i] = mm; l.v[i] = ll;
                    }
                }
            }
            if ((inplane >> it) & 1u) {
                const int p = sp0 + 128 * it;
                sb[p] = h.u;
                if constexpr (NS >= 2) sb[2 * plane + p] = m.u;
                if constexpr (NS == 3) sb[4 * plane + p] = l.u;
            }
        }
    };

    // ---- fragment addresses
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int t = wn * TN + tn;
        const int py = t * SR + (j >> a.tw_log2), px = j & (TW - 1);
        bbase[tn] = kk * plane + py * S * LW + px * S;
    }
    // packed weights: [group][tap][m-tile][split][lane] 16-byte fragments
    const int MT = a.Mpad >> 5;
    const uint4* wfrag[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int mt = min((m0 >> 5) + wm * TM + tm, MT - 1);
        wfrag[tm] = reinterpret_cast<const uint4*>(a.wp) + (long)n * (a.wp_ns >> 2) + (long)mt * (NSW * 64) + lane;   // wp_ns: floats per image (0 = shared)
    }
    const long wstep = (long)MT * (NSW * 64);    // 16-byte units per (group, tap)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    Frag af[TM][NS], afn[TM][NS];
    Frag aq[AD > 0 ? AD : 1][TM][NS];
    auto load_a = [&](Frag (&dst)[TM][NS], long gt) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int s = 0; s < NS; ++s) dst[tm][s].u = wfrag[tm][gt * wstep + s * 64];
    };

    if constexpr (AD > 0) {
#pragma unroll
        for (int d = 0; d < AD; ++d) load_a(aq[d], min((long)d, (long)ngroups * TAPS - 1));
    } else {
        load_a(af, 0);
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < ngroups) load_group(p, p);
    store_group(0, 0, 0);
    __syncthreads();

    for (int g0 = 0; g0 < ngroups; g0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int g = g0 + u;
            if (g < ngroups) {
                const int buf = a.single_buf ? 0 : (g & 1);
                const uint4* sb = smem4 + buf * (2 * NS) * plane;
                // Short rings (0 < AD < 9): the next group's operand loads are issued at tap 9 - AD, i.e. AFTER the last weight-
                // fragment request this group still consumes -- every fragment wait of the group is then a wait on OLDER loads
                // (vmcnt counts in order) and the operand loads stay in flight under the remaining AD taps.  Issued at the top of
                // the group (AD = 0, and AD = 9 where every slot in use was requested a whole group earlier) the first wait on a
                // fragment requested after them drains them: with AD = 0 that is tap 1, so the high-resolution launches overlapped
                // their HBM loads with one tap of MFMAs only.
                constexpr int IGT = (PF == 1 && KH == 3 && AD > 0 && AD < TAPS) ? TAPS - AD : -1;
                if (PF == 1 && IGT < 0 && g + 1 < ngroups) load_group(g + 1, 0);
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int tapoff = (tap / KH) * LW + (tap % KH);
                    if (tap == IGT && g + 1 < ngroups) {
                        load_group(g + 1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // next (group, tap) weight fragments first, THEN the far-ahead operand loads: the wait for
                    // the fragments (in-order vmcnt) then leaves the operand loads in flight.
                    // (the last prefetch of the last group re-reads a valid slot)
                    const long gtn = min((long)g * TAPS + tap + 1, (long)ngroups * TAPS - 1);
                    if (AD == 0) load_a(afn, gtn);
                    Frag (&afc)[TM][NS] = AD > 0 ? aq[AD > 0 ? tap % (AD > 0 ? AD : 1) : 0] : af;
                    if (PF > 1 && tap == 0 && g + PF < ngroups) load_group(g + PF, u);   // set u is free: group g already sits in LDS
                    Frag bf[TN][NS];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int s = 0; s < NS; ++s) bf[tn][s].u = sb[s * 2 * plane + bbase[tn] + tapoff];
                    // small cross terms first, the dominant h*h last
                    constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};          // bx3: lh hl mm mh hm hh
                    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};                            // hx2: mh hm hh
#pragma unroll
                    for (int q = 0; q < NP; ++q)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) {
                                if constexpr (SCH == SCH_H1)
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[tm][0].hv, bf[tn][0].hv, acc[tm][tn], 0, 0, 0);
                                else if constexpr (SCH == SCH_HX2)
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afc[tm][HA[q]].hv, bf[tn][HB[q]].hv, acc[tm][tn], 0, 0, 0);
                                else
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc[tm][SA[q]].v, bf[tn][SB[q]].v, acc[tm][tn], 0, 0, 0);
                            }
                    if constexpr (AD > 0) {      // the slot is consumed: request the fragments AD steps ahead into it
                        load_a(aq[tap % (AD > 0 ? AD : 1)], min((long)g * TAPS + tap + AD, (long)ngroups * TAPS - 1));
                    } else {
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int s = 0; s < NS; ++s) af[tm][s] = afn[tm][s];
                    }
                }
                {
                    if (a.single_buf) __syncthreads();        // one LDS buffer: every wave is done reading group g
                    if (g + 1 < ngroups) store_group(g + 1, (u + 1) % PF, a.single_buf ? 0 : (buf ^ 1));
                    __syncthreads();
                }
            }
        }
    }

    if constexpr (EPI != EPI_PSHUF) {
        if (a.vec_epi) {
            __syncthreads();                              // every wave is done with the operand tiles
            conv_epilogue_vec<TM, TN, EPI>(a, acc, n, m0, wm, wn, oy0, ox0, lane, reinterpret_cast<float*>(smem4) + wave * (32 * 36));
            return;
        }
    }
    conv_epilogue<TM, TN, EPI>(a, acc, n, m0, wm, wn, oy0, ox0, j, kk);
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 / stride 1 / 2-way fp16 split with 16-byte operand staging.
//
// The generic kernel above stages pixels with one dword load per (lane, channel): on the K <= 256 layers the texture-
// address path, not the matrix pipe, is the limit (profiles/README.md, timeline probe).  A 1x1 tile has no halo, so here a
// lane owns 4 adjacent pixels x 8 channels = eight float4 loads (128-byte row segments per 8 lanes), converts them to the
// same [split][octet][pixel] 16-byte LDS slots, and one pass of the 256 threads stages KS = 8192 / NPX channels (64 for
// a 128-pixel tile) instead of 16: a quarter of the vector-memory instructions and a quarter of the barriers.  LDS slots
// are XOR-swizzled (slot ^ ((slot >> 4) & 3)) so that both the 4-slot-strided writes and the 32-contiguous fragment reads
// are conflict-free.  Same weight fragments, accumulator layout and epilogues as conv_bx3_kernel.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz1(int slot) { return slot ^ ((slot >> 4) & 3); }

template <int WM, int TM, int TN, int EPI, bool GATE, int SCH = SCH_HX2>
__global__ __launch_bounds__(256, 2) void conv1x1_hx2_kernel(ConvArgs a) {
    constexpr int NS = SCH == SCH_H1 ? 1 : 2, NSW = 2, NP = SCH == SCH_H1 ? 1 : 3;
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = TN * WN;
    constexpr int NPX = 32 * NT;          // pixels per tile
    constexpr int QUADS = NPX / 4;        // float4 pixel quads per tile
    constexpr int OCT = 256 / QUADS;      // 8-channel octets staged per pass: one (octet, quad) task per thread
    constexpr int KS = 8 * OCT;           // channels per stage
    constexpr int GPS = KS / 16;          // 16-channel MFMA groups per stage
    constexpr int PFD = GATE ? 1 : 2;     // stages of operand loads in flight (register sets)

    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, kk = lane >> 5;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2, TH = NT * SR;
    int logical;
    {   // XCD-aware block order, as in conv_bx3_kernel
        const int T = gridDim.x, b = blockIdx.x;
        const int q = T >> 3, r = T & 7, xcd = b & 7, slot = b >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int mtile = logical % a.mtiles, ptile = logical / a.mtiles;
    const int tx = ptile % a.tiles_x, ty = ptile / a.tiles_x;
    const int m0 = mtile * BM;
    const int n = blockIdx.z;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const long HWin = (long)a.H * a.W;

    // ---- staging task of this thread: octet so, pixel quad sq
    const int so = tid / QUADS, sq = tid % QUADS;
    const int p0 = 4 * sq;
    const int gy = oy0 + (p0 >> a.tw_log2), gx = ox0 + (p0 & (TW - 1));
    const bool pok = gy < a.H && gx < a.W;            // W % 4 == 0: a quad is inside or outside as a whole
    const long goff = pok ? (long)gy * a.W + gx : 0;
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* ks_n = a.kscale ? a.kscale + (long)n * a.kscale_ns : nullptr;
    const int ngroups = (a.Cin + 15) >> 4;
    const int nstages = (a.Cin + KS - 1) / KS;

    float4 rin[PFD][8];
    float4 rin2[GATE ? PFD : 1][GATE ? 8 : 1];
    float rks[PFD][8];
    auto load_stage = [&](int st, int set) {
        const int cbase = st * KS + so * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ci = cbase + i;
            rin[set][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            rks[set][i] = 1.f;
            if (GATE) rin2[set][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pok && ci < a.Cin) {
                const float* src = in_n + (long)ci * HWin + goff;
                rin[set][i] = *reinterpret_cast<const float4*>(src);
                if (GATE) rin2[set][i] = *reinterpret_cast<const float4*>(src + a.gate_off);
                if (ks_n) rks[set][i] = ks_n[ci];
            }
        }
    };
    int wslot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wslot[i] = so * NPX + swz1(p0 + i);
    auto store_stage = [&](int set, int buf) {
        uint4* sb = smem4 + buf * (NS * OCT * NPX);
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            Frag h, m;
            float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 q4 = rin[set][i];
                float v = px == 0 ? q4.x : (px == 1 ? q4.y : (px == 2 ? q4.z : q4.w));
                if (GATE) {
                    const float4 g4 = rin2[set][i];
                    v *= px == 0 ? g4.x : (px == 1 ? g4.y : (px == 2 ? g4.z : g4.w));
                }
                vv[i] = v * rks[set][i];
            }
            if constexpr (NS == 2) {
                unsigned hd[4], md[4];                    // one fp32 value for head and residual (see conv_bx3_kernel)
#pragma unroll
                for (int i = 0; i < 4; ++i) tdr_split2_f16(vv[2 * i], vv[2 * i + 1], hd[i], md[i]);
                h.u = make_uint4(hd[0], hd[1], hd[2], hd[3]);
                m.u = make_uint4(md[0], md[1], md[2], md[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) h.hv[i] = (_Float16)vv[i];
            }
            sb[wslot[px]] = h.u;
            if constexpr (NS == 2) sb[OCT * NPX + wslot[px]] = m.u;
        }
    };

    // ---- fragment addresses
    int bslot[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bslot[tn] = kk * NPX + swz1(32 * (wn * TN + tn) + j);
    const int MT = a.Mpad >> 5;
    const uint4* wfrag[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int mt = min((m0 >> 5) + wm * TM + tm, MT - 1);
        wfrag[tm] = reinterpret_cast<const uint4*>(a.wp) + (long)n * (a.wp_ns >> 2) + (long)mt * (NSW * 64) + lane;
    }
    const long wstep = (long)MT * (NSW * 64);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    Frag af[TM][NS], afn[TM][NS];
    auto load_a = [&](Frag (&dst)[TM][NS], long g) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int s = 0; s < NS; ++s) dst[tm][s].u = wfrag[tm][g * wstep + s * 64];
    };

    load_a(af, 0);
#pragma unroll
    for (int p = 0; p < PFD; ++p)
        if (p < nstages) load_stage(p, p);
    store_stage(0, 0);
    __syncthreads();

    for (int s0 = 0; s0 < nstages; s0 += PFD) {
#pragma unroll
        for (int u = 0; u < PFD; ++u) {
            const int st = s0 + u;
            if (st < nstages) {
                const int buf = st & 1;
                const uint4* sb = smem4 + buf * (NS * OCT * NPX);
#pragma unroll
                for (int gg = 0; gg < GPS; ++gg) {
                    const int g = st * GPS + gg;
                    if (g < ngroups) {
                        load_a(afn, min(g + 1, ngroups - 1));
                        // set u is free (stage st already sits in LDS): the loads of stage st + PFD go out behind the
                        // first weight-fragment prefetch, so the in-order wait for the fragments leaves them in flight
                        if (gg == 0 && st + PFD < nstages) load_stage(st + PFD, u);
                        Frag bf[TN][NS];
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                            for (int s = 0; s < NS; ++s) bf[tn][s].u = sb[s * (OCT * NPX) + 2 * gg * NPX + bslot[tn]];
                        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};   // mh hm hh
#pragma unroll
                        for (int q = 0; q < NP; ++q)
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) {
                                    if constexpr (SCH == SCH_H1)
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tm][0].hv, bf[tn][0].hv, acc[tm][tn], 0, 0, 0);
                                    else
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tm][HA[q]].hv, bf[tn][HB[q]].hv, acc[tm][tn], 0, 0, 0);
                                }
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int s = 0; s < NS; ++s) af[tm][s] = afn[tm][s];
                    }
                }
                if (st + 1 < nstages) store_stage((u + 1) % PFD, buf ^ 1);
                __syncthreads();
            }
        }
    }

    if constexpr (EPI != EPI_PSHUF) {
        if (a.vec_epi) {
            conv_epilogue_vec<TM, TN, EPI>(a, acc, n, m0, wm, wn, oy0, ox0, lane, reinterpret_cast<float*>(smem4) + wave * (32 * 36));
            return;
        }
    }
    conv_epilogue<TM, TN, EPI>(a, acc, n, m0, wm, wn, oy0, ox0, j, kk);
}

template <int WM, int TM, int TN, int EPI, bool GATE, int SCH>
int launch_c1_hx2(const ConvArgs& a, int N, hipStream_t st) {
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = TN * WN;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2, TH = NT * SR;
    ConvArgs b = a;
    b.tiles_x = tdr_cdiv(a.OW, TW);
    const int tiles_y = tdr_cdiv(a.OH, TH);
    b.mtiles = tdr_cdiv(a.Cout, BM);
    dim3 grid(b.tiles_x * tiles_y * b.mtiles, 1, N);
    b.single_buf = 0;
    const size_t lds = 2 * 32768;        // two stages of 2 splits x OCT octets x NPX pixels x 16 B = 32 KiB
    auto kern = conv1x1_hx2_kernel<WM, TM, TN, EPI, GATE, SCH>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b);
    TDR_LAUNCH_CHECK("conv1x1_hx2_kernel");
    return TDR_OK;
}

// 16-byte staging needs whole, aligned pixel quads; it pays from four stages of K on and for launches of a single
// round of workgroups (probe_conv1x1.py / kernel traces on MI355X: 256->512 @64x64 N=4 30.1 -> 27.0 us, 512->256 37.9
// -> 32.2 us; 128->256 @256x256 with two stages 123 -> 134 us, 64->128 @512x512 with one stage 189 -> 226 us; 2048
// workgroups of 256->256 @128x128 87 -> 96 us: with 64 KiB of LDS only two workgroups share a CU, and the many-round
// launches want the four or five of the 16-channel pipeline of conv_bx3_kernel)
inline bool c1_hx2_ok(const ConvArgs& a, int npx, int bm, int N) {
    static const bool off = tdr_tune_env("TDR_C1_OLD") != nullptr;
    static const int min_stages = tdr_tune_env("TDR_C1_STAGES") ? atoi(tdr_tune_env("TDR_C1_STAGES")) : 4;
    static const long max_blocks = tdr_tune_env("TDR_C1_BLOCKS") ? atol(tdr_tune_env("TDR_C1_BLOCKS")) : 512;
    const int ks = 8192 / npx;
    const long blocks = (long)tdr_cdiv((long)a.OH * a.OW, npx) * tdr_cdiv(a.Cout, bm) * N;
    return !off && blocks <= max_blocks && (a.Cin + ks - 1) / ks >= min_stages && a.pad == 0 && a.W % 4 == 0 && a.in_ns % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 &&
           a.H == a.OH && a.W == a.OW;
}

template <int KH, int S, int WM, int TM, int TN, int EPI, bool GATE, int SCH>
int launch_bx_cfg_s(const ConvArgs& a, int N, hipStream_t st) {
    if constexpr (KH == 1 && S == 1 && SCH != SCH_BX3 && EPI != EPI_PSHUF)
        if (c1_hx2_ok(a, 32 * TN * (4 / WM), 32 * TM * WM, N)) return launch_c1_hx2<WM, TM, TN, EPI, GATE, SCH>(a, N, st);
    constexpr int NS = SCH == SCH_BX3 ? 3 : (SCH == SCH_HX2 ? 2 : 1);
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM;
    constexpr int NT = TN * WN;
    const int TW = 1 << a.tw_log2, SR = 32 >> a.tw_log2, TH = NT * SR;
    const int LH = (TH - 1) * S + KH, LW = (TW - 1) * S + KH;
    ConvArgs b = a;
    // short K loops (<= 2 channel groups) of the 3x3 kernels: one LDS buffer instead of two halves the 65 KB footprint, so
    // four workgroups instead of two share a CU and hide each other's load / store latencies (tuning aid: TDR_BX_SINGLE)
    static const int single_env = tdr_tune_env("TDR_BX_SINGLE") ? atoi(tdr_tune_env("TDR_BX_SINGLE")) : 2;
    b.single_buf = (KH == 3 && (a.Cin + 15) / 16 <= single_env) ? 1 : 0;
    size_t lds = (size_t)(b.single_buf ? 1 : 2) * (2 * NS) * LH * LW * 16;
    if (lds < 4 * 32 * 36 * sizeof(float)) lds = 4 * 32 * 36 * sizeof(float);   // the vector epilogue's four wave-private 32 x 36 patches
    b.tiles_x = tdr_cdiv(a.OW, TW);
    const int tiles_y = tdr_cdiv(a.OH, TH);
    b.mtiles = tdr_cdiv(a.Cout, BM);
    dim3 grid(b.tiles_x * tiles_y * b.mtiles, 1, N);
    // Multi-round launches (and the two-m-tile kernels, which spill with 9 slots): a 3-slot ring -- fragments requested three taps
    // ahead (an L2 round trip is longer than one tap of MFMAs) and the operand prefetch deferred to tap 6 (see the main loop).
    // Same-box step 59.1 -> 58.4 ms: 32 -> 32 @512^2 216 -> 196 us, 128 -> 128 @128^2 145 -> 129 us, 64 -> 64 @256^2 unchanged
    // (profiles/r3/tried_and_dropped.txt has the counterpart experiments); TDR_RING3=0 restores the double-buffered fragments.
    if constexpr (KH == 3 && S == 1 && SCH == SCH_HX2 && EPI == EPI_STD && !GATE) {
        static const int ring3 = tdr_tune_env("TDR_RING3") ? atoi(tdr_tune_env("TDR_RING3")) : 1;
        static const long ring_blocks3 = tdr_tune_env("TDR_RING_BLOCKS") ? atol(tdr_tune_env("TDR_RING_BLOCKS")) : 512;
        if (ring3 && ((long)grid.x * N > ring_blocks3 || TM != 1)) {
            auto kern = conv_bx3_kernel<KH, S, WM, TM, TN, EPI, GATE, SCH, 3>;
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_set = true;
            }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b);
            TDR_LAUNCH_CHECK("conv_bx3_kernel(ring3)");
            return TDR_OK;
        }
    }
    if constexpr (KH == 3 && S == 1 && TM == 1 && SCH == SCH_HX2 && EPI == EPI_STD) {   // weight-fragment ring: single-round launches
        static const long ring_blocks = tdr_tune_env("TDR_RING_BLOCKS") ? atol(tdr_tune_env("TDR_RING_BLOCKS")) : 512;
        if ((long)grid.x * N <= ring_blocks) {
            auto kern = conv_bx3_kernel<KH, S, WM, TM, TN, EPI, GATE, SCH, 9>;
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_set = true;
            }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b);
            TDR_LAUNCH_CHECK("conv_bx3_kernel(ring)");
            return TDR_OK;
        }
    }
    auto kern = conv_bx3_kernel<KH, S, WM, TM, TN, EPI, GATE, SCH>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, b);
    TDR_LAUNCH_CHECK("conv_bx3_kernel");
    return TDR_OK;
}

// tile-configuration override for profiles/autotune_conv.py: [0] 1x1 kernels, [1] 3x3 / 2x2 kernels; 0 = heuristic
int g_force_cfg[2] = {tdr_tune_env("TDR_BX_CFG1") ? atoi(tdr_tune_env("TDR_BX_CFG1")) : 0, tdr_tune_env("TDR_BX_CFG3") ? atoi(tdr_tune_env("TDR_BX_CFG3")) : 0};

template <int KH, int S, int WM, int TM, int TN, int EPI, bool GATE>
int launch_bx_cfg(const ConvArgs& a, int N, hipStream_t st) {
    if (a.scheme == SCH_HX2) return launch_bx_cfg_s<KH, S, WM, TM, TN, EPI, GATE, SCH_HX2>(a, N, st);
    if (a.scheme == SCH_H1) return launch_bx_cfg_s<KH, S, WM, TM, TN, EPI, GATE, SCH_H1>(a, N, st);
    return launch_bx_cfg_s<KH, S, WM, TM, TN, EPI, GATE, SCH_BX3>(a, N, st);
}

template <int KH, int S, int EPI, bool GATE>
int launch_bx_shape(const ConvArgs& a, int N, hipStream_t st) {
    const long pix = (long)a.OH * a.OW;
    auto blocks = [&](int bm, int bn) { return (long)tdr_cdiv(a.Cout, bm) * tdr_cdiv(pix, bn) * N; };
    if constexpr (S == 2) {   // the input halo of a stride-2 tile is 4x the output pixels: keep the pixel tile small
        if (a.Cout <= 32) return launch_bx_cfg<KH, S, 1, 1, 1, EPI, GATE>(a, N, st);  // 32 x 128
        if (a.Cout <= 64 || blocks(128, 64) < 512) return launch_bx_cfg<KH, S, 2, 1, 1, EPI, GATE>(a, N, st);  // 64 x 64
        return launch_bx_cfg<KH, S, 2, 2, 1, EPI, GATE>(a, N, st);                     // 128 x 64
    } else {
        // largest tile that still gives every CU its two resident workgroups (512 blocks)
        if (a.Cout <= 32) {
            if (blocks(32, 256) >= 512) return launch_bx_cfg<KH, S, 1, 1, 2, EPI, GATE>(a, N, st);   // 32 x 256
            return launch_bx_cfg<KH, S, 1, 1, 1, EPI, GATE>(a, N, st);                               // 32 x 128
        }
        // short K loops over many pixels (the high-resolution levels) are staging-bound: the small 64 x 128 tile keeps
        // four workgroups per CU in flight (profiles/autotune_conv.py: 5-15 % on Cin <= 128 1x1 layers at >= 256^2)
        // (3-way split only: with the cheaper 2-plane fp16 staging the larger tiles win again, autotune_nafnet_hx2.log)
        if (g_force_cfg[KH == 1 ? 0 : 1] == 0 && a.scheme == SCH_BX3 && a.Cin <= (KH == 1 ? 128 : 64) && blocks(64, 128) >= 2048)
            return launch_bx_cfg<KH, S, 2, 1, 2, EPI, GATE>(a, N, st);
        if constexpr (KH == 1) {   // two register sets of prefetched operands: 128 x 128 keeps the kernel under 256 VGPRs
            const int force = g_force_cfg[0];   // tuning aid (tdr_conv_force_cfg / TDR_BX_CFG1)
            if (force == 1) return launch_bx_cfg<KH, S, 2, 2, 2, EPI, GATE>(a, N, st);
            if (force == 2) return launch_bx_cfg<KH, S, 2, 1, 4, EPI, GATE>(a, N, st);
            if (force == 3) return launch_bx_cfg<KH, S, 2, 1, 2, EPI, GATE>(a, N, st);
            if (force == 4) return launch_bx_cfg<KH, S, 1, 1, 2, EPI, GATE>(a, N, st);
            if (force == 5) return launch_bx_cfg<KH, S, 4, 2, 2, EPI, GATE>(a, N, st);    // 256 x 64
            if (a.Cout > 64 && blocks(128, 128) >= 512) return launch_bx_cfg<KH, S, 2, 2, 2, EPI, GATE>(a, N, st);
            // long K over few pixels (Restormer's 32 x 32 level: 2042 -> 768 @32x32, N = 8: 160 us at 768 workgroups of 64 x 128,
            // 94 us at 384 of 256 x 64): one round of tall tiles reads each pixel column once per 256 output channels
            if (a.Cin >= 768 && a.Cout >= 256 && blocks(256, 64) >= 256 && blocks(256, 64) <= 512)
                return launch_bx_cfg<KH, S, 4, 2, 2, EPI, GATE>(a, N, st);
        } else {
            const int force3 = g_force_cfg[1];   // tuning aid (tdr_conv_force_cfg / TDR_BX_CFG3)
            if (force3 == 1) return launch_bx_cfg<KH, S, 2, 2, 4, EPI, GATE>(a, N, st);
            if (force3 == 2) return launch_bx_cfg<KH, S, 2, 1, 4, EPI, GATE>(a, N, st);
            if (force3 == 3) return launch_bx_cfg<KH, S, 2, 1, 2, EPI, GATE>(a, N, st);
            if (force3 == 4) return launch_bx_cfg<KH, S, 2, 2, 2, EPI, GATE>(a, N, st);
            // 128 x 256 (two m-tiles per wave: 256 VGPRs, the 3-slot ring spills 836 B there) only where it leaves at least two rounds of
            // workgroups or the 64-row tile would waste rows: 128 -> 128 @128^2 runs 64 x 256 (1024 workgroups, no spills) 5 - 8 % faster
            const bool tall_ok = blocks(128, 256) >= 1024 || a.Cout % 128 != 0 || a.scheme != SCH_HX2;
            if (a.Cout > 64 && blocks(128, 256) >= 512 && tall_ok) return launch_bx_cfg<KH, S, 2, 2, 4, EPI, GATE>(a, N, st);  // 128 x 256
        }
        // weight fragments are re-read per 32-pixel column of the wave tile: wide pixel tiles (TN = 4) halve that L2->VGPR
        // stream -- but only with two resident workgroups per CU: one wave per SIMD cannot overlap its own LDS reads and
        // operand conversion with its MFMAs (512 -> 512 @ 32x32, N = 8: 321 us at 256 blocks of 64 x 256, 232 us at 512 of 64 x 128)
        if (blocks(64, 256) >= 512) return launch_bx_cfg<KH, S, 2, 1, 4, EPI, GATE>(a, N, st);       // 64 x 256
        return launch_bx_cfg<KH, S, 2, 1, 2, EPI, GATE>(a, N, st);                                   // 64 x 128
    }
}

// one thread per 16-byte fragment (all three splits)
__global__ void pack_weights_bx3_kernel(const float* __restrict__ w, long w_bs, int Cout, int Cin, int KH, int mode, int M,
                                        int Kch, int KHe, int MT, long total, uint4* __restrict__ wp, long wp_bs16) {
    const float* wb = w + (long)blockIdx.y * w_bs;              // blockIdx.y: matrix of a batch (per-image weights)
    uint4* ob = wp + (long)blockIdx.y * wp_bs16;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        tdr_pack_bx3_frag(wb, Cin, KH, mode, M, Kch, KHe, MT, i, ob);
}

__global__ void pack_weights_hx2_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int mode, int M, int Kch, int KHe,
                                        int MT, long total, uint4* __restrict__ wp) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        tdr_pack_hx2_frag(w, Cin, KH, mode, M, Kch, KHe, MT, i, wp);
}

}  // namespace

extern "C" int tdr_conv_force_cfg(int kh, int cfg) {
    g_force_cfg[kh == 1 ? 0 : 1] = cfg;
    return TDR_OK;
}

extern "C" int64_t tdr_packed_weight_bytes_bx3(int M, int Kch, int KH_eff) {
    const long MT = (M + 31) / 32;
    return (long)((Kch + 15) / 16) * KH_eff * KH_eff * MT * 3 * 1024;
}

extern "C" int tdr_pack_weights_bx3(const float* w, int Cout, int Cin, int KH, int mode, void* wp, void* stream) {
    TDR_REQUIRE(w && wp, "tdr_pack_weights_bx3: null pointer");
    TDR_REQUIRE(mode >= 0 && mode <= 3, "tdr_pack_weights_bx3: bad mode %d", mode);
    int M, Kch, KHe;
    if (mode == 0) { M = Cout; Kch = Cin; KHe = KH; }
    else if (mode == 1) { M = Cin; Kch = Cout; KHe = KH; }
    else if (mode == 2) { TDR_REQUIRE(KH == 2, "mode 2 needs a 2x2 kernel"); M = 4 * Cin; Kch = Cout; KHe = 1; }
    else { TDR_REQUIRE(KH == 3, "mode 3 needs a 3x3 kernel"); M = 4 * Cin; Kch = Cout; KHe = 2; }
    const int MT = (M + 31) / 32;
    const long total = (long)((Kch + 15) / 16) * KHe * KHe * MT * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_bx3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, 0L, Cout, Cin, KH, mode, M,
                       Kch, KHe, MT, total, (uint4*)wp, 0L);
    TDR_LAUNCH_CHECK("pack_weights_bx3_kernel");
    return TDR_OK;
}

extern "C" int64_t tdr_packed_weight_bytes_hx2(int M, int Kch, int KH_eff) {
    const long MT = (M + 31) / 32;
    return (long)((Kch + 15) / 16) * KH_eff * KH_eff * MT * 2 * 1024;
}

extern "C" int tdr_pack_weights_hx2(const float* w, int Cout, int Cin, int KH, int mode, void* wp, void* stream) {
    TDR_REQUIRE(w && wp, "tdr_pack_weights_hx2: null pointer");
    TDR_REQUIRE(mode >= 0 && mode <= 3, "tdr_pack_weights_hx2: bad mode %d", mode);
    int M, Kch, KHe;
    if (mode == 0) { M = Cout; Kch = Cin; KHe = KH; }
    else if (mode == 1) { M = Cin; Kch = Cout; KHe = KH; }
    else if (mode == 2) { TDR_REQUIRE(KH == 2, "mode 2 needs a 2x2 kernel"); M = 4 * Cin; Kch = Cout; KHe = 1; }
    else { TDR_REQUIRE(KH == 3, "mode 3 needs a 3x3 kernel"); M = 4 * Cin; Kch = Cout; KHe = 2; }
    const int MT = (M + 31) / 32;
    const long total = (long)((Kch + 15) / 16) * KHe * KHe * MT * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_hx2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, KH, mode, M, Kch,
                       KHe, MT, total, (uint4*)wp);
    TDR_LAUNCH_CHECK("pack_weights_hx2_kernel");
    return TDR_OK;
}

// B matrices at once (per-image weights of the MDTA core): matrix b is w + b*w_stride (floats), packed to wp + b*per_b bytes,
// per_b = tdr_packed_weight_bytes_bx3(M, Kch, KH_eff)
extern "C" int tdr_pack_weights_bx3_batch(const float* w, int64_t w_stride, int B, int Cout, int Cin, int KH, int mode, void* wp,
                                          void* stream) {
    TDR_REQUIRE(w && wp && B > 0, "tdr_pack_weights_bx3_batch: bad argument");
    TDR_REQUIRE(mode == 0 || mode == 1, "tdr_pack_weights_bx3_batch: mode 0 or 1 only (got %d)", mode);
    const int M = mode == 0 ? Cout : Cin, Kch = mode == 0 ? Cin : Cout;
    const int MT = (M + 31) / 32;
    const long total = (long)((Kch + 15) / 16) * KH * KH * MT * 64;
    const long per_b16 = tdr_packed_weight_bytes_bx3(M, Kch, KH) / 16;
    const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(pack_weights_bx3_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, w, (long)w_stride, Cout, Cin,
                       KH, mode, M, Kch, KH, MT, total, (uint4*)wp, per_b16);
    TDR_LAUNCH_CHECK("pack_weights_bx3_kernel(batch)");
    return TDR_OK;
}

// called by tdr_conv_forward (tdr_conv_mfma.hip) when the descriptor carries bx3-packed weights
int tdr_conv_forward_bx3(const TdrConvDesc* d, void* stream) {
    TDR_REQUIRE(d->dil == 1 && d->wp_ns % 4 == 0, "tdr_conv_forward: split-bf16 path needs dil=1 (and wp_ns a multiple of 4 floats)");
    ConvArgs a;
    a.in = d->in; a.in_ns = d->in_ns; a.Cin = d->Cin; a.H = d->H; a.W = d->W;
    a.wp = (const float*)d->wp; a.wp_ns = d->wp_ns; a.Mpad = d->Mpad; a.Cout = d->Cout;
    a.scheme = d->wp_fmt == 3 ? SCH_H1 : (d->wp_fmt == 2 ? SCH_HX2 : SCH_BX3);   // 3: an hx2 pack read as plain fp16 (TDR_MATH=h1)
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
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    a.vec_epi = (d->OW % 4 == 0 && d->out_ns % 4 == 0 && al16(d->out) && (!d->res || (d->res_ns % 4 == 0 && al16(d->res))) &&
                 (!d->mask || (d->mask_ns % 4 == 0 && al16(d->mask))) && (!d->aux || (d->aux_ns % 4 == 0 && al16(d->aux)))) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const int N = d->N;
    const int key = d->KH * 100 + d->stride * 10 + d->epi;
    const bool g = d->gate != 0;
    switch (key) {
        case 110: return g ? launch_bx_shape<1, 1, EPI_STD, true>(a, N, st) : launch_bx_shape<1, 1, EPI_STD, false>(a, N, st);
        case 111: if (!g) return launch_bx_shape<1, 1, EPI_GATEBWD, false>(a, N, st); break;
        case 112: if (!g) return launch_bx_shape<1, 1, EPI_PSHUF, false>(a, N, st); break;
        case 310: if (!g) return launch_bx_shape<3, 1, EPI_STD, false>(a, N, st); break;
        case 312: if (!g) return launch_bx_shape<3, 1, EPI_PSHUF, false>(a, N, st); break;
        case 320: if (!g) return launch_bx_shape<3, 2, EPI_STD, false>(a, N, st); break;
        case 220: if (!g) return launch_bx_shape<2, 2, EPI_STD, false>(a, N, st); break;
        case 212: if (!g) return launch_bx_shape<2, 1, EPI_PSHUF, false>(a, N, st); break;
        default: break;
    }
    tdr_set_error("tdr_conv_forward(bx3): unsupported (KH=%d stride=%d epi=%d gate=%d)", d->KH, d->stride, d->epi, d->gate);
    return TDR_ERR_UNSUPPORTED;
}
