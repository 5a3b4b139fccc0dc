// Weight gradient of the 1x1 convolutions (the NAFBlock chains: conv1 / conv3 / conv4 / conv5, ups; reference
// models/archs/network_nafnet_guided_arch.py:183-205,216-238 and their autograd) with a DECOUPLED operand stream, gfx950.
//
//   G[co][ci] = sum_{n, pixel} dout[n][co][pixel] * x[n][ci][pixel]            (GATE: x[ci] * x[Cin + ci], SimpleGate input of conv5)
//
// Both operands are fp32 [C][HW] rows with the contraction index (the pixel) contiguous, so an MFMA fragment (lane = channel row,
// 8 consecutive k) is 32 contiguous bytes of one row.  The round 1-4 kernel (wgrad_bx3_kernel<KH = 1>, tdr_wgrad_bx3.hip) staged a
// 32-pixel tile as [barrier, split every value into planes + write them to LDS, barrier, 48 MFMAs per wave] with the next tile's
// loads held in registers.  Measured (profiles/r5/probe_wgrad1x1_v*.log, sweep_c.log): this kernel 43.9 - 46 us against 47.5 - 51 us per
// 512 x 256 @ 64^2 launch, -0.4 ms per step -- a modest gain: both designs stay at ~37 % matrix-pipe utilisation (an in-order wave
// runs its split VALU and its MFMAs back to back; profiles/r5/tried_and_dropped.txt has the variants).  Here:
//   * the RAW fp32 rows of a 32-pixel stage (BM + BN rows x 128 B) go to an LDS ring by LDS-DMA (global_load_lds_dwordx4: no VGPRs,
//     no VALU, issued a whole stage ahead, waited for with a counted vmcnt) -- coalesced 128-byte row segments, the 16-byte
//     chunks of a row XOR-permuted by (row >> 1) & 7 on the GLOBAL side so that the lane-linear LDS image is conflict-free for
//   * the fragment reads: each wave reads its own raw fragments (2 x ds_read_b128 per fragment), splits them IN REGISTERS
//     (3 x bf16: 44 VALU per fragment, or 2 x fp16) and feeds the MFMAs directly -- no plane writes, one barrier per stage, and
//     the split VALU of a wave runs in the shadow of its own / its SIMD partner's MFMAs.
// Same products, same partial layout ([split][co][ci]) and the same fixed-order reduction as the other weight-gradient kernels
// (tdr_wgrad_mfma.hip reduces; deterministic).  The bias gradient rides along as per-lane sums of the dout fragments.
#include "tdr_common.h"
#include "tdr_wgrad_common.h"
#include "../../include/tdr.h"
#include <stdlib.h>

typedef __bf16 g1bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 g1f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned g1u32x4 __attribute__((ext_vector_type(4)));

namespace {

enum { G1_BX3 = 0, G1_HX2 = 1 };      // WgArgs.scheme: 3-way bf16 split (6 products) / 2-way fp16 split (3 products)

constexpr int G1_PX = 32;             // pixels per stage (two k-steps of 16)

// one fragment: 8 consecutive fp32 values of a row -> NS packed 16-bit planes
template <int SCH>
__device__ __forceinline__ void g1_split8(const f32x4& a, const f32x4& b, g1u32x4 (&p)[SCH == G1_BX3 ? 3 : 2]) {
    if constexpr (SCH == G1_BX3) {
        unsigned h[4], m[4], l[4];
        tdr_split3_bf16<false>(a[0], a[1], h[0], m[0], l[0]);
        tdr_split3_bf16<false>(a[2], a[3], h[1], m[1], l[1]);
        tdr_split3_bf16<false>(b[0], b[1], h[2], m[2], l[2]);
        tdr_split3_bf16<false>(b[2], b[3], h[3], m[3], l[3]);
        p[0] = (g1u32x4){h[0], h[1], h[2], h[3]};
        p[1] = (g1u32x4){m[0], m[1], m[2], m[3]};
        p[2] = (g1u32x4){l[0], l[1], l[2], l[3]};
    } else {
        unsigned h[4], m[4];
        tdr_split2_f16<false>(a[0], a[1], h[0], m[0]);
        tdr_split2_f16<false>(a[2], a[3], h[1], m[1]);
        tdr_split2_f16<false>(b[0], b[1], h[2], m[2]);
        tdr_split2_f16<false>(b[2], b[3], h[3], m[3]);
        p[0] = (g1u32x4){h[0], h[1], h[2], h[3]};
        p[1] = (g1u32x4){m[0], m[1], m[2], m[3]};
    }
}

// LDS-DMA by inline asm: hipcc's own bookkeeping would put s_waitcnt vmcnt(0) in front of every LDS read that follows a builtin
// LDS-DMA; issued this way the pieces are waited for by the counted vmcnt in front of the stage barrier only
__device__ __forceinline__ void g1_glds(const char* src, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
}

// Workgroup = 2 x 2 waves, wave (wm, wn) owns TMW x TNW 32 x 32 tiles: BM = 64 TMW output channels x BN = 64 TNW input channels.
// RING stages of raw rows in LDS: stage s lives in slot s % RING.
template <int TMW, int TNW, bool GATE, int SCH, int RING>
__global__ __launch_bounds__(256, 2) void wgrad1x1_dma_kernel(WgArgs a) {
    constexpr int NS = SCH == G1_BX3 ? 3 : 2, NP = SCH == G1_BX3 ? 6 : 3;
    constexpr int BM = 64 * TMW, BN = 64 * TNW, NB = GATE ? 2 : 1;
    constexpr int ROWS = BM + NB * BN;                 // raw rows per stage: dout rows | input rows | (gate partners)
    constexpr int STAGE = ROWS * 128;                  // bytes
    constexpr int PIECES = ROWS / 8;                   // 1 KiB LDS-DMA instructions per stage (8 rows x 128 B each)
    static_assert(PIECES % 4 == 0, "every wave issues the same number of pieces");
    constexpr int PPW = PIECES / 4;
    static_assert(RING == 2 || RING == 3, "ring of 2 or 3 stages");

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int j = lane & 31, kg = lane >> 5;

    const int split = blockIdx.x;
    const int n = split / a.spi;
    const int s_begin = (split % a.spi) * a.tps;                // stages (32 pixels each) of this block: [s_begin, s_end)
    const int s_end = min(s_begin + a.tps, a.tpi);
    const int nst = s_end - s_begin;
    const int co0 = blockIdx.y * BM, ci0 = blockIdx.z * BN;
    const long HW = (long)a.OH * a.OW;
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* do_n = a.dout + (long)n * a.dout_ns;

    // ---- LDS-DMA sources of this thread: piece q = i * 4 + wave covers stage rows 8q .. 8q + 7; lane -> (row 8q + lane / 8,
    // LDS chunk position lane % 8), which holds GLOBAL chunk (position ^ ((row >> 1) & 7)) of that row's 32 pixels
    const char* src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = 8 * (i * 4 + wave) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const float* base;
        if (row < BM) base = do_n + (long)min(co0 + row, a.Cout - 1) * HW;
        else if (row < BM + BN) base = in_n + (long)min(ci0 + row - BM, a.Cin - 1) * HW;
        else base = in_n + (long)min(ci0 + row - BM - BN, a.Cin - 1) * HW + a.gate_off;
        src[i] = reinterpret_cast<const char*>(base + (long)s_begin * G1_PX + 4 * c);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
    auto issue = [&](int s) {                                   // stage s (relative to s_begin) -> slot s % RING
        const unsigned dst = lds0 + (unsigned)((s % RING) * STAGE) + (unsigned)(wave * 1024);
        const long adv = (long)s * (G1_PX * 4);
#pragma unroll
        for (int i = 0; i < PPW; ++i) g1_glds(src[i] + adv, __builtin_amdgcn_readfirstlane(dst + i * 4096));
    };

    // ---- fragment addresses inside a stage: row r, pixels 16 u + 8 kg .. + 7 = chunks 4u + 2kg, 4u + 2kg + 1
    int a_off[TMW], b_off[TNW], a_sw[TMW], b_sw[TNW];
#pragma unroll
    for (int x = 0; x < TMW; ++x) {
        const int r = (wm * TMW + x) * 32 + j;
        a_off[x] = r * 128; a_sw[x] = (r >> 1) & 7;
    }
#pragma unroll
    for (int y = 0; y < TNW; ++y) {
        const int r = BM + (wn * TNW + y) * 32 + j;
        b_off[y] = r * 128; b_sw[y] = (r >> 1) & 7;
    }

    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int x = 0; x < TMW; ++x)
#pragma unroll
        for (int y = 0; y < TNW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    float dsum[TMW];
#pragma unroll
    for (int x = 0; x < TMW; ++x) dsum[x] = 0.f;
    const bool want_db = a.dbpart != nullptr && blockIdx.z == 0 && wn == 0;      // wave-uniform

    constexpr int SA[6] = {SCH == G1_HX2 ? 1 : 2, 0, SCH == G1_HX2 ? 0 : 1, 1, 0, 0};      // hx2: mh hm hh ; bx3: lh hl mm mh hm hh
    constexpr int SB[6] = {0, SCH == G1_HX2 ? 1 : 2, SCH == G1_HX2 ? 0 : 1, 0, 1, 0};
    auto mma = [](const g1u32x4& x, const g1u32x4& y, const f32x16& c) {
        if constexpr (SCH == G1_BX3)
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g1bf16x8, x), __builtin_bit_cast(g1bf16x8, y), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(g1f16x8, x), __builtin_bit_cast(g1f16x8, y), c, 0, 0, 0);
    };

    // ---- prologue: RING - 1 stages in flight
#pragma unroll
    for (int s = 0; s < RING - 1; ++s)
        if (s < nst) issue(s);

    for (int s = 0; s < nst; ++s) {
        // stage s has landed (this wave's pieces: everything but the youngest RING - 2 stages), then every wave's (barrier); the
        // barrier also says every wave is done reading stage s - 1, whose slot the next issue overwrites
        if (RING == 3 && s + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + RING - 1 < nst) issue(s + RING - 1);
        const unsigned char* st = smem_raw + (s % RING) * STAGE;
        // software pipeline inside the stage: the raw fragments of k-step 1 are read and split WHILE the MFMAs of k-step 0 run (one
        // MFMA, then a handful of the split's VALU: an in-order wave otherwise spends split + MFMA time back to back)
        g1u32x4 af[2][TMW][NS], bf[2][TNW][NS];
        auto frags = [&](int u, g1u32x4 (&fa)[TMW][NS], g1u32x4 (&fb)[TNW][NS]) {
            const int c0 = 4 * u + 2 * kg;
#pragma unroll
            for (int x = 0; x < TMW; ++x) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + a_off[x] + ((c0 ^ a_sw[x]) << 4));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + a_off[x] + (((c0 + 1) ^ a_sw[x]) << 4));
                if (want_db) dsum[x] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
                g1_split8<SCH>(v0, v1, fa[x]);
            }
#pragma unroll
            for (int y = 0; y < TNW; ++y) {
                f32x4 v0 = *reinterpret_cast<const f32x4*>(st + b_off[y] + ((c0 ^ b_sw[y]) << 4));
                f32x4 v1 = *reinterpret_cast<const f32x4*>(st + b_off[y] + (((c0 + 1) ^ b_sw[y]) << 4));
                if constexpr (GATE) {
                    v0 *= *reinterpret_cast<const f32x4*>(st + b_off[y] + BN * 128 + ((c0 ^ b_sw[y]) << 4));
                    v1 *= *reinterpret_cast<const f32x4*>(st + b_off[y] + BN * 128 + (((c0 + 1) ^ b_sw[y]) << 4));
                    asm volatile("" : "+v"(v0), "+v"(v1));          // every plane from the same rounded product
                }
                g1_split8<SCH>(v0, v1, fb[y]);
            }
        };
        auto mmas = [&](const g1u32x4 (&fa)[TMW][NS], const g1u32x4 (&fb)[TNW][NS]) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int x = 0; x < TMW; ++x)
#pragma unroll
                    for (int y = 0; y < TNW; ++y) acc[x][y] = mma(fa[x][SA[p]], fb[y][SB[p]], acc[x][y]);
        };
        frags(0, af[0], bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        frags(1, af[1], bf[1]);
        mmas(af[0], bf[0]);
        {
            constexpr int NM = NP * TMW * TNW;
            constexpr int VPM = ((TMW + TNW) * (SCH == G1_BX3 ? 44 : 12) + (GATE ? 8 * TNW : 0) + 8 * TMW + NM - 1) / NM + 1;   // VALU per MFMA slot
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TMW + TNW) * (GATE ? 2 : 1), 0);      // the raw fragment reads first
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        mmas(af[1], bf[1]);
    }

    // ---- bias gradient partial: row j of tile x = the two lane halves' sums (fixed order)
    if (want_db) {
#pragma unroll
        for (int x = 0; x < TMW; ++x) {
            const float t = dsum[x] + __shfl_xor(dsum[x], 32, 64);
            const int co = co0 + (wm * TMW + x) * 32 + j;
            if (kg == 0 && co < a.Cout) a.dbpart[(long)split * a.Cout + co] = t;
        }
    }
    // ---- partial[split][co][ci]
    float* part = a.part + (long)split * a.Cout * a.Cin;
#pragma unroll
    for (int x = 0; x < TMW; ++x)
#pragma unroll
        for (int y = 0; y < TNW; ++y) {
            const int ci = ci0 + (wn * TNW + y) * 32 + j;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wm * TMW + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (co < a.Cout) part[(long)co * a.Cin + ci] = acc[x][y][r];
            }
        }
}

template <int TMW, int TNW, bool GATE, int SCH, int RING>
int launch_g1(const WgArgs& a, int N, hipStream_t st) {
    constexpr int BM = 64 * TMW, BN = 64 * TNW;
    constexpr size_t lds = (size_t)RING * (BM + (GATE ? 2 : 1) * BN) * 128;
    static_assert(lds <= 160 * 1024, "LDS");
    dim3 grid(N * a.spi, tdr_cdiv(a.Cout, BM), tdr_cdiv(a.Cin, BN));
    auto kern = wgrad1x1_dma_kernel<TMW, TNW, GATE, SCH, RING>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad1x1_dma_kernel");
    return TDR_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// SPLIT-ONCE kernel (round 5, second design): the DMA kernel above splits every operand value in every wave that multiplies it
// (2 x 2 waves: each value twice, 7.3 VALU per MFMA -- measured VALU-pipe-bound: 2.45 us per (workgroup, 32-pixel stage) at any
// occupancy and any contraction length, profiles/r5/probe_wgrad1x1_longk.log).  Here a workgroup of 8 waves splits each value of a
// stage ONCE: thread t loads 4 pixels (one dwordx4; 8 lanes = one 128-byte row segment) of rows t/8 + 64 i, splits them in
// registers and writes the bf16 (fp16) planes of the NEXT stage into the other LDS buffer while the MFMAs of the current stage run:
// 22 split VALU + 3 ds_write_b64 per load, 88 VALU per 24 MFMAs of a wave.  The 8 waves are 2 k-groups x (2 x 2) 64 x 64
// quadrants of the 128 x 128 output tile: wave (kq, wm, wn) multiplies k-step kq (16 pixels) of every stage, the two k-groups are
// summed through LDS in a fixed order at the end (in-block K split: the partial volume of a 256-block launch, two waves per SIMD).
// Plane image of a k-step: [plane][row][16 pixels] bf16 = 32-byte rows, the two 16-byte chunks of a row swapped on rows with
// bit 3 set so that the four 16-lane groups of a ds_read_b128 (lanes {0-3, 12-15, 20-27}, ..) touch 64 distinct banks; the second
// k-step's image sits 64 bytes off a multiple of 128 so that the 16 lanes of a ds_write_b64 group (2 rows x 8 pieces) do too.
// KQ = 1: the same pipeline on 4 waves and 16-pixel stages (no in-block K split): the footprint of the LDS-DMA kernel (256 threads, 49 KiB),
// for launches that should share a CU with the kernels of another stream.
// GRP: one launch over many problems of ONE shape (tdr_wgrad1x1_group): blockIdx.x = problem * grp_bpp + split, the problem's operand /
// partial pointers come from a table in device memory.
// TNW: 32-column tiles per wave along the input channels: 2 = 128 x 128 output tiles, 1 = 128 x 64 (layers with 64 input channels: the
// 64 x 64-tile LDS-DMA form splits every value in the wave that multiplies it -- 14.7 VALU per MFMA, 0.07 of the matrix ceiling there).
// TMW: 32-row tiles per wave along the output channels: 2 = 128-row tiles; 4 = 256 rows (with TNW = 4: 256 x 256 output tiles, 16 accumulators
// per wave in AGPRs, one wave per SIMD): every operand value is split by half as many workgroups and fetched half as often -- the grouped
// launch measured 2.2 x its algorithmic HBM bytes on 128 x 128 tiles (the tiles of a pair do not stay in step inside an XCD).
template <bool GATE, int SCH, int KQ, bool GRP = false, int TNW = 2, int TMW = 2>
__global__ __launch_bounds__(256 * KQ) void wgrad1x1_sp_kernel(WgArgs a) {
    static_assert(KQ == 1 || TMW == 2, "the in-block K split is built for 128-row tiles");
    constexpr int BM = 64 * TMW, BN = 64 * TNW, NI = TMW + TNW;        // rows of the tile; 64-row groups a thread loads from
    constexpr int NS = SCH == G1_BX3 ? 3 : 2, NP = SCH == G1_BX3 ? 6 : 3;
    constexpr int ROWS = BM + BN;                      // BM dout rows | BN input rows
    constexpr int PLANE = ROWS * 32;                   // bytes of one plane of one k-step
    constexpr int KSTR = NS * PLANE + 64;              // k-step image stride
    constexpr int BUF = KQ * KSTR;                     // one stage
    constexpr int PXS = 16 * KQ;                       // pixels per stage
    constexpr int NL = GATE ? NI + TNW : NI;           // dwordx4 loads per thread and stage

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = KQ == 2 ? wave >> 2 : 0, wm = (wave >> 1) & 1, wn = wave & 1;
    const int j = lane & 31, kg = lane >> 5;

    // GRP: 1-D grid in chunks of 8 T blocks (T = output tiles of a problem): block r of a chunk is tile r / 8 of pair r % 8 -- the
    // hardware deals consecutive workgroups round-robin to the 8 XCDs, so the T tiles of one (problem, image, split) pair land on ONE
    // XCD at about the same time and share its L2 (each operand row is needed by Cout / 128 or Cin / 128 of them); in (pair, tile)
    // order they ran on different XCDs at different times and every tile fetched its rows from HBM (2.7 x the bytes).
    int prob = 0, split = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if constexpr (GRP) {
        const int T = a.tiles_x * a.tiles_y;
        const int chunk = (int)blockIdx.x / (8 * T), r = (int)blockIdx.x - chunk * (8 * T);
        const int pair = chunk * 8 + (r & 7), tile = r >> 3;
        if (pair >= a.grp_pairs) return;
        prob = pair / a.grp_bpp; split = pair - prob * a.grp_bpp;
        by = tile % a.tiles_x; bz = tile / a.tiles_x;
        const TdrWg1GroupEntry e = static_cast<const TdrWg1GroupEntry*>(a.grp_tab)[prob];
        a.in = e.in; a.dout = e.dout; a.part = e.part; a.dbpart = e.dbpart;
    }
    const int n = split / a.spi;
    const int s_begin = (split % a.spi) * a.tps;                // (the plan counts 32-pixel stages)
    const int s_end = min(s_begin + a.tps, a.tpi);
    const int nst = (s_end - s_begin) * (2 / KQ);
    const int co0 = by * BM, ci0 = bz * BN;
    const long HW = (long)a.OH * a.OW;
    const float* in_n = a.in + (long)n * a.in_ns;
    const float* do_n = a.dout + (long)n * a.dout_ns;

    // ---- loader role: piece lp (4 pixels) of rows lr + 64 i
    const int lp = tid & (4 * KQ - 1), lr = tid >> (KQ == 2 ? 3 : 2);
    const float* src[NL];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = lr + 64 * i;
        const float* base = i < TMW ? do_n + (long)min(co0 + row, a.Cout - 1) * HW : in_n + (long)min(ci0 + row - BM, a.Cin - 1) * HW;
        src[i] = base + (long)s_begin * G1_PX + 4 * lp;
    }
    if constexpr (GATE) {
#pragma unroll
        for (int i = TMW; i < NI; ++i) src[i + TNW] = src[i] + a.gate_off;
    }
    const int w_off = (lp >> 2) * KSTR + lr * 32 + ((((lp >> 1) & 1) ^ ((lr >> 3) & 1)) << 4) + (lp & 1) * 8;     // + i * 64 * 32 + plane * PLANE

    // ---- multiplier role
    int a_off[TMW], b_off[TNW];
#pragma unroll
    for (int x = 0; x < TMW; ++x) a_off[x] = kq * KSTR + ((wm * TMW + x) * 32 + j) * 32 + ((kg ^ ((j >> 3) & 1)) << 4);
#pragma unroll
    for (int y = 0; y < TNW; ++y) b_off[y] = kq * KSTR + (BM + (wn * TNW + y) * 32 + j) * 32 + ((kg ^ ((j >> 3) & 1)) << 4);
    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int x = 0; x < TMW; ++x)
#pragma unroll
        for (int y = 0; y < TNW; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    float dsum[TMW];
#pragma unroll
    for (int x = 0; x < TMW; ++x) dsum[x] = 0.f;
    const bool want_db = a.dbpart != nullptr && bz == 0;      // workgroup-uniform

    constexpr int SA[6] = {SCH == G1_HX2 ? 1 : 2, 0, SCH == G1_HX2 ? 0 : 1, 1, 0, 0};      // hx2: mh hm hh ; bx3: lh hl mm mh hm hh
    constexpr int SB[6] = {0, SCH == G1_HX2 ? 1 : 2, SCH == G1_HX2 ? 0 : 1, 0, 1, 0};
    auto mma = [](const g1u32x4& x, const g1u32x4& y, const f32x16& c) {
        if constexpr (SCH == G1_BX3)
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g1bf16x8, x), __builtin_bit_cast(g1bf16x8, y), c, 0, 0, 0);
        else
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(g1f16x8, x), __builtin_bit_cast(g1f16x8, y), c, 0, 0, 0);
    };

    auto load = [&](int s, f32x4 (&r)[NL]) {
        const long adv = (long)min(s, nst - 1) * PXS;       // past the end: a harmless re-read (keeps the loop branch-free)
#pragma unroll
        for (int i = 0; i < NL; ++i) r[i] = *reinterpret_cast<const f32x4*>(src[i] + adv);
    };
    auto split_store = [&](f32x4 (&r)[NL], int wr, bool live) {
        unsigned char* dst = smem_raw + wr + w_off;
        // stages past the end are split like the others (branch-free pipeline) with their dout planes zeroed: they add nothing
        const float zf = live ? 1.f : 0.f, lf = want_db ? zf : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 v = r[i];
            if (i < TMW) {
                v *= zf;
                dsum[i] = __builtin_fmaf(lf, (v[0] + v[1]) + (v[2] + v[3]), dsum[i]);
            }
            else if constexpr (GATE) {
                v *= r[i + TNW];
                asm volatile("" : "+v"(v));                  // every plane from the same rounded product
            }
            unsigned p0[NS], p1[NS];
            if constexpr (SCH == G1_BX3) {
                tdr_split3_bf16<false>(v[0], v[1], p0[0], p0[1], p0[2]);
                tdr_split3_bf16<false>(v[2], v[3], p1[0], p1[1], p1[2]);
            } else {
                tdr_split2_f16<false>(v[0], v[1], p0[0], p0[1]);
                tdr_split2_f16<false>(v[2], v[3], p1[0], p1[1]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<uint2*>(dst + i * (64 * 32) + s * PLANE) = make_uint2(p0[s], p1[s]);
        }
    };
    auto read_frags = [&](int rd, g1u32x4 (&fa)[TMW][NS], g1u32x4 (&fb)[TNW][NS]) {
        const unsigned char* st = smem_raw + rd;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int x = 0; x < TMW; ++x) fa[x][s] = *reinterpret_cast<const g1u32x4*>(st + a_off[x] + s * PLANE);
#pragma unroll
            for (int y = 0; y < TNW; ++y) fb[y][s] = *reinterpret_cast<const g1u32x4*>(st + b_off[y] + s * PLANE);
        }
    };
    auto mmas = [&](const g1u32x4 (&fa)[TMW][NS], const g1u32x4 (&fb)[TNW][NS]) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int x = 0; x < TMW; ++x)
#pragma unroll
                for (int y = 0; y < TNW; ++y) acc[x][y] = mma(fa[x][SA[p]], fb[y][SB[p]], acc[x][y]);
    };
    // the split of stage s + 1 rides in the shadow of the MFMAs of stage s: per MFMA a handful of VALU, a plane store every other one
    auto interleave = [&]() {
        constexpr int NM = NP * TMW * TNW;
        constexpr int VPM = (NI * (SCH == G1_BX3 ? 22 : 6) + (GATE ? 4 * TNW : 0) + 16 + NM - 1) / NM + 1;
        __builtin_amdgcn_sched_group_barrier(0x100, (TMW + TNW) * NS, 0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            if (i % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };

    // ---- two stage images in LDS: iteration s multiplies stage s, splits stage s + 1 (raw since iteration s - 1 / s - 2) into the
    // other image and issues the loads of stage s + DIST
    f32x4 raw0[NL], raw1[NL];
    g1u32x4 fa[TMW][NS], fb[TNW][NS];
    load(0, raw0);
    load(1, raw1);
    split_store(raw0, 0, true);
    __syncthreads();
    for (int s = 0; s < nst; s += 2) {           // (an odd count runs one extra, zeroed stage)
        load(s + 2, raw0);
        read_frags(0, fa, fb);
        mmas(fa, fb);
        split_store(raw1, BUF, s + 1 < nst);
        interleave();
        __syncthreads();
        load(s + 3, raw1);
        read_frags(BUF, fa, fb);
        mmas(fa, fb);
        split_store(raw0, 0, s + 2 < nst);
        interleave();
        __syncthreads();
    }

    // ---- bias gradient partial: the 8 pieces of a row live in 8 consecutive lanes (fixed-order butterfly)
    if (want_db) {
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            float t = dsum[i];
            t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64);
            if constexpr (KQ == 2) t += __shfl_xor(t, 4, 64);
            const int co = co0 + lr + 64 * i;
            if (lp == 0 && co < a.Cout) a.dbpart[(long)split * a.Cout + co] = t;
        }
    }
    // ---- k-group 1 hands its sums to k-group 0 through LDS (the stage buffers are free: the loop ended on a barrier)
    float* red = reinterpret_cast<float*>(smem_raw);            // [quadrant][64 values][64 lanes] = 64 KiB (the stage images are dead)
    const int wq = wave & 3;
    if (KQ == 2 && kq == 1) {
#pragma unroll
        for (int x = 0; x < TMW; ++x)
#pragma unroll
            for (int y = 0; y < TNW; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wq * 4 + x * 2 + y) * 16 + r) * 64 + lane] = acc[x][y][r];
    }
    if constexpr (KQ == 2) {
        __syncthreads();
        if (kq == 1) return;
    }
    float* part = a.part + (long)split * a.Cout * a.Cin;
#pragma unroll
    for (int x = 0; x < TMW; ++x)
#pragma unroll
        for (int y = 0; y < TNW; ++y) {
            const int ci = ci0 + (wn * TNW + y) * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = KQ == 2 ? acc[x][y][r] + red[((wq * 4 + x * 2 + y) * 16 + r) * 64 + lane] : acc[x][y][r];
                const int co = co0 + (wm * TMW + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (co < a.Cout && ci < a.Cin) part[(long)co * a.Cin + ci] = v;
            }
        }
}

template <bool GATE, int SCH, int KQ>
int launch_sp(const WgArgs& a, int N, hipStream_t st) {
    constexpr int NS = SCH == G1_BX3 ? 3 : 2;
    constexpr size_t lds0 = (size_t)2 * KQ * (NS * 256 * 32 + 64);
    constexpr size_t lds = (KQ == 2 && lds0 < 65536) ? 65536 : lds0;
    dim3 grid(N * a.spi, tdr_cdiv(a.Cout, 128), tdr_cdiv(a.Cin, 128));
    auto kern = wgrad1x1_sp_kernel<GATE, SCH, KQ>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256 * KQ), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad1x1_sp_kernel");
    return TDR_OK;
}

int g1_sp() {
    // TDR_WG1_SP: 0 = the LDS-DMA kernel, 1 = 8 waves with the in-block K split, 2 (default) = 4 waves.
    // Standalone the 8-wave form is the fastest (256 -> 512 @ 64^2, N = 4, time per launch incl. the reduction: 36.7 us; 4 waves 41.6;
    // LDS-DMA 44.3 -- 1.54 vs 1.93 us per stage, fixed cost 10 vs 14 us: profiles/r5/probe_wgrad1x1_fixed.log), and on one stream the
    // step gains 0.4 ms with it (70.5 vs 70.9 ms).  But the default step runs the deferred leaves NEXT TO the MASA-encoder backward, and
    // there a 512-thread / 96 KiB workgroup shares a CU with nothing: the overlap that is worth 2.6 ms with a 256-thread / 64 KiB
    // kernel shrinks to 1.1 ms (68.3 -> 69.4 ms same box, profiles/r5/sweep_e.log).  The 4-wave form keeps the footprint of the
    // LDS-DMA kernel (256 threads, 49 KiB) and its overlap: 67.63 -> 67.31 ms (sweep_g.log).
    static const int v = tdr_tune_env("TDR_WG1_SP") ? atoi(tdr_tune_env("TDR_WG1_SP")) : 2;
    return v;
}

int g1_ring() {
    static const int r = tdr_tune_env("TDR_WG1_RING") ? atoi(tdr_tune_env("TDR_WG1_RING")) : 2;
    return r == 3 ? 3 : 2;
}

template <int TMW, int TNW, int SCH>
int launch_g1_gr(const WgArgs& a, const TdrWgradDesc* d, hipStream_t st) {
    const bool g = d->gate != 0;
    if (g1_ring() == 3 && !g) return launch_g1<TMW, TNW, false, SCH, 3>(a, d->N, st);
    return g ? launch_g1<TMW, TNW, true, SCH, 2>(a, d->N, st) : launch_g1<TMW, TNW, false, SCH, 2>(a, d->N, st);
}

// ---- grouped launch: the deferred leaf weight gradients of a whole level (same N, Cin, Cout, HW, gate) in ONE launch + ONE reduction.
// A per-problem launch is 256 workgroups x 16 stages: a third of its time is ramp, prologue, partial write and the reduction launch
// (profiles/r5/probe_wgrad1x1_fixed.log: 10 - 14 us fixed next to 16 x 1.5 - 1.9 us), and one workgroup per CU leaves every stall
// uncovered.  Grouped, a workgroup takes a whole image of one tile (>= 32 stages, 1 / 8 of the partial volume), thousands of
// workgroups keep 2 - 3 resident per CU, and the fixed-order reduction of all problems is one launch.
template <int KL>
__global__ __launch_bounds__(256) void wgrad1x1_grp_reduce_kernel(const TdrWg1GroupEntry* __restrict__ tab, long elems, int nsplit, int Cout, int nb_main) {
    const TdrWg1GroupEntry e = tab[blockIdx.y];
    const bool second = (int)blockIdx.x >= nb_main;
    const long ne = second ? Cout : elems;
    const float* p = second ? e.dbpart : e.part;
    float* o = second ? e.db : e.g;
    const long i = ((int)blockIdx.x - (second ? nb_main : 0)) * 256L + threadIdx.x;
    if (i >= ne || o == nullptr) return;
    float s0 = 0.f;
    for (int k = 0; k < nsplit; k += 8) {            // fixed order, eight partials in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k + u < nsplit ? p[(long)(k + u) * ne + i] : 0.f;
        s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    o[i] = s0;
}

struct G1GroupPlan { int spi, tps, tpi, bpp; };
struct G1Tile { int tmw, tnw; };

// output tile of a grouped launch: 128 x 128 (128 x 64 for 64 input channels).  TDR_WG1_GRP_BIG=1: 256 x 256 / 256 x 128 where the channel
// counts fill them -- half the split VALU and half the operand fetches per MFMA, but one wave per SIMD (16 accumulator tiles): measured
// SLOWER in the step (64.4 -> 64.9 / 65.5 ms same box, profiles/r5/sweep_o.log), so opt-in
G1Tile g1_group_tile(const TdrWgradDesc* d) {
    static const bool big = tdr_tune_env("TDR_WG1_GRP_BIG") && atoi(tdr_tune_env("TDR_WG1_GRP_BIG")) == 1;
    G1Tile t;
    t.tmw = (big && d->Cout >= 256 && d->Cin > 64) ? 4 : 2;
    t.tnw = d->Cin <= 64 ? 1 : ((t.tmw == 4 && d->Cin >= 256) ? 4 : 2);      // (instantiated: 4x4, 4x2, 2x2, 2x1)
    return t;
}

G1GroupPlan g1_group_plan(const TdrWgradDesc* d, int nprob) {
    G1GroupPlan p;
    const long HW = (long)d->OH * d->OW;
    p.tpi = (int)(HW / G1_PX);
    const G1Tile tl = g1_group_tile(d);
    const long tiles = (long)tdr_cdiv(d->Cout, 64 * tl.tmw) * tdr_cdiv(d->Cin, 64 * tl.tnw);
    // one image per workgroup unless that leaves the chip short of workgroups (few problems / few tiles): then split the images,
    // never below 8 stages per workgroup
    static const long want = tdr_tune_env("TDR_WG1_GRP_WANT") ? atol(tdr_tune_env("TDR_WG1_GRP_WANT")) : 512;      // (256 / 512 / 1024 / 2048: 67.06 / 67.12 / 67.33 / 67.39 ms per step, profiles/r5/sweep_k.log)
    long spi = 1;
    while ((long)nprob * d->N * spi * tiles < want && p.tpi / (spi * 2) >= 8) spi *= 2;
    p.tps = tdr_cdiv(p.tpi, spi);
    p.spi = tdr_cdiv(p.tpi, p.tps);
    p.bpp = d->N * p.spi;
    return p;
}

template <bool GATE, int SCH, int TNW, int TMW>
int launch_grp(const WgArgs& a, int nprob, hipStream_t st) {
    constexpr int NS = SCH == G1_BX3 ? 3 : 2;
    constexpr size_t lds0 = (size_t)2 * (NS * (64 * TMW + 64 * TNW) * 32 + 64);
    // TDR_WG1_GRP_LDS: pad the LDS request (bytes) to cap the workgroups resident per CU (tuning aid: fewer pairs in flight per XCD L2)
    static const size_t pad = tdr_tune_env("TDR_WG1_GRP_LDS") ? (size_t)atol(tdr_tune_env("TDR_WG1_GRP_LDS")) : 0;
    const size_t lds = lds0 < pad ? pad : lds0;
    const int T = a.tiles_x * a.tiles_y;
    dim3 grid((unsigned)tdr_cdiv(a.grp_pairs, 8) * 8 * T);
    auto kern = wgrad1x1_sp_kernel<GATE, SCH, 1, true, TNW, TMW>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad1x1_sp_kernel<grouped>");
    return TDR_OK;
}

}  // namespace

extern "C" int tdr_wgrad1x1_group_supported(const TdrWgradDesc* d) {
    if (!d || (d->math != 1 && d->math != 2) || d->per_image) return 0;
    if (d->KH != 1 || d->stride != 1 || d->pad != 0 || d->H != d->OH || d->W != d->OW) return 0;
    const long HW = (long)d->OH * d->OW;
    if (HW % G1_PX != 0 || d->in_ns % 4 != 0 || d->dout_ns % 4 != 0) return 0;
    return d->Cin >= 64 && d->Cout > 64;          // (Cin = 64: 128 x 64 output tiles)
}

// floats of workspace PER PROBLEM: [bpp][Cout][Cin] partials, then [bpp][Cout] bias-gradient partials
extern "C" int64_t tdr_wgrad1x1_group_ws_floats(const TdrWgradDesc* d, int nprob) {
    const G1GroupPlan p = g1_group_plan(d, nprob);
    return (int64_t)p.bpp * d->Cout * d->Cin + (int64_t)p.bpp * d->Cout;
}

// d: the common shape (N, Cin, Cout, OH, OW, in_ns, dout_ns, gate, math); its pointers are ignored.  table: TdrWg1GroupEntry[nprob] in
// DEVICE memory (in, dout, part = the problem's workspace, dbpart = part + bpp * Cout * Cin or NULL, g, db or NULL).
extern "C" int tdr_wgrad1x1_group(const TdrWgradDesc* d, int nprob, const void* table, void* stream) {
    TDR_REQUIRE(d && table && nprob > 0, "tdr_wgrad1x1_group: null argument");
    TDR_REQUIRE(tdr_wgrad1x1_group_supported(d), "tdr_wgrad1x1_group: shape not supported (1x1, whole 32-pixel stages, channels > 64, split arithmetic)");
    const G1GroupPlan p = g1_group_plan(d, nprob);
    WgArgs a;
    a.in = nullptr; a.in_ns = d->in_ns; a.Cin = d->Cin; a.H = d->H; a.W = d->W;
    a.gate_off = (long)d->Cin * d->H * d->W;
    a.dout = nullptr; a.dout_ns = d->dout_ns; a.Cout = d->Cout; a.OH = d->OH; a.OW = d->OW;
    const G1Tile tl = g1_group_tile(d);
    a.pad = 0; a.tw_log2 = 5; a.tiles_x = tdr_cdiv(d->Cout, 64 * tl.tmw); a.tiles_y = tdr_cdiv(d->Cin, 64 * tl.tnw);      // (grouped: the output tile grid)
    a.tpi = p.tpi; a.tps = p.tps; a.spi = p.spi;
    a.part = nullptr; a.dbpart = nullptr;
    a.scheme = d->math == 2 ? 1 : 0;
    a.grp_tab = table; a.grp_bpp = p.bpp; a.grp_pairs = nprob * p.bpp;
    hipStream_t st = (hipStream_t)stream;
    const bool g = d->gate != 0;
    int rc;
#define G1_GRP(GATE_, SCH_)                                                                                                       \
    (tl.tmw == 4 ? (tl.tnw == 4 ? launch_grp<GATE_, SCH_, 4, 4>(a, nprob, st) : launch_grp<GATE_, SCH_, 2, 4>(a, nprob, st))      \
                 : (tl.tnw == 2 ? launch_grp<GATE_, SCH_, 2, 2>(a, nprob, st) : launch_grp<GATE_, SCH_, 1, 2>(a, nprob, st)))
    if (a.scheme == 1) rc = g ? G1_GRP(true, G1_HX2) : G1_GRP(false, G1_HX2);
    else rc = g ? G1_GRP(true, G1_BX3) : G1_GRP(false, G1_BX3);
#undef G1_GRP
    if (rc != TDR_OK) return rc;
    const long elems = (long)d->Cout * d->Cin;
    const int nb_main = tdr_cdiv(elems, 256), nb2 = tdr_cdiv(d->Cout, 256);
    hipLaunchKernelGGL(wgrad1x1_grp_reduce_kernel<1>, dim3(nb_main + nb2, nprob), dim3(256), 0, st, static_cast<const TdrWg1GroupEntry*>(table), elems,
                       p.bpp, d->Cout, nb_main);
    TDR_LAUNCH_CHECK("wgrad1x1_grp_reduce_kernel");
    return TDR_OK;
}

namespace {
}  // namespace

// 1x1 / stride 1 / pad 0 on a split scheme, rows of whole 32-pixel stages, channel counts that fill whole 64-row tiles
bool tdr_wgrad_1x1_supported(const TdrWgradDesc* d) {
    static const bool off = tdr_tune_env("TDR_WG1") && atoi(tdr_tune_env("TDR_WG1")) == 0;     // A/B aid: 0 = the staged kernel of rounds 1-4
    if (off || (d->math != 1 && d->math != 2)) return false;
    if (d->KH != 1 || d->stride != 1 || d->pad != 0 || d->H != d->OH || d->W != d->OW) return false;
    const long HW = (long)d->OH * d->OW;
    if (HW % G1_PX != 0 || d->in_ns % 4 != 0 || d->dout_ns % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(d->in) & 15) || (reinterpret_cast<uintptr_t>(d->dout) & 15)) return false;
    return d->Cin >= 64 && d->Cout >= 64;
}

WgPlan tdr_wgrad_1x1_plan(const TdrWgradDesc* d) {
    WgPlan p;
    p.tw_log2 = 5;
    p.cfg = (d->Cout > 64 && d->Cin > 64) ? 0 : 1;              // 128 x 128 | 64 x 64 output tiles
    p.BMc = p.cfg == 0 ? 128 : 64; p.BNc = p.BMc;
    p.WKw = 1;
    const long HW = (long)d->OH * d->OW;
    p.tiles_x = (int)(HW / G1_PX); p.tiles_y = 1;
    p.tpi = p.tiles_x;                                           // "tiles" = 32-pixel stages of the flattened image
    const long out_tiles = (long)tdr_cdiv(d->Cout, p.BMc) * tdr_cdiv(d->Cin, p.BNc);
    // bx3: 256 blocks (one per CU; half the split-K partials: 68.66 -> 68.24 ms per step same box, profiles/r5/sweep_c.log); hx2: one round of 2 per CU
    static const long want_env = tdr_tune_env("TDR_WG1_WANT") ? atol(tdr_tune_env("TDR_WG1_WANT")) : 0;
    const long want_total = want_env > 0 ? want_env : (d->math == 1 ? 256 : 512);
    long want = want_total / out_tiles;
    if (want < 1) want = 1;
    long spi = (want + d->N - 1) / d->N;
    if (spi > p.tpi / 4) spi = p.tpi / 4;                       // at least 4 stages per block
    if (spi < 1) spi = 1;
    p.tps = tdr_cdiv(p.tpi, spi);
    p.spi = tdr_cdiv(p.tpi, p.tps);
    return p;
}

int tdr_wgrad_1x1_launch(const WgArgs& a, const WgPlan& p, const TdrWgradDesc* d, hipStream_t st) {
    if (p.cfg == 0 && g1_sp() == 1) {
        const bool g = d->gate != 0;
        if (a.scheme == 1) return g ? launch_sp<true, G1_HX2, 2>(a, d->N, st) : launch_sp<false, G1_HX2, 2>(a, d->N, st);
        return g ? launch_sp<true, G1_BX3, 2>(a, d->N, st) : launch_sp<false, G1_BX3, 2>(a, d->N, st);
    }
    if (p.cfg == 0 && g1_sp() == 2) {
        const bool g = d->gate != 0;
        if (a.scheme == 1) return g ? launch_sp<true, G1_HX2, 1>(a, d->N, st) : launch_sp<false, G1_HX2, 1>(a, d->N, st);
        return g ? launch_sp<true, G1_BX3, 1>(a, d->N, st) : launch_sp<false, G1_BX3, 1>(a, d->N, st);
    }
    if (a.scheme == 1) return p.cfg == 0 ? launch_g1_gr<2, 2, G1_HX2>(a, d, st) : launch_g1_gr<1, 1, G1_HX2>(a, d, st);
    return p.cfg == 0 ? launch_g1_gr<2, 2, G1_BX3>(a, d, st) : launch_g1_gr<1, 1, G1_BX3>(a, d, st);
}
