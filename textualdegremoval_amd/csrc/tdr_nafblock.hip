// Fused second half of a NAFBlock for the deep U-Net level (c = 256 at 64x64: 28 of the 46 blocks of BASELINE
// configs[1]) -- reference models/archs/network_nafnet_guided_arch.py:226-238:
//
//     y   = inp + conv3(g * sca) * beta                 (g = SimpleGate(conv2(conv1(norm1(inp)))), sca per image/channel)
//     yn  = norm2(y)                                    (LayerNorm2d over channels, nafnet_arch_utils.py:264-300)
//     t4  = conv4(yn)
//     out = y + conv5(t4[:, :c] * t4[:, c:]) * gamma
//
// Everything here is per pixel, so one workgroup owns 64 pixels x ALL channels and walks the chain without leaving
// the CU: three back-to-back implicit GEMMs on the 2-way fp16 split (3 x v_mfma_f32_32x32x16_f16 per fp32 product,
// fp32 accumulate -- same arithmetic and packed weights as conv1x1_hx2_kernel), LayerNorm as a cross-wave reduction
// through LDS, the SimpleGate product register-local (wave w owns channels [64w, 64w+64) and [256+64w, 256+64w+64) of
// t4).  The four separate launches (conv3, norm2, conv4, conv5) move 201 MB per block at this level and are single-round,
// latency-bound kernels (26.8 + 13.2 + 36.9 + 26.2 us); fused, the tile is read once (g, inp) and every tensor the
// backward pass keeps (y, mu, rstd, yn, t4) is written once on the way: 118 MB, one launch, no phase of one kernel
// waiting for the tail of the previous one.
//
// Layout notes (all as in tdr_conv_bx3.hip): A = packed weight fragments [group][mt][split][lane] read L2 -> VGPR with
// a ring of PF groups in flight; B = activations in LDS as [split][octet][pixel] 16-byte slots (8 channels x f16), XOR
// swizzled; accumulators in the gfx950 32x32 C/D layout (lane (j, kk): pixel j, rows (r&3) + 8(r>>2) + 4kk).
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 nbf16x8 __attribute__((ext_vector_type(8)));
union HFrag {
    uint4 u;
    f16x8 hv;
    nbf16x8 bv;
};

// Operand split schemes of the fused chains (the same two as tdr_conv_bx3.hip, same packed weights):
//   SCH_HX2  x = h + m, fp16 each, 3 products (mh hm hh)              -- operands inside the fp16 window (TDR_MATH=hx2)
//   SCH_BX3  x = h + m + l, bf16 each, 6 products (lh hl mm mh hm hh) -- 24-bit operands, fp32 range, no loss scale (TDR_MATH=bx3)
// The intermediates of a chain live in LDS as NS planes of 16-byte slots; with three planes the K = 2C operand of the backward
// kernels (192 KiB at C = 256) is staged one K half at a time (KHALF).
enum { SCH_BX3 = 0, SCH_HX2 = 1 };
template <int SCH> struct SchT {
    static constexpr int NS = SCH == SCH_BX3 ? 3 : 2;
    static constexpr int NP = SCH == SCH_BX3 ? 6 : 3;
};
// two fp32 values -> their NS packed split planes (both planes of a value start from the same pinned fp32 value)
template <int SCH>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&p)[SchT<SCH>::NS]) {
    if constexpr (SCH == SCH_HX2) tdr_split2_f16(x0, x1, p[0], p[1]);
    else tdr_split3_bf16(x0, x1, p[0], p[1], p[2]);
}

constexpr int NPX = 64;                       // pixels per workgroup

// Phase timeline (probe builds only: make probe -> libtdr_hip_probe.so with -DTDR_NB_PROBE; profiles/probe_nafblock_timeline.py):
// s_memtime stamps of waves 0 and 5 of workgroups (0, 0) and (37, 2) at the phase boundaries of the chain kernels.
#ifdef TDR_NB_PROBE
__device__ unsigned long long nb_probe_buf[4 * 16];
#define NB_STAMP(k)                                                                                         \
    do {                                                                                                    \
        const int wg_ = (blockIdx.x == 0 && blockIdx.y == 0) ? 0 : ((blockIdx.x == 37 && blockIdx.y == 2) ? 1 : -1); \
        if (wg_ >= 0 && (wave == 0 || wave == 5) && lane == 0)                                              \
            nb_probe_buf[(wg_ * 2 + (wave == 5)) * 16 + (k)] = __builtin_amdgcn_s_memtime();                \
    } while (0)
#else
#define NB_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ int swz(int slot) { return slot ^ ((slot >> 4) & 3); }
__device__ __forceinline__ int row_of(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }

// x -> (h, m), h = rn_f16(x), m = rn_f16(x - h).  The value is pinned in a VGPR first: when x is a product a * b the
// compiler is otherwise free to form h from the exact product (v_fma_mixlo_f16) and the residual from the rounded one (or
// the other way round) -- at an fp16 tie of the rounded product the two disagree about the neighbour and h + m is off by a
// whole ulp of h (measured: 1e-4 outliers in the conv4 data gradient).
__device__ __forceinline__ void split_hm(float x, _Float16& h, _Float16& m) {
    asm volatile("" : "+v"(x));
    h = (_Float16)x;
    m = (_Float16)(x - (float)h);
}

// acc[tm][tn] += W[mt_of(tm)] (K = 16 * NG channels) x B(LDS planes).  PF groups of A fragments in flight.
// side(g) is called once per 16-channel group, between the MFMAs: the caller's global stores of the PREVIOUS phase's
// tiles ride there, a few per group, so the store stream drains under the matrix work instead of in front of it.
template <int SCH, int TMW, int NG, int PF, typename MtOf, typename Side>
__device__ __forceinline__ void gemm_split(f32x16 (&acc)[TMW][2], const uint4* __restrict__ wp, int MT, MtOf mt_of, const uint4* sB,
                                           int noct, int lane, int rot, Side side) {
    constexpr int NS = SchT<SCH>::NS, NP = SchT<SCH>::NP;
    // rot: every workgroup walks the K groups from a different starting group.  All workgroups of the launch stream the
    // SAME weight fragments; started together they would ask the same few L2 lines at the same moment.
    static_assert((NG & (NG - 1)) == 0, "NG must be a power of two");
    const int j = lane & 31, kk = lane >> 5;
    const uint4* wl = wp + lane;
    HFrag af[PF][TMW][NS];
    auto load_a = [&](int slot, int g) {
        const int gr = (g + rot) & (NG - 1);
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
            for (int s = 0; s < NS; ++s) af[slot][tm][s].u = wl[((long)gr * MT + mt_of(tm)) * (NS * 64) + s * 64];
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) load_a(p, p < NG ? p : NG - 1);
    const int b0 = swz(j), b1 = swz(32 + j);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        HFrag bf[2][NS];
        const uint4* sg = sB + (2 * ((g + rot) & (NG - 1)) + kk) * NPX;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            bf[0][s].u = sg[s * noct * NPX + b0];
            bf[1][s].u = sg[s * noct * NPX + b1];
        }
        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};                          // hx2: m*h, h*m, h*h
        constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SB[6] = {0, 2, 1, 0, 1, 0};        // bx3: lh hl mm mh hm hh (small cross terms first)
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    if constexpr (SCH == SCH_HX2)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[g % PF][tm][HA[q]].hv, bf[tn][HB[q]].hv, acc[tm][tn], 0, 0, 0);
                    else
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g % PF][tm][SA[q]].bv, bf[tn][SB[q]].bv, acc[tm][tn], 0, 0, 0);
                }
        if (g + PF < NG) load_a(g % PF, g + PF);
        // pin the request here: left free, the scheduler sinks each fragment load to just in front of its use (to shorten live
        // ranges) and the loop degenerates into load -> s_waitcnt vmcnt(0) -> two MFMAs -> load ... (101 vmcnt(0) in naf_tail_bwd<256>).
        // These kernels sit at the 256-VGPR limit of two waves per SIMD: pinned, PF = 3 - 4 groups spilled more (step +0.6 ms), PF = 2
        // (12 - 24 MFMAs of cover, about an L2 round trip) spills less than the unpinned code did and is 0.1 ms faster
        __builtin_amdgcn_sched_barrier(0);
        side(g);
    }
}

// fp32 values of one accumulator tile (channel rows of octet-halves) -> the two f16 planes of the LDS B operand
template <int SCH>
__device__ __forceinline__ void tile_to_planes(const float (&v)[16], uint4* sB, int noct, int oct0, int pix, int kk) {
    // rows r = 4q..4q+3 are elements 4kk..4kk+3 of octet oct0 + q
    constexpr int NS = SchT<SCH>::NS;
    char* base = reinterpret_cast<char*>(sB);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned p0[NS], p1[NS];
        split_pair<SCH>(v[4 * q], v[4 * q + 1], p0);
        split_pair<SCH>(v[4 * q + 2], v[4 * q + 3], p1);
        const long slot = (long)(oct0 + q) * NPX + swz(pix);
#pragma unroll
        for (int s = 0; s < NS; ++s)
            *reinterpret_cast<uint2*>(base + ((long)s * noct * NPX + slot) * 16 + 8 * kk) = make_uint2(p0[s], p1[s]);
    }
}

// 8 channels x 4 adjacent pixels of fp32 (one float4 per channel, optionally times a per-channel scale) -> the NS planes of octet
// `oct`, pixels 4q .. 4q + 3 of the LDS B operand
template <int SCH, bool SCALE>
__device__ __forceinline__ void stage_octet(const float4 (&v)[8], const float (&sc)[8], uint4* sB, int noct, int oct, int q) {
    constexpr int NS = SchT<SCH>::NS;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        unsigned pl[4][NS];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float e0 = px == 0 ? v[2 * i].x : (px == 1 ? v[2 * i].y : (px == 2 ? v[2 * i].z : v[2 * i].w));
            float e1 = px == 0 ? v[2 * i + 1].x : (px == 1 ? v[2 * i + 1].y : (px == 2 ? v[2 * i + 1].z : v[2 * i + 1].w));
            if constexpr (SCALE) { e0 *= sc[2 * i]; e1 *= sc[2 * i + 1]; }
            split_pair<SCH>(e0, e1, pl[i]);
        }
        const int slot = oct * NPX + swz(4 * q + px);
#pragma unroll
        for (int s = 0; s < NS; ++s) sB[s * noct * NPX + slot] = make_uint4(pl[0][s], pl[1][s], pl[2][s], pl[3][s]);
    }
}

// sum over the NW waves' partials of one pixel column (fixed order)
template <int NW>
__device__ __forceinline__ float wave_partials_sum(const float* rp) {
    float t = rp[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += rp[w * NPX];
    return t;
}

struct TailArgs {
    const float* g; long g_ns;
    const float* sca;                 // [N][C]
    const float* x; long x_ns;
    const uint4 *w3, *w4, *w5;        // packed hx2 fragments (PACK_FWD): M = C, 2C, C; K = C
    const float *b3, *beta, *lnw, *lnb, *b4, *b5, *gamma;
    float eps;
    float* y; long y_ns;
    float *mu, *rs;                   // [N][HW]
    float* yn; long yn_ns;
    float* t4; long t4_ns;
    float* out; long out_ns;
    int HW;
    int c_out;                        // rows of conv5 actually produced (the `[:, :chan]` slice of the fusion blocks, :719,727)
};

// 2C threads = C/32 waves (C = 256: 8 waves, two per SIMD): wave w owns the 32 channels [32w, 32w + 32) of the C-row GEMMs and, in conv4,
// also their SimpleGate partners [C + 32w, C + 32w + 32).  While one wave of a SIMD waits on LDS / L2 / the store
// queue its partner's MFMAs run.
template <int C, int SCH>
__global__ __launch_bounds__(2 * C, 2) void naf_tail_fwd_kernel(TailArgs a) {
    static_assert(C == 256 || C == 128 || C == 64 || C == 32, "C / 32 waves x 32 channel rows");
    constexpr int NS = SchT<SCH>::NS;
    constexpr int NOCT = C / 8;               // octets of the K = C operands
    constexpr int NG = C / 16;
    constexpr int NW = C / 32;
    // three planes at the 256-register limit: tiles that only wait for their side stores inside the next GEMM get spilled, and a spill
    // reload behind queued global stores waits for every one of them (vmcnt is in order) -- there the saved tensors leave right away
    constexpr bool EARLY_STORES = SCH == SCH_BX3 && C >= 64;
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    uint4* sB = smem4;                                        // NS planes x NOCT x 64 px x 16 B = 64 / 96 KiB at C = 256
    float* red = reinterpret_cast<float*>(smem4 + NS * NOCT * NPX);   // [2][8 waves][64 px]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const int n = blockIdx.y;
    const long p0 = (long)blockIdx.x * NPX;
    const long HW = a.HW;
    const int m0 = 32 * wave;                                 // first channel row of this wave
    const int rot = (int)(blockIdx.x * 5);   // (not a function of the image index: batch-permutation equivariance stays bit-exact)
    // row r of this lane: channel m0 + row_of(r, kk); element offset of (row r, pixel j of sub-tile tn) in an [*, HW] image
    auto off = [&](int r, int tn) { return (long)(m0 + row_of(r, kk)) * HW + 32 * tn; };

    NB_STAMP(0);
    // ---- residual tile (inp) in accumulator layout: requested first, consumed after the first GEMM
    float xr[2][16];
    {
        const float* xp = a.x + (long)n * a.x_ns + p0 + j;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[tn][r] = xp[off(r, tn)];
    }
    // ---- stage B = g * sca as f16 planes: thread (oct, q) owns pixels 4q..4q+3 of octet oct
    {
        const int q = tid & 15, oct = tid >> 4;
        const float* gp = a.g + (long)n * a.g_ns + p0 + 4 * q;
        const float* sp = a.sca + (long)n * C;
        float4 v[8];
        float sc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = *reinterpret_cast<const float4*>(gp + (long)(8 * oct + i) * HW);
            sc[i] = sp[8 * oct + i];
        }
        stage_octet<SCH, true>(v, sc, sB, NOCT, oct, q);
    }
    // per-row vectors of this lane's 16 channel rows
    float b3v[16], bev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        b3v[r] = a.b3[m0 + row_of(r, kk)];
        bev[r] = a.beta[m0 + row_of(r, kk)];
    }
    __syncthreads();
    NB_STAMP(1);

    // ---- conv3: y = (W3 (g*sca) + b3) * beta + inp
    f32x16 acc[1][2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][tn][r] = 0.f;
    gemm_split<SCH, 1, NG, 2>(acc, a.w3, C / 32, [&](int) { return wave; }, sB, NOCT, lane, rot, [](int) {});
    NB_STAMP(2);

    float yv[2][16];
    float psum[2] = {0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (acc[0][tn][r] + b3v[r]) * bev[r] + xr[tn][r];
            yv[tn][r] = v;
            psum[tn] += v;
        }
    // ---- norm2: mean, then centred second moment (two passes over the register tile)
    float mean[2], rstd[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        psum[tn] += __shfl_xor(psum[tn], 32, 64);
        if (kk == 0) red[wave * NPX + 32 * tn + j] = psum[tn];
    }
    __syncthreads();                                          // (all waves are past their conv3 reads of sB here)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const float* rp = red + 32 * tn + j;
        mean[tn] = wave_partials_sum<NW>(rp) * (1.f / C);
    }
    float pvar[2] = {0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = yv[tn][r] - mean[tn];
            pvar[tn] += d * d;
        }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        pvar[tn] += __shfl_xor(pvar[tn], 32, 64);
        if (kk == 0) red[(NW + wave) * NPX + 32 * tn + j] = pvar[tn];
    }
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const float* rp = red + NW * NPX + 32 * tn + j;
        const float var = wave_partials_sum<NW>(rp) * (1.f / C);
        rstd[tn] = 1.f / sqrtf(var + a.eps);
        if (wave == 0 && kk == 0) {
            a.mu[(long)n * HW + p0 + 32 * tn + j] = mean[tn];
            a.rs[(long)n * HW + p0 + 32 * tn + j] = rstd[tn];
        }
    }
    // yn = (y - mu) * rstd * w + b : split into the LDS operand now; y and yn leave for HBM under conv4's MFMAs
    float ynv[2][16];
    {
        float lw[16], lb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            lw[r] = a.lnw[m0 + row_of(r, kk)];
            lb[r] = a.lnb[m0 + row_of(r, kk)];
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ynv[tn][r] = (yv[tn][r] - mean[tn]) * rstd[tn] * lw[r] + lb[r];
            tile_to_planes<SCH>(ynv[tn], sB, NOCT, 4 * wave, 32 * tn + j, kk);
            if constexpr (EARLY_STORES) {
                float* yp = a.y + (long)n * a.y_ns + p0 + j;
                float* ynp = a.yn + (long)n * a.yn_ns + p0 + j;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    yp[off(r, tn)] = yv[tn][r];
                    ynp[off(r, tn)] = ynv[tn][r];
                }
            }
        }
    }
    __syncthreads();
    NB_STAMP(3);

    // ---- conv4: t4 = W4 yn + b4 ; rows [32w, 32w+32) and their gate partners [C + 32w, C + 32w + 32)
    f32x16 acc4[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc4[tm][tn][r] = 0.f;
    {
        float* yp = a.y + (long)n * a.y_ns + p0 + j;
        float* ynp = a.yn + (long)n * a.yn_ns + p0 + j;
        gemm_split<SCH, 2, NG, 2>(acc4, a.w4, 2 * C / 32, [&](int tm) { return tm * (C / 32) + wave; }, sB, NOCT, lane, rot, [&](int g) {
            // 64 dword stores (y, yn: 2 sub-tiles x 16 rows each) spread evenly over the NG groups
            if constexpr (EARLY_STORES) return;
            constexpr int IPG = 32 / NG;
#pragma unroll
            for (int e = 0; e < IPG; ++e) {
                const int idx = g * IPG + e, tn = idx >> 4, r = idx & 15;
                yp[off(r, tn)] = yv[tn][r];
                ynp[off(r, tn)] = ynv[tn][r];
            }
        });
    }
    NB_STAMP(4);
    float b4v[2][16];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) b4v[tm][r] = a.b4[tm * C + m0 + row_of(r, kk)];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc4[tm][tn][r] += b4v[tm][r];
    __syncthreads();                                          // every wave has finished reading the yn planes
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc4[0][tn][r] * acc4[1][tn][r];      // SimpleGate (:170-175)
        tile_to_planes<SCH>(v, sB, NOCT, 4 * wave, 32 * tn + j, kk);
    }
    float b5v[16], gav[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        b5v[r] = m0 < a.c_out ? a.b5[m0 + row_of(r, kk)] : 0.f;
        gav[r] = m0 < a.c_out ? a.gamma[m0 + row_of(r, kk)] : 0.f;
    }
    __syncthreads();
    NB_STAMP(5);

    // ---- conv5: out = (W5 gate + b5) * gamma + y ; the t4 tile leaves for HBM under its MFMAs.  Only the first c_out rows
    // exist (fusion blocks keep `[:, :chan]`): the waves above them just store their t4 tiles.
    float* tp = a.t4 + (long)n * a.t4_ns + p0 + j;
    if constexpr (EARLY_STORES) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) tp[(long)tm * C * HW + off(r, tn)] = acc4[tm][tn][r];
    }
    if (m0 < a.c_out) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][tn][r] = 0.f;
        gemm_split<SCH, 1, NG, 2>(acc, a.w5, a.c_out / 32, [&](int) { return wave; }, sB, NOCT, lane, rot, [&](int g) {
            if constexpr (EARLY_STORES) return;
            constexpr int IPG = 64 / NG;                                         // 64 stores spread evenly over the NG groups
#pragma unroll
            for (int e = 0; e < IPG; ++e) {
                const int idx = g * IPG + e, tm = idx >> 5, tn = (idx >> 4) & 1, r = idx & 15;
                tp[(long)tm * C * HW + off(r, tn)] = acc4[tm][tn][r];
            }
        });
        NB_STAMP(6);
        float* op = a.out + (long)n * a.out_ns + p0 + j;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) op[off(r, tn)] = (acc[0][tn][r] + b5v[r]) * gav[r] + yv[tn][r];
        NB_STAMP(7);
    } else if constexpr (!EARLY_STORES) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) tp[(long)tm * C * HW + off(r, tn)] = acc4[tm][tn][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// First half of a NAFBlock up to the depthwise conv (:216-219): xn = norm1(inp), t1 = conv1(xn).  Same tile ownership as
// the tail kernel: the workgroup reads its 64 pixels of inp once (accumulator layout), reduces the LayerNorm statistics
// across its waves, writes xn / mu / rstd (the backward pass and conv1's weight gradient keep them) under conv1's MFMAs
// and leaves with t1 -- replacing ln_fwd + conv1x1 (14 + 40 us at the 64x64 level) and one pass over xn.
// ---------------------------------------------------------------------------------------------------------------
struct HeadFwdArgs {
    const float* x; long x_ns;
    const float *lnw, *lnb;
    float eps;
    const uint4* w1;                  // packed hx2 fragments (PACK_FWD): M = 2C, K = C
    const float* b1;
    float *mu, *rs;                   // [N][HW]
    float* xn; long xn_ns;
    float* t1; long t1_ns;
    int HW;
};

template <int C, int SCH>
__global__ __launch_bounds__(2 * C, 2) void naf_head_fwd_kernel(HeadFwdArgs a) {
    constexpr int NOCT = C / 8, NG = C / 16, NW = C / 32, NS = SchT<SCH>::NS;
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    uint4* sB = smem4;
    float* red = reinterpret_cast<float*>(smem4 + NS * NOCT * NPX);   // [2][NW][64 px]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const int n = blockIdx.y;
    const long p0 = (long)blockIdx.x * NPX;
    const long HW = a.HW;
    const int m0 = 32 * wave;
    const int rot = (int)(blockIdx.x * 5);
    auto off = [&](int r, int tn) { return (long)(m0 + row_of(r, kk)) * HW + 32 * tn; };

    float xv[2][16];
    float psum[2] = {0.f, 0.f};
    {
        const float* xp = a.x + (long)n * a.x_ns + p0 + j;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xv[tn][r] = xp[off(r, tn)];
                psum[tn] += xv[tn][r];
            }
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        psum[tn] += __shfl_xor(psum[tn], 32, 64);
        if (kk == 0) red[wave * NPX + 32 * tn + j] = psum[tn];
    }
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) mean[tn] = wave_partials_sum<NW>(red + 32 * tn + j) * (1.f / C);
    float pvar[2] = {0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = xv[tn][r] - mean[tn];
            pvar[tn] += d * d;
        }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        pvar[tn] += __shfl_xor(pvar[tn], 32, 64);
        if (kk == 0) red[(NW + wave) * NPX + 32 * tn + j] = pvar[tn];
    }
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const float var = wave_partials_sum<NW>(red + NW * NPX + 32 * tn + j) * (1.f / C);
        rstd[tn] = 1.f / sqrtf(var + a.eps);
        if (wave == 0 && kk == 0) {
            a.mu[(long)n * HW + p0 + 32 * tn + j] = mean[tn];
            a.rs[(long)n * HW + p0 + 32 * tn + j] = rstd[tn];
        }
    }
    float xnv[2][16];
    {
        float lw[16], lb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            lw[r] = a.lnw[m0 + row_of(r, kk)];
            lb[r] = a.lnb[m0 + row_of(r, kk)];
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xnv[tn][r] = (xv[tn][r] - mean[tn]) * rstd[tn] * lw[r] + lb[r];
            tile_to_planes<SCH>(xnv[tn], sB, NOCT, 4 * wave, 32 * tn + j, kk);
        }
    }
    __syncthreads();

    // ---- conv1: t1 = W1 xn + b1 ; rows [32w, 32w + 32) and [C + 32w, C + 32w + 32); xn leaves for HBM under the MFMAs
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    {
        float* xnp = a.xn + (long)n * a.xn_ns + p0 + j;
        gemm_split<SCH, 2, NG, 2>(acc, a.w1, 2 * C / 32, [&](int tm) { return tm * (C / 32) + wave; }, sB, NOCT, lane, rot, [&](int g) {
            constexpr int IPG = 32 / NG;
#pragma unroll
            for (int e = 0; e < IPG; ++e) {
                const int idx = g * IPG + e, tn = idx >> 4, r = idx & 15;
                xnp[off(r, tn)] = xnv[tn][r];
            }
        });
    }
    float* tp = a.t1 + (long)n * a.t1_ns + p0 + j;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = a.b1[tm * C + m0 + row_of(r, kk)];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) tp[(long)tm * C * HW + off(r, tn)] = acc[tm][tn][r] + bv[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the same chain, data gradients only (the weight gradients stay on tdr_conv_wgrad, which reads the dt4
// this kernel writes):
//     dg2 = W5^T (dout * gamma)                          conv5 data gradient
//     dt4 = [dg2 * t4[C:], dg2 * t4[:C]]                 SimpleGate backward (:170-175)
//     dyn = W4^T dt4                                     conv4 data gradient
//     dy  = LayerNorm2d backward(dyn; y, mu, rstd, w) + dout           (nafnet_arch_utils.py:283-300, + the y + x*gamma skip)
//     per-workgroup partial sums of the LayerNorm parameter gradients (gw = sum dyn * yhat, gb = sum dyn)
// replacing the launches conv1x1(GATEBWD) + conv1x1 + ln_bwd_cached (26 + 30 + 19 us at the 64x64 level).
// ---------------------------------------------------------------------------------------------------------------
struct TailBwdArgs {
    const float* dout; long dout_ns;
    const float* gamma;
    const float* t4; long t4_ns;
    const float* y; long y_ns;
    const float *mu, *rs, *lnw;
    const uint4 *w5t, *w4t;           // packed hx2 fragments, mode DGRAD_S1: M = C, K = C ; M = C, K = 2C
    float* dt4; long dt4_ns;
    const float* res; long res_ns;    // residual-branch gradient added to the LayerNorm data gradient
    float* dy; long dy_ns;
    float* part;                      // [gridDim.y * gridDim.x][2][C] LayerNorm parameter-gradient partials
    int HW;
    int c_out;                        // channels of dout (= rows of conv5 that exist); C or C / 2
    // optional conv3 data-gradient stage (tail only): dgp = s[n] * (W3^T (beta * dy)); the pooled-gradient term of the SCA
    // branch is added by the depthwise backward when it reads dgp (tdr_dwsg_bwd_biased)
    const uint4* w3t; const float* beta; const float* sca; float* dgp; long dgp_ns;
};

__device__ __forceinline__ float half_sum32(float v) {      // sum over the 32 lanes of a wave half (same kk)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// HEAD = true is the first half of the block instead (:216-225 backward): dxn = W1^T dt1 (conv1 data gradient, K = 2C,
// `dout` = dt1 [N, 2C, HW], `w4t` = conv1's DGRAD_S1 fragments) followed by norm1's backward + the y-branch gradient in
// `res`: the same K = 2C GEMM + LayerNorm epilogue without the conv5 / SimpleGate front.
template <int C, bool HEAD, int SCH>
__global__ __launch_bounds__(2 * C, 2) void naf_tail_bwd_kernel(TailBwdArgs a) {
    static_assert(C == 256 || C == 128 || C == 64 || C == 32, "C / 32 waves x 32 channel rows");
    constexpr int NW = C / 32;
    constexpr int NS = SchT<SCH>::NS;
    // three planes: the K = 2C operand (conv4 / conv1 data gradient) goes through LDS one K half (C channels) at a time
    constexpr bool KHALF = SCH == SCH_BX3;
    constexpr int KOCT = KHALF ? C / 8 : 2 * C / 8;           // octets of the largest operand resident at once
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    uint4* sB = smem4;                                        // NS planes x KOCT octets x 64 px x 16 B: 128 KiB (hx2) / 96 KiB (bx3) at C = 256
    float* red = reinterpret_cast<float*>(smem4 + NS * KOCT * NPX);   // [2][8 waves][64 px]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const int n = blockIdx.y;
    const long p0 = (long)blockIdx.x * NPX;
    const long HW = a.HW;
    const int m0 = 32 * wave;
    const int rot = (int)(blockIdx.x * 5);   // (not a function of the image index: batch-permutation equivariance stays bit-exact)
    auto off = [&](int r, int tn) { return (long)(m0 + row_of(r, kk)) * HW + 32 * tn; };
    const float one8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};

    NB_STAMP(0);
    f32x16 acc[1][2];
    float da[HEAD ? 1 : 2][16], db[HEAD ? 1 : 2][16];
    float4 vh[(HEAD && KHALF) ? 8 : 1];                       // HEAD + KHALF: the second K half of dt1, requested before the first GEMM
    if constexpr (HEAD) {
        // ---- stage B = dt1 (K = 2C): thread (oct, q) owns pixels 4q..4q+3 of octets oct and oct + C/8
        const int q = tid & 15, oct0 = tid >> 4;
        const float* gp = a.dout + (long)n * a.dout_ns + p0 + 4 * q;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int oct = oct0 + pass * (C / 8);
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(gp + (long)(8 * oct + i) * HW);
            if constexpr (KHALF) {
                if (pass == 0) stage_octet<SCH, false>(v, one8, sB, KOCT, oct0, q);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) vh[i] = v[i];
                }
            } else {
                stage_octet<SCH, false>(v, one8, sB, KOCT, oct, q);
            }
        }
    } else {
    // ---- gate operands of this wave's rows (t4[c], t4[C + c]) in accumulator layout
    float ta[2][16], tb[2][16];
    {
        const float* tp = a.t4 + (long)n * a.t4_ns + p0 + j;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ta[tn][r] = tp[off(r, tn)];
                tb[tn][r] = tp[(long)C * HW + off(r, tn)];
            }
    }
    // ---- stage B = dout * gamma (K = c_out)
    const int NOCT = a.c_out / 8;
    if ((tid >> 4) < NOCT) {
        const int q = tid & 15, oct = tid >> 4;
        const float* gp = a.dout + (long)n * a.dout_ns + p0 + 4 * q;
        float4 v[8];
        float sc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = *reinterpret_cast<const float4*>(gp + (long)(8 * oct + i) * HW);
            sc[i] = a.gamma[8 * oct + i];
        }
        stage_octet<SCH, true>(v, sc, sB, NOCT, oct, q);
    }
    __syncthreads();
    NB_STAMP(1);

    // ---- conv5 data gradient: dg2 rows [32w, 32w + 32)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][tn][r] = 0.f;
    if (a.c_out == C) gemm_split<SCH, 1, C / 16, 2>(acc, a.w5t, C / 32, [&](int) { return wave; }, sB, C / 8, lane, rot, [](int) {});
    else gemm_split<SCH, 1, C / 32, 2>(acc, a.w5t, C / 32, [&](int) { return wave; }, sB, C / 16, lane, rot, [](int) {});
    NB_STAMP(2);
    __syncthreads();                                          // every wave is done with the dout planes

    // ---- SimpleGate backward; dt4 rows c -> octets [4w, 4w+4), rows C + c -> octets [C/8 + 4w, ...) of the K = 2C operand
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            da[tn][r] = acc[0][tn][r] * tb[tn][r];
            db[tn][r] = acc[0][tn][r] * ta[tn][r];
        }
        tile_to_planes<SCH>(da[tn], sB, KOCT, 4 * wave, 32 * tn + j, kk);
        if constexpr (!KHALF) tile_to_planes<SCH>(db[tn], sB, KOCT, C / 8 + 4 * wave, 32 * tn + j, kk);
        if constexpr (KHALF) {
            // three planes: the dt4 tile leaves for HBM right here instead of riding inside the K = 2C GEMM -- only the second half's
            // 32 values stay in registers across the first half's MFMAs (spill reloads behind queued stores wait for every store)
            float* dp = a.dt4 + (long)n * a.dt4_ns + p0 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dp[off(r, tn)] = da[tn][r];
                dp[(long)C * HW + off(r, tn)] = db[tn][r];
            }
        }
    }
    }   // !HEAD
    // LayerNorm operands of this wave's rows (requested ahead of the GEMM that produces their partner; with three planes and the
    // gate tiles still live that is 48 registers too many -- 406 spilled -- so there they are requested between the two K halves)
    float yh[2][16], lw[16];
    float mean_[2], rstd_[2];
    auto load_ln_operands = [&]() {
        const float* yp = a.y + (long)n * a.y_ns + p0 + j;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            mean_[tn] = a.mu[(long)n * HW + p0 + 32 * tn + j];
            rstd_[tn] = a.rs[(long)n * HW + p0 + 32 * tn + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) yh[tn][r] = yp[off(r, tn)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) lw[r] = a.lnw[m0 + row_of(r, kk)];
    };
    constexpr bool LN_LATE = KHALF && !HEAD;
    if constexpr (!LN_LATE) load_ln_operands();
    __syncthreads();
    NB_STAMP(3);

    // ---- conv4 data gradient: dyn rows [32w, 32w + 32), K = 2C ; the dt4 tile leaves for HBM under its MFMAs
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][tn][r] = 0.f;
    if constexpr (KHALF) {
        // K half 0 (channels [0, C)) sits in LDS; then the planes are overwritten with K half 1 (channels [C, 2C)) and the
        // accumulation continues with the second half of the packed fragments ([group][m-tile][plane][lane]: group C/16 onwards)
        const uint4* w_hi = a.w4t + (long)(C / 16) * (C / 32) * (NS * 64);
        if constexpr (HEAD) {
            gemm_split<SCH, 1, C / 16, 2>(acc, a.w4t, C / 32, [&](int) { return wave; }, sB, KOCT, lane, rot, [](int) {});
            __syncthreads();
            stage_octet<SCH, false>(vh, one8, sB, KOCT, tid >> 4, tid & 15);
            __syncthreads();
            gemm_split<SCH, 1, C / 16, 2>(acc, w_hi, C / 32, [&](int) { return wave; }, sB, KOCT, lane, rot, [](int) {});
        } else {
            // a real two-trip loop (one GEMM body): unrolled, the register allocator kept the LDS / fragment addresses of the first
            // half alive for the second and spilled 350 registers at C = 256
            const uint4* wk = a.w4t;
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                if (half) {
                    __syncthreads();
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) tile_to_planes<SCH>(db[tn], sB, KOCT, 4 * wave, 32 * tn + j, kk);
                    load_ln_operands();
                    __syncthreads();
                }
                gemm_split<SCH, 1, C / 16, 2>(acc, wk, C / 32, [&](int) { return wave; }, sB, KOCT, lane, rot, [](int) {});
                wk = w_hi;
            }
        }
    } else if constexpr (HEAD) {
        gemm_split<SCH, 1, 2 * C / 16, 2>(acc, a.w4t, C / 32, [&](int) { return wave; }, sB, 2 * C / 8, lane, rot, [](int) {});
    } else {
        float* dp = a.dt4 + (long)n * a.dt4_ns + p0 + j;
        gemm_split<SCH, 1, 2 * C / 16, 2>(acc, a.w4t, C / 32, [&](int) { return wave; }, sB, 2 * C / 8, lane, rot, [&](int g) {
            constexpr int IPG = 32 / (2 * C / 16);             // 64 dword stores spread evenly over the groups
#pragma unroll
            for (int e = 0; e < IPG; ++e) {
                const int idx = g * IPG + e, tn = idx >> 4, r = idx & 15;
                dp[off(r, tn)] = da[tn][r];
                dp[(long)C * HW + off(r, tn)] = db[tn][r];
            }
        });
    }
    NB_STAMP(4);
    // ---- LayerNorm backward: g = dyn * w ; dx = (g - yhat * mean_c(g * yhat) - mean_c(g)) * rstd ; + dout
    float gv[2][16];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    float pw[16], pb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { pw[r] = 0.f; pb[r] = 0.f; }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[0][tn][r];
            const float yhat = (yh[tn][r] - mean_[tn]) * rstd_[tn];
            yh[tn][r] = yhat;
            pw[r] += d * yhat;
            pb[r] += d;
            const float g = d * lw[r];
            gv[tn][r] = g;
            s1[tn] += g * yhat;
            s2[tn] += g;
        }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        s1[tn] += __shfl_xor(s1[tn], 32, 64);
        s2[tn] += __shfl_xor(s2[tn], 32, 64);
        if (kk == 0) {
            red[wave * NPX + 32 * tn + j] = s1[tn];
            red[(NW + wave) * NPX + 32 * tn + j] = s2[tn];
        }
    }
    // parameter-gradient partials of this workgroup: sum over its 64 pixels, one value per channel row
    {
        float* pp = a.part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float w_ = half_sum32(pw[r]), b_ = half_sum32(pb[r]);
            if (j == 0) {
                pp[m0 + row_of(r, kk)] = w_;
                pp[C + m0 + row_of(r, kk)] = b_;
            }
        }
    }
    __syncthreads();
    {
        const float* dop = a.res + (long)n * a.res_ns + p0 + j;
        float* dyp = a.dy + (long)n * a.dy_ns + p0 + j;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const float* r1 = red + 32 * tn + j;
            const float* r2 = red + NW * NPX + 32 * tn + j;
            const float m1 = wave_partials_sum<NW>(r1) * (1.f / C);
            const float m2 = wave_partials_sum<NW>(r2) * (1.f / C);
            float res[16];
            const bool has_res = HEAD || m0 < a.c_out;                    // the skip gradient exists for the first c_out channels only
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = has_res ? dop[off(r, tn)] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) gv[tn][r] = (gv[tn][r] - yh[tn][r] * m1 - m2) * rstd_[tn] + res[r];      // = dy
            if (HEAD || !a.w3t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dyp[off(r, tn)] = gv[tn][r];
            }
        }
    }
    if constexpr (!HEAD) {
        if (a.w3t) {
            // ---- conv3 data gradient on the way out (:226-230 backward): u = W3^T (beta * dy), dgp = u * sca[n]; dy itself
            // leaves for HBM under the MFMAs.  (sB: every wave passed the barrier after conv4's GEMM.)
            float be[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) be[r] = a.beta[m0 + row_of(r, kk)];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = gv[tn][r] * be[r];
                tile_to_planes<SCH>(v, sB, C / 8, 4 * wave, 32 * tn + j, kk);
            }
            __syncthreads();
            NB_STAMP(5);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][tn][r] = 0.f;
            float* dyp = a.dy + (long)n * a.dy_ns + p0 + j;
            gemm_split<SCH, 1, C / 16, 2>(acc, a.w3t, C / 32, [&](int) { return wave; }, sB, C / 8, lane, rot, [&](int g) {
                constexpr int IPG = 32 / (C / 16);
#pragma unroll
                for (int e = 0; e < IPG; ++e) {
                    const int idx = g * IPG + e, tn = idx >> 4, r = idx & 15;
                    dyp[off(r, tn)] = gv[tn][r];
                }
            });
            NB_STAMP(6);
            float sc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = a.sca[(long)n * C + m0 + row_of(r, kk)];
            float* gp = a.dgp + (long)n * a.dgp_ns + p0 + j;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) gp[off(r, tn)] = acc[0][tn][r] * sc[r];
            NB_STAMP(7);
        }
    }
}

}  // namespace

#ifdef TDR_NB_PROBE
extern "C" int tdr_nb_probe_read(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(nb_probe_buf), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int tdr_naf_tail_supported(int C, int HW) { return ((C == 256 || C == 128 || C == 64 || C == 32) && HW % 64 == 0) ? 1 : 0; }

#define NAF_DISPATCH_C(C_, KERNEL_EXPR, lds_, a_, d_, stream_)                                                                   \
    do {                                                                                                                         \
        auto kern = KERNEL_EXPR;                                                                                                 \
        static bool attr_set = false;                                                                                            \
        if (!attr_set) {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                     \
        }                                                                                                                        \
        hipLaunchKernelGGL(kern, dim3((d_)->HW / NPX, (d_)->N), dim3(2 * C_), lds_, (hipStream_t)stream_, a_);                    \
    } while (0)

// channel count x split scheme (`bx` in scope: the weight packs are the 3-way bf16 split); MID: template arguments between C and SCH
#define NAF_COMMA ,
#define NAF_DISPATCH_CS(KERN, MID, lds_, a_, d_, stream_)                                                            \
    do {                                                                                                             \
        if (bx) {                                                                                                    \
            if ((d_)->C == 256) NAF_DISPATCH_C(256, (KERN<256 MID, SCH_BX3>), lds_, a_, d_, stream_);                \
            else if ((d_)->C == 128) NAF_DISPATCH_C(128, (KERN<128 MID, SCH_BX3>), lds_, a_, d_, stream_);           \
            else if ((d_)->C == 64) NAF_DISPATCH_C(64, (KERN<64 MID, SCH_BX3>), lds_, a_, d_, stream_);              \
            else NAF_DISPATCH_C(32, (KERN<32 MID, SCH_BX3>), lds_, a_, d_, stream_);                                 \
        } else {                                                                                                     \
            if ((d_)->C == 256) NAF_DISPATCH_C(256, (KERN<256 MID, SCH_HX2>), lds_, a_, d_, stream_);                \
            else if ((d_)->C == 128) NAF_DISPATCH_C(128, (KERN<128 MID, SCH_HX2>), lds_, a_, d_, stream_);           \
            else if ((d_)->C == 64) NAF_DISPATCH_C(64, (KERN<64 MID, SCH_HX2>), lds_, a_, d_, stream_);              \
            else NAF_DISPATCH_C(32, (KERN<32 MID, SCH_HX2>), lds_, a_, d_, stream_);                                 \
        }                                                                                                            \
    } while (0)

extern "C" int tdr_naf_tail_fwd(const TdrNafTailDesc* d, void* stream) {
    TDR_REQUIRE(d && d->g && d->sca && d->x && d->w3 && d->w4 && d->w5 && d->b3 && d->beta && d->lnw && d->lnb && d->b4 && d->b5 &&
                    d->gamma && d->y && d->mu && d->rs && d->yn && d->t4 && d->out,
                "tdr_naf_tail_fwd: null pointer");
    TDR_REQUIRE(tdr_naf_tail_supported(d->C, d->HW), "tdr_naf_tail_fwd: needs C in {32, 64, 128, 256} and HW %% 64 == 0 (got C=%d HW=%d)", d->C, d->HW);
    TDR_REQUIRE(d->w_fmt == 2 || d->w_fmt == 1, "tdr_naf_tail_fwd: weights must be packed with tdr_pack_weights_hx2 / _bx3 (mode FWD)");
    TDR_REQUIRE(d->HW % 4 == 0 && d->g_ns % 4 == 0 && (reinterpret_cast<uintptr_t>(d->g) & 15) == 0, "tdr_naf_tail_fwd: g must be 16-byte aligned");
    TailArgs a;
    a.g = d->g; a.g_ns = d->g_ns; a.sca = d->sca; a.x = d->x; a.x_ns = d->x_ns;
    a.w3 = reinterpret_cast<const uint4*>(d->w3); a.w4 = reinterpret_cast<const uint4*>(d->w4); a.w5 = reinterpret_cast<const uint4*>(d->w5);
    a.b3 = d->b3; a.beta = d->beta; a.lnw = d->lnw; a.lnb = d->lnb; a.b4 = d->b4; a.b5 = d->b5; a.gamma = d->gamma;
    a.eps = d->eps;
    a.y = d->y; a.y_ns = d->y_ns; a.mu = d->mu; a.rs = d->rs; a.yn = d->yn; a.yn_ns = d->yn_ns; a.t4 = d->t4; a.t4_ns = d->t4_ns;
    a.out = d->out; a.out_ns = d->out_ns; a.HW = d->HW;
    a.c_out = d->c_out > 0 ? d->c_out : d->C;
    TDR_REQUIRE(a.c_out == d->C || (a.c_out * 2 == d->C && a.c_out % 32 == 0), "tdr_naf_tail_fwd: c_out must be C or C / 2 (a multiple of 32)");
    const bool bx = d->w_fmt == 1;
    const size_t lds = (size_t)(bx ? 3 : 2) * (d->C / 8) * NPX * 16 + (size_t)2 * (d->C / 32) * NPX * sizeof(float);      // planes + red[2][C / 32 waves][64 px]
    NAF_DISPATCH_CS(naf_tail_fwd_kernel, , lds, a, d, stream);
    TDR_LAUNCH_CHECK("naf_tail_fwd_kernel");
    return TDR_OK;
}

extern "C" int tdr_naf_head_fwd(const TdrNafHeadFwdDesc* d, void* stream) {
    TDR_REQUIRE(d && d->x && d->lnw && d->lnb && d->w1 && d->b1 && d->mu && d->rs && d->xn && d->t1, "tdr_naf_head_fwd: null pointer");
    TDR_REQUIRE(tdr_naf_tail_supported(d->C, d->HW), "tdr_naf_head_fwd: needs C in {32, 64, 128, 256} and HW %% 64 == 0 (got C=%d HW=%d)", d->C, d->HW);
    TDR_REQUIRE(d->w_fmt == 2 || d->w_fmt == 1, "tdr_naf_head_fwd: weights must be packed with tdr_pack_weights_hx2 / _bx3 (mode FWD)");
    HeadFwdArgs a;
    a.x = d->x; a.x_ns = d->x_ns; a.lnw = d->lnw; a.lnb = d->lnb; a.eps = d->eps;
    a.w1 = reinterpret_cast<const uint4*>(d->w1); a.b1 = d->b1;
    a.mu = d->mu; a.rs = d->rs; a.xn = d->xn; a.xn_ns = d->xn_ns; a.t1 = d->t1; a.t1_ns = d->t1_ns; a.HW = d->HW;
    const bool bx = d->w_fmt == 1;
    const size_t lds = (size_t)(bx ? 3 : 2) * (d->C / 8) * NPX * 16 + (size_t)2 * (d->C / 32) * NPX * sizeof(float);      // planes + red[2][C / 32 waves][64 px]
    NAF_DISPATCH_CS(naf_head_fwd_kernel, , lds, a, d, stream);
    TDR_LAUNCH_CHECK("naf_head_fwd_kernel");
    return TDR_OK;
}

extern "C" int64_t tdr_naf_tail_bwd_ws_floats(int N, int C, int HW) {
    const int nparts = N * (HW / NPX);
    return (int64_t)nparts * 2 * C + tdr_pair_sum_mid_floats(nparts, C);
}

extern "C" int tdr_naf_tail_bwd(const TdrNafTailBwdDesc* d, void* stream) {
    TDR_REQUIRE(d && d->dout && d->gamma && d->t4 && d->y && d->mu && d->rs && d->lnw && d->w5t && d->w4t && d->dt4 && d->dy && d->ws,
                "tdr_naf_tail_bwd: null pointer");
    TDR_REQUIRE((d->gw != nullptr) == (d->gb != nullptr), "tdr_naf_tail_bwd: gw and gb are given together or not at all");
    TDR_REQUIRE(tdr_naf_tail_supported(d->C, d->HW), "tdr_naf_tail_bwd: needs C in {32, 64, 128, 256} and HW %% 64 == 0 (got C=%d HW=%d)", d->C, d->HW);
    TDR_REQUIRE(d->w_fmt == 2 || d->w_fmt == 1, "tdr_naf_tail_bwd: weights must be packed with tdr_pack_weights_hx2 / _bx3 (mode DGRAD_S1)");
    TDR_REQUIRE(d->dout_ns % 4 == 0 && (reinterpret_cast<uintptr_t>(d->dout) & 15) == 0, "tdr_naf_tail_bwd: dout must be 16-byte aligned");
    TailBwdArgs a;
    a.dout = d->dout; a.dout_ns = d->dout_ns; a.gamma = d->gamma; a.t4 = d->t4; a.t4_ns = d->t4_ns; a.y = d->y; a.y_ns = d->y_ns;
    a.mu = d->mu; a.rs = d->rs; a.lnw = d->lnw;
    a.w5t = reinterpret_cast<const uint4*>(d->w5t); a.w4t = reinterpret_cast<const uint4*>(d->w4t);
    a.dt4 = d->dt4; a.dt4_ns = d->dt4_ns; a.dy = d->dy; a.dy_ns = d->dy_ns; a.part = d->ws; a.HW = d->HW;
    a.res = d->dout; a.res_ns = d->dout_ns;
    a.c_out = d->c_out > 0 ? d->c_out : d->C;
    TDR_REQUIRE(a.c_out == d->C || (a.c_out * 2 == d->C && a.c_out % 32 == 0), "tdr_naf_tail_bwd: c_out must be C or C / 2 (a multiple of 32)");
    a.w3t = reinterpret_cast<const uint4*>(d->w3t); a.beta = d->beta; a.sca = d->sca; a.dgp = d->dgp; a.dgp_ns = d->dgp_ns;
    TDR_REQUIRE(!d->w3t || (d->beta && d->sca && d->dgp), "tdr_naf_tail_bwd: the conv3 stage needs beta, sca and dgp");
    const bool bx = d->w_fmt == 1;       // three planes: one K half (C / 8 octets) resident at a time
    const size_t lds = (size_t)(bx ? 3 * (d->C / 8) : 2 * (2 * d->C / 8)) * NPX * 16 + (size_t)2 * (d->C / 32) * NPX * sizeof(float);      // planes + red[2][C / 32 waves][64 px]
    NAF_DISPATCH_CS(naf_tail_bwd_kernel, NAF_COMMA false, lds, a, d, stream);
    TDR_LAUNCH_CHECK("naf_tail_bwd_kernel");
    if (!d->gw) return TDR_OK;            // the caller finishes the LayerNorm parameter gradients itself (tdr_pair_sum_partials on ws)
    return tdr_pair_sum_partials(d->ws, d->N * (d->HW / NPX), d->C, d->gw, d->gb, d->ws + (long)d->N * (d->HW / NPX) * 2 * d->C, stream);
}

extern "C" int tdr_naf_head_bwd(const TdrNafHeadBwdDesc* d, void* stream) {
    TDR_REQUIRE(d && d->dt1 && d->x && d->mu && d->rs && d->lnw && d->w1t && d->res && d->dx && d->ws, "tdr_naf_head_bwd: null pointer");
    TDR_REQUIRE((d->gw != nullptr) == (d->gb != nullptr), "tdr_naf_head_bwd: gw and gb are given together or not at all");
    TDR_REQUIRE(tdr_naf_tail_supported(d->C, d->HW), "tdr_naf_head_bwd: needs C in {32, 64, 128, 256} and HW %% 64 == 0 (got C=%d HW=%d)", d->C, d->HW);
    TDR_REQUIRE(d->w_fmt == 2 || d->w_fmt == 1, "tdr_naf_head_bwd: weights must be packed with tdr_pack_weights_hx2 / _bx3 (mode DGRAD_S1)");
    TDR_REQUIRE(d->dt1_ns % 4 == 0 && (reinterpret_cast<uintptr_t>(d->dt1) & 15) == 0, "tdr_naf_head_bwd: dt1 must be 16-byte aligned");
    TailBwdArgs a;
    a.dout = d->dt1; a.dout_ns = d->dt1_ns; a.gamma = nullptr; a.t4 = nullptr; a.t4_ns = 0; a.y = d->x; a.y_ns = d->x_ns;
    a.mu = d->mu; a.rs = d->rs; a.lnw = d->lnw;
    a.w5t = nullptr; a.w4t = reinterpret_cast<const uint4*>(d->w1t);
    a.dt4 = nullptr; a.dt4_ns = 0; a.res = d->res; a.res_ns = d->res_ns; a.dy = d->dx; a.dy_ns = d->dx_ns; a.part = d->ws; a.HW = d->HW;
    a.c_out = d->C;
    a.w3t = nullptr; a.beta = nullptr; a.sca = nullptr; a.dgp = nullptr; a.dgp_ns = 0;
    const bool bx = d->w_fmt == 1;
    const size_t lds = (size_t)(bx ? 3 * (d->C / 8) : 2 * (2 * d->C / 8)) * NPX * 16 + (size_t)2 * (d->C / 32) * NPX * sizeof(float);      // planes + red[2][C / 32 waves][64 px]
    NAF_DISPATCH_CS(naf_tail_bwd_kernel, NAF_COMMA true, lds, a, d, stream);
    TDR_LAUNCH_CHECK("naf_head_bwd_kernel");
    if (!d->gw) return TDR_OK;
    return tdr_pair_sum_partials(d->ws, d->N * (d->HW / NPX), d->C, d->gw, d->gb, d->ws + (long)d->N * (d->HW / NPX) * 2 * d->C, stream);
}
