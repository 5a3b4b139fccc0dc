// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions on PRE-SPLIT operands (P16 tensors, see tdr_conv_p16.hip), gfx950.
//
//   G[co][ci][ky][kx] = sum_{n,y,x} dout[n][co][y][x] * in[n][ci][y+ky-1][x+kx-1]
//
// MFMA view (v_mfma_f32_32x32x16_f16, 2-way fp16 split: products mh, hm, hh, fp32 accumulation): A[i = co][k = pixel],
// B[k = pixel][j = ci], one 32x32 accumulator per tap.  The contraction runs over PIXELS while a P16 slot holds 8 CHANNELS of one
// pixel -- the transposition the fp32 kernel (tdr_wgrad_bx3.hip) did with global loads of 8 consecutive pixels per lane, an
// operand split per element and v_alignbit assembly of the shifted fragments (VALU issue time ~ MFMA time) is done here by the
// LDS itself: ds_read_b64_tr_b16 hands lane n of a 16-lane group element n%4 of the 8 bytes lane 4e + n/4 pointed at
// (profiles/probes/tr16_probe.hip), so with lane s of a group pointing at pixel s/4, channels 4*(s%4).. of the tile, lane n
// receives channel n of four consecutive pixels: an MFMA operand fragment is two such reads, any tap shift is an address
// offset (slots are per pixel), and no VALU instruction touches an operand.
//
// A workgroup owns a (co-tile, ci-tile) pair and walks down a 32-column strip of one image: per step of RS rows it LDS-DMAs
// RS new input rows (+2 halo rows at the top of its chunk) and RS gradient rows into row rings ([octet][plane][column]
// slots, lane-linear), runs 2*RS k-steps of 27 MFMAs per wave (+2 against a fragment of ones for the bias gradient), one
// barrier per step.  Split-K partials [split][co][ci][9] are reduced in fixed order by wgrad_p16_reduce_kernel: deterministic.
//
// Replaces (reference): autograd's weight / bias gradient of the ResidualBlock convolutions of the MASA encoder
// (models/archs/network_nafnet_guided_arch.py:44-59,110-143).
#include <stdlib.h>
#include <stdint.h>
#include "tdr_common.h"
#include "../../include/tdr.h"

typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));

namespace {

struct WgP16Args {
    const uint4* in; const uint4* dout;       // P16 tensors
    int Cin, Cout, H, W, Hp, Wp;
    int strips, chunks, rc;                    // 32-column strips per image, row chunks per strip, rows per chunk
    float* part; float* dbpart;
};

#define WGP_GLDS(gptr, lptr)                                                                          \
    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(gptr),          \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ ws16x8 tr_frag(const char* p, int off0, int off1) {
    const ws16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)(p + off0));
    const ws16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)(p + off1));
    const ws16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return v;
}
// one matrix product of two 16-bit fragments: fp16 (pair planes) or bf16 (triple planes)
template <int NS>
__device__ __forceinline__ f32x16 wg_mfma(ws16x8 x, ws16x8 y, f32x16 c) {
    if constexpr (NS == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, x), __builtin_bit_cast(wbf16x8, y), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8, x), __builtin_bit_cast(wf16x8, y), c, 0, 0, 0);
}

// WM x WN x WK = 4 waves; wave (wm, wn) owns the 32 co x 32 ci tile pair, WK waves share it and split the k-steps of a step.
// NS: planes of the operand tensors (2: fp16 pair, products mh hm hh; 3: bf16 triple, products lh hl mm mh hm hh -- the plane
// formats of tdr_conv_p16.hip).
template <int WM, int WN, int WK, int NS = 2>
__global__ __launch_bounds__(256, 2) void wgrad3x3_p16_kernel(WgP16Args a) {
    static_assert(WM * WN * WK == 4, "4 waves");
    constexpr int RS = WK == 4 ? 2 : 1;                  // rows per step (2 k-steps of 16 pixels per row)
    static_assert(2 * RS % WK == 0, "k-steps of a step split evenly over the K waves");
    constexpr int BMo = 32 * WM, BNi = 32 * WN, NOo = BMo / 8, NOi = BNi / 8;
    constexpr int XS = 2 * RS + 2, DS = 2 * RS;          // ring slots (rows)
    constexpr int XR = ((NOi * NS * 34 + 63) / 64) * 64; // slots per input row (padded to whole pieces)
    constexpr int DR = NOo * NS * 32;                    // slots per gradient row
    constexpr int XP = XR / 64, DP = DR / 64;            // pieces per row

    extern __shared__ __attribute__((aligned(1024))) uint4 smem4[];
    uint4* sX = smem4;                  // [XS][XR]
    uint4* sD = smem4 + XS * XR;        // [DS][DR]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    int split = blockIdx.x;
    const int chunk = split % a.chunks; int t = split / a.chunks;
    const int strip = t % a.strips;
    const int n = t / a.strips;
    const int co0 = blockIdx.y * BMo, ci0 = blockIdx.z * BNi;
    const int y0 = chunk * a.rc, y1 = min(y0 + a.rc, a.H);
    const int x0 = strip * 32;
    const long PS = (long)a.Hp * a.Wp;
    const int Gi = a.Cin >> 3, Go = a.Cout >> 3;

    // ---- LDS-DMA sources: piece p of a row covers flat slots p*64 + lane of [octet][plane][column]; this wave issues pieces
    // wave, wave + 4, ..  The per-lane byte offsets are row-invariant; the row term is added per issue.
    constexpr int XPW = (XP + 3) / 4, DPW = (DP + 3) / 4;
    long xoff[XPW], doff[DPW];
    bool dcol[DPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        // input row piece: columns beyond the padded row are clamped to the right border (zero)
        const int f = min((wave + 4 * i) * 64 + lane, NOi * NS * 34 - 1);
        const int op = f / 34, c = f - op * 34;
        const int oc = min((ci0 >> 3) + (op / NS), Gi - 1);
        xoff[i] = ((((long)n * Gi + oc) * NS + (op % NS)) * PS + min(x0 + c, a.Wp - 1)) * 16;
    }
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        // gradient row piece: pixels outside the image must contribute ZERO: they are read from the (0, 0) border slot
        const int f = (wave + 4 * i) * 64 + lane;
        const int op = f >> 5, c = f & 31;
        const int oc = min((co0 >> 3) + (op / NS), Go - 1);
        dcol[i] = x0 + c < a.W;
        doff[i] = ((((long)n * Go + oc) * NS + (op % NS)) * PS) * 16;
    }
    const char* xbase = reinterpret_cast<const char*>(a.in);
    const char* dbase = reinterpret_cast<const char*>(a.dout);
    // LDS-DMA by inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read that follows a __builtin LDS-DMA it
    // cannot prove disjoint (here: all of them), which would serialise the row loads with the MFMAs of the step.  Pieces
    // issued this way are invisible to its bookkeeping; they are waited for by the vmcnt(0) + barrier that ends every step.
    auto glds = [&](const char* src, uint4* dst) {
        const unsigned l = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(l) : "memory");
    };
    auto issue_x_row = [&](int prow) {      // padded input row prow -> ring slot prow % XS
        uint4* dst = sX + (prow % XS) * XR;
        const long rowb = (long)min(prow, a.Hp - 1) * a.Wp * 16;
#pragma unroll
        for (int i = 0; i < XPW; ++i)
            if (wave + 4 * i < XP) glds(xbase + xoff[i] + rowb, dst + (wave + 4 * i) * 64);
    };
    auto issue_d_row = [&](int y) {
        uint4* dst = sD + (y % DS) * DR;
        const bool yok = y < a.H;
        const long rowb = ((long)(y + 1) * a.Wp + x0 + 1) * 16;
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            if (wave + 4 * i < DP) {
                const int c = ((wave + 4 * i) * 64 + lane) & 31;
                glds(dbase + doff[i] + ((yok && dcol[i]) ? rowb + c * 16 : 0), dst + (wave + 4 * i) * 64);
            }
    };

    // ---- fragment addressing (see the header): lane = 16*g16 + nn; as a loader it points at pixel nn/4, channels 4*(nn%4)..
    const int nn = lane & 15, g16 = lane >> 4;
    const int jpx = nn >> 2, q = nn & 3, chalf = g16 & 1, kk = g16 >> 1;
    const int oct = chalf * 2 + (q >> 1);
    const int a_lane = ((wm * 4 + oct) * NS * 32 + 8 * kk + jpx) * 16 + (q & 1) * 8;     // + plane*32*16 + (16*ks + 4*r)*16
    const int b_lane = ((wn * 4 + oct) * NS * 34 + 8 * kk + jpx) * 16 + (q & 1) * 8;     // + plane*34*16 + (16*ks + 4*r + kx)*16

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bool want_db = a.dbpart != nullptr && blockIdx.z == 0 && wn == 0;      // wave-uniform
    ws16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = NS == 3 ? (short)0x3f80 : (short)0x3c00;      // 1.0 as bf16 / fp16

    // ---- prologue: halo rows y0, y0+1 (+ the first step's new rows) and the first step's gradient rows
    issue_x_row(y0); issue_x_row(y0 + 1);
#pragma unroll
    for (int r = 0; r < RS; ++r) { issue_x_row(y0 + 2 + r); issue_d_row(y0 + r); }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    for (int y = y0; y < y1; y += RS) {
        // rows of the next step (clamped loads past the chunk are harmless: their ring slots are not read again)
#pragma unroll
        for (int r = 0; r < RS; ++r) { issue_x_row(y + RS + 2 + r); issue_d_row(y + RS + r); }
#pragma unroll
        for (int u = 0; u < 2 * RS / WK; ++u) {
            const int ku = wk + u * WK;                 // k-step of the step: row ku / 2, pixels 16 * (ku % 2) ..
            const int yr = y + (ku >> 1), ks = ku & 1;
            if (yr < y1) {
                const char* pd = reinterpret_cast<const char*>(sD + (yr % DS) * DR) + a_lane + ks * 256;
                ws16x8 ap[NS];                       // gradient planes h, m (, l)
#pragma unroll
                for (int s = 0; s < NS; ++s) ap[s] = tr_frag(pd, s * 32 * 16, s * 32 * 16 + 64);
                if (want_db) {
#pragma unroll
                    for (int s = NS - 1; s >= 0; --s) accb = wg_mfma<NS>(ap[s], ones, accb);
                }
                // small cross terms first, the dominant h*h last
                constexpr int NPR = NS == 3 ? 6 : 3;
                constexpr int PA[6] = {NS == 3 ? 2 : 1, 0, NS == 3 ? 1 : 0, 1, 0, 0};      // pair: mh hm hh ; triple: lh hl mm mh hm hh
                constexpr int PB[6] = {0, NS == 3 ? 2 : 1, NS == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const char* px = reinterpret_cast<const char*>(sX + ((yr + ky) % XS) * XR) + b_lane + ks * 256;
                    ws16x8 bp[NS][3];
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) bp[s][kx] = tr_frag(px, s * 34 * 16 + kx * 16, s * 34 * 16 + kx * 16 + 64);
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = wg_mfma<NS>(ap[PA[pr]], bp[PB[pr]][kx], acc[ky * 3 + kx]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- waves that split K inside the block are summed through LDS in a fixed order (wk = 1, 2, ..)
    const int j = lane & 31, kg = lane >> 5;
    if constexpr (WK > 1) {
        float* red = reinterpret_cast<float*>(smem4);           // 10 x 16 x 64 floats = 40 KiB (the rings are free now)
#pragma unroll
        for (int w = 1; w < WK; ++w) {
            __syncthreads();
            if (wk == w) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(tp * 16 + r) * 64 + lane] = acc[tp][r];
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(9 * 16 + r) * 64 + lane] = accb[r];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tp][r] += red[(tp * 16 + r) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 16; ++r) accb[r] += red[(9 * 16 + r) * 64 + lane];
            }
        }
        if (wk != 0) return;
    }
    // partial[split][co][ci][tap]
    float* part = a.part + (long)split * a.Cout * a.Cin * 9;
    const int ci = ci0 + wn * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (co < a.Cout && ci < a.Cin) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) part[((long)co * a.Cin + ci) * 9 + tp] = acc[tp][r];
        }
        if (want_db && j == 0 && co < a.Cout) a.dbpart[(long)split * a.Cout + co] = accb[r];
    }
}

// out[e] = sum_s part[s][e] in fixed order; blocks >= nb_main reduce the bias-gradient partials.  Block = 64 elements x KL
// partial-lanes, eight partials in flight per thread (as wgrad_reduce_kernel of tdr_wgrad_mfma.hip).
template <int KL>
__global__ __launch_bounds__(64 * KL) void wgrad_p16_reduce_kernel(const float* __restrict__ part, long elems, int nsplit, float* __restrict__ out,
                                                                   int nb_main, const float* __restrict__ part2, long elems2, float* __restrict__ out2) {
    __shared__ float red[KL][64];
    const int lane = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const bool second = (int)blockIdx.x >= nb_main;
    const float* pbase = second ? part2 : part;
    const long ne = second ? elems2 : elems;
    float* o = second ? out2 : out;
    const long e = ((int)blockIdx.x - (second ? nb_main : 0)) * 64L + lane;
    const float* p = pbase + (e < ne ? e : ne - 1);
    float s0 = 0.f;
    for (int k = kl; k < nsplit; k += 8 * KL) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k + u * KL < nsplit ? p[(long)(k + u * KL) * ne] : 0.f;
        s0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    red[kl][lane] = s0;
    __syncthreads();
    if (kl == 0 && e < ne) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < KL; ++q) t += red[q][lane];
        o[e] = t;
    }
}

struct WgP16Plan { int cfg, bm, bn, strips, chunks, rc, nsplit; };

WgP16Plan wgp_plan(const TdrWgradP16Desc* d) {
    WgP16Plan p;
    p.cfg = (d->Cin <= 32 && d->Cout <= 32) ? 1 : 0;
    p.bm = p.cfg == 1 ? 32 : 64; p.bn = p.bm;
    const int rs = p.cfg == 1 ? 2 : 1;
    p.strips = tdr_cdiv(d->W, 32);
    const long out_tiles = (long)tdr_cdiv(d->Cout, p.bm) * tdr_cdiv(d->Cin, p.bn);
    static const long want_total = tdr_tune_env("TDR_WGP_WANT") ? atol(tdr_tune_env("TDR_WGP_WANT")) : 512;
    long want = want_total / out_tiles;
    if (want < 1) want = 1;
    const long rows = (long)d->N * p.strips * d->H;
    long rc = (rows + want - 1) / want;
    if (rc < 8) rc = 8;                                   // amortise the two halo rows of a chunk
    rc = (rc + rs - 1) / rs * rs;
    if (rc > d->H) rc = (d->H + rs - 1) / rs * rs;
    p.rc = (int)rc;
    p.chunks = tdr_cdiv(d->H, p.rc);
    p.nsplit = d->N * p.strips * p.chunks;
    return p;
}

template <int WM, int WN, int WK, int NS>
int launch_wgp(const WgP16Args& a, const WgP16Plan& p, const TdrWgradP16Desc* d, hipStream_t st) {
    constexpr int RS = WK == 4 ? 2 : 1, NOo = 4 * WM, NOi = 4 * WN;
    constexpr int XR = ((NOi * NS * 34 + 63) / 64) * 64, DR = NOo * NS * 32;
    size_t lds = (size_t)((2 * RS + 2) * XR + 2 * RS * DR) * 16;
    if (WK > 1 && lds < 10 * 16 * 64 * 4) lds = 10 * 16 * 64 * 4;
    dim3 grid(p.nsplit, tdr_cdiv(d->Cout, 32 * WM), tdr_cdiv(d->Cin, 32 * WN));
    auto kern = wgrad3x3_p16_kernel<WM, WN, WK, NS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    TDR_LAUNCH_CHECK("wgrad3x3_p16_kernel");
    return TDR_OK;
}

}  // namespace

extern "C" int64_t tdr_wgrad3x3_p16_ws_floats(const TdrWgradP16Desc* d) {
    const WgP16Plan p = wgp_plan(d);
    return (int64_t)p.nsplit * d->Cout * d->Cin * 9 + (int64_t)p.nsplit * d->Cout;
}

extern "C" int tdr_wgrad3x3_p16(const TdrWgradP16Desc* d, void* stream) {
    TDR_REQUIRE(d && d->in16 && d->dout16 && d->g && d->ws, "tdr_wgrad3x3_p16: null pointer");
    TDR_REQUIRE(d->Cin % 16 == 0 && d->Cout % 16 == 0, "tdr_wgrad3x3_p16: channel counts must be multiples of 16 (%d, %d)", d->Cin, d->Cout);
    const WgP16Plan p = wgp_plan(d);
    const int64_t need = tdr_wgrad3x3_p16_ws_floats(d);
    TDR_REQUIRE(d->ws_floats >= need, "tdr_wgrad3x3_p16: workspace too small (%lld < %lld)", (long long)d->ws_floats, (long long)need);
    WgP16Args a;
    a.in = (const uint4*)d->in16; a.dout = (const uint4*)d->dout16;
    a.Cin = d->Cin; a.Cout = d->Cout; a.H = d->H; a.W = d->W; a.Hp = d->H + 2; a.Wp = d->W + 2;
    a.strips = p.strips; a.chunks = p.chunks; a.rc = p.rc;
    a.part = d->ws;
    a.dbpart = d->db ? d->ws + (int64_t)p.nsplit * d->Cout * d->Cin * 9 : nullptr;
    hipStream_t st = (hipStream_t)stream;
    TDR_REQUIRE(d->fmt == 0 || d->fmt == 1 || d->fmt == 2, "tdr_wgrad3x3_p16: plane format %d (1: bf16 triple, 2: fp16 pair)", d->fmt);
    int rc;
    if (d->fmt == 1) rc = p.cfg == 1 ? launch_wgp<1, 1, 4, 3>(a, p, d, st) : launch_wgp<2, 2, 1, 3>(a, p, d, st);
    else rc = p.cfg == 1 ? launch_wgp<1, 1, 4, 2>(a, p, d, st) : launch_wgp<2, 2, 1, 2>(a, p, d, st);
    if (rc != TDR_OK) return rc;
    const long elems = (long)d->Cout * d->Cin * 9;
    const int nb_main = tdr_cdiv(elems, 64), nb2 = d->db ? tdr_cdiv(d->Cout, 64) : 0;
    if (p.nsplit <= 8)
        hipLaunchKernelGGL(wgrad_p16_reduce_kernel<1>, dim3(nb_main + nb2), dim3(64), 0, st, d->ws, elems, p.nsplit, d->g, nb_main, a.dbpart,
                           (long)d->Cout, d->db);
    else if (p.nsplit <= 64)
        hipLaunchKernelGGL(wgrad_p16_reduce_kernel<4>, dim3(nb_main + nb2), dim3(256), 0, st, d->ws, elems, p.nsplit, d->g, nb_main, a.dbpart,
                           (long)d->Cout, d->db);
    else
        hipLaunchKernelGGL(wgrad_p16_reduce_kernel<16>, dim3(nb_main + nb2), dim3(1024), 0, st, d->ws, elems, p.nsplit, d->g, nb_main, a.dbpart,
                           (long)d->Cout, d->db);
    TDR_LAUNCH_CHECK("wgrad_p16_reduce_kernel");
    return TDR_OK;
}
