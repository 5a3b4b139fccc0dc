// Weight value lookup shared by the packing kernels (fp32 rows, split-bf16 fragments, multi-tensor).
#pragma once
#include "tdr_common.h"

// GEMM-row m, contraction channel c, tap index of the *effective* kernel (see include/tdr.h, tdr_pack_weights):
//   mode 0 FWD, 1 DGRAD_S1 (transposed + flipped), 2 DGRAD_2x2S2 (as 1x1, m = ci*4+a*2+b), 3 DGRAD_3x3S2 (as 2x2)
__device__ __forceinline__ float tdr_pack_value(const float* __restrict__ w, int Cin, int KH, int mode, int m, int c, int tap) {
    const int taps = KH * KH;
    if (mode == 0) return w[((long)m * Cin + c) * taps + tap];
    if (mode == 1) return w[((long)c * Cin + m) * taps + (taps - 1 - tap)];
    if (mode == 2) return w[((long)c * Cin + (m >> 2)) * 4 + (m & 3)];
    const int ci = m >> 2, aa = (m >> 1) & 1, bb = m & 1, u = tap >> 1, vv = tap & 1;
    const int ky = aa == 0 ? (u == 0 ? 1 : -1) : (u == 0 ? 2 : 0);
    const int kx = bb == 0 ? (vv == 0 ? 1 : -1) : (vv == 0 ? 2 : 0);
    return (ky >= 0 && kx >= 0) ? w[((long)c * Cin + ci) * 9 + ky * 3 + kx] : 0.f;
}

__device__ __forceinline__ void tdr_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);
}

typedef __bf16 tdr_bf16x8 __attribute__((ext_vector_type(8)));
union TdrFrag {
    uint4 u;
    unsigned d[4];
    tdr_bf16x8 v;
};

// element i of the fp32 row layout Wp[chunk][tap][ck][Mpad]
__device__ __forceinline__ float tdr_pack_f32_elem(const float* __restrict__ w, int Cin, int KH, int mode, int CK, int M,
                                                   int Kch, int KHe, int Mpad, long i) {
    const int taps_e = KHe * KHe;
    const int m = (int)(i % Mpad);
    long r = i / Mpad;
    const int ck = (int)(r % CK); r /= CK;
    const int tap = (int)(r % taps_e);
    const int c = (int)(r / taps_e) * CK + ck;
    return (m < M && c < Kch) ? tdr_pack_value(w, Cin, KH, mode, m, c, tap) : 0.f;
}

// fragment i (= ((group*taps + tap)*MT + mt)*64 + lane) of the split layout Wp3[group][tap][mt][split][lane][8]
__device__ __forceinline__ void tdr_pack_bx3_frag(const float* __restrict__ w, int Cin, int KH, int mode, int M, int Kch,
                                                  int KHe, int MT, long i, uint4* __restrict__ wp) {
    const int taps_e = KHe * KHe;
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int mt = (int)(r % MT); r /= MT;
    const int tap = (int)(r % taps_e);
    const int grp = (int)(r / taps_e);
    const int m = mt * 32 + (lane & 31);
    TdrFrag h, mm, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = grp * 16 + 8 * (lane >> 5) + e;
        const float v = (m < M && c < Kch) ? tdr_pack_value(w, Cin, KH, mode, m, c, tap) : 0.f;
        __bf16 a0, a1, a2;
        tdr_split3(v, a0, a1, a2);
        h.v[e] = a0; mm.v[e] = a1; l.v[e] = a2;
    }
    uint4* o = wp + ((i >> 6) * 3) * 64 + lane;
    o[0] = h.u; o[64] = mm.u; o[128] = l.u;
}

// 3x3 weights, forward (MODE 0) or stride-1 data-gradient layout (MODE 1), with ONE thread per (group, m-tile, lane) producing the triple-plane
// fragments of ALL nine taps (i = (grp * MT + mt) * 64 + lane; same bits as tdr_pack_bx3_frag).  One thread per (fragment, tap) gathers 8 scalars
// 36 bytes apart (forward) or whole rows apart (data gradient) NINE times over -- once per tap -- through different workgroups: 9 x the sector
// traffic of the 27 M 3x3 weights of the headline network.  Here a lane's 72 forward weights are one contiguous run (16-byte loads), and the nine
// taps of a data-gradient channel are nine consecutive floats read back to back.
template <int MODE>
__device__ __forceinline__ void tdr_pack_bx3_alltaps9(const float* __restrict__ w, int Cin, int M, int Kch, int MT, long i, uint4* __restrict__ wp) {
    constexpr int TAPS = 9;
    const int lane = (int)(i & 63);
    const long r = i >> 6;
    const int mt = (int)(r % MT);
    const int grp = (int)(r / MT);
    const int m = mt * 32 + (lane & 31);
    const int c0 = grp * 16 + 8 * (lane >> 5);
    float v[8 * TAPS];                       // v[e * TAPS + t] = tdr_pack_value(w, Cin, 3, MODE, m, c0 + e, t)
    if (MODE == 0) {
        const long base = ((long)m * Cin + c0) * TAPS;
        if (m < M && c0 + 8 <= Kch && ((reinterpret_cast<uintptr_t>(w + base) & 15) == 0)) {
#pragma unroll
            for (int q = 0; q < 2 * TAPS; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(w + base + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int t = 0; t < TAPS; ++t) v[e * TAPS + t] = (m < M && c0 + e < Kch) ? w[base + e * TAPS + t] : 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long base = ((long)(c0 + e) * Cin + m) * TAPS;          // w[c][m][:]: transposed; the taps are flipped below
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[e * TAPS + t] = (m < M && c0 + e < Kch) ? w[base + (TAPS - 1 - t)] : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        TdrFrag h, mm, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 a0, a1, a2;
            tdr_split3(v[e * TAPS + t], a0, a1, a2);
            h.v[e] = a0; mm.v[e] = a1; l.v[e] = a2;
        }
        uint4* o = wp + ((((long)grp * TAPS + t) * MT + mt) * 3) * 64 + lane;
        o[0] = h.u; o[64] = mm.u; o[128] = l.u;
    }
}

// the same fragment in the 2-way fp16 split layout Wp2[group][tap][mt][split(h, m)][lane][8 x f16]
typedef _Float16 tdr_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void tdr_pack_hx2_frag(const float* __restrict__ w, int Cin, int KH, int mode, int M, int Kch,
                                                  int KHe, int MT, long i, uint4* __restrict__ wp) {
    const int taps_e = KHe * KHe;
    const int lane = (int)(i & 63);
    long r = i >> 6;
    const int mt = (int)(r % MT); r /= MT;
    const int tap = (int)(r % taps_e);
    const int grp = (int)(r / taps_e);
    const int m = mt * 32 + (lane & 31);
    tdr_f16x8 h, mm;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = grp * 16 + 8 * (lane >> 5) + e;
        const float v = (m < M && c < Kch) ? tdr_pack_value(w, Cin, KH, mode, m, c, tap) : 0.f;
        const _Float16 a0 = (_Float16)v;
        h[e] = a0;
        mm[e] = (_Float16)(v - (float)a0);
    }
    uint4* o = wp + ((i >> 6) * 2) * 64 + lane;
    o[0] = __builtin_bit_cast(uint4, h);
    o[64] = __builtin_bit_cast(uint4, mm);
}

// Forward-layout (mode 0) hx2 pack with ONE thread per (group, m-tile, lane) producing the fragments of ALL taps: the 8 channels x taps weights
// a lane needs are contiguous in w[m][c][tap] (8 * taps floats), so they arrive as 16-byte loads of which every byte is used, where one thread
// per (fragment, tap) gathers 8 scalars 4 * taps bytes apart -- 64 different cache lines per load instruction, the texture-address path being
// the bound (pack_multi_kernel: 0.36 ms per step of the headline network at 0.6 TB/s).  Same bits as tdr_pack_hx2_frag.
// i = (grp * MT + mt) * 64 + lane
template <int TAPS>
__device__ __forceinline__ void tdr_pack_hx2_fwd_alltaps(const float* __restrict__ w, int Cin, int M, int Kch, int MT, long i,
                                                         uint4* __restrict__ wp) {
    const int lane = (int)(i & 63);
    const long r = i >> 6;
    const int mt = (int)(r % MT);
    const int grp = (int)(r / MT);
    const int m = mt * 32 + (lane & 31);
    const int c0 = grp * 16 + 8 * (lane >> 5);
    float v[8 * TAPS];
    const long base = ((long)m * Cin + c0) * TAPS;
    if (m < M && c0 + 8 <= Kch && ((reinterpret_cast<uintptr_t>(w + base) & 15) == 0)) {
#pragma unroll
        for (int q = 0; q < 2 * TAPS; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(w + base + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[e * TAPS + t] = (m < M && c0 + e < Kch) ? w[base + e * TAPS + t] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        unsigned h[4], mm[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) tdr_split2_f16<false>(v[(2 * e) * TAPS + t], v[(2 * e + 1) * TAPS + t], h[e], mm[e]);
        uint4* o = wp + ((((long)grp * TAPS + t) * MT + mt) * 2) * 64 + lane;
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[64] = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    }
}
