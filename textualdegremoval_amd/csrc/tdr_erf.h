// fp32 erf shared by the GDFN stencils (tdr_dwsg.hip) and the token-major ViT GEMM epilogue (tdr_tok16.hip)
#pragma once
#include <hip/hip_runtime.h>

// erf in fp32 with < 1 ulp error (two minimax polynomials, one exp: N. Juffa's erff; checked against scipy over [-6, 6] at
// 3 * 10^-6 spacing: 0.974 ulp).  The library erff costs ~100 FMAs and several exps per value, which made the GDFN kernels
// VALU-bound (profiles/README.md, round 2); both branches are evaluated and selected, no divergence.
__device__ __forceinline__ float erf_1ulp(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __expf(r), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    const float small = fmaf(q, a, a);
    return t > 0.927734375f ? big : small;
}
