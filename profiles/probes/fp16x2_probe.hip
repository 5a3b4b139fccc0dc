// Probe for DESIGN.md section 9 item 1: fp32 emulation by a 2-way fp16 split on the gfx950 f16 matrix cores
// (x = h + m, 4 cross products hh, hm, mh, mm, fp32 accumulate) against the 3-way bf16 split (6 products), the
// exact fp32 MFMA chain and an fp64 host reference -- numerics (incl. gradient-sized operands with and without an
// exact power-of-two pre-scale) and MFMA-only rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 fp16x2_probe.hip -o fp16x2_probe && ./fp16x2_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x; const float r = x - (float)h; m = (__bf16)r; l = (__bf16)(r - (float)m);
}
__device__ __forceinline__ void split2h(float x, _Float16& h, _Float16& m) {
    h = (_Float16)x; m = (_Float16)(x - (float)h);
}

// mode 0: fp32 MFMA; 1: bf16x3 (6 products); 2: fp16x2 (4 products); 3: fp16x2 without mm (3 products); 4: plain fp16
// sa / sb: exact power-of-two pre-scales of the operands (result divided by sa*sb)
__global__ void gemm_probe(const float* A, const float* B, float* C, int M, int N, int K, int mode, float sa, float sb) {
    const int lane = threadIdx.x;
    const int tm = blockIdx.y * 32, tn = blockIdx.x * 32;
    const int j = lane & 31, kk = lane >> 5;
    f32x16 acc = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(long)(tm + j) * K + k + kk], B[(long)(k + kk) * N + tn + j], acc, 0, 0, 0);
    } else if (mode == 1) {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __bf16 h, m, l;
                split3(A[(long)(tm + j) * K + k + 8 * kk + i], h, m, l); ah[i] = h; am[i] = m; al[i] = l;
                split3(B[(long)(k + 8 * kk + i) * N + tn + j], h, m, l); bh[i] = h; bm[i] = m; bl[i] = l;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            f16x8 ah, am, bh, bm;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 h, m;
                split2h(A[(long)(tm + j) * K + k + 8 * kk + i] * sa, h, m); ah[i] = h; am[i] = m;
                split2h(B[(long)(k + 8 * kk + i) * N + tn + j] * sb, h, m); bh[i] = h; bm[i] = m;
            }
            if (mode == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bm, acc, 0, 0, 0);
            if (mode != 4) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    const float inv = mode >= 2 ? 1.0f / (sa * sb) : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) C[(long)(tm + (r & 3) + 8 * (r >> 2) + 4 * kk) * N + tn + j] = acc[r] * inv;
}

template <int NPROD, bool F16>
__global__ __launch_bounds__(256) void rate_probe(float* out, int reps) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a[3], b[3]; f16x8 c[2], d[2];
    for (int s = 0; s < 3; ++s) for (int i = 0; i < 8; ++i) { a[s][i] = (__bf16)(float)(threadIdx.x + s + i); b[s][i] = (__bf16)(float)(threadIdx.x * 3 + s - i); }
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 8; ++i) { c[s][i] = (_Float16)(float)(threadIdx.x + s + i); d[s][i] = (_Float16)(float)(threadIdx.x * 3 + s - i); }
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (F16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c[q & 1], d[(q >> 1) & 1], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q % 3], b[q / 3 % 3], acc[t], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double frand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(frand() + 1e-12)) * cos(6.283185307179586 * frand()); }

int main() {
    const int M = 64, N = 64, K = 4608;
    const char* names[] = {"fp32 MFMA 32x32x2 (exact chain)", "bf16x3 (6 products)", "fp16x2 (4 products)", "fp16x2 w/o m*m (3 products)", "plain fp16 (1 product)"};
    struct Case { const char* what; double amag, bmag; float sa, sb; };
    Case cases[] = {{"weights 0.05 x relu activations O(1), no pre-scale", 0.05, 1.0, 1.f, 1.f},
                    {"weights 0.05 x gradients 3e-7 (dpred-sized), no pre-scale", 0.05, 3e-7, 1.f, 1.f},
                    {"weights 0.05 x gradients 3e-7, gradients pre-scaled by 2^20", 0.05, 3e-7, 1.f, 1048576.f},
                    {"activations O(1) x gradients 3e-7 with 1e3 dynamic range inside the tensor, pre-scale 2^20", 1.0, 3e-7, 1.f, 1048576.f}};
    for (int ci = 0; ci < 4; ++ci) {
        const Case& cs = cases[ci];
        std::vector<float> A(M * K), B(K * N);
        srand(4321 + ci);
        for (auto& v : A) v = (float)(nrand() * cs.amag);
        for (auto& v : B) { double x = nrand(); if (ci == 0) x = x > 0 ? x : 0; if (ci == 3) x *= pow(10.0, -3.0 * frand()); v = (float)(x * cs.bmag); }
        std::vector<double> ref(M * N), mag(M * N);
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double s = 0, sa = 0;
                for (int k = 0; k < K; ++k) { double p = (double)A[m * K + k] * (double)B[k * N + n]; s += p; sa += fabs(p); }
                ref[m * N + n] = s; mag[m * N + n] = sa;
            }
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, M * N * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        printf("K=%d  %s\n", K, cs.what);
        for (int mode = 0; mode < 5; ++mode) {
            hipLaunchKernelGGL(gemm_probe, dim3(N / 32, M / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K, mode, cs.sa, cs.sb);
            CK(hipDeviceSynchronize());
            std::vector<float> Cc(M * N);
            CK(hipMemcpy(Cc.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
            double worst = 0, rms = 0;
            for (int i = 0; i < M * N; ++i) { double e = fabs((double)Cc[i] - ref[i]) / mag[i]; worst = fmax(worst, e); rms += e * e; }
            printf("   %-34s max err / sum|ab| = %.3e   rms = %.3e\n", names[mode], worst, sqrt(rms / (M * N)));
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    float* out; CK(hipMalloc(&out, 2048 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 2000, blocks = 2048;
    auto time = [&](auto kern, int nprod, const char* nm) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, reps); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double macs = (double)blocks * 4 * reps * 4 * 32 * 32 * 16;        // fp32-equivalent MACs (one per group of nprod products)
        printf("rate %-28s %8.1f fp32-equivalent TFLOP/s  (%d MFMAs per fp32 product)\n", nm, 2 * macs / (ms * 1e-3) / 1e12, nprod);
    };
    time(rate_probe<6, false>, 6, "bf16x3");
    time(rate_probe<4, true>, 4, "fp16x2");
    time(rate_probe<3, true>, 3, "fp16x2 w/o m*m");
    return 0;
}
