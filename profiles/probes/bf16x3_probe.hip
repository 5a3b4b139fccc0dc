// Probe: numerics and rate of fp32 emulation by a 3-way bf16 split on the gfx950 bf16 matrix cores
// (x = h + m + l, 6 cross products hh, hm, mh, hl, lh, mm, fp32 accumulate) against the exact
// fp32 MFMA chain (v_mfma_f32_32x32x2_f32) and an fp64 host reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 bf16x3_probe.hip -o bf16x3_probe && ./bf16x3_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;
    m = (__bf16)r;
    const float r2 = r - (float)m;
    l = (__bf16)r2;
}

// C[32 x 32] tiles; A is [M][K] row-major, B is [K][N] row-major.  One wave per 32x32 tile.
// mode 0: fp32 MFMA; 1: bf16x3 single accumulator; 2: bf16x3, hh in one accumulator and the 5 small
// products in another; 3: plain bf16 (1 product); 4: bf16x2 (3 products)
__global__ void gemm_probe(const float* A, const float* B, float* C, int M, int N, int K, int mode) {
    const int lane = threadIdx.x;
    const int tm = blockIdx.y * 32, tn = blockIdx.x * 32;
    const int j = lane & 31, kk = lane >> 5;
    f32x16 acc = {0}, acc2 = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) {
            const float a = A[(long)(tm + j) * K + k + kk];
            const float b = B[(long)(k + kk) * N + tn + j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __bf16 h, m, l;
                split3(A[(long)(tm + j) * K + k + 8 * kk + i], h, m, l);
                ah[i] = h; am[i] = m; al[i] = l;
                split3(B[(long)(k + 8 * kk + i) * N + tn + j], h, m, l);
                bh[i] = h; bm[i] = m; bl[i] = l;
            }
            if (mode == 1) {
                // small terms first, so they are summed before meeting the large one
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            } else if (mode == 2) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            } else if (mode == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        C[(long)(tm + row) * N + tn + j] = acc[r] + acc2[r];
    }
}

// rate: REP back-to-back MFMA groups on 4 independent accumulators, one wave per SIMD x blocks
template <int MODE>
__global__ __launch_bounds__(256) void rate_probe(float* out, int reps) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a[3], b[3];
    for (int s = 0; s < 3; ++s) for (int i = 0; i < 8; ++i) { a[s][i] = (__bf16)(float)(threadIdx.x + s + i); b[s][i] = (__bf16)(float)(threadIdx.x * 3 + s - i); }
    float fa = threadIdx.x, fb = threadIdx.x * 0.5f;
    for (int it = 0; it < reps; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[t], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double frand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(frand() + 1e-12)) * cos(6.283185307179586 * frand()); }

int main() {
    const int M = 64, N = 64;
    const char* names[] = {"fp32 MFMA 32x32x2 (exact chain)", "bf16x3, 6 products, one accumulator", "bf16x3, hh / small split accumulators",
                           "plain bf16 (1 product)", "bf16x2 (3 products)"};
    for (int K : {288, 1024, 4608}) {
        for (int dist = 0; dist < 2; ++dist) {
            std::vector<float> A(M * K), B(K * N);
            srand(1234 + K + dist);
            for (auto& v : A) v = (float)(nrand() * 0.05);
            for (auto& v : B) { double x = nrand(); v = (float)(dist == 0 ? x : (x > 0 ? x : 0)); }   // dist 1: post-ReLU activations
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
            printf("K=%d  %s\n", K, dist == 0 ? "gaussian x gaussian" : "gaussian weights x relu activations");
            for (int mode = 0; mode < 5; ++mode) {
                hipLaunchKernelGGL(gemm_probe, dim3(N / 32, M / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K, mode);
                CK(hipDeviceSynchronize());
                std::vector<float> Cc(M * N);
                CK(hipMemcpy(Cc.data(), dC, M * N * 4, hipMemcpyDeviceToHost));
                double maxabs = 0, maxrel = 0, rms = 0;
                for (int i = 0; i < M * N; ++i) {
                    const double e = fabs((double)Cc[i] - ref[i]);
                    maxabs = fmax(maxabs, e); maxrel = fmax(maxrel, e / mag[i]); rms += (e / mag[i]) * (e / mag[i]);
                }
                printf("  %-42s max|err| %.3e   max err/sum|ab| %.3e   rms err/sum|ab| %.3e\n", names[mode], maxabs, maxrel, sqrt(rms / (M * N)));
            }
            CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
        }
    }
    // ---- rate
    float* dout;
    const int blocks = 256 * 2;
    CK(hipMalloc(&dout, blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        const int reps = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(rate_probe<0>, dim3(blocks), dim3(256), 0, 0, dout, reps);
            else hipLaunchKernelGGL(rate_probe<1>, dim3(blocks), dim3(256), 0, 0, dout, reps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double mfmas = (double)blocks * 4 * reps * (mode == 0 ? 32 : 24);
        const double hw_flops = mfmas * (mode == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16);
        const double eq_flops = mode == 0 ? hw_flops : hw_flops / 6.0;
        printf("rate %-28s %.3f ms   hardware %.1f TFLOP/s   fp32-equivalent %.1f TFLOP/s\n", mode == 0 ? "fp32 MFMA 32x32x2" : "bf16x3 (6 x 32x32x16 bf16)", ms,
               hw_flops / ms / 1e9, eq_flops / ms / 1e9);
    }
    return 0;
}
