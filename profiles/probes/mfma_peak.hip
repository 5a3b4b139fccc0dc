// Probe: sustained v_mfma_f32_32x32x2_f32 rate on this chip (clock/power ceiling for the fp32 conv kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, const char* tag) {
    float* out; hipMalloc(&out, 256 * 256 * 8 * 4 * sizeof(float));
    const int iters = 20000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<grid, 256>>>(out, 100, 1.f, 0.5f);
    hipEventRecord(e0);
    mfma_loop<NACC><<<grid, 256>>>(out, iters, 1.f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 32 * 32 * 2 * (double)NACC * iters * 4.0 * grid;
    printf("%s NACC=%d blocks/CU=%d: %.2f ms  %.1f TFLOP/s\n", tag, NACC, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<4>(1, "pure mfma f32 32x32x2");
    run<4>(2, "pure mfma f32 32x32x2");
    run<1>(1, "pure mfma f32 32x32x2");
    run<2>(2, "pure mfma f32 32x32x2");
    return 0;
}
