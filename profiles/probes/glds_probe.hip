// Probe: semantics of __builtin_amdgcn_global_load_lds(.., 16, ..) on gfx950: LDS destination = wave-uniform base (M0)
// + lane * 16 for active lanes; inactive lanes write nothing.  Prints mismatches (0 expected).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float* __restrict__ src, float* __restrict__ dst, int limit) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 256) lds[i] = -1.f;
    __syncthreads();
    // permuted source: lane l of wave w reads piece (w*64 + (l ^ 5)); only pieces < limit are loaded
    const int piece = wave * 64 + ((tid & 63) ^ 5);
    const float* g = src + (size_t)piece * 4;
    float* lbase = lds + __builtin_amdgcn_readfirstlane(wave * 64 * 4);
    if (piece < limit)
        __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)g,
                                         (__attribute__((address_space(3))) void*)lbase, 16, 0, 0);
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) dst[i] = lds[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 4096); hipMalloc(&d, 4096);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    const int limit = 200;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, s, d, limit);
    std::vector<float> o(1024);
    hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int piece = (t >> 6) * 64 + ((t & 63) ^ 5);
        for (int e = 0; e < 4; ++e) {
            const float expect = piece < limit ? (float)(piece * 4 + e) : -1.f;   // LDS slot of lane t holds what lane t loaded
            if (o[t * 4 + e] != expect) { if (bad < 8) printf("slot %d elem %d: got %g want %g\n", t, e, o[t * 4 + e], expect); ++bad; }
        }
    }
    printf("glds probe: %d mismatches\n", bad);
    return 0;
}
