// Probe: lane mapping of ds_read_b64_tr_b16 (gfx950).  Every lane reads 8 bytes at LDS address base + lane*8 (linear), the
// LDS holds the b16 value = its own element index; prints, for each result lane and element, which source element arrived.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
    int* d; hipMalloc(&d, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<int> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            const int src = h[l * 4 + e];        // source element index = src_lane*4 + src_elem
            printf("  (L%2d,e%d)", src / 4, src % 4);
            const int g = l / 16, n = l % 16;
            const int want = (g * 16 + 4 * e + n / 4) * 4 + n % 4;     // hypothesis: result[n][e] = src[4e + n/4][n%4] per 16-lane group
            if (src != want) ++bad;
        }
        printf("\n");
    }
    printf("tr16 probe: %d deviations from the hypothesis result[n][e] = src[4e + n/4][n%%4]\n", bad);
    return 0;
}
