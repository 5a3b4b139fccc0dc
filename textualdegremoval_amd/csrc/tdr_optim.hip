// Global-norm gradient clip + AdamW as two multi-tensor kernels
// (models/image_restoration_ref_model.py:172-178, 276-279: AdamW with the
// "masa" LR split, clip_grad_norm_(params, 0.01)).
// Tensors are addressed through device pointer tables; work is cut into
// CHUNK-element pieces by a host-built (tensor, chunk) table, so one launch
// covers all ~900 parameter tensors.  The norm is accumulated in double with a
// fixed two-stage order (deterministic).
#include "tdr_common.h"
#include "../../include/tdr.h"

namespace {
constexpr int CHUNK = 4096;   // elements per chunk (must match tdr_optim_chunk())
static_assert(CHUNK % 1024 == 0, "float4 path: 256 threads x 4 elements per pass");
__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ __launch_bounds__(256) void sumsq_kernel(const float* const* __restrict__ grads, const int64_t* __restrict__ sizes,
                                                   const int* __restrict__ chunk_tensor, const int* __restrict__ chunk_index,
                                                   double* __restrict__ partial) {
    __shared__ double red[4];
    const int t = chunk_tensor[blockIdx.x];
    const long base = (long)chunk_index[blockIdx.x] * CHUNK;
    const long n = sizes[t];
    const float* g = grads[t];
    double s = 0.0;
    const long end = min(base + CHUNK, n);
    if (al16(g) && end - base == CHUNK) {              // full chunk of a 16-byte aligned tensor: float4 loads
#pragma unroll
        for (int k = 0; k < CHUNK / 1024; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(g + base + 4 * (threadIdx.x + 256 * k));
            s += ((double)v.x * (double)v.x + (double)v.y * (double)v.y) + ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) {
            const float v = g[i];
            s += (double)v * (double)v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_finish_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// dst[t][i] = src[t][i] for every tensor of a (tensor, chunk) table: gathers ~900 freshly produced gradient
// tensors into the flat all-reduce / optimiser arena with ONE launch instead of one copy kernel per tensor.
__global__ __launch_bounds__(256) void multi_copy_kernel(const float* const* __restrict__ src, float* const* __restrict__ dst,
                                                        const int64_t* __restrict__ sizes, const int* __restrict__ chunk_tensor,
                                                        const int* __restrict__ chunk_index, float scale) {
    const int t = chunk_tensor[blockIdx.x];
    const long base = (long)chunk_index[blockIdx.x] * CHUNK;
    const long n = sizes[t];
    const float* s = src[t];
    float* d = dst[t];
    const long end = min(base + CHUNK, n);
    if (al16(s) && al16(d) && end - base == CHUNK) {
#pragma unroll
        for (int k = 0; k < CHUNK / 1024; ++k) {
            const long i = base + 4 * (threadIdx.x + 256 * k);
            float4 v = *reinterpret_cast<const float4*>(s + i);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            *reinterpret_cast<float4*>(d + i) = v;
        }
        return;
    }
    for (long i = base + threadIdx.x; i < end; i += 256) d[i] = s[i] * scale;   // scale: 1 / (power-of-two loss scale)
}

struct AdamArgs {
    float lr[4];
    float max_norm, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;
    int use_clip;
    const float* hp;    // optional device-resident {lr[4], bc1, bc2_sqrt} (graph replay: values change, launch does not)
};

__global__ __launch_bounds__(256) void adamw_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                   float* const* __restrict__ exp_avg, float* const* __restrict__ exp_avg_sq,
                                                   const int64_t* __restrict__ sizes, const int* __restrict__ group,
                                                   const int* __restrict__ chunk_tensor, const int* __restrict__ chunk_index,
                                                   const double* __restrict__ sumsq, AdamArgs a) {
    const int t = chunk_tensor[blockIdx.x];
    const long base = (long)chunk_index[blockIdx.x] * CHUNK;
    const long n = sizes[t];
    float* p = params[t];
    const float* g = grads[t];
    float* m = exp_avg[t];
    float* v = exp_avg_sq[t];
    const float lr = a.hp ? a.hp[group[t]] : a.lr[group[t]];
    const float bc1 = a.hp ? a.hp[4] : a.bc1;
    const float bc2_sqrt = a.hp ? a.hp[5] : a.bc2_sqrt;
    float coef = 1.f;
    if (a.use_clip) {
        const float total = (float)sqrt(sumsq[0]);
        coef = fminf(a.max_norm / (total + 1e-6f), 1.f);
    }
    const float step = lr / bc1;
    auto upd = [&](float g0, float& pv, float& mv, float& vv) {
        const float gv = g0 * coef;
        pv *= (1.f - lr * a.weight_decay);
        mv = mv + (1.f - a.beta1) * (gv - mv);                            // lerp, as torch.optim
        vv = a.beta2 * vv + (1.f - a.beta2) * gv * gv;
        const float denom = sqrtf(vv) / bc2_sqrt + a.eps;
        pv -= step * (mv / denom);
    };
    const long end = min(base + CHUNK, n);
    if (al16(p) && al16(g) && al16(m) && al16(v) && end - base == CHUNK) {   // full chunk, aligned tensors: float4 streams
#pragma unroll
        for (int k = 0; k < CHUNK / 1024; ++k) {
            const long i = base + 4 * (threadIdx.x + 256 * k);
            const float4 g4 = *reinterpret_cast<const float4*>(g + i);
            float4 p4 = *reinterpret_cast<const float4*>(p + i), m4 = *reinterpret_cast<const float4*>(m + i),
                   v4 = *reinterpret_cast<const float4*>(v + i);
            upd(g4.x, p4.x, m4.x, v4.x); upd(g4.y, p4.y, m4.y, v4.y); upd(g4.z, p4.z, m4.z, v4.z); upd(g4.w, p4.w, m4.w, v4.w);
            *reinterpret_cast<float4*>(p + i) = p4;
            *reinterpret_cast<float4*>(m + i) = m4;
            *reinterpret_cast<float4*>(v + i) = v4;
        }
        return;
    }
    for (long i = base + threadIdx.x; i < end; i += 256) {
        float pv = p[i], mv = m[i], vv = v[i];
        upd(g[i], pv, mv, vv);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}
}  // namespace

extern "C" int tdr_optim_chunk(void) { return CHUNK; }

extern "C" int tdr_grad_sumsq(const float* const* grads, const int64_t* sizes, const int* chunk_tensor, const int* chunk_index,
                              int n_chunks, double* partial, double* sumsq, void* stream) {
    TDR_REQUIRE(grads && sizes && chunk_tensor && chunk_index && partial && sumsq && n_chunks > 0, "tdr_grad_sumsq: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, st, grads, sizes, chunk_tensor, chunk_index, partial);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, partial, n_chunks, sumsq);
    TDR_LAUNCH_CHECK("grad_sumsq");
    return TDR_OK;
}

extern "C" int tdr_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                              const int64_t* sizes, const int* group, const int* chunk_tensor, const int* chunk_index,
                              int n_chunks, const double* sumsq, const float* group_lr, int n_groups, float max_norm,
                              int use_clip, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    TDR_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes && group && chunk_tensor && chunk_index && sumsq && group_lr,
                "tdr_adamw_step: null pointer");
    TDR_REQUIRE(n_groups >= 1 && n_groups <= 4 && step >= 1, "tdr_adamw_step: n_groups in 1..4 and step >= 1");
    AdamArgs a;
    for (int i = 0; i < 4; ++i) a.lr[i] = i < n_groups ? group_lr[i] : 0.f;
    a.max_norm = max_norm; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    a.use_clip = use_clip;
    a.hp = nullptr;
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, sizes,
                       group, chunk_tensor, chunk_index, sumsq, a);
    TDR_LAUNCH_CHECK("adamw_step");
    return TDR_OK;
}

// Same update with the per-step scalars read from device memory: hp = {lr[0..3], 1-beta1^step, sqrt(1-beta2^step)}.
// The launch arguments do not change from step to step, so the launch can be replayed from a hipGraph while the
// host refreshes `hp` with a 24-byte copy.
extern "C" int tdr_adamw_step_dev(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                  const int64_t* sizes, const int* group, const int* chunk_tensor, const int* chunk_index,
                                  int n_chunks, const double* sumsq, const float* hp, float max_norm, int use_clip, float beta1,
                                  float beta2, float eps, float weight_decay, void* stream) {
    TDR_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes && group && chunk_tensor && chunk_index && sumsq && hp,
                "tdr_adamw_step_dev: null pointer");
    AdamArgs a;
    for (int i = 0; i < 4; ++i) a.lr[i] = 0.f;
    a.max_norm = max_norm; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = 1.f; a.bc2_sqrt = 1.f;
    a.use_clip = use_clip;
    a.hp = hp;
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, sizes,
                       group, chunk_tensor, chunk_index, sumsq, a);
    TDR_LAUNCH_CHECK("adamw_step_dev");
    return TDR_OK;
}

extern "C" int tdr_multi_copy(const float* const* src, float* const* dst, const int64_t* sizes, const int* chunk_tensor,
                              const int* chunk_index, int n_chunks, float scale, void* stream) {
    TDR_REQUIRE(src && dst && sizes && chunk_tensor && chunk_index && n_chunks > 0, "tdr_multi_copy: bad argument");
    hipLaunchKernelGGL(multi_copy_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, src, dst, sizes, chunk_tensor,
                       chunk_index, scale);
    TDR_LAUNCH_CHECK("multi_copy");
    return TDR_OK;
}
