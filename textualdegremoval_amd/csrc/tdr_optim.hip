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
                                                   double* __restrict__ partial, const int* __restrict__ group) {
    __shared__ double red[4];
    const int t = chunk_tensor[blockIdx.x];
    if (group && group[t] < 0) {                       // frozen tensor (requires_grad False in the reference): not in the norm
        if (threadIdx.x == 0) partial[blockIdx.x] = 0.0;
        return;
    }
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

// The same reduction, then the step verdict (see TdrStepGuard in include/tdr.h): a non-finite gradient norm means an
// operand of the loss-scaled fp16-split backward pass left the fp16 range (or the forward pass produced inf/nan) --
// the step is skipped and the loss scale halved; after `growth_interval` finite steps the scale doubles again, up to
// max_scale.  Every rank sees the same all-reduced gradients, hence takes the same decision.
__global__ __launch_bounds__(256) void sumsq_finish_guard_kernel(const double* __restrict__ partial, int n, double* __restrict__ out,
                                                                 TdrStepGuard* __restrict__ g, double beta1, double beta2) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double tot = red[0];
        out[0] = tot;
        // growth_interval < 0: no verdict -- the step is applied whatever the norm is, as torch's clip_grad_norm_ + AdamW.step()
        // do (image_restoration_ref_model.py:276-279): the arithmetic modes without a loss scale (TDR_MATH=bx3 / f32) have no guard
        const bool ok = g->growth_interval < 0 || isfinite(tot);
        g->finite = ok ? 1 : 0;
        if (ok) {
            const int t = ++g->step;
            g->bc1 = (float)(1.0 - pow(beta1, (double)t));
            g->bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)t));
            if (g->growth_interval > 0 && ++g->good >= g->growth_interval) {
                g->good = 0;
                if (g->scale < g->max_scale) { g->scale *= 2.f; g->inv_scale *= 0.5f; }
            }
        } else {
            ++g->skipped;
            g->good = 0;
            if (g->scale > 1.f) { g->scale *= 0.5f; g->inv_scale *= 2.f; }
        }
    }
}

// dst[t][i] = src[t][i] for every tensor of a (tensor, chunk) table: gathers ~900 freshly produced gradient
// tensors into the flat all-reduce / optimiser arena with ONE launch instead of one copy kernel per tensor.
__global__ __launch_bounds__(256) void multi_copy_kernel(const float* const* __restrict__ src, float* const* __restrict__ dst,
                                                        const int64_t* __restrict__ sizes, const int* __restrict__ chunk_tensor,
                                                        const int* __restrict__ chunk_index, float scale,
                                                        const TdrStepGuard* __restrict__ guard) {
    if (guard) scale = guard->inv_scale;
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

// dst[t] = decay * dst[t] + (1 - decay) * src[t] over a (tensor, chunk) table: the EMA copy of the weights
// (models/base_model.py:54-62) in one launch.
__global__ __launch_bounds__(256) void multi_ema_kernel(const float* const* __restrict__ src, float* const* __restrict__ dst,
                                                       const int64_t* __restrict__ sizes, const int* __restrict__ chunk_tensor,
                                                       const int* __restrict__ chunk_index, float decay) {
    const int t = chunk_tensor[blockIdx.x];
    const long base = (long)chunk_index[blockIdx.x] * CHUNK;
    const long end = min(base + CHUNK, (long)sizes[t]);
    const float* s = src[t];
    float* d = dst[t];
    const float a = 1.f - decay;
    for (long i = base + threadIdx.x; i < end; i += 256) d[i] = d[i] * decay + s[i] * a;     // torch: mul_(decay).add_(src, alpha=1-decay)
}

struct AdamArgs {
    float lr[4];
    float max_norm, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt;
    int use_clip;
    int coupled_decay;  // 0: AdamW (decoupled decay); 1: torch.optim.Adam (weight_decay * p added to the gradient)
    const float* hp;    // optional device-resident {lr[4], bc1, bc2_sqrt} (graph replay: values change, launch does not)
    const TdrStepGuard* guard;   // optional: skip the update when !finite; bias corrections from the guard's own step count
};

__global__ __launch_bounds__(256) void adamw_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                   float* const* __restrict__ exp_avg, float* const* __restrict__ exp_avg_sq,
                                                   const int64_t* __restrict__ sizes, const int* __restrict__ group,
                                                   const int* __restrict__ chunk_tensor, const int* __restrict__ chunk_index,
                                                   const double* __restrict__ sumsq, AdamArgs a) {
    const int t = chunk_tensor[blockIdx.x];
    if (group[t] < 0) return;                           // frozen tensor
    if (a.guard && !a.guard->finite) return;            // skipped step (uniform over the grid)
    const long base = (long)chunk_index[blockIdx.x] * CHUNK;
    const long n = sizes[t];
    float* p = params[t];
    const float* g = grads[t];
    float* m = exp_avg[t];
    float* v = exp_avg_sq[t];
    const float lr = a.hp ? a.hp[group[t]] : a.lr[group[t]];
    const float bc1 = a.guard ? a.guard->bc1 : (a.hp ? a.hp[4] : a.bc1);
    const float bc2_sqrt = a.guard ? a.guard->bc2_sqrt : (a.hp ? a.hp[5] : a.bc2_sqrt);
    float coef = 1.f;
    if (a.use_clip) {
        const float total = (float)sqrt(sumsq[0]);
        coef = fminf(a.max_norm / (total + 1e-6f), 1.f);
    }
    const float step = lr / bc1;
    auto upd = [&](float g0, float& pv, float& mv, float& vv) {
        float gv = g0 * coef;
        if (a.coupled_decay) gv += a.weight_decay * pv;
        else pv *= (1.f - lr * a.weight_decay);
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
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, st, grads, sizes, chunk_tensor, chunk_index, partial,
                       (const int*)nullptr);
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, partial, n_chunks, sumsq);
    TDR_LAUNCH_CHECK("grad_sumsq");
    return TDR_OK;
}

extern "C" int tdr_grad_sumsq_guarded(const float* const* grads, const int64_t* sizes, const int* group, const int* chunk_tensor,
                                      const int* chunk_index, int n_chunks, double* partial, double* sumsq, TdrStepGuard* guard,
                                      float beta1, float beta2, void* stream) {
    TDR_REQUIRE(grads && sizes && group && chunk_tensor && chunk_index && partial && sumsq && guard && n_chunks > 0,
                "tdr_grad_sumsq_guarded: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, st, grads, sizes, chunk_tensor, chunk_index, partial, group);
    hipLaunchKernelGGL(sumsq_finish_guard_kernel, dim3(1), dim3(256), 0, st, partial, n_chunks, sumsq, guard, (double)beta1,
                       (double)beta2);
    TDR_LAUNCH_CHECK("grad_sumsq_guarded");
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
    a.coupled_decay = 0;
    a.hp = nullptr;
    a.guard = nullptr;
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
    a.coupled_decay = 0;
    a.hp = hp;
    a.guard = nullptr;
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, sizes,
                       group, chunk_tensor, chunk_index, sumsq, a);
    TDR_LAUNCH_CHECK("adamw_step_dev");
    return TDR_OK;
}

// The guarded step: learning rates from `hp` (4 floats, refreshed by the host every step), verdict / step count / bias
// corrections from the device-resident TdrStepGuard that tdr_grad_sumsq_guarded just updated.  group[t] < 0 freezes a
// tensor.  coupled_decay selects torch.optim.Adam's L2 term instead of AdamW's decoupled decay.
extern "C" int tdr_adamw_step_guarded(float* const* params, const float* const* grads, float* const* exp_avg,
                                      float* const* exp_avg_sq, const int64_t* sizes, const int* group, const int* chunk_tensor,
                                      const int* chunk_index, int n_chunks, const double* sumsq, const float* hp,
                                      const TdrStepGuard* guard, float max_norm, int use_clip, int coupled_decay, float beta1,
                                      float beta2, float eps, float weight_decay, void* stream) {
    TDR_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes && group && chunk_tensor && chunk_index && sumsq && hp && guard,
                "tdr_adamw_step_guarded: null pointer");
    AdamArgs a;
    for (int i = 0; i < 4; ++i) a.lr[i] = 0.f;
    a.max_norm = max_norm; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = 1.f; a.bc2_sqrt = 1.f;
    a.use_clip = use_clip;
    a.coupled_decay = coupled_decay;
    a.hp = hp;
    a.guard = guard;
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, sizes,
                       group, chunk_tensor, chunk_index, sumsq, a);
    TDR_LAUNCH_CHECK("adamw_step_guarded");
    return TDR_OK;
}

extern "C" int tdr_multi_copy(const float* const* src, float* const* dst, const int64_t* sizes, const int* chunk_tensor,
                              const int* chunk_index, int n_chunks, float scale, void* stream) {
    TDR_REQUIRE(src && dst && sizes && chunk_tensor && chunk_index && n_chunks > 0, "tdr_multi_copy: bad argument");
    hipLaunchKernelGGL(multi_copy_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, src, dst, sizes, chunk_tensor,
                       chunk_index, scale, (const TdrStepGuard*)nullptr);
    TDR_LAUNCH_CHECK("multi_copy");
    return TDR_OK;
}

extern "C" int tdr_multi_ema(const float* const* src, float* const* dst, const int64_t* sizes, const int* chunk_tensor,
                             const int* chunk_index, int n_chunks, float decay, void* stream) {
    TDR_REQUIRE(src && dst && sizes && chunk_tensor && chunk_index && n_chunks > 0, "tdr_multi_ema: bad argument");
    hipLaunchKernelGGL(multi_ema_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, src, dst, sizes, chunk_tensor,
                       chunk_index, decay);
    TDR_LAUNCH_CHECK("multi_ema");
    return TDR_OK;
}

extern "C" int tdr_multi_copy_guarded(const float* const* src, float* const* dst, const int64_t* sizes, const int* chunk_tensor,
                                      const int* chunk_index, int n_chunks, const TdrStepGuard* guard, void* stream) {
    TDR_REQUIRE(src && dst && sizes && chunk_tensor && chunk_index && guard && n_chunks > 0, "tdr_multi_copy_guarded: bad argument");
    hipLaunchKernelGGL(multi_copy_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, src, dst, sizes, chunk_tensor,
                       chunk_index, 1.f, guard);
    TDR_LAUNCH_CHECK("multi_copy_guarded");
    return TDR_OK;
}
