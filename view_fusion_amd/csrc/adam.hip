// Multi-tensor Adam: every parameter of the model updated by ONE launch (reference optimizer:
// torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, no weight decay / amsgrad), experiment.py:118-120;
// the LR comes from the host-side schedule utils/schedulers.py:10-14).  HBM-bound: 16 B read +
// 12 B written per parameter.  Same arithmetic as torch's single-tensor path:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"
#include "adam_update.h"

namespace {

struct AdamDesc {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long numel, first_block;
};

constexpr int ADAM_PER_BLOCK = 1024;   // 256 threads x 4 elements

// scal (or null): device float[3] {lr, bc1, bc2} read at run time instead of the launch arguments -- a HIP-graph replay
// of the training step takes the step-dependent scalars from memory the host refreshes before every replay
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamDesc* __restrict__ desc, int ntensors, float lr,
                                                         float b1, float b2, float eps, float bc1, float bc2,
                                                         const float* __restrict__ scal) {
    if (scal) { lr = scal[0]; bc1 = scal[1]; bc2 = scal[2]; }
    int lo = 0, hi = ntensors;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    const AdamDesc d = desc[lo];
    const long long base = ((long long)blockIdx.x - d.first_block) * ADAM_PER_BLOCK + 4 * threadIdx.x;
    if (base >= d.numel) return;
    const float step = lr / bc1, rs = 1.0f / sqrtf(bc2), omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    auto upd = [&](float& p, float g, float& m, float& v) { vf_adam_update(p, g, m, v, b1, b2, omb1, omb2, step, rs, eps); };
    if (base + 4 <= d.numel) {
        float4 p = *reinterpret_cast<float4*>(d.p + base);
        const float4 g = *reinterpret_cast<const float4*>(d.g + base);
        float4 m = *reinterpret_cast<float4*>(d.m + base);
        float4 v = *reinterpret_cast<float4*>(d.v + base);
        upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
        *reinterpret_cast<float4*>(d.p + base) = p;
        *reinterpret_cast<float4*>(d.m + base) = m;
        *reinterpret_cast<float4*>(d.v + base) = v;
    } else {
        for (long long i = base; i < d.numel; ++i) {
            float p = d.p[i], m = d.m[i], v = d.v[i];
            upd(p, d.g[i], m, v);
            d.p[i] = p; d.m[i] = m; d.v[i] = v;
        }
    }
}

}  // namespace

extern "C" {

// desc: device int64 [ntensors][6] rows {p, g, m, v, numel, first_block}; a block covers 1024
// elements of one tensor; all pointers 16-byte aligned.  bc1 = 1-beta1^t, bc2 = 1-beta2^t.
int vf_adam_multi(const void* desc, int ntensors, long total_blocks, float lr, float beta1, float beta2, float eps,
                  float bc1, float bc2, void* stream) {
    if (ntensors <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const AdamDesc*)desc, ntensors, lr, beta1, beta2, eps, bc1, bc2, (const float*)nullptr);
    VF_RETURN_LAST_ERROR();
}

__global__ void adam_set_scalars_kernel(float* dst, float a, float b, float c) {
    dst[0] = a; dst[1] = b; dst[2] = c;
}

// dst[0..2] = {lr, bc1, bc2}: the values travel as launch arguments (copied at enqueue time), so the host may run any
// number of replays ahead of the GPU without a staging buffer to guard
int vf_adam_set_scalars(float* dst, float lr, float bc1, float bc2, void* stream) {
    if (!dst) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(adam_set_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, lr, bc1, bc2);
    VF_RETURN_LAST_ERROR();
}

// the same update with {lr, 1-beta1^t, 1-beta2^t} read from device memory (scalars: float[3]) when the kernel runs: the
// form a captured training step uses (the host rewrites the three floats before every graph replay)
int vf_adam_multi_dev(const void* desc, int ntensors, long total_blocks, const float* scalars, float beta1, float beta2,
                      float eps, void* stream) {
    if (ntensors <= 0 || total_blocks <= 0) return 0;
    if (!scalars) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const AdamDesc*)desc, ntensors, 0.f, beta1, beta2, eps, 1.f, 1.f, scalars);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
