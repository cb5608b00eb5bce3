// Small HBM-bound helpers of the UNet: sinusoidal (level, angle) embedding, Swish,
// channel concat / split.  Reference: model/unet.py:142-157 (encoding), :180-182 (Swish),
// :134 (skip concat).
#include "common.h"

namespace {

// out[s][0:cnt)=sin(level*f), [cnt:2cnt)=cos(level*f), [2cnt:3cnt)=sin(angle*f), [3cnt:4cnt)=cos(angle*f)
// f_k = exp(-ln(1e4) * k / cnt), cnt = dim / 4.
__global__ void sincos_embed_kernel(const float* __restrict__ level, const float* __restrict__ angle,
                                    float* __restrict__ out, int S, int dim) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * dim) return;
    const int s = idx / dim, j = idx - s * dim;
    const int cnt = dim >> 2, half = dim >> 1;
    const int jj = j < half ? j : j - half;
    const float v = j < half ? level[s] : angle[s];
    const int k = jj < cnt ? jj : jj - cnt;
    const float step = (float)k / (float)cnt;
    const float arg = v * expf(-9.210340371976184f * step);
    out[idx] = jj < cnt ? sinf(arg) : cosf(arg);
}

__global__ void swish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = silu_f(x[i]);
}

__global__ void swish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                 size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float z = x[i];
        const float sg = sigmoid_f(z);
        dx[i] = dy[i] * (sg * (1.0f + z * (1.0f - sg)));
    }
}

// nn.Dropout(p) of Block (unet.py:207-216, training mode only): y = x * (u >= p) / (1 - p) with the uniform
// draws u in [0,1) supplied by the caller (the host's RNG; the same kernel applied to dy is the backward).
__global__ void dropout_kernel(const float4* __restrict__ x, const float4* __restrict__ u, float4* __restrict__ y,
                               size_t n4, float p, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 a = x[i], r = u[i];
    y[i] = make_float4(r.x >= p ? a.x * scale : 0.f, r.y >= p ? a.y * scale : 0.f, r.z >= p ? a.z * scale : 0.f,
                       r.w >= p ? a.w * scale : 0.f);
}

// out[s] = [a[s] | b[s]] along channels; na4 / nb4 = float4 per sample of a / b.
// split != 0 runs the copy the other way (out -> a, b).
__global__ void concat_kernel(float4* __restrict__ a, float4* __restrict__ b, float4* __restrict__ out, int na4,
                              int nb4, int split) {
    const int s = blockIdx.y;
    const int n4 = na4 + nb4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4* o = out + (size_t)s * n4 + i;
        float4* src = i < na4 ? a + (size_t)s * na4 + i : b + (size_t)s * nb4 + (i - na4);
        if (split) *src = *o; else *o = *src;
    }
}

}  // namespace

extern "C" {

int vf_sincos_embed(const float* level, const float* angle, float* out, int S, int dim, void* stream) {
    if (S <= 0) return 0;
    if (dim & 3) return (int)hipErrorInvalidValue;
    const int n = S * dim;
    hipLaunchKernelGGL(sincos_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, level, angle,
                       out, S, dim);
    VF_RETURN_LAST_ERROR();
}

int vf_swish_fwd(const float* x, float* y, long n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(swish_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       (size_t)n);
    VF_RETURN_LAST_ERROR();
}

int vf_swish_bwd(const float* x, const float* dy, float* dx, long n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(swish_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy,
                       dx, (size_t)n);
    VF_RETURN_LAST_ERROR();
}

// n a multiple of 4; 0 <= p < 1.
int vf_dropout(const float* x, const float* u, float* y, long n, float p, void* stream) {
    if (n <= 0) return 0;
    if ((n & 3) || !(p >= 0.f && p < 1.f)) return (int)hipErrorInvalidValue;
    const size_t n4 = (size_t)n >> 2;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, (const float4*)u, (float4*)y, n4, p, 1.0f / (1.0f - p));
    VF_RETURN_LAST_ERROR();
}

// a: [S][Ca*HW], b: [S][Cb*HW], out: [S][(Ca+Cb)*HW]; per-sample sizes in floats, multiples of 4.
int vf_concat_channels(float* a, float* b, float* out, int S, long na, long nb, int split, void* stream) {
    if (S <= 0) return 0;
    if ((na & 3) || (nb & 3)) return (int)hipErrorInvalidValue;
    const int na4 = (int)(na >> 2), nb4 = (int)(nb >> 2);
    int bx = (na4 + nb4 + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(concat_kernel, dim3(bx, S), dim3(256), 0, (hipStream_t)stream, (float4*)a, (float4*)b,
                       (float4*)out, na4, nb4, split);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
