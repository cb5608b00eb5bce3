// ViewFusion glue around the UNet: ragged view stacking (+ q_sample), softmax-over-views
// noise composition (+ MSE and its backward), and the fused reverse-diffusion step tail.
// Reference: model/view_fusion.py:162-164 (q_sample), :244-263 / :95-115 (stacking),
// :265-298 / :116-150 (compose / mean ablation / loss), :70-84,152-177 (posterior, p_sample).
//
// All HBM-bound and tiny next to the UNet; their point is that NOTHING here needs a host
// sync: ragged view counts come in as a device prefix-sum array `off[B+1]`.
#include "common.h"

namespace {

__device__ __forceinline__ int sample_of_view(const int* __restrict__ off, int B, int v) {
    int lo = 0, hi = B;                       // find b with off[b] <= v < off[b+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

// x[v] = [ y_cond[b][v-off[b]] | y_t[b] ]  with optional q_sample on the fly:
//   y_t' = sqrt(level_b) * y_t + sqrt(1-level_b) * noise.
// The conditioning part has Cc channels (3 = an RGB view; 6 = the `relative` configs' view pair,
// experiment.py:274-283 / configs/relative-small-v100-4.yaml:22), the noisy target always 3.
// grid (chunks, S); nc4 = Cc*HW/4 and n4 = 3*HW/4 float4 per image.
__global__ void stack_views_kernel(const float4* __restrict__ y_cond, const float4* __restrict__ y_t,
                                   const float4* __restrict__ noise, const float* __restrict__ level,
                                   const float* __restrict__ angle, const int* __restrict__ off,
                                   float4* __restrict__ x, float* __restrict__ level_s, float* __restrict__ angle_s,
                                   int B, int Nmax, int nc4, int n4, int copy_cond) {
    const int v = blockIdx.y;
    const int b = sample_of_view(off, B, v);
    const int j = v - off[b];
    const float lv = level[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        level_s[v] = lv;
        angle_s[v] = angle[b];
    }
    const float sa = sqrtf(lv), sb = sqrtf(1.0f - lv);
    const float4* c = y_cond + ((size_t)b * Nmax + j) * nc4;
    const float4* t = y_t + (size_t)b * n4;
    const float4* z = noise ? noise + (size_t)b * n4 : nullptr;
    float4* o = x + (size_t)v * (nc4 + n4);
    if (copy_cond)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc4; i += gridDim.x * blockDim.x) o[i] = c[i];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 y = t[i];
        if (z) {
            const float4 e = z[i];
            y.x = sa * y.x + sb * e.x;
            y.y = sa * y.y + sb * e.y;
            y.z = sa * y.z + sb * e.z;
            y.w = sa * y.w + sb * e.w;
        }
        o[nc4 + i] = y;
    }
}

// Composed noise for one float4 of (b, c, pixels): softmax over the sample's views of the
// logits (channels 3..5) weighting the per-view noise (channels 0..2); or the plain mean.
__device__ __forceinline__ float4 compose4(const float* __restrict__ out, int Cout, int HW, int v0, int v1, int c,
                                           int p, int weighting, float4* mx_out, float4* inv_out) {
    const size_t vs = (size_t)Cout * HW;
    const float* e0 = out + (size_t)v0 * vs + (size_t)c * HW + p;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!weighting) {
        for (int v = v0; v < v1; ++v) {
            const float4 e = *reinterpret_cast<const float4*>(e0 + (size_t)(v - v0) * vs);
            acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
        }
        const float inv = 1.0f / (float)(v1 - v0);
        return make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
    const float* l0 = e0 + (size_t)3 * HW;
    float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int v = v0; v < v1; ++v) {
        const float4 l = *reinterpret_cast<const float4*>(l0 + (size_t)(v - v0) * vs);
        mx.x = fmaxf(mx.x, l.x); mx.y = fmaxf(mx.y, l.y); mx.z = fmaxf(mx.z, l.z); mx.w = fmaxf(mx.w, l.w);
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v = v0; v < v1; ++v) {
        const float4 l = *reinterpret_cast<const float4*>(l0 + (size_t)(v - v0) * vs);
        const float4 e = *reinterpret_cast<const float4*>(e0 + (size_t)(v - v0) * vs);
        const float wx = expf(l.x - mx.x), wy = expf(l.y - mx.y), wz = expf(l.z - mx.z), ww = expf(l.w - mx.w);
        sum.x += wx; sum.y += wy; sum.z += wz; sum.w += ww;
        acc.x += wx * e.x; acc.y += wy * e.y; acc.z += wz * e.z; acc.w += ww * e.w;
    }
    const float4 inv = make_float4(1.0f / sum.x, 1.0f / sum.y, 1.0f / sum.z, 1.0f / sum.w);
    if (mx_out) { *mx_out = mx; *inv_out = inv; }
    return make_float4(acc.x * inv.x, acc.y * inv.y, acc.z * inv.z, acc.w * inv.w);
}

// grid (chunks, B).  Writes noise_hat[B][3][HW]; optional weights[B][maxV][3][HW] (zero padded);
// optional per-block partial sums of (target - noise_hat)^2 into loss_part[B*chunks].
__global__ __launch_bounds__(256) void compose_fwd_kernel(const float* __restrict__ out, const int* __restrict__ off,
                                                          const float* __restrict__ target,
                                                          float* __restrict__ noise_hat, float* __restrict__ weights,
                                                          float* __restrict__ loss_part, int Cout, int HW, int maxV,
                                                          int weighting) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int v0 = off[b], v1 = off[b + 1];
    const int n4 = 3 * HW / 4;
    float sq = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int c = (4 * i) / HW, p = 4 * i - c * HW;
        float4 mx, inv;
        const float4 nh = compose4(out, Cout, HW, v0, v1, c, p, weighting, &mx, &inv);
        const size_t o = (size_t)b * 3 * HW + 4 * (size_t)i;
        *reinterpret_cast<float4*>(noise_hat + o) = nh;
        if (target) {
            const float4 t = *reinterpret_cast<const float4*>(target + o);
            const float dx = t.x - nh.x, dy = t.y - nh.y, dz = t.z - nh.z, dw = t.w - nh.w;
            sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        if (weights && weighting) {
            const size_t vs = (size_t)Cout * HW;
            for (int j = 0; j < maxV; ++j) {
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (v0 + j < v1) {
                    const float4 l = *reinterpret_cast<const float4*>(out + (size_t)(v0 + j) * vs +
                                                                     (size_t)(3 + c) * HW + p);
                    w = make_float4(expf(l.x - mx.x) * inv.x, expf(l.y - mx.y) * inv.y, expf(l.z - mx.z) * inv.z,
                                    expf(l.w - mx.w) * inv.w);
                }
                *reinterpret_cast<float4*>(weights + (((size_t)b * maxV + j) * 3) * HW + 4 * (size_t)i) = w;
            }
        }
    }
    if (loss_part) {
        sq = block_sum<256>(sq, red);
        if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = sq;
    }
}

__global__ void loss_finish_kernel(const float* __restrict__ part, float* __restrict__ loss, int n, float scale) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float a = 0.f;
        for (int i = 0; i < n; ++i) a += part[i];
        *loss = a * scale;
    }
}

// d(out) for loss = mean((target - noise_hat)^2) * gloss:
//   g = 2 (nh - target) / n * gloss;  d eps_v = w_v g;  d logit_v = w_v (eps_v - nh) g
// (mean ablation: d eps_v = g / count, logits untouched -> zero).
__global__ __launch_bounds__(256) void compose_mse_bwd_kernel(const float* __restrict__ out,
                                                              const int* __restrict__ off,
                                                              const float* __restrict__ target,
                                                              const float* __restrict__ noise_hat,
                                                              const float* __restrict__ gloss,
                                                              float* __restrict__ dout, int Cout, int HW,
                                                              int weighting, float inv_n) {
    const int b = blockIdx.y;
    const int v0 = off[b], v1 = off[b + 1];
    const int n4 = 3 * HW / 4;
    const float gs = 2.0f * inv_n * gloss[0];
    const size_t vs = (size_t)Cout * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int c = (4 * i) / HW, p = 4 * i - c * HW;
        const size_t o = (size_t)b * 3 * HW + 4 * (size_t)i;
        const float4 nh = *reinterpret_cast<const float4*>(noise_hat + o);
        const float4 t = *reinterpret_cast<const float4*>(target + o);
        const float4 g = make_float4(gs * (nh.x - t.x), gs * (nh.y - t.y), gs * (nh.z - t.z), gs * (nh.w - t.w));
        if (!weighting) {
            const float inv = 1.0f / (float)(v1 - v0);
            const float4 d = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
            for (int v = v0; v < v1; ++v) {
                *reinterpret_cast<float4*>(dout + (size_t)v * vs + (size_t)c * HW + p) = d;
                if (Cout > 3)
                    *reinterpret_cast<float4*>(dout + (size_t)v * vs + (size_t)(3 + c) * HW + p) =
                        make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        float4 mx, inv;
        (void)compose4(out, Cout, HW, v0, v1, c, p, 1, &mx, &inv);
        for (int v = v0; v < v1; ++v) {
            const float* ep = out + (size_t)v * vs + (size_t)c * HW + p;
            const float4 e = *reinterpret_cast<const float4*>(ep);
            const float4 l = *reinterpret_cast<const float4*>(ep + (size_t)3 * HW);
            const float4 w = make_float4(expf(l.x - mx.x) * inv.x, expf(l.y - mx.y) * inv.y, expf(l.z - mx.z) * inv.z,
                                         expf(l.w - mx.w) * inv.w);
            float* dp = dout + (size_t)v * vs + (size_t)c * HW + p;
            *reinterpret_cast<float4*>(dp) = make_float4(w.x * g.x, w.y * g.y, w.z * g.z, w.w * g.w);
            *reinterpret_cast<float4*>(dp + (size_t)3 * HW) =
                make_float4(w.x * (e.x - nh.x) * g.x, w.y * (e.y - nh.y) * g.y, w.z * (e.z - nh.z) * g.z,
                            w.w * (e.w - nh.w) * g.w);
        }
    }
}

// One reverse step after the UNet: compose -> y0_hat = a_t y_t - b_t eps -> clamp ->
// mean = c1 y0_hat + c2 y_t -> y_{t-1} = mean + z * exp(0.5 logvar).
__global__ __launch_bounds__(256) void p_sample_tail_kernel(
    const float* __restrict__ out, const int* __restrict__ off, const float* __restrict__ y_t,
    const float* __restrict__ z, const long long* __restrict__ t, const float* __restrict__ sqrt_recip,
    const float* __restrict__ sqrt_recipm1, const float* __restrict__ logvar, const float* __restrict__ coef1,
    const float* __restrict__ coef2, float* __restrict__ y_next, float* __restrict__ mean_out,
    float* __restrict__ weights, int Cout, int HW, int maxV, int weighting, int clip) {
    const int b = blockIdx.y;
    const int v0 = off[b], v1 = off[b + 1];
    const int n4 = 3 * HW / 4;
    const long long tb = t[b];
    const float a_t = sqrt_recip[tb], b_t = sqrt_recipm1[tb], c1 = coef1[tb], c2 = coef2[tb];
    const float sd = expf(0.5f * logvar[tb]);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const int c = (4 * i) / HW, p = 4 * i - c * HW;
        float4 mx, inv;
        const float4 eps = compose4(out, Cout, HW, v0, v1, c, p, weighting, &mx, &inv);
        const size_t o = (size_t)b * 3 * HW + 4 * (size_t)i;
        const float4 y = *reinterpret_cast<const float4*>(y_t + o);
        float y0[4] = {a_t * y.x - b_t * eps.x, a_t * y.y - b_t * eps.y, a_t * y.z - b_t * eps.z,
                       a_t * y.w - b_t * eps.w};
        const float ys[4] = {y.x, y.y, y.z, y.w};
        float m[4], r[4];
        float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
        if (z) zz = *reinterpret_cast<const float4*>(z + o);
        const float zs[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (clip) y0[k] = fminf(fmaxf(y0[k], -1.0f), 1.0f);
            m[k] = c1 * y0[k] + c2 * ys[k];
            r[k] = m[k] + zs[k] * sd;
        }
        if (y_next) *reinterpret_cast<float4*>(y_next + o) = make_float4(r[0], r[1], r[2], r[3]);
        if (mean_out) *reinterpret_cast<float4*>(mean_out + o) = make_float4(m[0], m[1], m[2], m[3]);
        if (weights && weighting) {
            const size_t vs = (size_t)Cout * HW;
            for (int j = 0; j < maxV; ++j) {
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (v0 + j < v1) {
                    const float4 l = *reinterpret_cast<const float4*>(out + (size_t)(v0 + j) * vs +
                                                                     (size_t)(3 + c) * HW + p);
                    w = make_float4(expf(l.x - mx.x) * inv.x, expf(l.y - mx.y) * inv.y, expf(l.z - mx.z) * inv.z,
                                    expf(l.w - mx.w) * inv.w);
                }
                *reinterpret_cast<float4*>(weights + (((size_t)b * maxV + j) * 3) * HW + 4 * (size_t)i) = w;
            }
        }
    }
}

// level[b] = table[t[b]]  (extract(), view_fusion.py:314-317) or the training draw
// level = (g[t]-g[t-1])*u + g[t-1]  (view_fusion.py:231-237) when u != null.
__global__ void gather_level_kernel(const float* __restrict__ gammas, const long long* __restrict__ t,
                                    const float* __restrict__ u, float* __restrict__ level, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long long tb = t[b];
    const float hi = gammas[tb];
    if (u) {
        const float lo = gammas[tb - 1];
        level[b] = (hi - lo) * u[b] + lo;
    } else {
        level[b] = hi;
    }
}

// psnr[b] = 20 log10(1 / sqrt(mean((a-b)^2)))  -- one workgroup per image (utils/metrics.py:6-8)
__global__ __launch_bounds__(256) void psnr_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   float* __restrict__ out, int n) {
    __shared__ float red[4];
    const float4* a4 = reinterpret_cast<const float4*>(a + (size_t)blockIdx.x * n);
    const float4* b4 = reinterpret_cast<const float4*>(b + (size_t)blockIdx.x * n);
    float s = 0.f;
    for (int i = threadIdx.x; i < (n >> 2); i += 256) {
        const float4 u = a4[i], v = b4[i];
        const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z, dw = u.w - v.w;
        s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = 20.0f * log10f(1.0f / sqrtf(s / (float)n));
}

inline int chunks_for(int n4) {
    int c = (n4 + 255) / 256;
    return c < 1 ? 1 : (c > 64 ? 64 : c);
}

}  // namespace

extern "C" {

int vf_psnr(const float* generated, const float* target, float* out, int B, int n, void* stream) {
    if (B <= 0) return 0;
    if (n & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(psnr_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, generated, target, out, n);
    VF_RETURN_LAST_ERROR();
}

int vf_gather_level(const float* gammas, const long long* t, const float* u, float* level, int B, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(gather_level_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, gammas, t, u,
                       level, B);
    VF_RETURN_LAST_ERROR();
}

// y_cond [B][Nmax][3][HW], y_t [B][3][HW], noise [B][3][HW] or null, level/angle [B],
// off [B+1] -> x [S][6][HW], level_s/angle_s [S].
int vf_stack_views(const float* y_cond, const float* y_t, const float* noise, const float* level,
                   const float* angle, const int* off, float* x, float* level_s, float* angle_s, int B, int Nmax,
                   int Cc, int HW, int S, int copy_cond, void* stream) {
    if (S <= 0) return 0;
    if ((HW & 3) || Cc < 1) return (int)hipErrorInvalidValue;
    const int n4 = 3 * HW / 4, nc4 = Cc * HW / 4;
    hipLaunchKernelGGL(stack_views_kernel, dim3(chunks_for(nc4 > n4 ? nc4 : n4), S), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)y_cond, (const float4*)y_t, (const float4*)noise, level, angle, off,
                       (float4*)x, level_s, angle_s, B, Nmax, nc4, n4, copy_cond);
    VF_RETURN_LAST_ERROR();
}

// loss_part must hold B*64 floats when target != null.
int vf_compose_fwd(const float* unet_out, const int* off, const float* target, float* noise_hat, float* weights,
                   float* loss_part, float* loss, int B, int Cout, int HW, int maxV, int weighting, void* stream) {
    if (B <= 0) return 0;
    if ((HW & 3) || Cout < 3 || (weighting && Cout < 6)) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int ch = chunks_for(3 * HW / 4);
    hipLaunchKernelGGL(compose_fwd_kernel, dim3(ch, B), dim3(256), 0, st, unet_out, off, target, noise_hat, weights,
                       target ? loss_part : nullptr, Cout, HW, maxV, weighting);
    if (target)
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, loss_part, loss, B * ch,
                           1.0f / ((float)B * 3.0f * (float)HW));
    VF_RETURN_LAST_ERROR();
}

int vf_compose_mse_bwd(const float* unet_out, const int* off, const float* target, const float* noise_hat,
                       const float* gloss, float* dout, int B, int Cout, int HW, int weighting, void* stream) {
    if (B <= 0) return 0;
    if ((HW & 3) || Cout < 3 || (weighting && Cout < 6)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(compose_mse_bwd_kernel, dim3(chunks_for(3 * HW / 4), B), dim3(256), 0, (hipStream_t)stream,
                       unet_out, off, target, noise_hat, gloss, dout, Cout, HW, weighting,
                       1.0f / ((float)B * 3.0f * (float)HW));
    VF_RETURN_LAST_ERROR();
}

int vf_p_sample_tail(const float* unet_out, const int* off, const float* y_t, const float* z, const long long* t,
                     const float* sqrt_recip_gammas, const float* sqrt_recipm1_gammas,
                     const float* posterior_log_variance, const float* posterior_mean_coef1,
                     const float* posterior_mean_coef2, float* y_next, float* mean_out, float* weights, int B,
                     int Cout, int HW, int maxV, int weighting, int clip, void* stream) {
    if (B <= 0) return 0;
    if ((HW & 3) || Cout < 3 || (weighting && Cout < 6)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(p_sample_tail_kernel, dim3(chunks_for(3 * HW / 4), B), dim3(256), 0, (hipStream_t)stream,
                       unet_out, off, y_t, z, t, sqrt_recip_gammas, sqrt_recipm1_gammas, posterior_log_variance,
                       posterior_mean_coef1, posterior_mean_coef2, y_next, mean_out, weights, Cout, HW, maxV,
                       weighting, clip);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
