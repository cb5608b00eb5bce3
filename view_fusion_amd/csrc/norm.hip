// GroupNorm(+Swish) forward / backward and the small row / column reductions.
//
// Replaces the reference's nn.GroupNorm(32, C, eps=1e-5) -> Swish pairs
// (model/unet.py:207-218, :254, :180-182).  All kernels are HBM-bound:
//   fwd : one workgroup per (view, group); the whole group (<= 32768 floats) is held in
//         registers, so x is read ONCE and y written once (two-pass mean / variance in
//         registers -> no E[x^2]-E[x]^2 cancellation).
//   bwd : per-(view,channel) row sums (one wave per row), then a streaming dx pass.
// Layout: NCHW fp32, a group is cpg*H*W contiguous floats.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld_nt(const float4* p) {          // streaming (non-temporal) 16-byte load
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

// "cat" inputs: the normalised tensor is the channel concatenation [x1 (C1 channels) | x2 (C - C1)] of two
// contiguous NCHW tensors (the decoder's skip connections, reference unet.py:134) that is never materialised.
// float4 index i4 of group (s, g) -> its address in whichever tensor holds that channel.
__device__ __forceinline__ const float4* cat_ptr(const float* x1, const float* x2, int C1, int C, int s, int c,
                                                 int HW) {
    return reinterpret_cast<const float4*>(c < C1 ? x1 + ((size_t)s * C1 + c) * HW
                                                  : x2 + ((size_t)s * (C - C1) + (c - C1)) * HW);
}

template <int NV, int NT>
__global__ __launch_bounds__(NT) void gn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta,
                                                    float* __restrict__ y, float* __restrict__ mean_out,
                                                    float* __restrict__ rstd_out, int C, int HW, int cpg,
                                                    float eps, int silu) {
    __shared__ float red[NT / 64];
    const int G = C / cpg;
    const int sg = blockIdx.x;
    const int s = sg / G, g = sg - s * G;
    const size_t base = ((size_t)s * C + (size_t)g * cpg) * HW;
    const int n = cpg * HW, n4 = n >> 2;
    // HW is a power of two (checked by the launcher): channel-of-element is a shift, not an integer division
    // (a runtime division is ~30 VALU instructions, and there is one per float4 of payload)
    const int hwsh = 31 - __clz(HW >> 2), hwmask = (HW >> 2) - 1;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    float4* y4 = reinterpret_cast<float4*>(y + base);

    // Loads are unconditional on clamped indices and issued back to back (data, then the per-channel affine
    // parameters): a load inside an `if` makes the compiler drain the load queue (s_waitcnt vmcnt(0)) at every
    // iteration, which serialises the round trips.  Out-of-range slots are zeroed by a select afterwards.
    float4 v[NV];
    float gam[NV], bet[NV];
    if (x2) {              // concatenated input: per-access source tensor
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idc = min((int)threadIdx.x + i * NT, n4 - 1);
            v[i] = cat_ptr(x, x2, C1, C, s, g * cpg + (idc >> hwsh), HW)[idc & hwmask];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = x4[min((int)threadIdx.x + i * NT, n4 - 1)];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = g * cpg + (min((int)threadIdx.x + i * NT, n4 - 1) >> hwsh);
        gam[i] = gamma[c];
        bet[i] = beta[c];
    }
#ifdef VF_GN_PRESTATS
    // DIAGNOSTIC build only (tools/gn_prestats_ab.sh, round 6): the statistics are handed in, as they would be if the
    // producing conv's epilogue had left them -- what is left is a pure streaming normalise.  Prices review item 5.
    const float mean = mean_out[sg], rstd = rstd_out[sg];
    (void)red; (void)n; (void)eps;
#else
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((int)threadIdx.x + i * NT >= n4) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = block_sum<NT>(sum, red) / (float)n;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        const float q = (a * a + b * b) + (c * c + d * d);
        sq += (int)threadIdx.x + i * NT < n4 ? q : 0.f;
    }
    const float var = block_sum<NT>(sq, red) / (float)n;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
        mean_out[sg] = mean;
        rstd_out[sg] = rstd;
    }
#endif
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * NT;
        if (idx < n4) {
            const float ga = gam[i] * rstd;
            const float be = bet[i] - mean * ga;
            float4 o;
            o.x = v[i].x * ga + be;
            o.y = v[i].y * ga + be;
            o.z = v[i].z * ga + be;
            o.w = v[i].w * ga + be;
            if (silu) {
                o.x = silu_f(o.x);
                o.y = silu_f(o.y);
                o.z = silu_f(o.z);
                o.w = silu_f(o.w);
            }
            y4[idx] = o;
        }
    }
}

__device__ __forceinline__ float dsilu_mul(float z, float dy) {
    const float sg = sigmoid_f(z);
    return dy * (sg * (1.0f + z * (1.0f - sg)));
}

// One wave per (view, channel) row: a = sum(dz), b = sum(dz * xhat).
__global__ __launch_bounds__(256) void gn_bwd_rows_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          float* __restrict__ dbeta_part,
                                                          float* __restrict__ dgamma_part, int rows, int C,
                                                          int HW, int cpg, int silu) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int s = row / C, c = row - s * C;
    const int sg = s * (C / cpg) + c / cpg;
    const float mu = mean[sg], r = rstd[sg], ga = gamma[c], be = beta[c];
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)row * HW);
    const float4* d4 = reinterpret_cast<const float4*>(dy + (size_t)row * HW);
    float a = 0.f, b = 0.f;
    for (int i = lane; i < (HW >> 2); i += 64) {
        const float4 xv = x4[i], dv = d4[i];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - mu) * r;
            const float dz = silu ? dsilu_mul(xh * ga + be, ds[j]) : ds[j];
            a += dz;
            b += dz * xh;
        }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        dbeta_part[row] = a;
        dgamma_part[row] = b;
    }
}

// dx = rstd * (dz*gamma - (s1 + xhat*s2)/n),  s1 = sum_c gamma_c a[s][c], s2 = sum_c gamma_c b[s][c].
__global__ __launch_bounds__(256) void gn_bwd_dx_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd,
                                                        const float* __restrict__ dbeta_part,
                                                        const float* __restrict__ dgamma_part,
                                                        float* __restrict__ dx, int C, int HW, int cpg,
                                                        int silu) {
    const int G = C / cpg;
    const int sg = blockIdx.x;
    const int s = sg / G, g = sg - s * G;
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < cpg; ++j) {
        const int c = g * cpg + j;
        const float ga = gamma[c];
        s1 += ga * dbeta_part[s * C + c];
        s2 += ga * dgamma_part[s * C + c];
    }
    const int n = cpg * HW, n4 = n >> 2;
    const int hwsh = 31 - __clz(HW >> 2);
    const float inv_n = 1.0f / (float)n;
    s1 *= inv_n;
    s2 *= inv_n;
    const float mu = mean[sg], r = rstd[sg];
    const size_t base = ((size_t)s * C + (size_t)g * cpg) * HW;
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* d4 = reinterpret_cast<const float4*>(dy + base);
    float4* o4 = reinterpret_cast<float4*>(dx + base);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n4; i += gridDim.y * 256) {
        const int c = g * cpg + (i >> hwsh);
        const float ga = gamma[c], be = beta[c];
        const float4 xv = x4[i], dv = d4[i];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - mu) * r;
            const float dz = silu ? dsilu_mul(xh * ga + be, ds[j]) : ds[j];
            o[j] = r * (dz * ga - (s1 + xh * s2));
        }
        o4[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}


// Single-pass backward: the whole (view, group) lives in registers (x and dy read ONCE), like the
// forward.  Requires H*W >= 256 so that the 64 float4 of one wave-wide access sit in one channel:
// per-channel sums are then wave reductions accumulated by lane 0 into a per-(wave, channel) LDS
// slot in a fixed order (deterministic), combined after one barrier.
// Optional `addend` (same shape as x) is added to dx: fuses the gradient of a second use of x
// (residual / skip branch) that autograd would otherwise sum with a separate kernel.
// SEG = lanes that share a channel in one 64-lane access: 64 when a channel row has >= 64 float4 (HW >= 256),
// 16 on 8x8 maps (HW = 64: four channels per access, reduced per 16-lane segment).
template <int NV, int NT, int SEG = 64>
__global__ __launch_bounds__(NT) void gn_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                          int C1, const float* __restrict__ dy,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ addend,
                                                          const float* __restrict__ addend2, float* __restrict__ dx,
                                                          float* __restrict__ dx2,
                                                          float* __restrict__ dgamma_part,
                                                          float* __restrict__ dbeta_part,
                                                          float* __restrict__ dx_rowsum, int C, int HW, int cpg,
                                                          int silu) {
    constexpr int NWV = NT / 64;
    extern __shared__ float part[];                  // [NWV][cpg][3]
    const int G = C / cpg;
    const int sg = blockIdx.x;
    const int s = sg / G, g = sg - s * G;
    const size_t base = ((size_t)s * C + (size_t)g * cpg) * HW;
    const int n = cpg * HW, n4 = n >> 2;
    const int hwsh = 31 - __clz(HW >> 2), hwmask = (HW >> 2) - 1;      // HW is a power of two: shifts, not divisions
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* d4 = reinterpret_cast<const float4*>(dy + base);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float mu = mean[sg], r = rstd[sg];

    for (int i = threadIdx.x; i < NWV * cpg * 3; i += NT) part[i] = 0.f;
    __syncthreads();

    // unconditional loads on clamped indices, all issued before the first use (see gn_fwd_kernel)
    float4 xv[NV], dv[NV];
    float gam[NV], bet[NV];
    if (x2) {              // concatenated input (see cat_ptr): per-access source tensor
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idc = min((int)threadIdx.x + i * NT, n4 - 1);
            xv[i] = cat_ptr(x, x2, C1, C, s, g * cpg + (idc >> hwsh), HW)[idc & hwmask];
            dv[i] = d4[idc];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idc = min((int)threadIdx.x + i * NT, n4 - 1);
            // (last use of both tensors in the backward pass: non-temporal loads leave the cache to dx, which the next
            // kernel reads -- gn_bwd 2.80 -> 2.73 ms per step)
            xv[i] = ld_nt(x4 + idc);
            dv[i] = ld_nt(d4 + idc);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (SEG == 64) {   // wave-uniform channel of this access: scalar loads, no vector registers
            const int c = __builtin_amdgcn_readfirstlane(g * cpg + (min(wid * 64 + i * NT, n4 - 1) >> hwsh));
            gam[i] = gamma[c];
            bet[i] = beta[c];
        } else {
            const int c = g * cpg + (min((int)threadIdx.x + i * NT, n4 - 1) >> hwsh);
            gam[i] = gamma[c];
            bet[i] = beta[c];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx0 = wid * 64 + i * NT + (SEG == 64 ? 0 : (lane & ~(SEG - 1)));   // first index of this lane's segment
        float a = 0.f, b = 0.f, xsum = 0.f;
        int cl = 0;
        if (idx0 < n4) {
            cl = idx0 >> hwsh;                        // channel within the group (uniform over the segment)
            const float ga = gam[i], be = bet[i];
            float xs[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
            float ds[4] = {dv[i].x, dv[i].y, dv[i].z, dv[i].w};
            const bool ok = threadIdx.x + i * NT < n4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xs[j] - mu) * r;
                float dz = silu ? dsilu_mul(xh * ga + be, ds[j]) : ds[j];
                if (!ok) dz = 0.f;
                xs[j] = xh; ds[j] = dz;
                a += dz; b += dz * xh;
                xsum += ok ? xh : 0.f;
            }
            xv[i] = make_float4(xs[0], xs[1], xs[2], xs[3]);
            dv[i] = make_float4(ds[0], ds[1], ds[2], ds[3]);
        }
        a = seg_sum<SEG>(a);
        b = seg_sum<SEG>(b);
        if (dx_rowsum) xsum = seg_sum<SEG>(xsum);
        if ((lane & (SEG - 1)) == 0 && idx0 < n4) {
            part[(wid * cpg + cl) * 3 + 0] += a;
            part[(wid * cpg + cl) * 3 + 1] += b;
            part[(wid * cpg + cl) * 3 + 2] += xsum;
        }
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f, myA = 0.f, myX = 0.f;
    for (int c = 0; c < cpg; ++c) {
        float A = 0.f, B = 0.f, X = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            A += part[(w * cpg + c) * 3]; B += part[(w * cpg + c) * 3 + 1]; X += part[(w * cpg + c) * 3 + 2];
        }
        const float ga = gamma[g * cpg + c];
        s1 += ga * A; s2 += ga * B;
        if (threadIdx.x == c) {
            dbeta_part[(size_t)s * C + g * cpg + c] = A;
            dgamma_part[(size_t)s * C + g * cpg + c] = B;
            myA = ga * A; myX = X;
        }
    }
    const float inv_n = 1.0f / (float)n;
    s1 *= inv_n; s2 *= inv_n;
    // sum over the map of this channel's dx, in closed form from the sums above (the gradient of a
    // per-(view, channel) bias added in front of this GroupNorm; excludes `addend`)
    if (dx_rowsum && threadIdx.x < cpg)
        dx_rowsum[(size_t)s * C + g * cpg + threadIdx.x] = r * (myA - (float)HW * s1 - s2 * myX);
    float4* o4 = reinterpret_cast<float4*>(dx + base);
    const float4* a4 = addend ? reinterpret_cast<const float4*>(addend + base) : nullptr;
    // dx into the dy registers, then the addend (all its loads back to back) into the x registers
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float ga = gam[i];
        dv[i].x = r * (dv[i].x * ga - (s1 + xv[i].x * s2));
        dv[i].y = r * (dv[i].y * ga - (s1 + xv[i].y * s2));
        dv[i].z = r * (dv[i].z * ga - (s1 + xv[i].z * s2));
        dv[i].w = r * (dv[i].w * ga - (s1 + xv[i].w * s2));
    }
    if (x2) {              // cat: the second-consumer gradients and dx live in two tensors like the input
        if (addend) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int idc = min((int)threadIdx.x + i * NT, n4 - 1);
                xv[i] = cat_ptr(addend, addend2, C1, C, s, g * cpg + (idc >> hwsh), HW)[idc & hwmask];
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) { dv[i].x += xv[i].x; dv[i].y += xv[i].y; dv[i].z += xv[i].z; dv[i].w += xv[i].w; }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + i * NT;
            if (idx < n4)
                const_cast<float4*>(cat_ptr(dx, dx2, C1, C, s, g * cpg + (idx >> hwsh), HW))[idx & hwmask] = dv[i];
        }
        return;
    }
    if (a4) {
#pragma unroll
        for (int i = 0; i < NV; ++i) xv[i] = a4[min((int)threadIdx.x + i * NT, n4 - 1)];
#pragma unroll
        for (int i = 0; i < NV; ++i) { dv[i].x += xv[i].x; dv[i].y += xv[i].y; dv[i].z += xv[i].z; dv[i].w += xv[i].w; }
    }
    if (addend2) {         // plain input: gradient of a third consumer of x (the decoder's skip connection)
        const float4* b4 = reinterpret_cast<const float4*>(addend2 + base);
#pragma unroll
        for (int i = 0; i < NV; ++i) xv[i] = b4[min((int)threadIdx.x + i * NT, n4 - 1)];
#pragma unroll
        for (int i = 0; i < NV; ++i) { dv[i].x += xv[i].x; dv[i].y += xv[i].y; dv[i].z += xv[i].z; dv[i].w += xv[i].w; }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * NT;
        if (idx < n4) o4[idx] = dv[i];
    }
}

// dx += addend for the two-kernel fallback path
__global__ void add_inplace_kernel(float4* __restrict__ y, const float4* __restrict__ a, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { float4 v = y[i]; const float4 t = a[i]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; y[i] = v; }
}

// Sum of n4 float4 by one wave (lane partial; caller wave_sums): 8 independent loads in flight per lane, fixed
// order.  A plain `for { v = p[i]; a += v }` loop compiles to load -> s_waitcnt vmcnt(0) -> add per iteration,
// i.e. one memory round trip per 1 KB of the row.
__device__ __forceinline__ float lane_sum_f4(const float4* __restrict__ p, int n4, int lane) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = lane;
    for (; i + 7 * 64 < n4; i += 8 * 64) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[i + 64 * j];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    if (i < n4) {                                    // tail: same 8 slots, clamped loads, masked adds
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[min(i + 64 * j, n4 - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += i + 64 * j < n4 ? (v[j].x + v[j].y) + (v[j].z + v[j].w) : 0.f;
    }
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

// out[row] = sum_j x[row][j]   (one wave per row; row length multiple of 4)
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                     int rows, int len) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)row * len);
    const float a = wave_sum(lane_sum_f4(x4, len >> 2, lane));
    if (lane == 0) out[row] = a;
}

// Gradients of the conv epilogue's bias[c] and view_bias[s][c] from dy[S][C][HW] in ONE launch:
//   dvb[s][c] = sum_p dy[s][c][p]      db[c] = sum_s dvb[s][c]
// One workgroup per channel; wave w handles views w, w+4, ...; fixed order -> deterministic.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db,
                                                        float* __restrict__ dvb, int S, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float tot = 0.f;
    const int n4 = HW >> 2;
    if (n4 <= 64) {                                  // short rows: the rows of 8 views in flight per wave
        for (int s = wid; s < S; s += 32) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = reinterpret_cast<const float4*>(dy + ((size_t)min(s + 4 * j, S - 1) * C + c) * HW)[min(lane, n4 - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = wave_sum(lane < n4 ? (v[j].x + v[j].y) + (v[j].z + v[j].w) : 0.f);
                if (s + 4 * j < S) {
                    if (lane == 0 && dvb) dvb[(size_t)(s + 4 * j) * C + c] = a;
                    tot += a;
                }
            }
        }
    } else {
        for (int s = wid; s < S; s += 4) {
            const float4* p = reinterpret_cast<const float4*>(dy + ((size_t)s * C + c) * HW);
            const float a = wave_sum(lane_sum_f4(p, n4, lane));
            if (lane == 0 && dvb) dvb[(size_t)s * C + c] = a;
            tot += a;
        }
    }
    if (lane == 0) red[wid] = tot;
    __syncthreads();
    if (threadIdx.x == 0 && db) db[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[b][c] = sum_s part[b][s][c]   (deterministic: fixed 4-way row split + fixed tree)
// block = 64 columns x 4 row groups; grid (ceil(C/64), batch)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                     int S, int C) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const float* p = part + (size_t)blockIdx.y * S * C;
    float a = 0.f;
    if (c < C) {                                     // 8 rows in flight per thread, fixed order
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = ry; s < S; s += 32) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = p[(size_t)min(s + 4 * j, S - 1) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += s + 4 * j < S ? t[j] : 0.f;
        }
        a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    red[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && c < C) out[(size_t)blockIdx.y * C + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// Many column sums in ONE launch (the GroupNorm weight / bias gradients of a whole backward pass): entry e of the
// device table = {part [batch][S][C], out [batch][C], S, C, batch, first 64-column block}; the block grid covers all
// entries' (64-column block, batch row) pairs.  Same arithmetic and summation order as colsum_kernel.
struct ColsumDesc {
    const float* part;
    float* out;
    long long S, C, batch, first_block;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(const ColsumDesc* __restrict__ desc, int n) {
    __shared__ float red[4][64];
    const long long vb = blockIdx.x;
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first_block <= vb) lo = mid; else hi = mid;
    }
    const ColsumDesc d = desc[lo];
    const int S = (int)d.S, C = (int)d.C;
    const int nbx = (C + 63) / 64;
    const int rel = (int)(vb - d.first_block);
    const int bx = rel % nbx, by = rel / nbx;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = bx * 64 + cx;
    const float* p = d.part + (size_t)by * S * C;
    float a = 0.f;
    if (c < C) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = ry; s < S; s += 32) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = p[(size_t)min(s + 4 * j, S - 1) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += s + 4 * j < S ? t[j] : 0.f;
        }
        a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    red[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && c < C) d.out[(size_t)by * C + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

template <int NV, int NT>
int launch_gn_fwd(const float* x, const float* x2, int C1, const float* gamma, const float* beta, float* y, float* mean,
                  float* rstd, int S, int C, int HW, int cpg, float eps, int silu, hipStream_t st) {
    hipLaunchKernelGGL((gn_fwd_kernel<NV, NT>), dim3(S * (C / cpg)), dim3(NT), 0, st, x, x2, C1, gamma, beta, y, mean,
                       rstd, C, HW, cpg, eps, silu);
    VF_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

int vf_gn_cat_fwd(const float* x, const float* x2, int C1, const float* gamma, const float* beta, float* y, float* mean,
                  float* rstd, int S, int C, int HW, int groups, float eps, int silu, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (S <= 0) return 0;
    // HW: a power of two >= 4 (square power-of-two maps; the kernels index channels by shifts)
    if (C % groups != 0 || HW < 4 || (HW & (HW - 1)) || (x2 && (C1 <= 0 || C1 >= C))) return (int)hipErrorInvalidValue;
    const int cpg = C / groups;
    const long n4 = (long)cpg * HW / 4;
#define VF_GN(NV, NT) return launch_gn_fwd<NV, NT>(x, x2, C1, gamma, beta, y, mean, rstd, S, C, HW, cpg, eps, silu, st)
    if (n4 <= 64) VF_GN(1, 64);
    if (n4 <= 256) VF_GN(1, 256);
    if (n4 <= 512) VF_GN(2, 256);
    if (n4 <= 1024) VF_GN(4, 256);
    if (n4 <= 2048) VF_GN(8, 256);
    if (n4 <= 3072) VF_GN(12, 256);
    if (n4 <= 4096) VF_GN(16, 256);
    if (n4 <= 6144) VF_GN(12, 512);
    if (n4 <= 8192) VF_GN(16, 512);
    if (n4 <= 16384) VF_GN(16, 1024);
    if (n4 <= 32768) VF_GN(32, 1024);
#undef VF_GN
    return (int)hipErrorInvalidValue;
}

int vf_gn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int S,
              int C, int HW, int groups, float eps, int silu, void* stream) {
    return vf_gn_cat_fwd(x, nullptr, C, gamma, beta, y, mean, rstd, S, C, HW, groups, eps, silu, stream);
}

// 1 when vf_gn_bwd fills `dx_rowsum` at this shape (the single-pass kernel; the two-kernel path does not)
int vf_gn_bwd_emits_rowsum(int C, int HW, int groups) {
    if (groups <= 0 || C % groups != 0 || (HW & 3)) return 0;
    const long n4g = (long)(C / groups) * HW / 4;
    return ((HW >= 256 && n4g <= 8192) || (HW == 64 && n4g <= 1024)) ? 1 : 0;
}

int vf_gn_cat_bwd(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* mean,
                  const float* rstd, const float* dy, const float* addend, const float* addend2, float* dx, float* dx2,
                  float* dgamma_part, float* dbeta_part, float* dx_rowsum, int S, int C, int HW, int groups, int silu,
                  void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (S <= 0) return 0;
    if (C % groups != 0 || HW < 4 || (HW & (HW - 1))) return (int)hipErrorInvalidValue;
    if (x2 && (C1 <= 0 || C1 >= C || !dx2 || (addend && !addend2) || !vf_gn_bwd_emits_rowsum(C, HW, groups)))
        return (int)hipErrorInvalidValue;                // cat inputs: single-pass kernels only
    const int cpg = C / groups;
    const int rows = S * C;
    const long n4g = (long)cpg * HW / 4;
    if (HW == 64 && n4g <= 1024) {       // 8x8 maps: 16-lane channel segments
#define VF_GNB16(NV)                                                                                       \
    {                                                                                                      \
        hipLaunchKernelGGL((gn_bwd_fused_kernel<NV, 256, 16>), dim3(S * groups), dim3(256), 4 * cpg * 3 * 4, st, x, x2, C1, dy, \
                           gamma, beta, mean, rstd, addend, addend2, dx, dx2, dgamma_part, dbeta_part, dx_rowsum, C, HW, cpg, silu); \
        VF_RETURN_LAST_ERROR();                                                                            \
    }
        if (n4g <= 256) VF_GNB16(1)
        if (n4g <= 512) VF_GNB16(2)
        VF_GNB16(4)
#undef VF_GNB16
    }
    if (HW >= 256 && n4g <= 8192) {
#define VF_GNB(NV, NT)                                                                                     \
    {                                                                                                      \
        hipLaunchKernelGGL((gn_bwd_fused_kernel<NV, NT>), dim3(S * groups), dim3(NT), (NT / 64) * cpg * 3 * 4, st, \
                           x, x2, C1, dy, gamma, beta, mean, rstd, addend, addend2, dx, dx2, dgamma_part, dbeta_part, \
                           dx_rowsum, C, HW, cpg, silu);                                                   \
        VF_RETURN_LAST_ERROR();                                                                            \
    }
        if (n4g <= 256) VF_GNB(1, 256)
        if (n4g <= 512) VF_GNB(2, 256)
        if (n4g <= 1024) VF_GNB(4, 256)
        if (n4g <= 2048) VF_GNB(8, 256)
        if (n4g <= 4096) VF_GNB(8, 512)
        if (n4g <= 6144) VF_GNB(12, 512)
        VF_GNB(8, 1024)
#undef VF_GNB
    }
    hipLaunchKernelGGL(gn_bwd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, dy, gamma, beta, mean, rstd,
                       dbeta_part, dgamma_part, rows, C, HW, cpg, silu);
    const int n4 = cpg * HW / 4;
    int chunks = (n4 + 1023) / 1024;  // >= 4 float4 per thread
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(gn_bwd_dx_kernel, dim3(S * groups, chunks), dim3(256), 0, st, x, dy, gamma, beta, mean,
                       rstd, dbeta_part, dgamma_part, dx, C, HW, cpg, silu);
    const size_t t4 = (size_t)S * C * HW / 4;
    if (addend)
        hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, st, (float4*)dx,
                           (const float4*)addend, t4);
    if (addend2)
        hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, st, (float4*)dx,
                           (const float4*)addend2, t4);
    VF_RETURN_LAST_ERROR();
}

int vf_gn_bwd(const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
              const float* dy, const float* addend, float* dx, float* dgamma_part, float* dbeta_part,
              float* dx_rowsum, int S, int C, int HW, int groups, int silu, void* stream) {
    return vf_gn_cat_bwd(x, nullptr, C, gamma, beta, mean, rstd, dy, addend, nullptr, dx, nullptr, dgamma_part,
                         dbeta_part, dx_rowsum, S, C, HW, groups, silu, stream);
}

int vf_rowsum(const float* x, float* out, int rows, int len, void* stream) {
    if (rows <= 0) return 0;
    if (len & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, out, rows, len);
    VF_RETURN_LAST_ERROR();
}

int vf_bias_grad(const float* dy, float* db, float* dvb, int S, int C, int HW, void* stream) {
    if (S <= 0 || C <= 0) return 0;
    if (HW & 3) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, db, dvb, S, C, HW);
    VF_RETURN_LAST_ERROR();
}

// desc: device int64 [n][6] rows {part, out, S, C, batch, first_block}; total_blocks = sum of ceil(C/64) * batch
int vf_colsum_multi(const void* desc, int n, long total_blocks, void* stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const ColsumDesc*)desc, n);
    VF_RETURN_LAST_ERROR();
}

int vf_colsum(const float* part, float* out, int batch, int S, int C, void* stream) {
    if (C <= 0 || batch <= 0) return 0;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, part, out, S,
                       C);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
