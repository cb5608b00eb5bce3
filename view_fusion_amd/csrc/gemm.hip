// Strided batched fp32 GEMM on v_mfma_f32_32x32x2_f32 plus row softmax.
//
// Serves the contractions that are not convolutions:
//   * self-attention  Q^T K / sqrt(C), P V^T and their four backward products
//     (reference model/unet.py:267-274, materialised scores like the reference)
//   * nn.Linear of the noise/angle embedding MLP and FeatureWiseAffine (unet.py:27-32,165)
//     and their weight / input gradients.
// C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][k][n] (+ bias[n]) (+ beta * C[b][m][n])
// with arbitrary element strides, so transposed operands need no copies.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32, LDT = 65;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    long sAb, sAm, sAk, sBb, sBk, sBn, sCb, sCm, sCn;
    float alpha, beta;
    int xcd_batches;      // bgemm_v2: number of whole groups of eight batch items dealt one item per XCD (0: plain order)
};

__global__ __launch_bounds__(256) void bgemm_kernel(GemmArgs g) {
    __shared__ float Al[BK * LDT];
    __shared__ float Bl[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1, wn = wid >> 1, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, b = blockIdx.z;
    const float* A = g.A + (long)b * g.sAb;
    const float* B = g.B + (long)b * g.sBb;
    const bool a_m_fast = g.sAm == 1, b_n_fast = g.sBn == 1;

    f32x16 acc = {0};
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int m, k;
            if (a_m_fast) { m = tid & 63; k = (tid >> 6) + 4 * i; }
            else          { k = tid & 31; m = (tid >> 5) + 8 * i; }
            float v = 0.f;
            if (m0 + m < g.M && k0 + k < g.K) v = A[(long)(m0 + m) * g.sAm + (long)(k0 + k) * g.sAk];
            Al[k * LDT + m] = v;
            int n, kb;
            if (b_n_fast) { n = tid & 63; kb = (tid >> 6) + 4 * i; }
            else          { kb = tid & 31; n = (tid >> 5) + 8 * i; }
            float u = 0.f;
            if (n0 + n < g.N && k0 + kb < g.K) u = B[(long)(k0 + kb) * g.sBk + (long)(n0 + n) * g.sBn];
            Bl[kb * LDT + n] = u;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const float av = Al[(2 * s + lh) * LDT + wm * 32 + li];
            const float bv = Bl[(2 * s + lh) * LDT + wn * 32 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    float* C = g.C + (long)b * g.sCb;
    const int n = n0 + wn * 32 + li;
    if (n < g.N) {
        const float bn = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < g.M) {
                float* p = C + (long)m * g.sCm + (long)n * g.sCn;
                float v = g.alpha * acc[r] + bn;
                if (g.beta != 0.f) v += g.beta * *p;
                *p = v;
            }
        }
    }
}


// Fast path: both operands have a unit stride (along k or along m/n), all extents and the other
// strides are multiples of 4 floats.  16-byte global loads, LDS image oriented like the global
// one (so LDS writes are b128 too), next K-chunk prefetched in registers while the current one is
// multiplied, fragment reads one chunk-group ahead of the MFMAs.
// k order inside a group of 8: MFMA step e pairs k = e (lane half 0) with k = 4 + e (half 1), so a
// k-contiguous operand is fetched with one ds_read_b128 per 4 MFMAs.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void bgemm_v2_kernel(GemmArgs g) {
    constexpr int KS_ = BK + 4;                      // [m][k] image row stride: 4*odd -> conflict-free b128
    __shared__ __attribute__((aligned(16))) float Al[A_KC ? BM * KS_ : BK * BM];
    __shared__ __attribute__((aligned(16))) float Bl[B_KC ? BN * KS_ : BK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1, wn = wid >> 1, li = lane & 31, lh = lane >> 5;
    // Batched products (attention backward: 12-16 tiles per view, 96 views): consecutive workgroups go to different XCDs,
    // so in plain order the tiles of ONE view run on all eight XCDs and every XCD's L2 fetches that view's operands from
    // HBM (PMC: 110 MB per launch for 63 MB of operands + result).  Whole groups of eight views are dealt one view per
    // XCD (round 5); the views past the last whole group keep the plain order.
    int bx = blockIdx.x, by = blockIdx.y, b = blockIdx.z;
    if (g.xcd_batches) {
        const int nt = gridDim.x * gridDim.y, i = bx + gridDim.x * (by + gridDim.y * b);
        if (i < (g.xcd_batches << 3) * nt) {
            const int j = i >> 3, tile = j % nt;
            b = ((j / nt) << 3) + (i & 7);
            bx = tile % gridDim.x; by = tile / gridDim.x;
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    const float* A = g.A + (long)b * g.sAb;
    const float* B = g.B + (long)b * g.sBb;

    float4 ar[2], br[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (A_KC) {
                const int m = (tid >> 3) + 32 * i, kq = tid & 7;
                if (m0 + m < g.M && k0 + 4 * kq < g.K)
                    v = *reinterpret_cast<const float4*>(A + (long)(m0 + m) * g.sAm + k0 + 4 * kq);
            } else {
                const int k = (tid >> 4) + 16 * i, mq = tid & 15;
                if (k0 + k < g.K && m0 + 4 * mq < g.M)
                    v = *reinterpret_cast<const float4*>(A + (long)(k0 + k) * g.sAk + m0 + 4 * mq);
            }
            ar[i] = v;
            float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
            if (B_KC) {
                const int n = (tid >> 3) + 32 * i, kq = tid & 7;
                if (n0 + n < g.N && k0 + 4 * kq < g.K)
                    u = *reinterpret_cast<const float4*>(B + (long)(n0 + n) * g.sBn + k0 + 4 * kq);
            } else {
                const int k = (tid >> 4) + 16 * i, nq = tid & 15;
                if (k0 + k < g.K && n0 + 4 * nq < g.N)
                    u = *reinterpret_cast<const float4*>(B + (long)(k0 + k) * g.sBk + n0 + 4 * nq);
            }
            br[i] = u;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (A_KC) *reinterpret_cast<float4*>(Al + ((tid >> 3) + 32 * i) * KS_ + 4 * (tid & 7)) = ar[i];
            else      *reinterpret_cast<float4*>(Al + ((tid >> 4) + 16 * i) * BM + 4 * (tid & 15)) = ar[i];
            if (B_KC) *reinterpret_cast<float4*>(Bl + ((tid >> 3) + 32 * i) * KS_ + 4 * (tid & 7)) = br[i];
            else      *reinterpret_cast<float4*>(Bl + ((tid >> 4) + 16 * i) * BN + 4 * (tid & 15)) = br[i];
        }
    };
    auto frag_a = [&](int grp) -> float4 {
        if (A_KC) return *reinterpret_cast<const float4*>(Al + (wm * 32 + li) * KS_ + 8 * grp + 4 * lh);
        const float* p = Al + (8 * grp + 4 * lh) * BM + wm * 32 + li;
        return make_float4(p[0], p[BM], p[2 * BM], p[3 * BM]);
    };
    auto frag_b = [&](int grp) -> float4 {
        if (B_KC) return *reinterpret_cast<const float4*>(Bl + (wn * 32 + li) * KS_ + 8 * grp + 4 * lh);
        const float* p = Bl + (8 * grp + 4 * lh) * BN + wn * 32 + li;
        return make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);
    };

    f32x16 acc = {0};
    load(0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();
        store();
        __syncthreads();
        if (k0 + BK < g.K) load(k0 + BK);
        float4 a_cur = frag_a(0), b_cur = frag_b(0);
#pragma unroll
        for (int grp = 0; grp < BK / 8; ++grp) {
            float4 a_nxt = a_cur, b_nxt = b_cur;
            if (grp + 1 < BK / 8) { a_nxt = frag_a(grp + 1); b_nxt = frag_b(grp + 1); }
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt; b_cur = b_nxt;
        }
    }
    float* C = g.C + (long)b * g.sCb;
    const int n = n0 + wn * 32 + li;
    if (n < g.N) {
        const float bn = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < g.M) {
                float* p = C + (long)m * g.sCm + n;
                float v = g.alpha * acc[r] + bn;
                if (g.beta != 0.f) v += g.beta * *p;
                *p = v;
            }
        }
    }
}

// One wave per row; cols <= 64*MAXV.
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (size_t)row * cols;
    float v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < cols ? xr[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        v[i] = expf(v[i] - mx);
        sum += v[i];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float* yr = y + (size_t)row * cols;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < cols) yr[c] = v[i] * inv;
    }
}

// dx = y * (dy - sum(y*dy))
template <int MAXV>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* yr = y + (size_t)row * cols;
    const float* dr = dy + (size_t)row * cols;
    float yv[MAXV], dv[MAXV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        yv[i] = c < cols ? yr[c] : 0.f;
        dv[i] = c < cols ? dr[c] : 0.f;
        dot += yv[i] * dv[i];
    }
    dot = wave_sum(dot);
    float* xr = dx + (size_t)row * cols;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < cols) xr[c] = yv[i] * (dv[i] - dot);
    }
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// Grouped time-embedding affine (FeatureWiseAffine, reference model/unet.py:160-177): the UNet's 30 residual
// blocks each own a Linear(K -> C_g) applied to the SAME (S, K) embedding.  All of them are evaluated in one
// launch (and their weight / input gradients in two) instead of 30 (60) latency-bound tiny GEMMs.
// Descriptor rows (device int64): {W_g, b_g, C_g, out_off_g (floats), c_off_g (first global channel)};
// outputs / output gradients of group g live at  out + out_off_g  as a contiguous (S, C_g) matrix.
namespace {

struct TADesc {
    const float* w;
    const float* b;
    long long C, out_off, c_off;
};

struct TAGrad {
    float* dw;
    float* db;
};

__device__ __forceinline__ int ta_group(const TADesc* __restrict__ d, int ng, int cg) {
    int lo = 0, hi = ng;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (d[mid].c_off <= cg) lo = mid; else hi = mid;
    }
    return lo;
}

// out_g[s][c] = b_g[c] + sum_k emb[s][k] W_g[c][k]        one thread per output, c fastest
__global__ __launch_bounds__(256) void time_affine_fwd_kernel(const TADesc* __restrict__ desc, int ng,
                                                              const float* __restrict__ emb, float* __restrict__ out,
                                                              int S, int K, int CT) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= CT * S) return;
    const int cg = idx % CT, s = idx / CT;
    const TADesc d = desc[ta_group(desc, ng, cg)];
    const int c = cg - (int)d.c_off;
    const float4* e4 = reinterpret_cast<const float4*>(emb + (size_t)s * K);
    const float4* w4 = reinterpret_cast<const float4*>(d.w + (size_t)c * K);
    float a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < K / 4; k += 2) {
        const float4 e0 = e4[k], w0 = w4[k], e1 = e4[k + 1], w1 = w4[k + 1];
        a0 += (e0.x * w0.x + e0.y * w0.y) + (e0.z * w0.z + e0.w * w0.w);
        a1 += (e1.x * w1.x + e1.y * w1.y) + (e1.z * w1.z + e1.w * w1.w);
    }
    out[d.out_off + (size_t)s * d.C + c] = (a0 + a1) + (d.b ? d.b[c] : 0.f);
}

// dW[cg][k] = sum_s dE_g[s][c] emb[s][k],  db[cg] = sum_s dE_g[s][c]      one thread per (cg, k), k fastest
__global__ __launch_bounds__(256) void time_affine_bwd_w_kernel(const TADesc* __restrict__ desc, int ng,
                                                                const float* __restrict__ emb,
                                                                const float* __restrict__ de, float* __restrict__ dw,
                                                                float* __restrict__ db,
                                                                const TAGrad* __restrict__ gdst, int S, int K, int CT) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= CT * K) return;
    const int k = idx % K, cg = idx / K;
    const int gi = ta_group(desc, ng, cg);
    const TADesc d = desc[gi];
    const int c = cg - (int)d.c_off;
    if (gdst) {                                  // per-layer destinations instead of the flat [CT][K] / [CT] buffers
        dw = gdst[gi].dw - (size_t)d.c_off * K;
        db = gdst[gi].db - d.c_off;
    }
    const float* g = de + d.out_off + c;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, bsum[4] = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 3 < S; s += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gv = g[(size_t)(s + j) * d.C];
            a[j] += gv * emb[(size_t)(s + j) * K + k];
            bsum[j] += gv;
        }
    }
    for (; s < S; ++s) {
        const float gv = g[(size_t)s * d.C];
        a[0] += gv * emb[(size_t)s * K + k];
        bsum[0] += gv;
    }
    dw[(size_t)cg * K + k] = (a[0] + a[1]) + (a[2] + a[3]);
    if (k == 0) db[cg] = (bsum[0] + bsum[1]) + (bsum[2] + bsum[3]);
}

// demb[s][k] = sum_g sum_c dE_g[s][c] W_g[c][k].  Stage 1: workgroup (s, split): 256 threads = K lanes x (256/K)
// channel strides over every TA_NSPLIT-th stride of channels, partial sums combined through LDS in a fixed order
// into part[split][s][k]; stage 2 adds the TA_NSPLIT partials in order.
constexpr int TA_NSPLIT = 16;

__global__ __launch_bounds__(256) void time_affine_bwd_x_kernel(const TADesc* __restrict__ desc, int ng,
                                                                const float* __restrict__ de,
                                                                float* __restrict__ partial, int S, int K) {
    __shared__ float part[256];
    const int s = blockIdx.x, sp = blockIdx.y, kk = threadIdx.x % K, p = threadIdx.x / K, np = 256 / K;
    float a0 = 0.f, a1 = 0.f;
    for (int g = 0; g < ng; ++g) {
        const TADesc d = desc[g];
        const float* gr = de + d.out_off + (size_t)s * d.C;
        const int step = np * TA_NSPLIT;
        int c = sp * np + p;
        for (; c + step < (int)d.C; c += 2 * step) {
            a0 += gr[c] * d.w[(size_t)c * K + kk];
            a1 += gr[c + step] * d.w[(size_t)(c + step) * K + kk];
        }
        if (c < (int)d.C) a0 += gr[c] * d.w[(size_t)c * K + kk];
    }
    part[threadIdx.x] = a0 + a1;
    __syncthreads();
    if (p == 0) {
        float t = 0.f;
        for (int q = 0; q < np; ++q) t += part[q * K + kk];
        partial[((size_t)sp * S + s) * K + kk] = t;
    }
}

__global__ __launch_bounds__(256) void time_affine_bwd_x_sum_kernel(const float* __restrict__ partial,
                                                                    float* __restrict__ demb, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t[TA_NSPLIT];
#pragma unroll
    for (int q = 0; q < TA_NSPLIT; ++q) t[q] = partial[(size_t)q * n + i];
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < TA_NSPLIT; ++q) a += t[q];
    demb[i] = a;
}

}  // namespace

extern "C" {

int vf_bgemm(const float* A, const float* B, float* C, const float* bias, int batch, int M, int N, int K,
             long sAb, long sAm, long sAk, long sBb, long sBk, long sBn, long sCb, long sCm, long sCn,
             float alpha, float beta, void* stream) {
    if (batch <= 0 || M <= 0 || N <= 0) return 0;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.sAb = sAb; g.sAm = sAm; g.sAk = sAk; g.sBb = sBb; g.sBk = sBk; g.sBn = sBn;
    g.sCb = sCb; g.sCm = sCm; g.sCn = sCn; g.alpha = alpha; g.beta = beta;
    const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
    hipStream_t st = (hipStream_t)stream;
    static const bool xcd = !(getenv("VF_GEMM_XCD") && getenv("VF_GEMM_XCD")[0] == '0');     // (tuning aid)
    g.xcd_batches = xcd ? batch / 8 : 0;
    auto mul4 = [](long v) { return (v & 3) == 0; };
    const bool a_kc = sAk == 1, a_mc = sAm == 1, b_kc = sBk == 1, b_nc = sBn == 1;
    const bool fast = (a_kc || a_mc) && (b_kc || b_nc) && sCn == 1 && mul4(K) && mul4(sAb) && mul4(sBb) &&
                      (a_kc ? mul4(sAm) : (mul4(sAk) && mul4(M))) && (b_kc ? mul4(sBn) : (mul4(sBk) && mul4(N))) &&
                      ((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(B)) & 15) == 0;
    if (fast) {
        if (a_kc && b_kc) hipLaunchKernelGGL((bgemm_v2_kernel<true, true>), grid, dim3(256), 0, st, g);
        else if (a_kc) hipLaunchKernelGGL((bgemm_v2_kernel<true, false>), grid, dim3(256), 0, st, g);
        else if (b_kc) hipLaunchKernelGGL((bgemm_v2_kernel<false, true>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((bgemm_v2_kernel<false, false>), grid, dim3(256), 0, st, g);
        VF_RETURN_LAST_ERROR();
    }
    hipLaunchKernelGGL(bgemm_kernel, grid, dim3(256), 0, st, g);
    VF_RETURN_LAST_ERROR();
}

int vf_softmax_fwd(const float* x, float* y, int rows, int cols, void* stream) {
    if (rows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((rows + 3) / 4), blk(256);
    if (cols <= 64) hipLaunchKernelGGL(softmax_fwd_kernel<1>, grid, blk, 0, st, x, y, rows, cols);
    else if (cols <= 256) hipLaunchKernelGGL(softmax_fwd_kernel<4>, grid, blk, 0, st, x, y, rows, cols);
    else if (cols <= 1024) hipLaunchKernelGGL(softmax_fwd_kernel<16>, grid, blk, 0, st, x, y, rows, cols);
    else if (cols <= 4096) hipLaunchKernelGGL(softmax_fwd_kernel<64>, grid, blk, 0, st, x, y, rows, cols);
    else return (int)hipErrorInvalidValue;
    VF_RETURN_LAST_ERROR();
}

int vf_softmax_bwd(const float* y, const float* dy, float* dx, int rows, int cols, void* stream) {
    if (rows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((rows + 3) / 4), blk(256);
    if (cols <= 64) hipLaunchKernelGGL(softmax_bwd_kernel<1>, grid, blk, 0, st, y, dy, dx, rows, cols);
    else if (cols <= 256) hipLaunchKernelGGL(softmax_bwd_kernel<4>, grid, blk, 0, st, y, dy, dx, rows, cols);
    else if (cols <= 1024) hipLaunchKernelGGL(softmax_bwd_kernel<16>, grid, blk, 0, st, y, dy, dx, rows, cols);
    else if (cols <= 4096) hipLaunchKernelGGL(softmax_bwd_kernel<64>, grid, blk, 0, st, y, dy, dx, rows, cols);
    else return (int)hipErrorInvalidValue;
    VF_RETURN_LAST_ERROR();
}


// desc: device int64 [ngroups][5] rows {W_g, b_g, C_g, out_off_g, c_off_g}; CT = sum of C_g; K = embedding width
// (multiple of 8, <= 256).  out / de: flat buffers holding the (S, C_g) matrices of all groups at out_off_g.
int vf_time_affine_fwd(const void* desc, int ngroups, const float* emb, float* out, int S, int K, int CT,
                       void* stream) {
    if (ngroups <= 0 || S <= 0) return 0;
    if (K % 8 != 0 || K > 256 || 256 % K != 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(time_affine_fwd_kernel, dim3((CT * S + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const TADesc*)desc, ngroups, emb, out, S, K, CT);
    VF_RETURN_LAST_ERROR();
}

long vf_time_affine_ws_floats(int S, int K) { return (long)TA_NSPLIT * S * K; }

// dw: [CT][K] (group g's weight gradient = rows c_off_g ..), db: [CT]; or, with gdst != NULL (ngroups rows of
// {float* dW_g [C_g][K], float* db_g [C_g]}), one destination per layer and dw/db unused; demb: [S][K] (or NULL;
// needs ws of vf_time_affine_ws_floats(S, K) floats)
int vf_time_affine_bwd(const void* desc, int ngroups, const float* emb, const float* de, float* dw, float* db,
                       const void* gdst, float* demb, float* ws, int S, int K, int CT, void* stream) {
    if (ngroups <= 0 || S <= 0) return 0;
    if (K % 8 != 0 || K > 256 || 256 % K != 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(time_affine_bwd_w_kernel, dim3((CT * K + 255) / 256), dim3(256), 0, st, (const TADesc*)desc,
                       ngroups, emb, de, dw, db, (const TAGrad*)gdst, S, K, CT);
    if (demb) {
        if (!ws) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(time_affine_bwd_x_kernel, dim3(S, TA_NSPLIT), dim3(256), 0, st, (const TADesc*)desc,
                           ngroups, de, ws, S, K);
        hipLaunchKernelGGL(time_affine_bwd_x_sum_kernel, dim3((S * K + 255) / 256), dim3(256), 0, st, ws, demb, S * K);
    }
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
