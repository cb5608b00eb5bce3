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

extern "C" {

int vf_bgemm(const float* A, const float* B, float* C, const float* bias, int batch, int M, int N, int K,
             long sAb, long sAm, long sAk, long sBb, long sBk, long sBn, long sCb, long sCm, long sCn,
             float alpha, float beta, void* stream) {
    if (batch <= 0 || M <= 0 || N <= 0) return 0;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.sAb = sAb; g.sAm = sAm; g.sAk = sAk; g.sBb = sBb; g.sBk = sBk; g.sBn = sBn;
    g.sCb = sCb; g.sCm = sCm; g.sCn = sCn; g.alpha = alpha; g.beta = beta;
    hipLaunchKernelGGL(bgemm_kernel, dim3((N + BN - 1) / BN, (M + BM - 1) / BM, batch), dim3(256), 0,
                       (hipStream_t)stream, g);
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

}  // extern "C"
