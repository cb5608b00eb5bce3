// fp32-accurate products on the bf16 matrix path (a measured basis for DESIGN.md section 7, not used by the product):
//   x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)   (3 x 8 mantissa bits)
//   a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)                       (6 products, fp32 accumulation)
// One 32x32x16 tile product per wave: (1) 8 x v_mfma_f32_32x32x2_f32, (2) 6 x v_mfma_f32_32x32x16_bf16 on the
// split operands, (3) 3 x (a1b1 + a1b2 + a2b1), (4) 1 x plain bf16 -- each compared with an fp64 host reference --
// and the time of (1), (2) and (2) including the split of the B operand on the VALU (A = weights: split at pack time).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16x3.hip -o tools/mfma_bf16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {          // two fp32 -> two bf16 (RNE), one instruction
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// split 8 fp32 (this lane's k = 8 (lane/32) .. +7 of one row / column) into three bf16x8
__device__ __forceinline__ void split8(const float* x, bf16x8& s1, bf16x8& s2, bf16x8& s3) {
    u32x4 p1, p2, p3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        p1[i] = pk_bf16(a, b);
        const float ra = a - bf_lo(p1[i]), rb = b - bf_hi(p1[i]);
        p2[i] = pk_bf16(ra, rb);
        p3[i] = pk_bf16(ra - bf_lo(p2[i]), rb - bf_hi(p2[i]));
    }
    s1 = __builtin_bit_cast(bf16x8, p1); s2 = __builtin_bit_cast(bf16x8, p2); s3 = __builtin_bit_cast(bf16x8, p3);
}

// A [32][16], B [16][32] row-major fp32; out [4 variants][32][32]
__global__ void check(const float* A, const float* B, float* out) {
    const int l = threadIdx.x, li = l & 31, lh = l >> 5;
    f32x16 d0 = (f32x16){0}, d1 = d0, d2 = d0, d3 = d0;
    for (int k = 0; k < 16; k += 2) d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * 16 + k + lh], B[(k + lh) * 32 + li], d0, 0, 0, 0);
    float ax[8], bx[8];
    for (int i = 0; i < 8; ++i) { ax[i] = A[li * 16 + 8 * lh + i]; bx[i] = B[(8 * lh + i) * 32 + li]; }
    bf16x8 a1, a2, a3, b1, b2, b3;
    split8(ax, a1, a2, a3); split8(bx, b1, b2, b3);
    // smallest terms first
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, d1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, d1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, d1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, d1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, d1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d1, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, d2, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, d2, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d3, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[0 * 1024 + row * 32 + li] = d0[r]; out[1 * 1024 + row * 32 + li] = d1[r];
        out[2 * 1024 + row * 32 + li] = d2[r]; out[3 * 1024 + row * 32 + li] = d3[r];
    }
}

// timing: MODE 0: 8 fp32 MFMAs per K=16 block; 1: 6 bf16 MFMAs, operands pre-split; 2: 6 bf16 MFMAs + split of B (8 values/lane)
template <int MODE>
__global__ __launch_bounds__(512, 2) void timeit(const float* src, float* out, int iters) {
    const int l = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
    float bx[8];
    for (int i = 0; i < 8; ++i) bx[i] = src[l * 8 + i];
    float a = src[l], b = src[64 + l];
    bf16x8 a1, a2, a3, b1, b2, b3;
    split8(bx, a1, a2, a3); split8(bx, b1, b2, b3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {                       // four K=16 blocks on four accumulators per iteration
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s], 0, 0, 0);
            } else {
                if (MODE == 2) { bx[s] += 1.0f; split8(bx, b1, b2, b3); }      // fresh B values every block
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[s], 0, 0, 0);
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[s], 0, 0, 0);
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[s], 0, 0, 0);
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[s], 0, 0, 0);
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[s], 0, 0, 0);
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[s], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const float* src, float* d, const char* name) {
    const int iters = 5000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((timeit<MODE>), dim3(256), dim3(512), 0, 0, src, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 256.0 * 8 * iters * 4 * 2.0 * 32 * 32 * 16;      // fp32-equivalent flops
    printf("%-58s %7.3f ms  %6.1f TFLOP/s fp32-equivalent\n", name, best, flop / best / 1e9);
}

int main() {
    float hA[512], hB[512], *dA, *dB, *dO, hO[4096];
    srand(3);
    for (int i = 0; i < 512; ++i) { hA[i] = (float)rand() / RAND_MAX * 2.f - 1.f; hB[i] = (float)rand() / RAND_MAX * 2.f - 1.f; }
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dO, 16384);
    (void)hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    (void)hipMemcpy(hO, dO, 16384, hipMemcpyDeviceToHost);
    const char* nm[4] = {"8 x fp32 MFMA", "6 x bf16 MFMA (3-way split)", "3 x bf16 MFMA (2-way split)", "1 x bf16 MFMA"};
    for (int v = 0; v < 4; ++v) {
        double emax = 0, rmax = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double r = 0; for (int k = 0; k < 16; ++k) r += (double)hA[i * 16 + k] * hB[k * 32 + j];
            emax = fmax(emax, fabs(hO[v * 1024 + i * 32 + j] - r)); rmax = fmax(rmax, fabs(r));
        }
        printf("%-30s max abs error vs fp64 %.3e (max |ref| %.2f)\n", nm[v], emax, rmax);
    }
    float *src, *d;
    (void)hipMalloc(&src, 4096 * 4); (void)hipMalloc(&d, 256 * 512 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 17) * 0.01f;
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(src, d, "fp32: 8 x v_mfma_f32_32x32x2_f32 per K=16 block");
    run<1>(src, d, "bf16x3: 6 x v_mfma_f32_32x32x16_bf16, operands pre-split");
    run<2>(src, d, "bf16x3: 6 MFMAs + 3-way split of B on the VALU (8 values/lane)");
    return 0;
}
