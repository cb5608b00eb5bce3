// Does the fp32 MFMA SHAPE change what the chip sustains?  (MI355X_MICROARCH.md, DVFS give-back item 7: for bf16 the
// 16x16x32 shape held a higher clock than 32x32x16 under load.)  Same MACs per wave-iteration for both shapes:
//   v_mfma_f32_32x32x2_f32 : 2048 MACs / 64 cycles     v_mfma_f32_16x16x4_f32 : 1024 MACs / 32 cycles
// Variants: operands in registers (varying per lane and per step) or re-read from LDS before every MFMA group, as the
// conv kernels do.  Each variant runs ~2 s back to back on random data; reports TFLOP/s and the in-kernel shader clock
// (s_memtime / s_memrealtime, 100 MHz reference).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.hip -o tools/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SHAPE 0: 32x32x2, 8 accumulator blocks (128 regs); SHAPE 1: 16x16x4, 32 accumulator blocks (128 regs).
// LDS = 0: operands from registers; 1: every operand group re-read from LDS (b128 for A, 4 x b32 for B).
template <int SHAPE, int LDS>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* out, long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float la[16 * 64 * 8];      // [k 16][row 64][8]
    __shared__ __attribute__((aligned(16))) float lb[16 * 8 * 64];      // [k 16][8][col 64]
    for (int i = threadIdx.x; i < 16 * 64 * 8; i += 512) { la[i] = src[i]; lb[i] = src[8192 + i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float ar[8], br[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ar[i] = src[(lane * 8 + i + wid * 97) & 8191]; br[i] = src[8192 + ((lane * 8 + i + wid * 131) & 8191)]; }
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x16){0};
        const int li = lane & 31, lh = lane >> 5;
        for (int it = 0; it < iters; ++it) {
            const int tog = (it & 1) << 12;              // alternate halves of the images: keeps the reads in the loop
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {                 // one "slice": 4 MFMAs (K = 8) on accumulator kk
                float a4[4], b4[4];
                if (LDS) {
                    const float4 av = *reinterpret_cast<const float4*>(la + ((((kk + 8 * (wid >> 2)) * 64 + (wid & 1) * 32 + li) * 8 + 4 * lh) ^ tog));
                    a4[0] = av.x; a4[1] = av.y; a4[2] = av.z; a4[3] = av.w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) b4[e] = lb[((kk + 8 * (wid >> 2)) * 512 + (4 * lh + e) * 64 + ((wid >> 1) & 1) * 32 + li) ^ tog];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a4[e] = ar[(kk + e) & 7]; b4[e] = br[(kk + 2 * e) & 7]; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[kk], 0, 0, 0);
            }
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0};
        const int l16 = lane & 15, lq = lane >> 4;
        for (int it = 0; it < iters; ++it) {
            const int tog = (it & 1) << 12;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {                 // one slice: a 32x32 block = 2x2 sub-blocks, K = 8 = 2 steps of 4
                float a2[2][2], b2[2][2];                    // [row half / col half][k step]
                if (LDS) {
                    // lane (l16, lq) needs k = lq and lq + 4 of rows l16 and 16 + l16: two float2-like pairs; read as
                    // scalars from the same [k][row][8] image (8 b32 reads per slice, like the 32x32 variant's 1 b128 + 4 b32)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            a2[h][st] = la[(((kk + 8 * (wid >> 2)) * 64 + (wid & 1) * 32 + 16 * h + l16) * 8 + lq + 4 * st) ^ tog];
                            b2[h][st] = lb[((kk + 8 * (wid >> 2)) * 512 + (lq + 4 * st) * 64 + ((wid >> 1) & 1) * 32 + 16 * h + l16) ^ tog];
                        }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int st = 0; st < 2; ++st) { a2[h][st] = ar[(kk + 2 * h + st) & 7]; b2[h][st] = br[(kk + h + 3 * st) & 7]; }
                }
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch)
                            acc[kk * 4 + rh * 2 + ch] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[rh][st], b2[ch][st], acc[kk * 4 + rh * 2 + ch], 0, 0, 0);
            }
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int LDS>
void run(const float* src, float* d, long long* clk) {
    const int blocks = 256, iters = 20000;                  // one 8-wave workgroup per CU, like the Winograd kernels
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    double best = 0, clock = 0;
    for (int rep = 0; rep < 50; ++rep) {                     // ~2+ s in total; report the LAST launch (steady state)
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(blocks), dim3(512), 0, 0, src, d, clk, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long h[2 * 256];
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double c = 0;
        for (int b = 0; b < blocks; ++b) c += (double)h[2 * b] / (double)h[2 * b + 1];
        clock = c / blocks * 0.1;                            // GHz (memrealtime ticks at 100 MHz)
        best = (double)blocks * 8 * iters * 8 * 4 * 4096.0 / ms / 1e9;
    }
    printf("shape %s  operands %s : %8.2f ms  %6.1f TFLOP/s  in-kernel clock %.3f GHz\n", SHAPE ? "16x16x4" : "32x32x2",
           LDS ? "LDS      " : "registers", ms, best, clock);
}

int main() {
    float *src, *d; long long* clk;
    (void)hipMalloc(&src, 16384 * 4); (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&clk, 2 * 256 * 8);
    float h[16384];
    srand(1);
    for (int i = 0; i < 16384; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>(src, d, clk); run<1, 0>(src, d, clk); run<0, 1>(src, d, clk); run<1, 1>(src, d, clk);
    run<0, 1>(src, d, clk); run<1, 1>(src, d, clk);
    return 0;
}
