// Can VALU work of one wave run under the fp32 MFMAs of its SIMD PARTNER when the two are out of phase?
// (tools/mfma_valu.hip: with both waves of a SIMD running the same fine-grained MFMA / VALU interleave, every VALU
// instruction adds its full issue time.)  One 512-thread workgroup per CU; per iteration every wave issues 4 dependent
// v_mfma_f32_32x32x2_f32 and NV independent v_add_f32 as two BLOCKS; waves 4-7 (the SIMD partners of waves 0-3) run the
// blocks in the opposite order when STAG = 1, so one partner's VALU block coincides with the other's MFMA block.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_stagger.hip -o tools/mfma_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int STAG>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const bool late = STAG && __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) >= 4;
    f32x16 acc = (f32x16){0};
    float a = src[lane], b = src[64 + lane];
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = src[128 + i * 64 + lane];
#define MF4 { _Pragma("unroll") for (int e = 0; e < 4; ++e) { __builtin_amdgcn_sched_barrier(0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); } __builtin_amdgcn_sched_barrier(0); }
#define VA  { _Pragma("unroll") for (int q = 0; q < NV; ++q) v[q & 31] = v[q & 31] + v[(q + 1) & 31]; __builtin_amdgcn_sched_barrier(0); }
    if (late) { for (int it = 0; it < iters; ++it) { VA; MF4; } }
    else      { for (int it = 0; it < iters; ++it) { MF4; VA; } }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV, int STAG>
void run(const float* src, float* d) {
    const int blocks = 256, iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, STAG>), dim3(blocks), dim3(512), 0, 0, src, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%2d v_add per 4 MFMAs, %s: %8.3f ms  (%.1f ns per wave-pair block of 8 MFMAs; 8 MFMAs alone = 216 ns)\n", NV,
           STAG ? "partners out of phase" : "partners in phase    ", best, best * 1e6 / iters);
}

int main() {
    float *src, *d;
    (void)hipMalloc(&src, 4096 * 4); (void)hipMalloc(&d, 256 * 512 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 17) * 0.01f;
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>(src, d); run<16, 0>(src, d); run<16, 1>(src, d); run<32, 0>(src, d); run<32, 1>(src, d); run<48, 0>(src, d); run<48, 1>(src, d);
    return 0;
}
