// The slab sums behind the split-K weight-gradient kernels, shared by conv.hip (direct / 1x1 kernels: one follow-up launch
// per layer, or -- round 6 -- a row of the deferred multi launch) and winograd44.hip (the table-driven launch that sums the
// slabs of ALL layers of a backward pass).  Fixed summation order, no float atomics: bit-reproducible.
#pragma once
#include "common.h"
#include <cstring>

// One row of the deferred slab-sum launch (vf_wino44_reduce_multi): 9 x int64.
//   nt == 0: a Winograd F(4x4) weight gradient (wino44_reduce_body: slabs of 9 transformed taps, bias sums behind them)
//   nt == 1 | 9: a direct / 1x1 weight gradient (wgrad_reduce_body: slabs [slab][tap nt][CoutP][CinQ]); bsum / db / db2 unused
struct W44Red {
    const float* ws;
    float* dw;
    const float* bsum;
    float* db;
    float* db2;
    int Cout, Cin, CoutP, CinQ, nslab, nmain, first, nt;
};
static_assert(sizeof(W44Red) == 72, "descriptor row = 9 x int64");

// dw[co][ci][tap] = sum_slab ws[slab][tap][co][ci]
// block = 64 consecutive (tap,co,ci) outputs x 4 slab groups (fixed 4-way split + fixed tree: deterministic), so that
// small weight tensors still spread over the chip.  bid = block index within this tensor.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, int nslab, int NT,
                                                  int Cout, int Cin, int CoutP, int CinQ, int bid) {
    __shared__ float red[4][64];
    const int ox = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int idx = bid * 64 + ox;                            // over (tap, co, ci), ci fastest
    const int total = NT * Cout * Cin;
    float acc = 0.f;
    int ci = 0, co = 0, tap = 0;
    if (idx < total) {
        ci = idx % Cin;
        const int t = idx / Cin;
        co = t % Cout;
        tap = t / Cout;
        const size_t stride = (size_t)NT * CoutP * CinQ;
        const float* p = ws + ((size_t)tap * CoutP + co) * CinQ + ci;
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // 8 slabs in flight per thread, fixed order
        for (int s = sy; s < nslab; s += 32) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = p[(size_t)min(s + 4 * j, nslab - 1) * stride];
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] += s + 4 * j < nslab ? t[j] : 0.f;
        }
        acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    red[sy][ox] = acc;
    __syncthreads();
    if (sy == 0 && idx < total)
        dw[((size_t)co * Cin + ci) * NT + tap] = (red[0][ox] + red[1][ox]) + (red[2][ox] + red[3][ox]);
}

// fills a row for a direct / 1x1 weight gradient; returns its workgroup count
inline int wgrad_reduce_row(long long* desc9, const float* ws, float* dw, int nslab, int NT, int Cout, int Cin, int CoutP,
                            int CinQ) {
    W44Red r;
    r.ws = ws; r.dw = dw; r.bsum = nullptr; r.db = nullptr; r.db2 = nullptr;
    r.Cout = Cout; r.Cin = Cin; r.CoutP = CoutP; r.CinQ = CinQ; r.nslab = nslab;
    r.nmain = (NT * Cout * Cin + 63) / 64; r.first = 0; r.nt = NT;
    memcpy(desc9, &r, sizeof(r));
    return r.nmain;
}
