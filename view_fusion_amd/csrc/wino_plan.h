// Tail plan shared by the persistent Winograd forward / dgrad kernels (winograd24.hip: nested F(2,3)xF(4,3);
// winograd44f.hip: F(4x4,3x3)): how a grid of T equal workgroup tiles is finished on 256 CUs.
#pragma once
#include <cstdlib>

namespace {

constexpr int WINO_SLOTS = 256;       // one workgroup per CU
constexpr double WINO_TAIL_G = 4.0;   // cost of the fix-up launch behind a split tail, in chunk-times
constexpr double WINO_TAIL_F = 2.5;   // per-part overhead of a K-split tail part, in chunk-times (see wino_tail_time)

// How a grid of T equal tiles is finished when T is not a multiple of the slot count: the last R = T mod 256
// tiles are split over K into `split` parts each; the parts run in ceil(R*split/256) rounds.  A part of `per` chunks
// costs per + F chunk-times, F = the prologue + epilogue + partial-output store of a workgroup in units of one chunk
// (measured: a tile's epilogue is 23 % of an 8-chunk tile; VF_WINO_TAIL_F overrides, tuning aid).  `split` is the value
// in [1, min(8, nch/4)] with the shortest tail (ties: fewer parts).
inline double wino_tail_overhead() {
    static const double f = getenv("VF_WINO_TAIL_F") ? atof(getenv("VF_WINO_TAIL_F")) : WINO_TAIL_F;
    return f;
}
inline double wino_tail_fixup() {     // the fix-up launch a split costs, in chunk-times (VF_WINO_TAIL_G overrides, tuning aid)
    static const double g = getenv("VF_WINO_TAIL_G") ? atof(getenv("VF_WINO_TAIL_G")) : WINO_TAIL_G;
    return g;
}
inline double wino_tail_time(int R, int sp, int nch, double F) {      // in units of one whole tile
    const int per = (nch + sp - 1) / sp;
    const double fix = (sp > 1 && F > 0.0) ? wino_tail_fixup() : 0.0;
    return ((double)((R * sp + WINO_SLOTS - 1) / WINO_SLOTS) * (per + F) + fix) / (nch + F);
}

inline void wino_tail_plan(int T, int nch, int* nfull, int* split, double F = wino_tail_overhead()) {
    *nfull = T;
    *split = 1;
    const int R = T % WINO_SLOTS;
    if (R == 0 || T / WINO_SLOTS >= 3) return;          // tail round costs < 1/4 of the launch: leave it
    int best = 1;
    for (int sp = 2; sp <= 8 && sp <= nch / 4; ++sp) {    // at least 4 chunks per part
        const int per = (nch + sp - 1) / sp;              // the kernel gives each part `per` chunks:
        if ((nch + per - 1) / per != sp) continue;        // no part may start beyond the last chunk
        if (wino_tail_time(R, sp, nch, F) < wino_tail_time(R, best, nch, F) - 1e-9) best = sp;
    }
    if (best < 2) return;
    *nfull = T - R;
    *split = best;
}

}  // namespace
