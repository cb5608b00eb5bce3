// One-shot gradient all-reduce over IPC-mapped peer arenas, fused with the Adam update (replaces
// DistributedDataParallel's bucketed NCCL all-reduce + optimizer.step(), reference experiment.py:104-107, 118-120,
// 292-293).  One process per GPU; every rank's gradient arena (reducer.GradArena: the backward kernels write dW / db /
// dgamma / dbeta straight into it) lives in IPC-exportable device memory and is mapped into every peer's address space;
// on one node the 8 GPUs are fully connected by xGMI links, so a peer read is a direct load over the link.
//
// Per arena segment k of iteration e (six segments, so the exchange overlaps the rest of the backward pass):
//   signal(ready, k, e)   one thread: system-scope release, then flag[k][me] = e in EVERY peer's flag block
//   wait(ready, k, e)     one wave spins (s_sleep, bounded by a wall-clock timeout) until flag[k][p] >= e for all p
//   reduce_adam(k)        every rank reads the segment from ALL W arenas in rank order, g = (g_0 + ... + g_{W-1}) / W,
//                         writes g to its local averaged-gradient buffer and applies Adam to ITS OWN copy of the
//                         parameters -- every rank computes the same sum in the same order, so the replicas stay
//                         bit-identical with no second exchange step, no sharded optimizer state (checkpoints keep
//                         torch's layout) and no all-gather of parameters
//   signal(done, k, e)    "I have read your segment k": a rank waits for it before iteration e + 1 overwrites the arena
// Link traffic: every rank pulls (W - 1) x 135.8 MB per iteration, 135.8 MB per link: ~0.9 ms of link time spread over
// six launches behind the backward pass (a reduce-scatter + all-gather would move 1/4 of that per link at W = 8 but needs
// a second cross-rank barrier per segment and either sharded Adam state or a parameter all-gather).
// Visibility: kernel boundaries release / acquire at system scope (a segment's gradients are complete when the producer
// kernels have finished; the reduce kernel starts after the wait kernel has seen the flags); the flags themselves are
// accessed with system-scope atomics.  Validated on this pool with two processes sharing ONE GPU (tests/
// test_gpu_two_rank.py); no run on two devices exists.
#include "common.h"
#include "adam_update.h"
#include <cstring>

namespace {

struct XAdamDesc {          // same rows as adam.hip: {p, g, exp_avg, exp_avg_sq, numel, first_block}; p == 0: average only
    float* p;
    const float* g;         // the slot in the LOCAL arena
    float* m;
    float* v;
    long long numel, first_block;
};

constexpr int XG_PER_BLOCK = 1024;
constexpr int XG_MAXW = 16;

struct XPeers {
    const float* base[XG_MAXW];   // arena base of rank r as mapped HERE (own rank: the local arena)
};

__global__ __launch_bounds__(256) void xgmi_reduce_adam_kernel(const XAdamDesc* __restrict__ desc, int ntensors,
                                                               XPeers peers, const float* my_base,
                                                               float* __restrict__ gavg_base, int world, float b1,
                                                               float b2, float eps, const float* __restrict__ scal) {
    const float lr = scal[0], bc1 = scal[1], bc2 = scal[2];
    int lo = 0, hi = ntensors;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    const XAdamDesc d = desc[lo];
    const long long base = ((long long)blockIdx.x - d.first_block) * XG_PER_BLOCK + 4 * threadIdx.x;
    if (base >= d.numel) return;
    const long long off = (d.g - my_base) + base;           // element offset inside every arena
    const float inv = 1.0f / (float)world;
    const float step = lr / bc1, rs = 1.0f / sqrtf(bc2), omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    // (adam.hip's arithmetic, bit for bit: one shared, explicitly rounded update)
    auto upd = [&](float& p, float g, float& m, float& v) { vf_adam_update(p, g, m, v, b1, b2, omb1, omb2, step, rs, eps); };
    if (base + 4 <= d.numel) {
        float4 g = *reinterpret_cast<const float4*>(peers.base[0] + off);
        for (int r = 1; r < world; ++r) {
            const float4 q = *reinterpret_cast<const float4*>(peers.base[r] + off);
            g.x += q.x; g.y += q.y; g.z += q.z; g.w += q.w;
        }
        g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
        *reinterpret_cast<float4*>(gavg_base + off) = g;
        if (d.p) {
            float4 p = *reinterpret_cast<float4*>(d.p + base);
            float4 m = *reinterpret_cast<float4*>(d.m + base);
            float4 v = *reinterpret_cast<float4*>(d.v + base);
            upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
            *reinterpret_cast<float4*>(d.p + base) = p;
            *reinterpret_cast<float4*>(d.m + base) = m;
            *reinterpret_cast<float4*>(d.v + base) = v;
        }
    } else {
        for (long long i = base; i < d.numel; ++i) {
            float g = peers.base[0][off + (i - base)];
            for (int r = 1; r < world; ++r) g += peers.base[r][off + (i - base)];
            g *= inv;
            gavg_base[off + (i - base)] = g;
            if (d.p) {
                float p = d.p[i], m = d.m[i], v = d.v[i];
                upd(p, g, m, v);
                d.p[i] = p; d.m[i] = m; d.v[i] = v;
            }
        }
    }
}

struct XFlags {
    unsigned* flags[XG_MAXW];     // flag block of rank r as mapped here: unsigned [nslots][world]
};

// flag[slot][me] = epoch in every rank's block.  The system-scope release in front orders it behind everything this
// stream has executed before (the producer kernels have completed: stream order; their writes are made visible to the
// other agents by the fence).
__global__ void xgmi_signal_kernel(XFlags f, int world, int me, int slot, unsigned epoch) {
    if (threadIdx.x != 0) return;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    __threadfence_system();
    for (int r = 0; r < world; ++r)
        __hip_atomic_store(f.flags[r] + (size_t)slot * world + me, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Spin until flag[slot][p] >= epoch for every p and every slot in [slot_lo, slot_hi), one lane per (slot, peer) pair.
// Bounded: after timeout_ticks of the 100 MHz wall clock the kernel gives up and raises status[0] (the host checks it at
// the end of the iteration: a missing peer becomes an exception, never a hung GPU).
__global__ __launch_bounds__(64) void xgmi_wait_kernel(const unsigned* my_flags, int world, int slot_lo, int slot_hi,
                                                       unsigned epoch, unsigned* status, unsigned long long timeout_ticks) {
    const int n = (slot_hi - slot_lo) * world;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool late = false;
    for (int i = threadIdx.x; i < n; i += 64) {
        const unsigned* w = my_flags + (size_t)slot_lo * world + i;
        // (epochs are compared as a signed distance: the counter may wrap)
        while ((int)(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { late = true; break; }
            __builtin_amdgcn_s_sleep(32);
        }
        if (late) break;
    }
    if (late) __hip_atomic_store(status, 1u + (unsigned)slot_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
}

}  // namespace

extern "C" {

// ---- IPC-exportable device memory: the gradient arena and the flag block of a rank
int vf_xgmi_alloc(void** ptr, long bytes) {
    if (!ptr || bytes <= 0) return (int)hipErrorInvalidValue;
    hipError_t e = hipMalloc(ptr, (size_t)bytes);
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(*ptr, 0, (size_t)bytes);
}

int vf_xgmi_free(void* ptr) { return ptr ? (int)hipFree(ptr) : 0; }

// handle64: 64 bytes (hipIpcMemHandle_t) a peer process opens with vf_xgmi_open
int vf_xgmi_export(void* ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (!ptr || !handle64) return (int)hipErrorInvalidValue;
    return (int)hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr);
}

int vf_xgmi_open(const void* handle64, void** ptr) {
    if (!ptr || !handle64) return (int)hipErrorInvalidValue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof(h));
    return (int)hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
}

int vf_xgmi_close(void* ptr) { return ptr ? (int)hipIpcCloseMemHandle(ptr) : 0; }

// peer_flags: HOST array of `world` device pointers (rank r's flag block as mapped in this process)
int vf_xgmi_signal(const void* const* peer_flags, int world, int rank, int slot, unsigned epoch, void* stream) {
    if (!peer_flags || world < 1 || world > XG_MAXW || rank < 0 || rank >= world || slot < 0) return (int)hipErrorInvalidValue;
    XFlags f;
    for (int r = 0; r < world; ++r) f.flags[r] = (unsigned*)peer_flags[r];
    hipLaunchKernelGGL(xgmi_signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, f, world, rank, slot, epoch);
    VF_RETURN_LAST_ERROR();
}

// my_flags: this rank's flag block; waits for slots [slot_lo, slot_hi); status: device unsigned, 0 = fine
int vf_xgmi_wait(const void* my_flags, int world, int slot_lo, int slot_hi, unsigned epoch, void* status,
                 long timeout_us, void* stream) {
    if (!my_flags || !status || world < 1 || world > XG_MAXW || slot_lo < 0 || slot_hi <= slot_lo) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(xgmi_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)my_flags, world,
                       slot_lo, slot_hi, epoch, (unsigned*)status, (unsigned long long)timeout_us * 100ull);
    VF_RETURN_LAST_ERROR();
}

// desc: device int64 [ntensors][6] rows {p | 0, g (slot in the local arena), exp_avg, exp_avg_sq, numel, first_block}
// (block = 1024 elements); peer_bases: HOST array of `world` arena base pointers in rank order (own rank: my_base);
// gavg_base: local buffer with the arena's layout that receives the averaged gradients; scalars: device {lr, bc1, bc2}.
int vf_xgmi_reduce_adam(const void* desc, int ntensors, long total_blocks, const void* const* peer_bases,
                        const float* my_base, float* gavg_base, int world, const float* scalars, float beta1,
                        float beta2, float eps, void* stream) {
    if (ntensors <= 0 || total_blocks <= 0) return 0;
    if (!desc || !peer_bases || !my_base || !gavg_base || !scalars || world < 1 || world > XG_MAXW)
        return (int)hipErrorInvalidValue;
    XPeers p;
    for (int r = 0; r < world; ++r) p.base[r] = (const float*)peer_bases[r];
    hipLaunchKernelGGL(xgmi_reduce_adam_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const XAdamDesc*)desc, ntensors, p, my_base, gavg_base, world, beta1, beta2, eps, scalars);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
