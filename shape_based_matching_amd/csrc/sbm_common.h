// sbm_common.h — types shared by the gfx950 kernels of the engine (per-template records, candidate records, load types).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/sbm_types.h"
#include "sbm_resize_table.h"

namespace sbm {

// Issue priority of the waves of the kernels that mostly WAIT (linear memories, coarse pass, refinement pass) above the
// gradient kernel's (default 0).  With several batches in flight these kernels share SIMDs with gradient waves that
// could issue a vector instruction every cycle; at equal priority their few instructions -- the ones that put the next
// loads in flight -- queue behind that stream, their waves live longer and hold their slots and registers longer.
// Measured (tools/r03_ab_prio.sh, three batches in flight): scene frame 115.2 -> 112.6 us per step, tiled frame 144.7 ->
// 136.9; priorities 1 and 3 alike, a raised priority for the small level-1 gradient launch adds nothing.  A kernel that
// runs alone is unaffected.
__device__ __forceinline__ void raise_wave_priority() { __builtin_amdgcn_s_setprio(1); }
// ... and the one-workgroup kernels between them (counter reset, list copy) above those: at the default priority a
// 64-thread reset queued behind another batch's coarse pass waits for it to END (BASELINE config 3 with two batches in
// flight: 45 -> 74 us per step when only the waiting kernels were raised)
__device__ __forceinline__ void top_wave_priority() { __builtin_amdgcn_s_setprio(3); }


// per (template, level) record on the device
struct DevTL {
    int32_t width, height, nf, feat_off;
};
// coarse candidate / refinement state (Match under construction)
// Carries everything the NEXT refinement pass needs (the template record of that level, its keep
// threshold, the ids of the final Match), so that pass starts its feature loads after one
// dependent read instead of three.
struct Cand {
    int32_t t;   // template index (into the uploaded list)
    int32_t x;   // Match::x at the level just processed
    int32_t y;
    int32_t raw; // integer similarity sum; < 0 = dropped by the per-level filter
    int32_t next_width, next_height, next_nf, next_feat_off; // DevTL of the level refined next
    int32_t next_keep;                                       // raw_keep of that level
    int32_t class_idx, template_id;
    int32_t pad;
};
static_assert(sizeof(Cand) == 48, "Cand layout");

__device__ __forceinline__ void cand_fill_next(Cand& c, const DevTL* __restrict__ tls, const int32_t* __restrict__ raw_keep,
                                               const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id,
                                               int L, int next_level)
{
    if (next_level >= 0) {
        const DevTL tn = tls[(size_t)c.t * L + next_level];
        c.next_width = tn.width;
        c.next_height = tn.height;
        c.next_nf = tn.nf;
        c.next_feat_off = tn.feat_off;
        c.next_keep = raw_keep[(size_t)c.t * L + next_level];
    }
    c.class_idx = class_idx[c.t];
    c.template_id = template_id[c.t];
    c.pad = 0;
}

// Per-frame counters of the template loop, three 128-byte lines per frame.  The candidate count is READ by every
// workgroup of the refinement pass while other workgroups arrive: arrivals are device-scope atomics, performed at the
// memory side, and a load of a line they are hitting waits behind them -- so the count, the arrival levels and the
// sub-arrivals each have a line of their own.
constexpr int CTR_STRIDE = 96;  // int32 per frame
constexpr int CTR_COUNT = 0;    // candidates of the coarse pass ([2,3]: refinement bytes, sbm_get_stats)
constexpr int CTR_ARRIVE = 32;  // [32] arrivals of sub-counters, [33] arrivals of frames (k_similarity_local groups)
constexpr int CTR_SUB = 64;     // [64..95] arrivals of workgroups

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ uint32_t ld_u32_any(const uint8_t* p) { return *(const u32_unaligned*)p; }
struct __attribute__((aligned(4))) u128_a4 { uint32_t x, y, z, w; }; // 16-byte load at 4-byte alignment

} // namespace sbm
