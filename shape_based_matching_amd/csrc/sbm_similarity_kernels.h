// sbm_similarity_kernels.h — gfx950 kernels of the template loop.
// Reference functions replaced (file:line in ddcr/shape_based_matching):
//   k_prep_features / k_prep_coarse_items   accessLinearMemory address arithmetic   line2Dup.cpp:782-805
//   k_similarity_coarse(_wave)   similarity / similarity_64 + candidate scan        line2Dup.cpp:807-858, 924-984, 1199-1216
//   k_similarity_map             similarity / similarity_64 (score map out)         same
//   k_similarity_local           similarityLocal(_64) + best-of-16x16 + filter      line2Dup.cpp:860-922, 986-1048, 1221-1293
//   k_emit_coarse                single-level pyramids: candidates are the matches  line2Dup.cpp:1199-1216
#pragma once
#include "sbm_common.h"
#include "sbm_lm_kernels.h" // response4: the refinement pass applies the response LUT to spread bytes

#include "sbm_local_bits.h"

namespace sbm {

// ---------------------------------------------------------------------------
// Similarity
// ---------------------------------------------------------------------------

// per-feature linear-memory offset for the current level geometry
__global__ __launch_bounds__(256) void k_prep_features(const uint32_t* __restrict__ fxy,
                                                       const uint8_t* __restrict__ flabel,
                                                       const uint8_t* __restrict__ flevel, int64_t n,
                                                       const int* __restrict__ Tl, const int* __restrict__ Wl,
                                                       const int* __restrict__ Hl,
                                                       const int64_t* __restrict__ stride_l,
                                                       int32_t* __restrict__ foff, int lc)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int l = flevel[i];
        const int T = Tl[l], W = Wl[l], H = Hl[l];
        const int x = fxy[i] & 0xffff, y = fxy[i] >> 16;
        int64_t off = (int64_t)flabel[i] * stride_l[l] + (int64_t)((y % T) * T + (x % T)) * W * H + (int64_t)(y / T) * W + x / T;
        // Coarsest level: a feature outside the image is skipped by the reference (:836-837) -- it adds nothing.  Point it
        // at the zero tail of the last orientation's plane (every position of the span reads 0 there), so that the coarse
        // pass needs neither the feature's coordinates nor a bounds test.  The refinement levels keep the plain offset:
        // there the test is on the feature moved by the candidate's patch origin.
        if (l == lc && (x >= W * T || y >= H * T)) off = 7 * stride_l[l] + (int64_t)T * T * W * H;
        foff[i] = (int32_t)off;
    }
}

// What one work item of the coarse pass needs about its template, in one 32-byte record (one scalar load instead of
// the chain active[] -> tls[] -> raw_min[]), and the template's coarsest-level feature offsets SORTED BY THEIR BYTE
// MISALIGNMENT (offset & 3): features of one class are summed as raw dwords and re-aligned once per batch instead of
// once per feature (a sum does not care about the order of its terms).  e01 / e2: where classes 0, 1, 2 end in the
// sorted list (class 3 ends at nf).  cfoff: the first 64 sorted offsets, indexed by the item's slot (a vector load that
// does not wait for the record).
struct CoarseItem {
    int32_t t, soff_base, nf, npos, rmin, e01, e2, pad;
};
__global__ __launch_bounds__(64) void k_prep_coarse_items(const int32_t* __restrict__ active, int n_active, const DevTL* __restrict__ tls,
                                                          int L, int lc, const int32_t* __restrict__ raw_min,
                                                          const int32_t* __restrict__ foff, const int32_t* __restrict__ soff_base,
                                                          int T, int W, int H, int zero_off, CoarseItem* __restrict__ items,
                                                          int32_t* __restrict__ soff, int32_t* __restrict__ cfoff)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_active) return;
    const int t = active[slot];
    const DevTL tl = tls[(size_t)t * L + lc];
    const int32_t* f = foff + tl.feat_off;
    int32_t* out = soff + soff_base[slot];
    int cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < tl.nf; ++i) ++cnt[f[i] & 3];
    int pos[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
    const int e0 = pos[1], e1 = pos[2], e2 = pos[3];
    for (int i = 0; i < tl.nf; ++i) { // stable counting sort over the four classes
        const int o = f[i];
        out[pos[o & 3]++] = o;
    }
    for (int i = 0; i < 64; ++i) cfoff[(size_t)slot * 64 + i] = i < tl.nf ? out[i] : zero_off;
    const int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
    CoarseItem it;
    it.t = t;
    it.soff_base = soff_base[slot];
    it.nf = tl.nf;
    it.npos = (H - hf) * W + (W - wf) + 1;
    it.rmin = raw_min[(size_t)t * L + lc];
    it.e01 = e0 | (e1 << 16);
    it.e2 = e2;
    it.pad = 0;
    items[slot] = it;
}

__device__ __forceinline__ int template_positions(const DevTL& tl, int W, int H, int T)
{
    int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
    return (H - hf) * W + (W - wf) + 1;
}

constexpr int SIM_POS_PER_BLOCK = 1024; // 256 lanes x 4 positions
constexpr int FB = 8;                   // features per batch: FB independent loads in flight per lane
constexpr int FBL = 16;                 // refinement pass: few features per wave, all in flight at once
constexpr int FB16 = 8;                 // same for the 16-byte variant (coarse pass)

// Sum `count` features starting at `f0` into packed 16-bit lanes.  Each lane
// owns 4 consecutive bytes at lm + foff[f] + lane_off.  The wave first loads 64
// features' (position, offset) records with one vector load per array, then
// broadcasts them one at a time with v_readlane: the inner loop is branch-free
// and keeps FB independent 4-byte loads in flight per lane.  MUST be called with
// all 64 lanes of the wave active (v_readlane reads lanes regardless of EXEC, and
// an inactive lane's record would be stale).  A feature that is
// out of the image (skipped by the reference, :836-837 / :878-879) or past the
// end of the range reads zero bytes from the linear memories' zero tail.
// Packed-u8 partial sums are carry-free for <= 63 addends (63 * 4 = 252) and
// are flushed into two packed-u16 words before that.
// COMPACT: lm is the single spread plane of a refinement-only level; the feature's orientation plane offset
// (label * lm_stride, part of foff) is taken out again and the response LUT is applied to the loaded bytes.
template <bool COMPACT = false>
__device__ __forceinline__ void accumulate_features(const uint8_t* __restrict__ lm,
                                                    const uint32_t* __restrict__ fxy,
                                                    const int32_t* __restrict__ foff, int f0, int count,
                                                    int rows, int cols, int ox, int oy, int lane_off,
                                                    const uint8_t* __restrict__ zero_addr, int zero_lane_off,
                                                    uint32_t& lo, uint32_t& hi,
                                                    const uint8_t* __restrict__ flabel = nullptr, int lm_stride = 0)
{
    const int lane = threadIdx.x & 63;
    const uint8_t* p_ok = lm + lane_off;
    const uint8_t* p_zero = zero_addr + zero_lane_off;
    uint32_t acc = 0;
    lo = hi = 0;
    int pending = 0;
    count = __builtin_amdgcn_readfirstlane(count); // wave-uniform by contract: keep the loop control scalar
    f0 = __builtin_amdgcn_readfirstlane(f0);
    for (int b = 0; b < count; b += 64) {
        int sel = -1, lab = 0;
        if (b + lane < count) {
            const uint32_t xy = fxy[f0 + b + lane];
            const int x = (int)(xy & 0xffff) + ox, y = (int)(xy >> 16) + oy;
            if (x >= 0 && y >= 0 && x < cols && y < rows) sel = foff[f0 + b + lane];
            if (COMPACT) {
                lab = flabel[f0 + b + lane];
                if (sel >= 0) sel -= lab * lm_stride;
            }
        }
        const int nb = count - b < 64 ? count - b : 64;
        // straight-line batches (no branches between the loads of a batch, so all of a batch's
        // loads are in flight together); the tail runs as batches of 4, 2, 1: no padding loads
        auto batch = [&](auto N, int u) {
            constexpr int n = decltype(N)::value;
            uint32_t v[n];
#pragma unroll
            for (int k = 0; k < n; ++k) {
                const int o = __builtin_amdgcn_readlane(sel, (u + k) & 63);
                v[k] = ld_u32_any(o >= 0 ? p_ok + o : p_zero);
            }
            if (COMPACT) {
#pragma unroll
                for (int k = 0; k < n; ++k) v[k] = response4(v[k], __builtin_amdgcn_readlane(lab, (u + k) & 63));
            }
#pragma unroll
            for (int k = 0; k < n; ++k) acc += v[k];
            pending += n;
            if (pending + FBL > 63) {
                lo += acc & 0x00ff00ffu;
                hi += (acc >> 8) & 0x00ff00ffu;
                acc = 0;
                pending = 0;
            }
        };
        int u = 0;
        for (; u + FBL <= nb; u += FBL) batch(std::integral_constant<int, FBL>{}, u);
        if (FBL > 8 && nb - u >= 8) { batch(std::integral_constant<int, 8>{}, u); u += 8; }
        if (nb - u >= 4) { batch(std::integral_constant<int, 4>{}, u); u += 4; }
        if (nb - u >= 2) { batch(std::integral_constant<int, 2>{}, u); u += 2; }
        if (nb - u >= 1) batch(std::integral_constant<int, 1>{}, u);
    }
    lo += acc & 0x00ff00ffu;
    hi += (acc >> 8) & 0x00ff00ffu;
}


__device__ __forceinline__ int unpack4(uint32_t lo, uint32_t hi, int k)
{
    return (k & 1) ? (int)((hi >> (8 * (k - 1))) & 0xffff) : (int)((lo >> (8 * k)) & 0xffff);
}

// 16 consecutive positions per lane: one 4-byte-aligned 16-byte load plus one
// 4-byte load per feature, re-aligned with v_alignbyte by the (wave-uniform)
// byte misalignment of the feature's linear-memory offset.  Measured on
// MI355X (tools/ld_probe.hip): 4-byte-aligned dwordx4 streams from L2 at
// ~12 TB/s, twice the rate of dword loads and of byte-misaligned dwordx4.
// j0 must be a multiple of 4.  Same calling convention as accumulate_features
// (all 64 lanes active).  lo[i]/hi[i]: packed u16 sums of bytes 4i..4i+3.
template <bool RESET = true>
__device__ __forceinline__ void accumulate_features16(const uint8_t* __restrict__ lm,
                                                      const uint32_t* __restrict__ fxy,
                                                      const int32_t* __restrict__ foff, int count,
                                                      int rows, int cols, int j0, bool lane_on, int zero_off,
                                                      uint32_t (&lo)[4], uint32_t (&hi)[4])
{
    const int lane = threadIdx.x & 63;
    const uint8_t* p = lm + j0;
    uint32_t acc[4] = {0, 0, 0, 0};
    if (RESET) { // RESET == false: add to the sums the caller already holds
#pragma unroll
        for (int i = 0; i < 4; ++i) lo[i] = hi[i] = 0;
    }
    int pending = 0;
    count = __builtin_amdgcn_readfirstlane(count); // wave-uniform by contract: keep the loop control scalar
    for (int b = 0; b < count; b += 64) {
        int sel = zero_off; // computed with every lane active: v_readlane below reads any lane's copy
        if (b + lane < count) {
            // both loads issued together (a load of foff under the bounds test would wait for fxy first)
            const uint32_t xy = fxy[b + lane];
            const int off = foff[b + lane];
            const int x = (int)(xy & 0xffff), y = (int)(xy >> 16);
            if (x < cols && y < rows) sel = off;
        }
        const int nb = count - b < 64 ? count - b : 64;
        // lanes past the template's span (lane_on == false) issue no loads at all; lane 63 fetches the
        // 4 bytes that follow its 16, every other lane takes them from its right neighbour's load (DPP)
        auto batch = [&](auto N, int u) {
            constexpr int n = decltype(N)::value;
            u128_a4 q[n];
            uint32_t e[n];
            int sh[n];
            if (lane_on) {
#pragma unroll
                for (int k = 0; k < n; ++k) {
                    const int o = __builtin_amdgcn_readlane(sel, (u + k) & 63);
                    sh[k] = o & 3;
                    const uint8_t* a = p + (o & ~3);
                    q[k] = *(const u128_a4*)a;
                    e[k] = 0;
                    if (lane == 63) e[k] = *(const uint32_t*)(a + 16);
                }
            } else {
#pragma unroll
                for (int k = 0; k < n; ++k) {
                    q[k] = u128_a4{0, 0, 0, 0};
                    e[k] = 0;
                    sh[k] = 0;
                }
            }
#pragma unroll
            for (int k = 0; k < n; ++k) {
                // wave_shl:1 -> lane i receives lane i+1's first dword; lane 63 keeps its own loaded value
                const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)e[k], (int)q[k].x, 0x130, 0xf, 0xf, false);
                acc[0] += __builtin_amdgcn_alignbyte(q[k].y, q[k].x, sh[k]);
                acc[1] += __builtin_amdgcn_alignbyte(q[k].z, q[k].y, sh[k]);
                acc[2] += __builtin_amdgcn_alignbyte(q[k].w, q[k].z, sh[k]);
                acc[3] += __builtin_amdgcn_alignbyte(nx, q[k].w, sh[k]);
            }
            pending += n;
            if (pending + FB16 > 63) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] += acc[i] & 0x00ff00ffu;
                    hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
                    acc[i] = 0;
                }
                pending = 0;
            }
        };
        int u = 0;
        for (; u + FB16 <= nb; u += FB16) batch(std::integral_constant<int, FB16>{}, u);
        if (nb - u >= 4) { batch(std::integral_constant<int, 4>{}, u); u += 4; }
        if (nb - u >= 2) { batch(std::integral_constant<int, 2>{}, u); u += 2; }
        if (nb - u >= 1) batch(std::integral_constant<int, 1>{}, u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] += acc[i] & 0x00ff00ffu;
        hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
    }
}

typedef unsigned short coarse_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t coarse_pk_max_u16(uint32_t a, uint32_t b) // v_pk_max_u16
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(coarse_us2, a), __builtin_bit_cast(coarse_us2, b)));
}

// The coarse pass of a wave that owns ALL nf features of its item, with exact pruning.
//
// Pruning: a feature adds at most 4, so once `done` features are in, a position whose partial sum is below
// rmin - 4*(nf - done) cannot reach rmin any more; after every batch the wave looks whether any of its 1024 positions is
// still alive and returns false if none is.  Positions past the template's span may hold anything: they can only keep
// the wave going, the caller's scan ignores them.  Returns true with the complete sums.
//
// Round 3: the feature offsets arrive sorted by byte misalignment (k_prep_coarse_items), a batch (up to 8 features)
// never crosses a class boundary, and its features are summed as RAW dwords -- the 16 bytes at the 4-byte-aligned
// address, the dword after them (lane 63 loads it, every other lane takes its right neighbour's first dword: a lane
// permutation commutes with the sum, so ONE DPP move per batch) -- then re-aligned once (4 v_alignbyte per batch instead
// of per feature).  While no position can exceed 255 (<= 60 features) the totals stay packed bytes and the aliveness test
// is a carry-free byte trick: with K = 128 - bound, byte + K has its top bit set iff byte >= bound, and byte + K never
// exceeds 128 + (4 nf - rmin) (needs 4 nf - rmin <= 127: thresholds of ~70 % and up at 128 features; below that, and
// once a bound passes 128, the sums are widened to packed u16 as before).
// soff: the template's sorted offsets (out-of-image features point at the zero tail); sel0: the first 64 of them.
__device__ __forceinline__ bool accumulate_features16_pruned(const uint8_t* __restrict__ lm, const int32_t* __restrict__ soff, int sel0,
                                                             int nf, int e0, int e1, int e2, int j0, bool lane_on, int zero_off,
                                                             int rmin, uint32_t (&lo)[4], uint32_t (&hi)[4])
{
    const int lane = threadIdx.x & 63;
    const uint8_t* p = lm + j0;
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = hi[i] = 0;
    nf = __builtin_amdgcn_readfirstlane(nf);
    rmin = __builtin_amdgcn_readfirstlane(rmin);
    e0 = __builtin_amdgcn_readfirstlane(e0);
    e1 = __builtin_amdgcn_readfirstlane(e1);
    e2 = __builtin_amdgcn_readfirstlane(e2);
    const int slack = 4 * nf - rmin;
    uint32_t acc8[4] = {0, 0, 0, 0}; // packed-byte totals of the features since the last widening
    int in8 = 0;                     // features in acc8
    int sel = sel0;
    int g = 0;
    while (g < nf) {
        if (g && (g & 63) == 0) sel = g + lane < nf ? soff[g + lane] : zero_off; // next 64 offsets
        const int cls = g < e0 ? 0 : (g < e1 ? 1 : (g < e2 ? 2 : 3));
        const int cend = cls == 0 ? e0 : (cls == 1 ? e1 : (cls == 2 ? e2 : nf));
        int lim = cend - g;                 // features left in this class ...
        const int chunk_left = 64 - (g & 63); // ... and in this 64-offset chunk
        lim = lim < chunk_left ? lim : chunk_left;
        uint32_t raw[5] = {0, 0, 0, 0, 0};
        int n = 0;
        auto batch = [&](auto N) {
            constexpr int nn = decltype(N)::value;
            if (lane_on) {
                u128_a4 q[nn];
                uint32_t e[nn];
#pragma unroll
                for (int k = 0; k < nn; ++k) {
                    const int o = __builtin_amdgcn_readlane(sel, (g + k) & 63);
                    const uint8_t* a = p + (o & ~3);
                    q[k] = *(const u128_a4*)a;
                    e[k] = 0;
                    if (lane == 63) e[k] = *(const uint32_t*)(a + 16);
                }
#pragma unroll
                for (int k = 0; k < nn; ++k) { // <= 8 addends of <= 4: no carry between the bytes
                    raw[0] += q[k].x;
                    raw[1] += q[k].y;
                    raw[2] += q[k].z;
                    raw[3] += q[k].w;
                    raw[4] += e[k];
                }
            }
            n = nn;
        };
        if (lim >= FB16) batch(std::integral_constant<int, FB16>{});
        else if (lim >= 4) batch(std::integral_constant<int, 4>{});
        else if (lim >= 2) batch(std::integral_constant<int, 2>{});
        else batch(std::integral_constant<int, 1>{});
        // the dword after the lane's 16 bytes: the right neighbour's first dword (lane 63: what it loaded itself)
        const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)raw[4], (int)raw[0], 0x130, 0xf, 0xf, false);
        acc8[0] += __builtin_amdgcn_alignbyte(raw[1], raw[0], cls);
        acc8[1] += __builtin_amdgcn_alignbyte(raw[2], raw[1], cls);
        acc8[2] += __builtin_amdgcn_alignbyte(raw[3], raw[2], cls);
        acc8[3] += __builtin_amdgcn_alignbyte(nx, raw[3], cls);
        g += n;
        in8 += n;
        const int bound = rmin - 4 * (nf - g);
        const bool test = bound > 0 && g < nf;
        const bool in_bytes = test && slack <= 127 && bound <= 128 && in8 == g; // nothing widened yet: the totals are the packed bytes
        if (in_bytes) {
            const uint32_t K = (uint32_t)(128 - bound) * 0x01010101u; // byte + K <= 128 + slack: no carry between the bytes
            const uint32_t hit = ((acc8[0] + K) | (acc8[1] + K) | (acc8[2] + K) | (acc8[3] + K)) & 0x80808080u;
            if (__builtin_amdgcn_ballot_w64(hit != 0u) == 0ull) return false;
        }
        if ((test && !in_bytes) || in8 > 52) {
            // widen: packed bytes into the packed-u16 totals (a byte may hold up to 4 * 60)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo[i] += acc8[i] & 0x00ff00ffu;
                hi[i] += (acc8[i] >> 8) & 0x00ff00ffu;
                acc8[i] = 0;
            }
            in8 = 0;
            if (test && !in_bytes) {
                const uint32_t m = coarse_pk_max_u16(coarse_pk_max_u16(coarse_pk_max_u16(lo[0], hi[0]), coarse_pk_max_u16(lo[1], hi[1])),
                                                     coarse_pk_max_u16(coarse_pk_max_u16(lo[2], hi[2]), coarse_pk_max_u16(lo[3], hi[3])));
                const int best = (int)((m & 0xffff) > (m >> 16) ? (m & 0xffff) : (m >> 16));
                if (__builtin_amdgcn_ballot_w64(best >= bound) == 0ull) return false;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] += acc8[i] & 0x00ff00ffu;
        hi[i] += (acc8[i] >> 8) & 0x00ff00ffu;
    }
    return true;
}

// plain byte copy (gathered match lists -> pinned host mirror); n multiple of 8 by construction
__global__ __launch_bounds__(256) void k_copy_bytes(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n)
{
    top_wave_priority();
    const size_t n8 = n / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256)
        ((uint64_t*)dst)[i] = ((const uint64_t*)src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = src[n8 * 8 + threadIdx.x];
}

// zero the per-call counters (one launch instead of two memsets)
__global__ void k_reset(int32_t* __restrict__ counters, int32_t* __restrict__ out_count)
{
    top_wave_priority();
    for (int i = threadIdx.x; i < CTR_STRIDE; i += blockDim.x) counters[i] = 0; // layout: sbm_common.h CTR_*
    if (threadIdx.x < 2) out_count[threadIdx.x] = 0;
}

// Which (position chunk, template slot, frame) a workgroup of the coarse pass works on.  Pure speed: any bijection
// of the grid is correct.
__device__ __forceinline__ void coarse_block_item(int& chunk_id, int& templ_slot, int& frame)
{
    chunk_id = blockIdx.x, templ_slot = blockIdx.y, frame = blockIdx.z;
    const bool frame_affinity = (gridDim.z & 7) == 0;
    if (frame_affinity) {
        // Workgroups are dealt to the 8 XCDs round-robin by linear id and each XCD has its own 4 MiB L2.  With a batch
        // of 8k frames give XCD x the frames x, x+8, ... one after the other: its L2 then holds ONE frame's linear
        // memories at a time (2 MiB on the bench configuration) instead of every frame's.
        const uint32_t lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const uint32_t xcd = lin & 7, j = lin >> 3, per = gridDim.x * gridDim.y;
        const uint32_t fl = j / per, rem = j - fl * per;
        frame = (int)(xcd + 8 * fl);
        templ_slot = (int)(rem / gridDim.x);
        chunk_id = (int)(rem - (uint32_t)templ_slot * gridDim.x);
    }
    // XCD-aware (chunk, template) assignment: give XCD x the position chunks [x*cpx, (x+1)*cpx) of EVERY
    // template, so that one L2 only ever sees the slice of the linear memories those chunks read
    // (chunk span + template extent) instead of all of them.
    if (!frame_affinity && (gridDim.x & 7) == 0) {
        const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int cpx = (int)gridDim.x >> 3;
        const int xcd = lin & 7, slot = lin >> 3;
        templ_slot = slot / cpx;
        // the chunk ranges rotate over the XCDs every 64 templates: ranges past the templates' span are
        // empty, and this spreads them evenly while an XCD still works on one slice for 64 templates
        chunk_id = ((xcd + (templ_slot >> 6)) & 7) * cpx + slot % cpx;
    }
}

// The prefix length k1 and the partial-sum bound thr1 of the coarse pass's exact pruning.  A feature adds at most 4,
// so after the first k1 of nf features a position whose partial sum is below rmin - 4*(nf - k1) cannot reach rmin any
// more.  k1 is the shortest prefix (multiple of 4, at least 8) for which that bound is at least 62.5 % of the
// prefix's own maximum 4*k1 (a weaker bound prunes little on busy images); k1 == nf means no pruning.
__device__ __forceinline__ void coarse_prune_split(int nf, int rmin, int& k1, int& thr1)
{
    k1 = nf, thr1 = 0;
    if (rmin > 0) {
        int k = ((4 * nf - rmin) * 5 + 7) >> 3;
        k = ((k < 8 ? 8 : k) + 3) & ~3;
        if (4 * k <= 3 * nf) {
            k1 = k;
            thr1 = rmin - 4 * (nf - k1);
        }
    }
}

// grid = (position chunks of 1024, active templates); block = 4 waves.  Every
// wave covers the same 1024 positions (16 per lane) for a contiguous quarter
// of the template's features, so four times as many loads are in flight per
// position chunk; the partial sums meet in LDS and each wave then scans a
// quarter of the positions.  Fused threshold scan: no score map is written;
// candidates are appended with an atomic counter.
constexpr int COARSE_POS_PER_BLOCK = 1024;
__global__ __launch_bounds__(256) void k_similarity_coarse(
    const uint8_t* __restrict__ lm, int64_t lm_stride, int rows, int cols, int T, int W, int H, int L, int lc,
    const DevTL* __restrict__ tls, const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,
    const int32_t* __restrict__ active, const int32_t* __restrict__ raw_min, const int32_t* __restrict__ raw_keep,
    const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id, Cand* __restrict__ cands,
    int32_t* __restrict__ counters, int cap, int64_t lm_fs)
{
    raise_wave_priority();
    // batch of frames: per-frame linear memories, candidate list and counters
    __shared__ uint32_t s_red[4][8][64];
    __shared__ int s_alive[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int chunk_id, templ_slot, frame;
    coarse_block_item(chunk_id, templ_slot, frame);
    lm += (size_t)frame * lm_fs;
    cands += (size_t)frame * cap;
    counters += (size_t)frame * CTR_STRIDE;
    const int t = active[templ_slot];
    const DevTL tl = tls[(size_t)t * L + lc];
    const int npos = template_positions(tl, W, H, T);
    const int rmin = raw_min[(size_t)t * L + lc];
    const int base = chunk_id * COARSE_POS_PER_BLOCK;
    const int total = W * H;
    if (base >= total) return;
    if (base >= npos && rmin > 0) return; // beyond the span every score is 0
    const int j0 = base + lane * 16;
    const int zero_off = (int)(7 * lm_stride + (int64_t)T * T * W * H); // zero tail of the last orientation
    // lanes past the span take no part in the loads (one extra lane does: its first dword is its
    // left neighbour's bytes 16..19)
    const bool lane_on = j0 < npos + 16;
    // features [fb, fe) over the block's 1024 positions: a quarter of them per wave, partial sums meet in LDS;
    // returns the packed sums of positions j0 + 4*wave .. +3 of every lane
    auto partial = [&](int fb, int fe, uint32_t& slo, uint32_t& shi) {
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        if (base < npos) { // block-uniform: every lane takes part (features are spread over the wave's lanes)
            const int chunk = (fe - fb + 3) >> 2;
            const int f0 = fb + wave * chunk;
            int cnt = fe - f0;
            cnt = cnt < 0 ? 0 : (cnt > chunk ? chunk : cnt);
            accumulate_features16(lm, fxy + tl.feat_off + f0, foff + tl.feat_off + f0, cnt, rows, cols, lane_on ? j0 : 0,
                                  lane_on, zero_off, lo, hi);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_red[wave][i][lane] = lo[i];
            s_red[wave][4 + i][lane] = hi[i];
        }
        __syncthreads();
        slo = s_red[0][wave][lane] + s_red[1][wave][lane] + s_red[2][wave][lane] + s_red[3][wave][lane];
        shi = s_red[0][4 + wave][lane] + s_red[1][4 + wave][lane] + s_red[2][4 + wave][lane] + s_red[3][4 + wave][lane];
    };
    // Exact pruning in two phases (coarse_prune_split): the block stops after the prefix unless one of its 1024
    // positions is still alive.  The candidate set is the same as without pruning: nothing is dropped that could
    // have reached rmin.
    int k1, thr1;
    coarse_prune_split(tl.nf, rmin, k1, thr1);
    uint32_t slo, shi;
    partial(0, k1, slo, shi);
    if (k1 < tl.nf) {
        bool alive = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) alive |= (j0 + 4 * wave + k < npos) && unpack4(slo, shi, k) >= thr1;
        if (lane == 0) s_alive[wave] = 0;
        if (__builtin_amdgcn_ballot_w64(alive) != 0ull && lane == 0) s_alive[wave] = 1;
        __syncthreads(); // also: every wave has read s_red before the second phase rewrites it
        if ((s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3]) == 0) return;
        uint32_t lo2, hi2;
        partial(k1, tl.nf, lo2, hi2);
        slo += lo2;
        shi += hi2;
    }
    const int offset = T / 2 + (T % 2 - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + 4 * wave + k;
        int raw = unpack4(slo, shi, k);
        if (j >= npos) raw = 0;
        if (j < total && raw >= rmin) {
            int idx = atomicAdd(&counters[0], 1);
            if (idx < cap) {
                Cand c;
                c.t = t;
                c.x = (j % W) * T + offset;
                c.y = (j / W) * T + offset;
                c.raw = raw;
                cand_fill_next(c, tls, raw_keep, class_idx, template_id, L, lc - 1);
                cands[idx] = c;
            }
        }
    }
}

// The coarse pass for large launches (batches of frames, thousands of templates): one WAVE per (position chunk,
// template, frame), four such items per workgroup (template slots 4*blockIdx.y .. +3), no LDS and no barrier.
// With the exact pruning (accumulate_features16_pruned: re-checked after every 8 features) most waves stop after
// two or three batches, so splitting a template's features over four waves (the kernel above, which keeps
// single-frame latency low) only multiplies the per-item overhead.  Same arguments, same candidates.
__global__ __launch_bounds__(256) void k_similarity_coarse_wave(
    const uint8_t* __restrict__ lm, int64_t lm_stride, int rows, int cols, int T, int W, int H, int L, int lc,
    const DevTL* __restrict__ tls, const int32_t* __restrict__ soff, const CoarseItem* __restrict__ items,
    const int32_t* __restrict__ cfoff, int n_active, const int32_t* __restrict__ raw_keep,
    const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id, Cand* __restrict__ cands,
    int32_t* __restrict__ counters, int cap, int64_t lm_fs)
{
    raise_wave_priority();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int chunk_id, slot4, frame;
    coarse_block_item(chunk_id, slot4, frame);
    const int templ_slot = __builtin_amdgcn_readfirstlane(slot4 * 4 + wave);
    if (templ_slot >= n_active) return;
    // The item's dependent chain is two loads deep: {template record (scalar cache), its first 64 feature offsets
    // (vector load)} -- both indexed by the slot -- then the linear memories.  (Round 2: active[] -> tls[] -> raw_min[]
    // and fxy[] / foff[] -> linear memories: a third of the launch's time at threshold 100.)
    const int sel0 = cfoff[(size_t)templ_slot * 64 + lane];
    const CoarseItem it = items[templ_slot];
    lm += (size_t)frame * lm_fs;
    cands += (size_t)frame * cap;
    counters += (size_t)frame * CTR_STRIDE;
    const int t = it.t;
    const int npos = it.npos;
    const int rmin = it.rmin;
    const int base = chunk_id * COARSE_POS_PER_BLOCK;
    const int total = W * H;
    if (base >= total) return;
    if (base >= npos && rmin > 0) return; // beyond the span every score is 0
    const int j0 = base + lane * 16;
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    if (base < npos) {
        const int zero_off = (int)(7 * lm_stride + (int64_t)T * T * W * H);
        const bool lane_on = j0 < npos + 16;
        if (!accumulate_features16_pruned(lm, soff + it.soff_base, sel0, it.nf, it.e01 & 0xffff, (int)((uint32_t)it.e01 >> 16), it.e2,
                                          lane_on ? j0 : 0, lane_on, zero_off, rmin, lo, hi))
            return; // no position of this item can reach rmin
    }
    const int offset = T / 2 + (T % 2 - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + 4 * i + k;
            int raw = unpack4(lo[i], hi[i], k);
            if (j >= npos) raw = 0;
            if (j < total && raw >= rmin) {
                int idx = atomicAdd(&counters[0], 1);
                if (idx < cap) {
                    Cand c;
                    c.t = t;
                    c.x = (j % W) * T + offset;
                    c.y = (j / W) * T + offset;
                    c.raw = raw;
                    cand_fill_next(c, tls, raw_keep, class_idx, template_id, L, lc - 1);
                    cands[idx] = c;
                }
            }
        }
    }
}

// same accumulation, score map out (stage entry point sbm_similarity)
__global__ __launch_bounds__(256) void k_similarity_map(const uint8_t* __restrict__ lm, int64_t lm_stride,
                                                        int rows, int cols, int T, int W, int H, DevTL tl,
                                                        const uint32_t* __restrict__ fxy,
                                                        const int32_t* __restrict__ foff,
                                                        uint16_t* __restrict__ dst)
{
    const int npos = template_positions(tl, W, H, T);
    const int total = W * H;
    const int base = blockIdx.x * SIM_POS_PER_BLOCK;
    const int j0 = base + threadIdx.x * 4;
    uint32_t lo = 0, hi = 0;
    if (base < npos) { // block-uniform, all lanes take part
        const uint8_t* zero_addr = lm + 7 * lm_stride + (int64_t)T * T * W * H;
        const int jl = j0 < npos ? j0 : 0;
        accumulate_features(lm, fxy + tl.feat_off, foff + tl.feat_off, 0, tl.nf, rows, cols, 0, 0, jl, zero_addr, 0, lo, hi);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        if (j < total) dst[j] = (uint16_t)(j < npos ? unpack4(lo, hi, k) : 0);
    }
}

// Arrival of the working workgroups of one group of `nfr` frames (k_similarity_local); true in EVERY thread of the one
// workgroup that arrives last -- or of workgroup 0 when nobody works.  Idle workgroups do not arrive (they must not
// queue up behind the working ones).  Arrivals on one 128-byte line serialise at ~25 ns each whatever the address inside
// it, so the arrivals are spread over the lines of the group's frames (counters + CTR_STRIDE f) in three levels:
//   workgroup b -> sub-counter (frame fs = b % nfr, sub = (b / nfr) % 32)   [CTR_SUB + sub] of frame fs
//   last of a sub-counter -> frame fs                                        [CTR_ARRIVE] of frame fs   (<= 32 arrivals)
//   last of a frame -> the group                                             [CTR_ARRIVE + 1] of frame 0 (<= nfr arrivals)
// No fence: the publishing workgroup only reads out_count (device-scope atomics); the records themselves become
// visible to the host and to later kernels at the end of the kernel.
__device__ __forceinline__ bool arrive_last(int32_t* __restrict__ counters, int nfr, int n_work, bool worked, int b)
{
    if (!worked && !(n_work == 0 && b == 0)) return false; // block-uniform
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = n_work == 0;
        if (!last) {
            const int S = nfr * 32, s = b % S, fs = s % nfr, sub = s / nfr;
            const int expect = (n_work - s + S - 1) / S;                 // working blocks on this sub-counter
            const int used = n_work < S ? n_work : S;                    // sub-counters in use
            const int subs_of_frame = (used - fs + nfr - 1) / nfr;       // ... of them on frame fs
            const int frames_used = n_work < nfr ? n_work : nfr;
            int32_t* cf = counters + (size_t)fs * CTR_STRIDE;
            if (atomicAdd(&cf[CTR_SUB + sub], 1) == expect - 1)
                if (atomicAdd(&cf[CTR_ARRIVE], 1) == subs_of_frame - 1) last = atomicAdd(&counters[CTR_ARRIVE + 1], 1) == frames_used - 1;
        }
        s_last = last ? 1 : 0;
    }
    __syncthreads();
    return s_last != 0;
}

// End of the emitting kernel: the last WORKING block to finish (arrival counter counters[4]) writes the
// overflow status and mirrors the final {count, overflow} pair to the caller's mirror (plain stores:
// it may be pinned host memory).  Only blocks that had a candidate arrive — atomics on one address
// serialise at ~25 ns each, so idle blocks must not queue up behind them; with no candidate at all
// block 0 publishes.  out_count itself is only ever updated with device-scope atomics, so the atomic
// read below sees every block's increments once all of them have arrived.
__device__ __forceinline__ void publish_counts(int32_t* __restrict__ counters, int n_all, int n_work_blocks, bool worked,
                                               int cand_cap, int32_t* __restrict__ out_count,
                                               int32_t* __restrict__ mirror_count, int slot)
{
    if (!worked && !(n_work_blocks == 0 && slot == 0)) return; // block-uniform
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = n_work_blocks == 0;
        if (!last) {
            // two-level arrival (32 sub-counters): arrivals on one address serialise at ~25 ns each
            const int sub = slot & 31;
            const int expect = (n_work_blocks - sub + 31) / 32; // working blocks with this residue
            const int groups = n_work_blocks < 32 ? n_work_blocks : 32;
            // no fence: the last block only reads out_count (device-scope atomics); the records themselves
            // become visible to the host and to later kernels at the end of the kernel
            if (atomicAdd(&counters[CTR_SUB + sub], 1) == expect - 1) last = atomicAdd(&counters[CTR_ARRIVE], 1) == groups - 1;
        }
        if (last) {
            const int n_out = atomicAdd(out_count, 0);
            const int flag = n_all > cand_cap ? 1 : 0;
            out_count[1] = flag;
            if (mirror_count) {
                mirror_count[0] = n_out;
                mirror_count[1] = flag;
            }
        }
    }
}

// ---- refinement pass on the strip-interleaved plane: four features per wave instruction ----------------------
// response4 with a per-lane orientation (the four 16-lane groups of a wave work on four different features)
// Round 4: seven instructions per dword instead of the ten the compiler made of the expression above -- three shifts, then
// a two-bit code per byte (bit 0 = the orientation's own bit, bit 1 = a circular neighbour's: one v_and, one v_bitop3
// "(b | c) & m", one v_lshl_or) looked up in the four-entry byte table {0, 4, 3, 4} by v_perm_b32 (a selector byte 0..3
// picks that byte of the table).
__device__ __forceinline__ uint32_t response4v(uint32_t v, uint32_t o, uint32_t o1, uint32_t o7)
{
    const uint32_t m = 0x01010101u;
    const uint32_t self = (v >> o) & m;
    uint32_t nb;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xa8" : "=v"(nb) : "v"(v >> o1), "v"(v >> o7), "v"(m)); // (b | c) & m
    const uint32_t code = (nb << 1) | self;
    return __builtin_amdgcn_perm(0u, 0x04030400u, code);
}

struct __attribute__((aligned(16))) u128_a16 { uint32_t x, y, z, w; };

// Lane = (feature slot g = lane >> 4, patch row r = lane & 15): a step of the wave adds FOUR features to the 16 x 16
// patch; a lane loads the two aligned 16-byte strip rows that hold its row of the patch (strip s and s + 1: 32 bytes
// of which the 16 starting at column c0 = gx0 & 15 are the patch row) and keeps 16 positions (4 packed dwords).
// The dword misalignment Q = c0 >> 2 picks WHICH of the eight loaded dwords are used and must not cost a per-lane
// select: the host sorts every template's features by (x / T) & 15, the class that fixes c0 for a given candidate
// (c0 = (class + ox / T) & 15), and wave w of the workgroup takes the four (cyclically consecutive) classes whose Q
// is w & 3 -- Q is a template parameter of the body.  Per feature that is 2 loads per FOUR features instead of 2 per
// feature, no per-feature scalar address arithmetic, and ~15 vector instructions instead of ~25.
// cls[0..16]: class start offsets inside the (sorted) feature range of this template level.
// Result: lo[i] / hi[i] = packed u16 sums of columns 4i .. 4i+3 of row r, valid in lanes 0..15 of every wave
// (the wave's share of the features), to be added over the workgroup's waves.
template <int LOG2T, int Q, int NPARTS>
__device__ __forceinline__ void accumulate_rows16_q(const uint8_t* __restrict__ lmc, const uint32_t* __restrict__ fxy,
                                                    const uint8_t* __restrict__ flabel, const uint16_t* __restrict__ cls,
                                                    int part, int rows, int cols, int ox, int oy, int W, int H,
                                                    uint32_t (&lo)[4], uint32_t (&hi)[4])
{
    constexpr int T = 1 << LOG2T;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, r = lane & 15;
    const int X15 = (ox >> LOG2T) & 15;
    const int k0 = (4 * Q - X15) & 15; // classes k0 .. k0+3 (mod 16) end up with dword misalignment Q
    int a0 = cls[k0], a1, b1 = 0;      // up to two runs of the sorted list: [a0, a1) and [0, b1)
    if (k0 <= 12) a1 = cls[k0 + 4];
    else a1 = cls[16], b1 = cls[k0 - 12];
    a0 = __builtin_amdgcn_readfirstlane(a0);
    a1 = __builtin_amdgcn_readfirstlane(a1);
    b1 = __builtin_amdgcn_readfirstlane(b1);
    const int len1 = a1 - a0, n = len1 + b1;
    const int per = ((((n + 3) >> 2) + NPARTS - 1) / NPARTS) << 2; // features per part (whole steps)
    int j0 = per * part, j1 = j0 + per;
    j0 = j0 < n ? j0 : n;
    j1 = j1 < n ? j1 : n;
    const uint32_t zero16 = (uint32_t)(((int64_t)T * T * W * H) >> 4); // the plane's zero tail (>= H*16 + 272 bytes)
    const uint32_t pitch = (uint32_t)H * 16u;                         // strip s -> strip s + 1, same row
    const uint32_t row_off = (uint32_t)r * 16u;
    uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = hi[i] = 0;
    int pending = 0;
    for (int jb = j0; jb < j1; jb += 64) {
        // lane = feature: (strip row address / 16, column misalignment | label)
        uint32_t recA = zero16, recB = 0;
        const int jj = jb + lane;
        if (jj < j1) {
            const int idx = jj < len1 ? a0 + jj : jj - len1;
            const uint32_t xy = fxy[idx];
            const uint32_t lab = flabel[idx];
            const int x = (int)(xy & 0xffff) + ox, y = (int)(xy >> 16) + oy;
            const int gx0 = x >> LOG2T, gy0 = y >> LOG2T;
            if (x >= 0 && y >= 0 && x < cols && y < rows)
                recA = (uint32_t)(lm_strip_offset(((y & (T - 1)) << LOG2T) | (x & (T - 1)), gy0, gx0 & ~15, W, H) >> 4);
            recB = (uint32_t)(gx0 & 3) | (lab << 2);
        }
        const int nbf = j1 - jb < 64 ? j1 - jb : 64;
        auto batch = [&](auto N, int u) { // N steps = 4 N features, u = first feature of the batch (multiple of 4)
            constexpr int nst = decltype(N)::value;
            u128_a16 A[nst], B[nst];
            uint32_t rb[nst];
#pragma unroll
            for (int k = 0; k < nst; ++k) {
                const int f = (u + 4 * k + g) << 2; // lanes past the last feature hold the zero record
                const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(f, (int)recA);
                rb[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(f, (int)recB);
                const uint8_t* p = lmc + ((size_t)a << 4) + row_off;
                A[k] = *(const u128_a16*)p;
                B[k] = *(const u128_a16*)(p + pitch);
            }
#pragma unroll
            for (int k = 0; k < nst; ++k) {
                const uint32_t sh = rb[k] & 3u, o = rb[k] >> 2, o1 = (o + 1u) & 7u, o7 = (o + 7u) & 7u;
                const uint32_t d0 = Q == 0 ? A[k].x : Q == 1 ? A[k].y : Q == 2 ? A[k].z : A[k].w;
                const uint32_t d1 = Q == 0 ? A[k].y : Q == 1 ? A[k].z : Q == 2 ? A[k].w : B[k].x;
                const uint32_t d2 = Q == 0 ? A[k].z : Q == 1 ? A[k].w : Q == 2 ? B[k].x : B[k].y;
                const uint32_t d3 = Q == 0 ? A[k].w : Q == 1 ? B[k].x : Q == 2 ? B[k].y : B[k].z;
                const uint32_t d4 = Q == 0 ? B[k].x : Q == 1 ? B[k].y : Q == 2 ? B[k].z : B[k].w;
                acc[0] += response4v(__builtin_amdgcn_alignbyte(d1, d0, sh), o, o1, o7);
                acc[1] += response4v(__builtin_amdgcn_alignbyte(d2, d1, sh), o, o1, o7);
                acc[2] += response4v(__builtin_amdgcn_alignbyte(d3, d2, sh), o, o1, o7);
                acc[3] += response4v(__builtin_amdgcn_alignbyte(d4, d3, sh), o, o1, o7);
            }
            pending += nst;
            if (pending + 4 > 63) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] += acc[i] & 0x00ff00ffu;
                    hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
                    acc[i] = 0;
                }
                pending = 0;
            }
        };
        int u = 0;
        for (; u + 16 <= nbf; u += 16) batch(std::integral_constant<int, 4>{}, u);
        if (nbf - u > 8) { batch(std::integral_constant<int, 2>{}, u); u += 8; }
        if (nbf - u > 4) { batch(std::integral_constant<int, 2>{}, u); u += 8; }
        else if (nbf - u > 0) batch(std::integral_constant<int, 1>{}, u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] += acc[i] & 0x00ff00ffu;
        hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
    }
}

// 16x16 patch of one candidate: lane = (row = lane>>2, 4 columns); the block's
// LOCAL_WAVES waves each take a contiguous slice of the features and the partial
// sums meet in LDS.  Result (packed u16) valid in wave 0.
constexpr int LOCAL_WAVES = 16;
// COMPACT: 0 eight response planes, 1 one plane of spread bytes, 2 the same strip-interleaved, 3 bit strips (LW = 1)
// T8: the strip path's stride (COMPACT == 2: T = 4 or 8, a template parameter of its inner loop) -- a launch instantiates
// only its own (half the code of the kernel, which is what its waves have to keep in the instruction cache)
template <int COMPACT = 0, int LW = LOCAL_WAVES, bool T8 = false>
__device__ __forceinline__ void local_patch(const uint8_t* __restrict__ lm, int64_t lm_stride, const DevTL tl,
                                            const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,
                                            int rows, int cols, int W, int H, int T, int ox, int oy,
                                            uint32_t (*s_part)[2][64], uint32_t& lo, uint32_t& hi,
                                            const uint8_t* __restrict__ flabel = nullptr,
                                            const uint16_t* __restrict__ cls = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (COMPACT == 2) {
        // strip plane: fxy / flabel are the class-sorted copies, cls the 17 class offsets of this template level;
        // wave w takes the features whose dword misalignment is w & 3 (and, with 16 waves, a quarter of those)
        static_assert(COMPACT != 2 || (LW & 3) == 0, "one wave (or LW / 4) per dword misalignment");
        const int q = __builtin_amdgcn_readfirstlane(wave) & 3, part = __builtin_amdgcn_readfirstlane(wave) >> 2;
        uint32_t l4[4], h4[4];
        const uint32_t* fx = fxy + tl.feat_off;
        const uint8_t* fl = flabel + tl.feat_off;
#define SBM_ROWS16(LOG2T_)                                                                                              \
        do {                                                                                                            \
            if (q == 0) accumulate_rows16_q<LOG2T_, 0, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4);      \
            else if (q == 1) accumulate_rows16_q<LOG2T_, 1, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4); \
            else if (q == 2) accumulate_rows16_q<LOG2T_, 2, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4); \
            else accumulate_rows16_q<LOG2T_, 3, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4);             \
        } while (0)
        if (T8) SBM_ROWS16(3);
        else SBM_ROWS16(2);
#undef SBM_ROWS16
        // the four feature slots of the wave (lanes r, r+16, r+32, r+48), then the waves of the workgroup; the last
        // read also turns (row, 16 columns) per lane into the (row, 4 columns) per lane the caller scans
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            l4[i] += (uint32_t)__shfl_xor((int)l4[i], 16, 64);
            h4[i] += (uint32_t)__shfl_xor((int)h4[i], 16, 64);
            l4[i] += (uint32_t)__shfl_xor((int)l4[i], 32, 64);
            h4[i] += (uint32_t)__shfl_xor((int)h4[i], 32, 64);
        }
        uint32_t(*s8)[8][16] = (uint32_t(*)[8][16])s_part; // same bytes: [LW][2][64] == [LW][8][16]
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s8[wave][i][lane] = l4[i];
                s8[wave][4 + i][lane] = h4[i];
            }
        }
        __syncthreads();
        if (wave == 0) {
            lo = hi = 0;
#pragma unroll
            for (int w = 0; w < LW; ++w) {
                lo += s8[w][lane & 3][lane >> 2];
                hi += s8[w][4 + (lane & 3)][lane >> 2];
            }
        }
        __syncthreads();
        return;
    }
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    const int delta = (oy / T) * W + ox / T + r * W + c4;
    const uint8_t* zero_addr = lm + (COMPACT ? 0 : 7) * lm_stride + (int64_t)T * T * W * H;
    const int chunk = (tl.nf + LW - 1) / LW;
    const int f0 = wave * chunk;
    int cnt = tl.nf - f0;
    cnt = cnt < 0 ? 0 : (cnt > chunk ? chunk : cnt);
    accumulate_features<COMPACT != 0>(lm, fxy + tl.feat_off, foff + tl.feat_off, f0, cnt, rows, cols, ox, oy, delta, zero_addr, 0, lo, hi,
                                      COMPACT ? flabel + tl.feat_off : nullptr, (int)lm_stride);
    s_part[wave][0][lane] = lo;
    s_part[wave][1][lane] = hi;
    __syncthreads();
    if (wave == 0) {
        lo = hi = 0;
#pragma unroll
        for (int w = 0; w < LW; ++w) {
            lo += s_part[w][0][lane];
            hi += s_part[w][1][lane];
        }
    }
    __syncthreads();
}

// ORDER: which candidate a workgroup takes.
//   0  grid = (frames, slots): workgroup (f, s) takes candidates s, s + slots, ... of frame f; the frame is the FAST
//      grid dimension, so the workgroups dispatched first are the low slots of every frame -- the ones that have a
//      candidate -- and the idle slots come last.  Best while the batch's strip planes fit the L2s.
//   2  grid = (workgroups, groups of up to 64 frames): the candidates of a group's frames form ONE list in frame-major
//      order and workgroup b takes entries b, b + gridDim.x, ...: the workgroups running at any moment are on one or
//      two frames, whose strip planes then stay in the L2s (with order 0 a 64-frame 1080p batch has 64 planes in
//      flight: 128 MiB against 8 x 4 MiB of L2), and the grid is as full as the batch's candidate count, not as each
//      frame's.  Costs a running sum of the frames' counts per wave before the first candidate record can be fetched
//      and a third arrival level: config 5 (64 frames, ~1400 candidates each) 1878 -> 1753 us per step, but 16 tiled
//      1024^2 frames 200 -> 207 us (tools/r03_local_order.sh), so the host picks by the planes' total size.
//   (the frame as the SLOW dimension of order 0's grid was measured too: between the two on config 5, 1825 us)
#define SBM_LOCAL_PARAMS                                                                                                    \
    const uint8_t* __restrict__ lm, int64_t lm_stride, int rows, int cols, int T, int W, int H, int L, int l,              \
    const DevTL* __restrict__ tls, const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,                     \
    const int32_t* __restrict__ raw_keep, const int32_t* __restrict__ class_idx,                                           \
    const int32_t* __restrict__ template_id, Cand* __restrict__ cands, int32_t* __restrict__ counters,                     \
    int cand_cap, int is_last, sbm_match_rec* __restrict__ out, int32_t* __restrict__ out_count,                           \
    int out_cap, sbm_match_rec* __restrict__ mirror_out, int32_t* __restrict__ mirror_count, int collect_stats,            \
    int64_t lm_fs, const uint8_t* __restrict__ flabel, const uint16_t* __restrict__ fcls, int frames
#define SBM_LOCAL_ARGS                                                                                                      \
    lm, lm_stride, rows, cols, T, W, H, L, l, tls, fxy, foff, raw_keep, class_idx, template_id, cands, counters, cand_cap,  \
    is_last, out, out_count, out_cap, mirror_out, mirror_count, collect_stats, lm_fs, flabel, fcls, frames

template <int COMPACT, int LW, int ORDER, bool T8 = false>
__device__ __forceinline__ void similarity_local_body(SBM_LOCAL_PARAMS)
{
    // The refinement pass runs at the gradient kernels' priority when it follows them (a match call: beside other batches'
    // gradient launches its few working workgroups at a raised priority cost more than they gain -- scene frame 111.9 ->
    // 110.2 us per step without, tools/r03_ab3.sh), and at the raised one in a template-loop-only call, where the other
    // batches in flight run coarse passes (flag bit 1 of collect_stats).
    if (collect_stats & 2) raise_wave_priority();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ORDER 2: lane = frame of the group: candidate counts, their running sums, and from those (frame, index) of entry g
    const int f_first = ORDER == 2 ? (int)blockIdx.y * 64 : 0;
    const int nfr = frames - f_first < 64 ? frames - f_first : 64;
    int n_all_l = 0, incl = 0, excl = 0;
    // ORDER 0: one frame per workgroup; the candidate count and the workgroup's first candidate are fetched together
    // (the record is read speculatively: the buffer always exists, the count decides whether it is used)
    int slot = ORDER == 0 ? blockIdx.y : blockIdx.x;
    const int n_slots = ORDER == 0 ? gridDim.y : gridDim.x;
    size_t frame_wg = ORDER == 0 ? blockIdx.x : 0;
    if (COMPACT == 3 && ORDER == 0 && (collect_stats & 4) && (gridDim.x & 7) == 0) {
        // Workgroups are dealt to the 8 XCDs round-robin by linear id (the frame is the fast grid dimension: XCD x gets the
        // frames x, x + 8, ... anyway).  Flag bit 2: give it those frames ONE AFTER THE OTHER instead of slot by slot, so that
        // its 4 MiB L2 holds one frame's bit strips while that frame's candidates are refined.  Pays on large batches of
        // large frames (64 x 1080p: 263 -> 228 us, and better than the list order); on 16 frames of 1024 x 1024, whose two
        // frames per XCD fit its L2 together, it only delays the second frame's candidates (21.8 -> 25.6 us).  Any
        // bijection of the grid is correct.
        const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y, j = lin >> 3, fl = j / gridDim.y;
        frame_wg = (lin & 7) + 8 * fl;
        slot = (int)(j - fl * gridDim.y);
    }
    collect_stats &= 1;
    Cand c_first;
    int tot, n_all_wg = 0;
    if (ORDER == 2) {
        if (lane < nfr) n_all_l = counters[(size_t)(f_first + lane) * CTR_STRIDE];
        const int n_l = n_all_l < cand_cap ? n_all_l : cand_cap;
        incl = n_l;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            incl += lane >= d ? o : 0;
        }
        excl = incl - n_l;
        tot = __builtin_amdgcn_readlane(incl, 63);
    } else {
        c_first = cands[frame_wg * cand_cap + (slot < cand_cap ? slot : 0)];
        n_all_wg = counters[frame_wg * CTR_STRIDE];
        tot = n_all_wg < cand_cap ? n_all_wg : cand_cap;
    }
    const int g_first = ORDER == 2 ? (int)blockIdx.x : slot, g_step = ORDER == 2 ? (int)gridDim.x : n_slots;
    __shared__ uint32_t s_part[LW][2][64];
    const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
    unsigned long long stat_bytes = 0;
    for (int g = g_first; g < tot; g += g_step) {
        size_t frame = frame_wg;
        int ci = g;
        if (ORDER == 2) {
            // first frame whose running sum passes g (g < tot: there is one); wave-uniform, and said so to the compiler
            const int fl = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(g < incl)));
            ci = __builtin_amdgcn_readfirstlane(g - __builtin_amdgcn_readlane(excl, fl));
            frame = (size_t)(f_first + fl);
        }
        Cand* const cand_p = cands + frame * cand_cap + ci;
        Cand c = ORDER != 2 && g == g_first ? c_first : *cand_p;
        if constexpr (COMPACT == 3) {
            // every lane read the same record: said so, the record and what is derived from it (clamps, the template record,
            // the Match) live in scalar registers across the feature loop instead of being spilled around it
            int32_t* w = (int32_t*)&c;
#pragma unroll
            for (int i = 0; i < (int)(sizeof(Cand) / 4); ++i) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
        }
        if (c.raw < 0) continue; // dropped at a coarser level (uniform per block)
        DevTL tl;
        tl.width = c.next_width;
        tl.height = c.next_height;
        tl.nf = c.next_nf;
        tl.feat_off = c.next_feat_off;
        int x = c.x * 2 + 1, y = c.y * 2 + 1;
        const int max_x = cols - tl.width - border, max_y = rows - tl.height - border;
        x = x < border ? border : x;
        y = y < border ? border : y;
        x = x > max_x ? max_x : x;
        y = y > max_y ? max_y : y;
        const int ox = (x / T - 8) * T, oy = (y / T - 8) * T;
        uint32_t best = 0;
        if constexpr (COMPACT == 3) {
            // bit strips (sbm_local_bits.h): one wave per candidate (LW == 1), the maximum comes back in key form
            const uint32_t* lb = (const uint32_t*)(lm + frame * lm_fs);
            const uint32_t* fx = fxy + tl.feat_off;
            const uint8_t* fl = flabel + tl.feat_off;
            // counter planes per slot (a slot counts ceil(nf / 4) features)
            if (tl.nf <= 252) best = local_best_bits<6>(lb, fx, fl, tl.nf, rows, cols, ox, oy, W, H);
            else if (tl.nf <= 2044) best = local_best_bits<9>(lb, fx, fl, tl.nf, rows, cols, ox, oy, W, H);
            else best = local_best_bits<12>(lb, fx, fl, tl.nf, rows, cols, ox, oy, W, H);
        } else {
        uint32_t lo, hi;
        local_patch<COMPACT, LW, T8>(lm + frame * lm_fs, lm_stride, tl, fxy, foff, rows, cols, W, H, T, ox, oy, s_part, lo, hi, flabel,
                                 COMPACT == 2 ? fcls + ((size_t)c.t * L + l) * 17 : nullptr);
        if (wave != 0) continue;
        // first maximum in row-major order, strict '>' from 0 (:1265-1282): maximise (raw, -position)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t raw = (uint32_t)unpack4(lo, hi, k);
            uint32_t p = (uint32_t)(lane * 4 + k);
            uint32_t key = (raw << 8) | (255u - p);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            uint32_t o = __shfl_xor(best, s, 64);
            best = o > best ? o : best;
        }
        }
        if (lane == 0) {
            if (frame == 0) stat_bytes += (unsigned long long)tl.nf * (COMPACT == 3 ? 128ull : 256ull); // refinement bytes (sbm_get_stats: frame 0)
            const int raw = (int)(best >> 8);
            int br = -1, bc = -1;
            if (raw > 0) {
                int p = 255 - (int)(best & 255u);
                br = p >> 4;
                bc = p & 15;
            }
            const int nx = (x / T - 8 + bc) * T + offset;
            const int ny = (y / T - 8 + br) * T + offset;
            const bool keep = raw >= c.next_keep;
            if (is_last) {
                if (keep) {
                    int idx = atomicAdd(out_count + frame * 2, 1);
                    if (idx < out_cap) {
                        sbm_match_rec m;
                        m.x = nx;
                        m.y = ny;
                        m.similarity = __fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * tl.nf));
                        m.raw = raw;
                        m.class_idx = c.class_idx;
                        m.template_id = c.template_id;
                        out[frame * out_cap + idx] = m;
                        if (mirror_out) mirror_out[frame * out_cap + idx] = m; // e.g. pinned host memory: no copy engine involved
                    }
                }
            } else {
                c.x = nx;
                c.y = ny;
                c.raw = keep ? raw : -1;
                if (keep) cand_fill_next(c, tls, raw_keep, class_idx, template_id, L, l - 1);
                *cand_p = c;
            }
        }
    }
    // one addition per workgroup, not per candidate (a per-candidate atomic on one address stretched the very launch
    // it was measuring)
    if (collect_stats && threadIdx.x == 0 && stat_bytes) atomicAdd((unsigned long long*)(counters + 2), stat_bytes);
    if (is_last && ORDER == 2) {
        // the last working workgroup of the group publishes every frame's count
        const int n_work = tot < (int)gridDim.x ? tot : (int)gridDim.x;
        const bool last = arrive_last(counters + (size_t)f_first * CTR_STRIDE, nfr, n_work, (int)blockIdx.x < n_work, (int)blockIdx.x);
        if (last && wave == 0 && lane < nfr) {
            const size_t frame = (size_t)(f_first + lane);
            const int n_out = atomicAdd(out_count + frame * 2, 0);
            const int flag = n_all_l > cand_cap ? 1 : 0;
            out_count[frame * 2 + 1] = flag;
            if (mirror_count) {
                mirror_count[frame * 2] = n_out;
                mirror_count[frame * 2 + 1] = flag;
            }
        }
    } else if (is_last) {
        const int n_work = tot < n_slots ? tot : n_slots;
        publish_counts(counters + frame_wg * CTR_STRIDE, n_all_wg, n_work, slot < tot, cand_cap, out_count + frame_wg * 2,
                       mirror_count ? mirror_count + frame_wg * 2 : nullptr, slot);
    }
}

template <int COMPACT, int LW, int ORDER, bool T8 = false>
__global__ __launch_bounds__(64 * LW) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_similarity_local(SBM_LOCAL_PARAMS)
{
    similarity_local_body<COMPACT, LW, ORDER, T8>(SBM_LOCAL_ARGS);
}

// the bit-strip form (sbm_local_bits.h): one wave per candidate, a window of 64 features = 32 loads per lane in flight --
// the register budget is the compiler's (launch bound 64 threads), not the byte form's six waves per SIMD
template <int ORDER>
__global__ __launch_bounds__(64) void k_similarity_local_bits(SBM_LOCAL_PARAMS)
{
    similarity_local_body<3, 1, ORDER, false>(SBM_LOCAL_ARGS);
}
#undef SBM_LOCAL_PARAMS
#undef SBM_LOCAL_ARGS

// stage entry point: one 16x16 patch
__global__ __launch_bounds__(64 * LOCAL_WAVES) void k_similarity_local_patch(const uint8_t* __restrict__ lm, int64_t lm_stride,
                                                                int rows, int cols, int T, int W, int H, DevTL tl,
                                                                const uint32_t* __restrict__ fxy,
                                                                const int32_t* __restrict__ foff, int cx, int cy,
                                                                uint16_t* __restrict__ dst)
{
    __shared__ uint32_t s_part[LOCAL_WAVES][2][64];
    const int ox = (cx / T - 8) * T, oy = (cy / T - 8) * T;
    uint32_t lo, hi;
    local_patch<0, LOCAL_WAVES>(lm, lm_stride, tl, fxy, foff, rows, cols, W, H, T, ox, oy, s_part, lo, hi);
    if (threadIdx.x < 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[threadIdx.x * 4 + k] = (uint16_t)unpack4(lo, hi, k);
    }
}

// single-level pyramids: coarse candidates are the final matches
__global__ __launch_bounds__(256) void k_emit_coarse(const Cand* __restrict__ cands,
                                                     int32_t* __restrict__ counters, int cand_cap,
                                                     const DevTL* __restrict__ tls, int L, int lc,
                                                     const int32_t* __restrict__ class_idx,
                                                     const int32_t* __restrict__ template_id,
                                                     sbm_match_rec* __restrict__ out,
                                                     int32_t* __restrict__ out_count, int out_cap,
                                                     sbm_match_rec* __restrict__ mirror_out,
                                                     int32_t* __restrict__ mirror_count)
{
    raise_wave_priority();
    { // batch of frames: frame = blockIdx.y
        const size_t frame = blockIdx.y;
        cands += frame * cand_cap;
        counters += frame * CTR_STRIDE;
        out += frame * out_cap;
        out_count += frame * 2;
        if (mirror_out) mirror_out += frame * out_cap;
        if (mirror_count) mirror_count += frame * 2;
    }
    const int n_all = counters[0];
    const int n = n_all < cand_cap ? n_all : cand_cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const Cand c = cands[i];
        const DevTL tl = tls[(size_t)c.t * L + lc];
        int idx = atomicAdd(out_count, 1);
        if (idx < out_cap) {
            sbm_match_rec m;
            m.x = c.x;
            m.y = c.y;
            m.similarity = __fdiv_rn(__fmul_rn((float)c.raw, 100.f), (float)(4 * tl.nf));
            m.raw = c.raw;
            m.class_idx = class_idx[c.t];
            m.template_id = template_id[c.t];
            out[idx] = m;
            if (mirror_out) mirror_out[idx] = m;
        }
    }
    {
        const int per = (int)gridDim.x * 256;
        const int n_work = (n + 255) / 256 < (int)gridDim.x ? (n + 255) / 256 : (int)gridDim.x;
        (void)per;
        publish_counts(counters, n_all, n_work, (int)blockIdx.x < n_work, cand_cap, out_count, mirror_count, (int)blockIdx.x);
    }
}


} // namespace sbm
