// sbm_coarse_bits.h — the coarse pass on BIT PLANES (round 4), gfx950.
// Reference functions replaced (file:line in ddcr/shape_based_matching):
//   k_pack_bitplanes          computeResponseMaps' value set {0, 3, 4} (line2Dup.cpp:632-635, 687) as two bits per position
//   k_similarity_coarse_bits  similarity / similarity_64 + candidate scan   line2Dup.cpp:807-858, 924-984, 1199-1216
//
// The reference adds one BYTE per (position, feature): response 4 if the feature's orientation o is set in the spread
// byte, 3 if a circular neighbour of o is, else 0.  That byte carries two bits of information:
//     any(o, p)   = response > 0        (plane o      of the bit linear memory)
//     exact(o, p) = response == 4       (plane 8 + o)
// and the reference's sum is   raw(p) = 3 * #any + #exact   over the template's features -- exactly.  Both planes use the
// byte linear memory's flat order (bit j of plane o <-> byte j of LM[o], zero tail included), so a feature's byte offset
// (k_prep_features) is also its bit offset and the row-overrun semantics of SURVEY 8a-6 carry over unchanged.
//
// Work item = (template, frame, 2016 consecutive positions): ONE wave, lane i owns positions 32 i .. 32 i + 31 (lane 63
// only supplies lane 62's upper dword).  Per feature one 4-byte load per lane, the upper dword from the right neighbour
// (DPP), one v_alignbit by the offset's bit misalignment (wave-uniform), and the 32 positions' bits enter a bit-sliced
// carry-save counter (Harley-Seal: 7 full adders per 8 features, two v_bitop3_b32 each; sbm_bitplane_math.h).
//
//   Pass A (screening, exact pruning): counts MISSES of the any-plane.  raw <= 4 * (nf - misses), so a position with more
//   than M = floor((4 nf - rmin) / 4) misses can never reach rmin.  The counter starts at 2^P - 1 - M: the carry out of
//   its top plane IS "dead", kept sticky; every 8 features the wave stops if none of its positions is alive.
//   Pass B (only for items with a survivor; only the lanes that hold one load): counts the exact-plane hits.
//   Then per surviving position: raw = 3 * (nf - misses) + exact; candidates with raw >= rmin are appended.
//
// Nothing that could reach rmin is dropped and raw is the reference's sum: the candidate multiset equals the byte
// kernels' (tests/test_gpu_coarse_pruning.py, tests/test_gpu_configs.py, tools/fuzz_match.py).
#pragma once
#include "sbm_common.h"
#include "sbm_similarity_kernels.h"
#include "sbm_bitplane_math.h"

namespace sbm {

constexpr int CB_POS = 63 * 32; // positions per work item

// byte planes -> bit planes, flat: thread = 32 consecutive positions of one orientation plane of one frame.
// n_dwords = dwords of one bit plane that hold data (the tail stays zero from the allocation).
__global__ __launch_bounds__(256) void k_pack_bitplanes(const uint8_t* __restrict__ lm, int64_t lm_stride, int64_t lm_fs,
                                                        uint32_t* __restrict__ blm, int64_t blm_fs_dwords, int n_dwords)
{
    raise_wave_priority();
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= n_dwords) return;
    const int o = blockIdx.y;
    const size_t frame = blockIdx.z;
    const uint4* src = (const uint4*)(lm + frame * lm_fs + (size_t)o * lm_stride + (size_t)d * 32);
    const uint4 a = src[0], b = src[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t any = 0, exact = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // response bytes are 0, 3 (0b011) or 4 (0b100): bit 2 = exact, bit 2 | bit 0 = any
        uint32_t e = (w[i] >> 2) & 0x01010101u, n = (w[i] | (w[i] >> 2)) & 0x01010101u;
        e |= e >> 7;  // byte 1's flag -> bit 1 (and byte 3's -> bit 17)
        e |= e >> 14; // byte 2's -> bit 2, byte 3's -> bit 3
        n |= n >> 7;
        n |= n >> 14;
        exact |= (e & 15u) << (4 * i);
        any |= (n & 15u) << (4 * i);
    }
    const int64_t plane_dwords = lm_stride >> 5;
    uint32_t* dst = blm + frame * blm_fs_dwords;
    dst[(int64_t)o * plane_dwords + d] = any;
    dst[(int64_t)(8 + o) * plane_dwords + d] = exact;
}

// spread bytes -> bit planes, flat (round 4): the producer of the match entry points on grids where the fused form of
// k_build_lm_rows does not apply (W * H % 256 != 0, e.g. 1920 x 1072 frames: 120 x 67 cells at the coarsest level).  The
// linear-memory launch then writes the coarsest level as ONE plane of spread bytes (1 byte per pixel instead of the 8
// response planes) and this kernel turns 32 of them per thread into the 16 planes' dwords with the fused producer's
// register transpose (bytes32_to_bitplanes): 1 byte per pixel read, 2 written, where k_pack_bitplanes reads 8.
__global__ __launch_bounds__(256) void k_pack_bitplanes_spread(const uint8_t* __restrict__ spread, int64_t spread_fs, int64_t lm_stride,
                                                               uint32_t* __restrict__ blm, int64_t blm_fs_dwords, int n_dwords)
{
    raise_wave_priority();
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= n_dwords) return;
    const size_t frame = blockIdx.y;
    const uint4* src = (const uint4*)(spread + frame * spread_fs + (size_t)d * 32);
    const uint4 a = src[0], b = src[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    bytes32_to_bitplanes(w);
    const int64_t plane_dwords = lm_stride >> 5;
    uint32_t* dst = blm + frame * blm_fs_dwords + d;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const uint32_t e = w[bitplane_reg(o)];
        dst[(int64_t)o * plane_dwords] = w[bitplane_reg((o + 7) & 7)] | e | w[bitplane_reg((o + 1) & 7)];
        dst[(int64_t)(8 + o) * plane_dwords] = e;
    }
}

// One pass over the template's nf features on one bit-plane set.  ub: the plane set of this frame (wave-uniform);
// li: this lane's first dword of the item (item base / 32 + DW * lane), or 0 for a lane whose loads do not matter (it
// then reads the features' own first dwords: one more cache line per load, no branch).  MISSES: count zero bits and
// stop when every position is dead, else count one bits.  DW: dwords (32 positions each) a lane owns.
// soff: the template's feature offsets (bits == bytes, see above); sel: the first 64 of them, one per lane.
// Returns false when pass A ended with no position alive.
typedef uint32_t cb_u32x2 __attribute__((ext_vector_type(2)));

template <int P, bool MISSES, bool WIDE, int DW>
__device__ __forceinline__ bool bit_pass(const uint32_t* __restrict__ ub, uint32_t li, const int32_t* __restrict__ soff, int sel, int nf,
                                         int zero_off, uint32_t (&c)[DW][P], uint32_t (&dead)[DW])
{
    const int lane = threadIdx.x & 63;
    const uint32_t li4 = li << 2;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ub, 0, 0x7fffffff, 0x00020000);
    // N = 8 or 16 features: loads first (all in flight together), then the arithmetic.  TAIL: features g + k >= nf are
    // padding (their offsets point at the zero tail) and must not count as misses.  A batch never crosses a multiple of
    // 64 features (the offsets of 64 features live in `sel`, one per lane).
    auto batch = [&](int g, auto n_, auto tail) {
        constexpr int N = decltype(n_)::value;
        constexpr bool TAIL = decltype(tail)::value;
        uint32_t lo[N][DW];
        int sh[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane(sel, (g + k) & 63);
            sh[k] = (int)(o & 31u);
            // buffer load: plane set (resource) + this lane's byte offset (VGPR) + the feature's dword (SGPR): no address arithmetic
            if (DW == 1) {
                lo[k][0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)li4, (int)((o >> 5) << 2), 0);
            } else {
                const cb_u32x2 v2 = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)li4, (int)((o >> 5) << 2), 0);
                lo[k][0] = v2.x;
                lo[k][DW - 1] = v2.y;
            }
        }
        constexpr int G = N == 32 ? 32 : 8; // slices per counter update (32: one ripple for four groups of eight)
#pragma unroll
        for (int h = 0; h < N; h += G) {
            uint32_t x[DW][G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                // the dword after the lane's own = lane i+1's first (wave_shl:1; lane 63 reads 0 -- it owns no position)
                const uint32_t nx = (uint32_t)__builtin_amdgcn_mov_dpp((int)lo[h + k][0], 0x130, 0xf, 0xf, true);
#pragma unroll
                for (int w = 0; w < DW; ++w) {
                    const uint32_t v = __builtin_amdgcn_alignbit(w + 1 < DW ? lo[h + k][w + 1] : nx, lo[h + k][w], (uint32_t)sh[h + k]);
                    x[w][k] = v; // pass A counts the ZERO bits: the inversion is folded into the adders' truth tables
                    if (TAIL && g + h + k >= nf) x[w][k] = MISSES ? ~0u : 0u; // padding counts nothing (wave-uniform)
                }
            }
#pragma unroll
            for (int w = 0; w < DW; ++w) {
                uint32_t carry;
                if constexpr (G == 32) carry = bitslice_add32<P, MISSES>(c[w], x[w]);
                else carry = bitslice_add8<P, MISSES>(c[w], x[w]);
                if (MISSES) dead[w] |= carry;
            }
        }
    };
    // the NEXT 64 offsets are fetched while the current 64 are being worked on
    int sel_next = 64 + lane < nf ? soff[64 + lane] : zero_off;
    int g = 0;
    while (g < nf) {
        if (g && (g & 63) == 0) {
            sel = sel_next;
            sel_next = g + 64 + lane < nf ? soff[g + 64 + lane] : zero_off;
        }
        // pass A: two batches of 8 first (most items end there: empty background dies after M + 1 <= 8 misses), then 16s --
        // twice the loads in flight for the items that go on; pass B never stops early: 16s throughout
        const int left = nf - g;
        // thousands of features (P >= 10, one dword per lane): 32 per batch once 32 are in -- 32 loads in flight and one
        // counter ripple per 32 features; the aliveness test then runs every 32 features (a few per cent of such an item)
        constexpr bool DEEP = WIDE && P >= 10 && DW == 1;
        if (DEEP && g >= 32 && left > 16) {
            if (left >= 32) batch(g, std::integral_constant<int, 32>{}, std::false_type{});
            else batch(g, std::integral_constant<int, 32>{}, std::true_type{});
            g += 32;
        } else if (!WIDE || (MISSES && g < 16) || left <= 8) {
            if (left >= 8) batch(g, std::integral_constant<int, 8>{}, std::false_type{});
            else batch(g, std::integral_constant<int, 8>{}, std::true_type{});
            g += 8;
        } else {
            if (left >= 16) batch(g, std::integral_constant<int, 16>{}, std::false_type{});
            else batch(g, std::integral_constant<int, 16>{}, std::true_type{});
            g += 16;
        }
        if (MISSES) {
            uint32_t alive = ~dead[0];
#pragma unroll
            for (int w = 1; w < DW; ++w) alive |= ~dead[w];
            if (__builtin_amdgcn_ballot_w64(alive != 0u) == 0ull) return false;
        }
    }
    return true;
}

// grid = (items of 2016 * DW positions, active templates / 4, frames); block = 4 waves = 4 template slots.
// blm: [frames][16 planes][lm_stride / 32 dwords]; P: counter planes, 2^P > the largest nf of the launch.
// DW = 2: a lane owns 64 positions (an 8-byte load per feature): half as many, twice as long work items -- for launches
// whose one-dword items would not all be resident at once (8 waves per SIMD): the second round of waves starts when the
// first ends, and a launch takes two "longest waves" instead of one (16 case1 frames: 25.2 -> see DESIGN.md).
template <int P, bool WIDE, int DW>
__global__ __launch_bounds__(256) void k_similarity_coarse_bits(
    const uint32_t* __restrict__ blm, int64_t lm_stride, int T, int W, int H, int L, int lc, const DevTL* __restrict__ tls,
    const int32_t* __restrict__ soff, const CoarseItem* __restrict__ items, const int32_t* __restrict__ cfoff, int n_active,
    const int32_t* __restrict__ raw_keep, const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id,
    Cand* __restrict__ cands, int32_t* __restrict__ counters, int cap, int64_t blm_fs_dwords)
{
    raise_wave_priority();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int chunk_id, slot4, frame;
    coarse_block_item(chunk_id, slot4, frame);
    const int templ_slot = __builtin_amdgcn_readfirstlane(slot4 * 4 + wave);
    if (templ_slot >= n_active) return;
    const int sel0 = cfoff[(size_t)templ_slot * 64 + lane];
    const CoarseItem it = items[templ_slot];
    const int nf = __builtin_amdgcn_readfirstlane(it.nf), rmin = __builtin_amdgcn_readfirstlane(it.rmin);
    const int npos = __builtin_amdgcn_readfirstlane(it.npos);
    constexpr int LP = 32 * DW; // positions per lane
    const int base = chunk_id * (CB_POS * DW);
    if (base >= npos || nf <= 0) return; // rmin >= 1 (host): positions past the span score 0 and are never candidates
    const int M = (4 * nf - rmin) >> 2;  // misses a position can afford
    if (4 * nf < rmin) return;
    const int zero_off = (int)(7 * lm_stride + (int64_t)T * T * W * H);
    const bool on = base + LP * lane < npos + LP; // own positions or the left neighbour's last dword
    const uint32_t* ub = blm + (size_t)frame * blm_fs_dwords;
    const uint32_t li = (uint32_t)((base >> 5) + DW * lane);
    const int32_t* so = soff + it.soff_base;

    uint32_t ca[DW][P], dead[DW];
    const int bias = (1 << P) - 1 - M;
#pragma unroll
    for (int w = 0; w < DW; ++w) {
        // this lane's positions base + LP lane + 32 w + b, b < nvalid (lane 63 only supplies lane 62's last dword)
        int nvalid = npos - base - LP * lane - 32 * w;
        nvalid = lane == 63 ? 0 : (nvalid < 0 ? 0 : (nvalid > 32 ? 32 : nvalid));
        dead[w] = ~(nvalid == 32 ? ~0u : ((1u << nvalid) - 1u));
#pragma unroll
        for (int p = 0; p < P; ++p) ca[w][p] = (bias >> p) & 1 ? ~0u : 0u;
    }
    if (!bit_pass<P, true, WIDE, DW>(ub, on ? li : 0u, so, sel0, nf, zero_off, ca, dead)) return;

    // survivors: exact hits of the lanes that hold one (and of their right neighbours, for the dword after their own)
    uint32_t alive_any = ~dead[0];
#pragma unroll
    for (int w = 1; w < DW; ++w) alive_any |= ~dead[w];
    const uint32_t left_alive = (uint32_t)__builtin_amdgcn_mov_dpp((int)alive_any, 0x138, 0xf, 0xf, true); // wave_shr:1
    uint32_t ce[DW][P], unused[DW];
#pragma unroll
    for (int w = 0; w < DW; ++w) {
        unused[w] = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) ce[w][p] = 0;
    }
    bit_pass<P, false, WIDE, DW>(ub + ((8 * lm_stride) >> 5), (alive_any | left_alive) != 0u ? li : 0u, so, sel0, nf, zero_off, ce, unused);

    cands += (size_t)frame * cap;
    counters += (size_t)frame * CTR_STRIDE;
    const int offset = T / 2 + (T % 2 - 1);
#pragma unroll
    for (int w = 0; w < DW; ++w) {
        uint32_t m = ~dead[w];
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            const int misses = bitslice_get<P>(ca[w], b) - bias;
            const int raw = 3 * (nf - misses) + bitslice_get<P>(ce[w], b);
            if (raw >= rmin) {
                const int j = base + LP * lane + 32 * w + b;
                const int idx = atomicAdd(&counters[0], 1);
                if (idx < cap) {
                    Cand c;
                    c.t = it.t;
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


// bit-sliced sum of two P-plane counters (no overflow by the caller's bound)
template <int P>
__device__ __forceinline__ void bitslice_sum(uint32_t (&a)[P], const uint32_t (&b)[P])
{
    uint32_t carry = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint32_t u = a[p] ^ b[p];
        const uint32_t s_ = u ^ carry;
        carry = (u & carry) | (~u & a[p]);
        a[p] = s_;
    }
}

// The LATENCY form of the coarse pass on bit planes (one frame, a few hundred templates): a workgroup of EIGHT waves owns
// one item of 2016 positions -- waves 0..3 count the any-plane hits of a quarter of the template's features each, waves
// 4..7 the exact-plane hits of the same quarters -- so an item is two or three rounds of loads deep instead of the
// eleven of the one-wave form (which stops early on most items but is as slow as its longest item: 11.8 us for a 1024^2
// frame x 360 templates whatever the content).  No pruning: every feature of every item is loaded (26 MB for that frame).
// The partial counters meet in LDS; wave 0 adds them (bit-sliced), compares the hit count with nf - M (positions that
// can reach rmin), and for those takes raw = 3 * any + exact.  Same candidates as every other coarse kernel.
// grid = (items of 2016 positions, active templates, frames); block = 512.
template <int P>
__global__ __launch_bounds__(512) void k_similarity_coarse_bits_block(
    const uint32_t* __restrict__ blm, int64_t lm_stride, int T, int W, int H, int L, int lc, const DevTL* __restrict__ tls,
    const int32_t* __restrict__ soff, const CoarseItem* __restrict__ items, int n_active, const int32_t* __restrict__ raw_keep,
    const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id, Cand* __restrict__ cands,
    int32_t* __restrict__ counters, int cap, int64_t blm_fs_dwords)
{
    raise_wave_priority();
    __shared__ uint32_t s_c[8][P][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunk_id = blockIdx.x, templ_slot = blockIdx.y, frame = blockIdx.z;
    const CoarseItem it = items[templ_slot];
    const int nf = __builtin_amdgcn_readfirstlane(it.nf), rmin = __builtin_amdgcn_readfirstlane(it.rmin);
    const int npos = __builtin_amdgcn_readfirstlane(it.npos);
    const int base = chunk_id * CB_POS;
    if (base >= npos || nf <= 0 || 4 * nf < rmin) return; // block-uniform
    const int M = (4 * nf - rmin) >> 2; // misses a position can afford
    const int zero_off = (int)(7 * lm_stride + (int64_t)T * T * W * H);
    const bool on = base + 32 * lane < npos + 32; // own positions or the left neighbour's upper dword
    const int pass = wave >> 2, q = wave & 3;
    const int nq = (nf + 3) >> 2, f0 = q * nq, cnt = nf - f0 < 0 ? 0 : (nf - f0 < nq ? nf - f0 : nq);
    const uint32_t* ub = blm + (size_t)frame * blm_fs_dwords + (pass ? ((8 * lm_stride) >> 5) : 0);
    const int32_t* so = soff + it.soff_base + f0;
    uint32_t c[1][P], unused[1] = {0u};
#pragma unroll
    for (int p = 0; p < P; ++p) c[0][p] = 0;
    if (cnt > 0) {
        const int sel0 = lane < cnt ? so[lane] : zero_off;
        bit_pass<P, false, true, 1>(ub, on ? (uint32_t)((base >> 5) + lane) : 0u, so, sel0, cnt, zero_off, c, unused);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) s_c[wave][p][lane] = c[0][p];
    __syncthreads();
    if (wave != 0) return;
    uint32_t h[P], t[P];
#pragma unroll
    for (int p = 0; p < P; ++p) h[p] = c[0][p];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
#pragma unroll
        for (int p = 0; p < P; ++p) t[p] = s_c[w][p][lane];
        bitslice_sum<P>(h, t);
    }
    // positions that can still reach rmin: any-hits >= nf - M (bit-sliced compare against a wave-uniform constant, LSB up)
    const int K = nf - M;
    uint32_t ge = ~0u;
#pragma unroll
    for (int p = 0; p < P; ++p) ge = ((K >> p) & 1) ? (h[p] & ge) : (h[p] | ge);
    int nvalid = npos - base - 32 * lane;
    nvalid = lane == 63 ? 0 : (nvalid < 0 ? 0 : (nvalid > 32 ? 32 : nvalid));
    uint32_t m = ge & (nvalid == 32 ? ~0u : ((1u << nvalid) - 1u));
    if (m == 0u) return;
#pragma unroll
    for (int p = 0; p < P; ++p) t[p] = s_c[4][p][lane];
#pragma unroll
    for (int w = 5; w < 8; ++w) {
        uint32_t u[P];
#pragma unroll
        for (int p = 0; p < P; ++p) u[p] = s_c[w][p][lane];
        bitslice_sum<P>(t, u);
    }
    cands += (size_t)frame * cap;
    counters += (size_t)frame * CTR_STRIDE;
    const int offset = T / 2 + (T % 2 - 1);
    while (m) {
        const int b = __builtin_ctz(m);
        m &= m - 1;
        const int raw = 3 * bitslice_get<P>(h, b) + bitslice_get<P>(t, b);
        if (raw >= rmin) {
            const int j = base + 32 * lane + b;
            const int idx = atomicAdd(&counters[0], 1);
            if (idx < cap) {
                Cand cd;
                cd.t = it.t;
                cd.x = (j % W) * T + offset;
                cd.y = (j / W) * T + offset;
                cd.raw = raw;
                cand_fill_next(cd, tls, raw_keep, class_idx, template_id, L, lc - 1);
                cands[idx] = cd;
            }
        }
    }
}

} // namespace sbm
