// sbm_local_bits.h — the refinement pass (similarityLocal, line2Dup.cpp:860-922, called from matchClass :1234-1290) on BIT
// STRIPS of a T = 4 level (round 4).
//
// The reference adds, for every feature of the template, the 16 x 16 response bytes around the candidate and takes the
// first maximum.  A response byte is 4 (the feature's orientation is in the spread byte), 3 (a neighbouring orientation is)
// or 0, so the sum is 3 #any + #exact, and both counts can be taken from one bit each.  k_build_lm_rows stores the level
// as dwords (any bits of 16 columns | exact bits << 16) per (sub-plane, orientation, strip of 16 columns, grid row)
// (build_lm_strip4_allty, lm_bits_offset: two strips side by side); a feature's patch is then 16 rows x two dwords (strips s
// and s + 1) = 128 bytes in one or two cache lines instead of 512 spread bytes in two to four, and four features cost a wave two loads, five alignment instructions and 1.75 carry-save adders instead of
// two 16-byte loads and ~36 instructions of byte arithmetic.
//
// One WAVE per candidate: lane = (feature slot g = lane >> 4, patch row r = lane & 15); c[p] = bit p of 32 counters, the
// row's 16 any-counts in bit positions 0..15, its 16 exact-counts in 16..31.  Feature records (byte offset of the strip
// row, column misalignment) are computed with lane = feature and handed to the slots by ds_bpermute, as in the byte form;
// a window of 64 features is ONE batch (32 loads per lane in flight) and the next window's records are fetched meanwhile.
// At the end the four slots' counters are added (two bit-sliced additions across lanes), raw = A + E + 2A is formed and
// maximised in bit-sliced form (bitslice_row_best), and the rows' keys (raw << 8 | 255 - position) meet in a 16-lane
// reduction.  P: planes per slot counter (a slot counts ceil(nf / 4) features at most).
#pragma once
#include "sbm_bitplane_math.h"
#include "sbm_lm_kernels.h" // lm_bits_offset

namespace sbm {

template <int P>
__device__ __forceinline__ uint32_t local_best_bits(const uint32_t* __restrict__ lb, const uint32_t* __restrict__ fxy,
                                                    const uint8_t* __restrict__ flabel, int nf, int rows, int cols, int ox, int oy,
                                                    int W, int H)
{
    constexpr int PM = P + 4;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const uint32_t zero_idx = (uint32_t)lm_bits_dwords(W, H); // the zero tail behind the 128 (sub-plane, orientation) planes
    const uint32_t pitch = (uint32_t)H * 4u;                  // strip s -> strip s + 1, same row
    const uint32_t row_off = (uint32_t)r * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)lb, 0, 0x7fffffff, 0x00020000);
    uint32_t c[PM];
#pragma unroll
    for (int p = 0; p < PM; ++p) c[p] = 0;
    uint32_t (&cp)[P] = *(uint32_t (*)[P])&c[0];
    // lane = feature.  The records of window jb + 64 are fetched while window jb is being added (the loop is a chain of
    // dependent round trips otherwise).
    auto fetch = [&](int jb, uint32_t& xy, uint32_t& lab) {
        const int jj = jb + lane;
        xy = 0;
        lab = 8; // no feature: the zero record
        if (jj < nf) {
            xy = fxy[jj];
            lab = flabel[jj];
        }
    };
    uint32_t nxy, nlab;
    fetch(0, nxy, nlab);
    for (int jb = 0; jb < nf; jb += 64) {
        // record: dword index of (plane, strip, first row) << 5 | column of the patch inside the strip (bit 4 stays 0: the
        // record itself is the shift count, of which the hardware reads five bits)
        uint32_t rec = zero_idx << 5;
        {
            const int x = (int)(nxy & 0xffff) + ox, y = (int)(nxy >> 16) + oy;
            const int gx0 = x >> 2, gy0 = y >> 2;
            if (nlab < 8 && x >= 0 && y >= 0 && x < cols && y < rows)
                rec = ((uint32_t)lm_bits_offset((((y & 3) << 2) | (x & 3)) * 8 + (int)nlab, gx0 >> 4, gy0, W, H) << 5) | (uint32_t)(gx0 & 15);
        }
        if (jb + 64 < nf) fetch(jb + 64, nxy, nlab);
        const int nbf = nf - jb < 64 ? nf - jb : 64;
        auto batch = [&](auto N, int u) { // N steps = 4 N features, u = first feature of the batch (multiple of 4)
            constexpr int nst = decltype(N)::value;
            uint32_t A[nst], B[nst], sh[nst];
#pragma unroll
            for (int k = 0; k < nst; ++k) {
                const int f = (u + 4 * k + g) << 2; // lanes past the last feature hold the zero record
                sh[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(f, (int)rec);
                const uint32_t a = ((sh[k] >> 5) << 2) + row_off;
                A[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)a, 0, 0);
                B[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)a, (int)pitch, 0);
            }
#pragma unroll
            for (int k = 0; k < nst; ++k) {
                const uint32_t any = __builtin_amdgcn_perm(B[k], A[k], 0x05040100u) >> (sh[k] & 31u); // 32 columns from the strip's first
                const uint32_t ex = __builtin_amdgcn_perm(B[k], A[k], 0x07060302u) >> (sh[k] & 31u);
                A[k] = __builtin_amdgcn_perm(ex, any, 0x05040100u);
            }
            if constexpr (nst == 16) {
                const uint32_t x0[8] = {A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7]};
                const uint32_t x1[8] = {A[8 % nst], A[9 % nst], A[10 % nst], A[11 % nst], A[12 % nst], A[13 % nst], A[14 % nst], A[15 % nst]};
                bitslice_add8<P>(cp, x0);
                bitslice_add8<P>(cp, x1);
            } else if constexpr (nst == 8) {
                const uint32_t x[8] = {A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7]};
                bitslice_add8<P>(cp, x);
            } else if constexpr (nst == 4) {
                bitslice_add4<P>(cp, A[0], A[1], A[2], A[3]);
            } else {
                bitslice_add2<P>(cp, A[0], A[1]);
            }
        };
        // a full window is one batch: all 32 loads of the lane in flight at once.  (Batches of 8 steps in 64 registers, eight
        // waves per SIMD instead of four, measure the same on 16 candidate-heavy frames -- 19.5 us: the launch is bound by
        // what a CU's vector memory path delivers from the L2s for dword gathers, ~11 bytes per clock -- and worse on
        // config 5's 64 frames, 293 against 261 us.)
        if (nbf > 32) batch(std::integral_constant<int, 16>{}, 0);
        else if (nbf > 16) batch(std::integral_constant<int, 8>{}, 0);
        else if (nbf > 8) batch(std::integral_constant<int, 4>{}, 0);
        else batch(std::integral_constant<int, 2>{}, 0);
    }
    // the four slots: lanes l, l ^ 16, l ^ 32 hold the same row
    uint32_t o[PM];
#pragma unroll
    for (int p = 0; p < PM; ++p) o[p] = p < P ? (uint32_t)__shfl_xor((int)c[p], 16, 64) : 0u;
    bitslice_add_planes<P>(c, o);
#pragma unroll
    for (int p = 0; p < PM; ++p) o[p] = p < P + 1 ? (uint32_t)__shfl_xor((int)c[p], 32, 64) : 0u;
    bitslice_add_planes<P + 1>(c, o);
    int first;
    const uint32_t raw = bitslice_row_best<P + 2>(c, &first);
    uint32_t best = (raw << 8) | (255u - (uint32_t)(r * 16 + first));
#pragma unroll
    for (int s = 8; s >= 1; s >>= 1) {
        const uint32_t ob = (uint32_t)__shfl_xor((int)best, s, 64);
        best = ob > best ? ob : best;
    }
    return best;
}

} // namespace sbm
