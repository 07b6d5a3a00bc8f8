// sbm_bitplane_math.h — register-level arithmetic on bit planes (round 4): the 32-byte -> 8-dword bit transpose of the
// producer and the bit-sliced carry-save counters of the consumer.  Plain 32-bit integer code, host and device (the
// host pass is what tests/emu compiles for the CPU suite).
//
// (1) 32 spread bytes -> 8 bit-plane dwords.
// Part of the fused producer of the coarsest level's bit planes (k_build_lm_rows, sbm_lm_kernels.h); the reference
// computes one response byte per orientation from the spread byte (computeResponseMaps, line2Dup.cpp:637-747), the
// coarse pass on bit planes (sbm_coarse_bits.h) needs "bit o of the spread byte" of 32 consecutive cells as one dword.
//
// d[i] (i = 0..7) holds the spread bytes of cells 4i .. 4i+3 (byte j = cell 4i + j).  An element is addressed by
// (register i = i2 i1 i0, position 8 j + o = j1 j0 o2 o1 o0); wanted is (register o, position 4 i + j = i2 i1 i0 j1 j0).
// Five exchanges of one register-index bit with one position bit do it; the two that move a byte-index bit are byte
// shuffles (v_perm_b32), the other three the classic masked-xor butterfly:
//     i2 <-> j1 (pairs r, r+4)   i1 <-> j0 (pairs r, r+2)   i0 <-> o2 (pairs r, r+1, shift 4)
//     j1 <-> o1 (pairs r, r+4, shift 2)   j0 <-> o0 (pairs r, r+2, shift 1)
// after which register r = (o1 o0 o2) holds orientation o: see bitplane_reg().  76 instructions for 32 cells x 8 planes.
#pragma once
#include <stdint.h>

namespace sbm {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t bp_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#define SBM_BP_FN __device__ __forceinline__
#else
// host pass of the same header (tests/emu compiles it for the CPU): v_perm_b32's documented byte selection
inline uint32_t bp_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 0xff;
        const uint32_t b = s < 8 ? (uint32_t)((src >> (8 * s)) & 0xff) : 0u; // selectors >= 8 are not used here
        r |= b << (8 * i);
    }
    return r;
}
#if defined(__HIPCC__)
#define SBM_BP_FN __host__ inline
#else
#define SBM_BP_FN inline
#endif
#endif

#if defined(__HIPCC__)
#define SBM_BP_HD __host__ __device__ __forceinline__
#else
#define SBM_BP_HD inline
#endif

// which register holds orientation o after bytes32_to_bitplanes
SBM_BP_HD constexpr int bitplane_reg(int o) { return (((o >> 1) & 1) << 2) | ((o & 1) << 1) | ((o >> 2) & 1); }

#if defined(__HIPCC__)
__host__ __device__ __forceinline__
#else
inline
#endif
void bytes32_to_bitplanes(uint32_t (&d)[8])
{
    // i2 <-> j1: a' = {a.b0, a.b1, b.b0, b.b1}, b' = {a.b2, a.b3, b.b2, b.b3}
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t a = d[r], b = d[r + 4];
        d[r] = bp_perm(b, a, 0x05040100u);
        d[r + 4] = bp_perm(b, a, 0x07060302u);
    }
    // i1 <-> j0: a' = {a.b0, b.b0, a.b2, b.b2}, b' = {a.b1, b.b1, a.b3, b.b3}
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (q & 1) | ((q >> 1) << 2);
        const uint32_t a = d[r], b = d[r + 2];
        d[r] = bp_perm(b, a, 0x06020400u);
        d[r + 2] = bp_perm(b, a, 0x07030501u);
    }
    // i0 <-> o2
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const uint32_t t = ((d[r] >> 4) ^ d[r + 1]) & 0x0f0f0f0fu;
        d[r + 1] ^= t;
        d[r] ^= t << 4;
    }
    // j1 <-> o1
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t t = ((d[r] >> 2) ^ d[r + 4]) & 0x33333333u;
        d[r + 4] ^= t;
        d[r] ^= t << 2;
    }
    // j0 <-> o0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (q & 1) | ((q >> 1) << 2);
        const uint32_t t = ((d[r] >> 1) ^ d[r + 2]) & 0x55555555u;
        d[r + 2] ^= t;
        d[r] ^= t << 1;
    }
}

// (2) Bit-sliced counters of the coarse pass on bit planes (sbm_coarse_bits.h): c[p] holds bit p of 32 independent counters
// (one per bit position).  Full adder on 32 slices: (h, l) = a + b + c = two v_bitop3_b32 on gfx950 (any function of three
// inputs is one instruction: parity 0x96, majority 0xe8; truth-table index = S0 * 4 + S1 * 2 + S2).  INV: b and c are to be
// counted INVERTED (the any-plane's zero bits are the misses) -- parity is unchanged by two inversions, and
// "majority of (a, ~b, ~c)" is table 0x71, so the inversion costs nothing.  The device pass writes the instruction itself
// (left to its own pattern matching the compiler builds a full adder from 3 - 4 instructions); the host pass (tests/emu) is C.
#if defined(__HIP_DEVICE_COMPILE__)
#define SBM_BITOP3(d, s0, s1, s2, tbl) asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:" tbl : "=v"(d) : "v"(s0), "v"(s1), "v"(s2))
template <bool INV>
__device__ __forceinline__ void bitslice_csa(uint32_t& h, uint32_t& l, uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t hh, ll;
    if (INV) SBM_BITOP3(hh, a, b, c, "0x71");
    else SBM_BITOP3(hh, a, b, c, "0xe8");
    SBM_BITOP3(ll, a, b, c, "0x96");
    h = hh;
    l = ll;
}
#else
template <bool INV>
inline void bitslice_csa(uint32_t& h, uint32_t& l, uint32_t a, uint32_t b, uint32_t c)
{
    if (INV) b = ~b, c = ~c;
    const uint32_t hh = (a & b) | (a & c) | (b & c), ll = a ^ b ^ c;
    h = hh;
    l = ll;
}
#endif

// add eight 1-bit slices x[0..7] (INV: their complements) to the counters c[0..P-1] (Harley-Seal: 7 full adders, then the
// eights ripple up); returns the carry out of the top plane
template <int P, bool INV = false>
SBM_BP_HD uint32_t bitslice_add8(uint32_t (&c)[P], const uint32_t (&x)[8])
{
    uint32_t t0, t1, f0, f1, e;
    bitslice_csa<INV>(t0, c[0], c[0], x[0], x[1]);
    bitslice_csa<INV>(t1, c[0], c[0], x[2], x[3]);
    bitslice_csa<false>(f0, c[1], c[1], t0, t1);
    bitslice_csa<INV>(t0, c[0], c[0], x[4], x[5]);
    bitslice_csa<INV>(t1, c[0], c[0], x[6], x[7]);
    bitslice_csa<false>(f1, c[1], c[1], t0, t1);
    bitslice_csa<false>(e, c[2], c[2], f0, f1);
#pragma unroll
    for (int p = 3; p < P; ++p) { // ripple the eights
        const uint32_t t = c[p] & e;
        c[p] ^= e;
        e = t;
    }
    return e;
}

// The same for 32 slices with ONE ripple: the four groups' eights meet in two more carry-save levels (planes 3 and 4) and
// only the resulting thirty-twos ripple up -- 4 x 14 + 6 + 2 (P - 5) instructions instead of 4 x (14 + 2 (P - 3)); what
// the large-template launches (P = 10, 13) are made of.  Needs P >= 6.
template <int P, bool INV = false>
SBM_BP_HD uint32_t bitslice_add32(uint32_t (&c)[P], const uint32_t (&x)[32])
{
    uint32_t e[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t t0, t1, f0, f1;
        bitslice_csa<INV>(t0, c[0], c[0], x[8 * g + 0], x[8 * g + 1]);
        bitslice_csa<INV>(t1, c[0], c[0], x[8 * g + 2], x[8 * g + 3]);
        bitslice_csa<false>(f0, c[1], c[1], t0, t1);
        bitslice_csa<INV>(t0, c[0], c[0], x[8 * g + 4], x[8 * g + 5]);
        bitslice_csa<INV>(t1, c[0], c[0], x[8 * g + 6], x[8 * g + 7]);
        bitslice_csa<false>(f1, c[1], c[1], t0, t1);
        bitslice_csa<false>(e[g], c[2], c[2], f0, f1);
    }
    uint32_t sa, sb, q;
    bitslice_csa<false>(sa, c[3], c[3], e[0], e[1]);
    bitslice_csa<false>(sb, c[3], c[3], e[2], e[3]);
    bitslice_csa<false>(q, c[4], c[4], sa, sb);
#pragma unroll
    for (int p = 5; p < P; ++p) { // ripple the thirty-twos
        const uint32_t t = c[p] & q;
        c[p] ^= q;
        q = t;
    }
    return q;
}

// four / two slices (the tails of the refinement pass's feature loop, sbm_local_bits.h)
template <int P>
SBM_BP_HD void bitslice_add4(uint32_t (&c)[P], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    uint32_t t0, t1, e;
    bitslice_csa<false>(t0, c[0], c[0], x0, x1);
    bitslice_csa<false>(t1, c[0], c[0], x2, x3);
    bitslice_csa<false>(e, c[1], c[1], t0, t1);
#pragma unroll
    for (int p = 2; p < P; ++p) { // ripple the fours
        const uint32_t t = c[p] & e;
        c[p] ^= e;
        e = t;
    }
}
template <int P>
SBM_BP_HD void bitslice_add2(uint32_t (&c)[P], uint32_t x0, uint32_t x1)
{
    uint32_t e;
    bitslice_csa<false>(e, c[0], c[0], x0, x1);
#pragma unroll
    for (int p = 1; p < P; ++p) { // ripple the twos
        const uint32_t t = c[p] & e;
        c[p] ^= e;
        e = t;
    }
}

// a[0..N) += b[0..N) as N-plane numbers; the carry out of plane N-1 goes to a[N] (the caller provides the room)
template <int N, int PM>
SBM_BP_HD void bitslice_add_planes(uint32_t (&a)[PM], const uint32_t (&b)[PM])
{
    static_assert(N < PM, "room for the carry");
    uint32_t carry = 0;
#pragma unroll
    for (int p = 0; p < N; ++p) bitslice_csa<false>(carry, a[p], a[p], b[p], carry);
    a[N] = carry;
}

// Refinement pass on bit strips (sbm_local_bits.h): s[0..N) are the bit-sliced counts of one patch row, "response > 0"
// counts in bit positions 0..15 (one per column), "response == 4" counts in positions 16..31.  The reference's sum of
// response bytes (4 / 3 / 0, similarityLocal line2Dup.cpp:860-922) is raw = 3 #any + #exact; returns, for the row, the largest raw
// (<= 2^(N+2) - 1) and through *first the lowest column holding it.  All in bit-sliced form: raw = A + E + 2A by two plane-wise
// additions, then the maximum from the top plane down (keep the columns whose bit is set whenever any kept column has it).
template <int N, int PM>
SBM_BP_HD uint32_t bitslice_row_best(const uint32_t (&s)[PM], int* first)
{
    static_assert(N + 2 <= PM, "room for raw");
    uint32_t r[PM], e[PM], a2[PM];
#pragma unroll
    for (int p = 0; p < PM; ++p) {
        r[p] = p < N ? s[p] : 0u;
        e[p] = p < N ? s[p] >> 16 : 0u;
        a2[p] = p >= 1 && p <= N ? s[p - 1] : 0u;
    }
    bitslice_add_planes<N>(r, e);       // A + E: N + 1 planes
    bitslice_add_planes<N + 1>(r, a2);  // + 2A:  N + 2 planes
    uint32_t mask = 0xffffu;
#pragma unroll
    for (int p = N + 1; p >= 0; --p) {
        const uint32_t t = mask & r[p];
        mask = t ? t : mask;
    }
    const int f = __builtin_ctz(mask); // mask != 0
    uint32_t v = 0;
#pragma unroll
    for (int p = 0; p < N + 2; ++p) v |= ((r[p] >> f) & 1u) << p;
    *first = f;
    return v;
}

// counter of bit position b
template <int P>
SBM_BP_HD int bitslice_get(const uint32_t (&c)[P], int b)
{
    int v = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) v |= (int)((c[p] >> b) & 1u) << p;
    return v;
}

} // namespace sbm
