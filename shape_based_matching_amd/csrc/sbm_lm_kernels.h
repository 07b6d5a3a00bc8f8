// sbm_lm_kernels.h — gfx950 kernels that turn a one-hot orientation map into linear memories.
// Reference functions replaced (file:line in ddcr/shape_based_matching):
//   k_build_lm_rows / k_build_lm   spread + computeResponseMaps + linearize, fused   line2Dup.cpp:616-630, 637-747, 749-777
//   k_spread / k_response / k_linearize   the same three, unfused (stage entry points)
//   k_expand_lm           the 8-plane form of a compact (spread-byte) level, on demand
#pragma once
#include "sbm_common.h"
#include "sbm_bitplane_math.h"

namespace sbm {

// ---------------------------------------------------------------------------
// Linear memories: spread + response + linearize fused
// ---------------------------------------------------------------------------
constexpr int LM_GX = 64; // grid cells per block along x

// response of 4 packed spread bytes for orientation o: 4 if bit o set, else 3
// if a circular neighbour bit is set, else 0 (== SIMILARITY_LUT, line2Dup.cpp:635)
__device__ __forceinline__ uint32_t response4(uint32_t v, int o)
{
    uint32_t self = (v >> o) & 0x01010101u;
    uint32_t nb = ((v >> ((o + 1) & 7)) | (v >> ((o + 7) & 7))) & 0x01010101u & ~self;
    return (self << 2) | (nb * 3u);
}

// One block = one row of grid cells (T pixel rows) x 64 cells.  The one-hot
// tile (+T-1 halo, zero outside the image = the clipped window of spread())
// is OR-reduced separably in LDS, stored there already in linear-memory order
// [ty][tx][gx], then each lane turns 4 consecutive cells into the 8
// orientation responses and writes 8 dwords, coalesced along gx.
__global__ __launch_bounds__(256) void k_build_lm(const uint8_t* __restrict__ q, int rows, int cols, int T,
                                                  int W, int H, uint8_t* __restrict__ lm, int64_t lm_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int gx0 = blockIdx.x * LM_GX, gy = blockIdx.y;
    const int tw = LM_GX * T;      // tile width in pixels
    const int lw = tw + T - 1;     // + halo
    const int lwp = (lw + 3) & ~3; // padded row pitch
    const int lh = 2 * T - 1;
    uint8_t* s_q = smem;                  // [lh][lwp]
    uint8_t* s_h = s_q + lh * lwp;        // [lh][tw]   horizontal OR
    uint8_t* s_sp = s_h + lh * tw;        // [T*T][LM_GX] spread, linearized order
    const int pr0 = gy * T, pc0 = gx0 * T;
    for (int idx = tid; idx < lh * lwp; idx += 256) {
        int r = idx / lwp, c = idx - r * lwp;
        int gr = pr0 + r, gc = pc0 + c;
        s_q[idx] = (c < lw && gr < rows && gc < cols) ? q[(size_t)gr * cols + gc] : (uint8_t)0;
    }
    __syncthreads();
    for (int idx = tid; idx < lh * tw; idx += 256) {
        int r = idx / tw, c = idx - r * tw;
        uint8_t v = 0;
        for (int d = 0; d < T; ++d) v |= s_q[r * lwp + c + d];
        s_h[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < T * tw; idx += 256) {
        int ty = idx / tw, c = idx - ty * tw;
        uint8_t v = 0;
        for (int d = 0; d < T; ++d) v |= s_h[(ty + d) * tw + c];
        int g = c / T, tx = c - g * T;
        s_sp[(ty * T + tx) * LM_GX + g] = v;
    }
    __syncthreads();
    const int64_t WH = (int64_t)W * H;
    const int items = T * T * (LM_GX / 4);
    const bool fast = (W & 3) == 0;
    for (int it = tid; it < items; it += 256) {
        int sub = it / (LM_GX / 4), g4 = it - sub * (LM_GX / 4);
        int gx = gx0 + g4 * 4;
        if (gx >= W) continue;
        uint32_t v = ((const uint32_t*)s_sp)[sub * (LM_GX / 4) + g4];
        int64_t dst = (int64_t)sub * WH + (int64_t)gy * W + gx;
        if (fast) {
#pragma unroll
            for (int o = 0; o < 8; ++o) *(uint32_t*)(lm + o * lm_stride + dst) = response4(v, o);
        } else {
            for (int o = 0; o < 8; ++o) {
                uint32_t rsp = response4(v, o);
                for (int k = 0; k < 4 && gx + k < W; ++k) lm[o * lm_stride + dst + k] = (uint8_t)(rsp >> (8 * k));
            }
        }
    }
}

// Strip-interleaved compact plane of a refinement-only level (needs W % 16 == 0): inside sub-plane
// sub = (y%T)*T + x%T the W x H grid is cut into W/16 column strips of 16 cells, and a strip is stored row after
// row, 16 bytes per row:  offset = sub*W*H + ((gx / 16) * H + gy) * 16 + gx % 16.
// similarityLocal (line2Dup.cpp:860-922) reads 16 x 16 cells per feature; row-major that is 16 pieces of 16 bytes in
// 16 different 128-byte lines, here it is two runs of 256 contiguous bytes at most (one when gx % 16 == 0).
__host__ __device__ __forceinline__ int64_t lm_strip_offset(int sub, int gy, int gx, int W, int H)
{
    return (int64_t)sub * W * H + ((int64_t)(gx >> 4) * H + gy) * 16 + (gx & 15);
}

// ---- register-only variant for T = 4 and T = 8 (the reference's strides) ----
// One lane owns 4 consecutive grid cells (4*T pixels) of one (ty, gy) pixel row
// r0 = gy*T + ty: it ORs the T source rows r0..r0+T-1 (16-byte loads), ORs T
// pixels forward with funnel shifts, transposes the 4 cells x T sub-columns in
// registers (v_perm), and for every tx and orientation stores one dword =
// responses of cells gx..gx+3.  A wave's store is one contiguous run of
// LM[o][ty*T+tx][gy*W + ...]: no LDS, no barrier, fully coalesced.
// Requires W % 4 == 0 and cols % 16 == 0 (host checks; else k_build_lm).
__device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi, lo, sel);
}

constexpr int LM_FULL_SPLIT = 4; // work items per (pixel row, 4 cells) of a level stored as 8 response planes

template <int T>
__device__ __forceinline__ void build_lm_rows_item(const uint8_t* __restrict__ q, int rows, int cols, int W, int H,
                                                   uint8_t* __restrict__ lm, int64_t lm_stride, int64_t item,
                                                   int compact, int split, uint32_t* s_tile)
{
    constexpr int NQ = T / 4 * 4; // dwords of own pixels per lane (4 cells * T px / 4)
    int r0, k, part = 0;
    bool active = true; // strip mode: every thread of the workgroup reaches the barrier below
    int tile_cb = 0, tile_gyb = 0, tile_ty = 0;
    if (compact == 2) {
        // strip-interleaved plane: a WORKGROUP = one ty x 16 grid rows x 64 cells (4 strips).  Thread (row = t / 16,
        // kk = t % 16) reads 4 cells of its row -- a wave's loads are 4 runs of 256 bytes -- and the spread dwords go
        // through LDS so that wave w then stores strip w of the tile: 16 strip rows of 16 bytes = one run of 256 bytes
        // per store.  (Round 2's first form let a wave own one strip end to end: contiguous stores, but each load
        // gathered 64-byte pieces of 16 different rows; 14.9 us of the 16-frame launch were level 0.)
        const int n_cb = (W + 63) >> 6, gyb_n = (H + 15) >> 4;
        const int t = (int)(item & 255);
        const int64_t blk = item >> 8;
        tile_cb = (int)(blk % n_cb);
        const int64_t rest = blk / n_cb;
        tile_gyb = (int)(rest % gyb_n);
        tile_ty = (int)(rest / gyb_n);
        const int gy_ = tile_gyb * 16 + (t >> 4);
        k = tile_cb * 16 + (t & 15);
        active = tile_ty < T && gy_ < H && k * 4 < W;
        r0 = active ? gy_ * T + tile_ty : 0;
        if (!active) k = 0;
    } else {
        const int lanes_per_row = W >> 2;
        int64_t it = item;
        if (!compact && split > 1) { // LM_FULL_SPLIT items per (pixel row, 4 cells): part = (tx half, orientation half)
            const int64_t base_items = (int64_t)rows * lanes_per_row;
            part = (int)(item / base_items);
            if (part >= LM_FULL_SPLIT) return;
            it = item - part * base_items;
        }
        // rows in (ty, gy) order: where a grid row has fewer than 64 lanes (W < 256: the coarsest level of a 1024^2 frame has
        // 16), the wave's 2 - 4 rows are then CONSECUTIVE grid rows of one ty, i.e. one contiguous run of each plane
        // (256 bytes per store instruction instead of four separate 64-byte pieces of four different planes)
        const int64_t row_id = it / lanes_per_row;
        k = (int)(it - row_id * lanes_per_row);
        if (row_id >= rows) return;
        r0 = (int)(row_id % H) * T + (int)(row_id / H); // pixel row gy * T + ty
    }
    const int gy = r0 / T, ty = r0 - gy * T;
    const int c0 = k * 4 * T; // first pixel column of this lane
    // vertical OR of rows r0 .. r0+T-1 (clipped at the bottom, :626-627), own pixels + T px of right halo
    uint32_t v[NQ + T / 4];
#pragma unroll
    for (int i = 0; i < NQ + T / 4; ++i) v[i] = 0;
#pragma unroll
    for (int d = 0; d < T; ++d) {
        const int r = r0 + d;
        if (r < rows && active) {
            const uint8_t* src = q + (size_t)r * cols + c0;
#pragma unroll
            for (int i = 0; i < NQ; i += 4) {
                const uint4 w = *(const uint4*)(src + 4 * i);
                v[i] |= w.x;
                v[i + 1] |= w.y;
                v[i + 2] |= w.z;
                v[i + 3] |= w.w;
            }
            if (c0 + 4 * NQ < cols) { // right halo (zero past the last column)
#pragma unroll
                for (int i = 0; i < T / 4; ++i) v[NQ + i] |= *(const uint32_t*)(src + 4 * (NQ + i));
            }
        }
    }
    // horizontal OR over T pixels forward: byte c |= bytes c+1 .. c+T-1
    uint32_t s[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        uint32_t a = v[i], b = v[i + 1];
        uint32_t o = a | __builtin_amdgcn_alignbyte(b, a, 1) | __builtin_amdgcn_alignbyte(b, a, 2) |
                     __builtin_amdgcn_alignbyte(b, a, 3);
        if (T == 8) {
            uint32_t c2 = v[i + 2];
            o |= b | __builtin_amdgcn_alignbyte(c2, b, 1) | __builtin_amdgcn_alignbyte(c2, b, 2) |
                 __builtin_amdgcn_alignbyte(c2, b, 3);
        }
        s[i] = o;
    }
    // s[] holds 4 cells x T sub-columns; gather, per tx, the 4 cells' bytes into one dword
    const int64_t WH = (int64_t)W * H;
    const int64_t cell = (int64_t)gy * W + k * 4;
    auto spread_dword = [&](int tx) {
        // cell j lives in dwords s[j*T/4 .. ], sub-column tx is byte (tx & 3) of dword j*(T/4) + (tx >> 2)
        const int dsel = tx >> 2, b = tx & 3;
        const uint32_t d0 = s[0 * (T / 4) + dsel], d1 = s[1 * (T / 4) + dsel], d2 = s[2 * (T / 4) + dsel],
                       d3 = s[3 * (T / 4) + dsel];
        // v_perm selector bytes: 0-3 pick from 'lo', 4-7 from 'hi'
        const uint32_t p01 = perm_b32(d1, d0, 0x0c0c0000u | ((4 + b) << 8) | b);        // {d0.b, d1.b, 0, 0}
        const uint32_t p23 = perm_b32(d3, d2, 0x00000c0cu | ((4 + b) << 24) | (b << 16)); // {0, 0, d2.b, d3.b}
        return p01 | p23;
    };
    if (compact == 2) { // strip-interleaved spread plane (lm_strip_offset): a 16 x 16 patch is 2 - 4 cache lines
        const int t = (int)(item & 255);
        const int kk = t & 15, row = t >> 4;
#pragma unroll
        for (int tx = 0; tx < T; ++tx) s_tile[tx * 256 + (kk >> 2) * 64 + row * 4 + (kk & 3)] = spread_dword(tx);
        __syncthreads();
        const int S = tile_cb * 4 + (t >> 6), gy2 = tile_gyb * 16 + ((t & 63) >> 2);
        if (tile_ty < T && S < (W >> 4) && gy2 < H) {
#pragma unroll
            for (int tx = 0; tx < T; ++tx)
                *(uint32_t*)(lm + lm_strip_offset(tile_ty * T + tx, gy2, S * 16 + (t & 3) * 4, W, H)) = s_tile[tx * 256 + t];
        }
    } else if (compact == 3) {
        // BIT PLANES of the coarsest level (round 4; read by k_similarity_coarse_bits): per orientation o one "response > 0"
        // plane (spread bits o-1, o, o+1) and one "response == 4" plane (bit o), 1 bit per position in the byte planes' flat
        // order -- 2 bytes per pixel instead of 8.  The wave's 64 lanes hold 256 CONSECUTIVE positions of sub-plane (ty, tx)
        // for every tx (host: W * H % 256 == 0, so a wave never straddles two ty); the spread dwords cross LDS once
        // ([tx][lane]), then lane (tx' = lane / 8, seg = lane % 8) reads the 32 bytes of cells 32 seg .. 32 seg + 31 of
        // sub-column tx', transposes them into 8 dwords (bit b of dword o = bit o of cell b: bytes32_to_bitplanes) and
        // stores 16 dwords.  T = 4: lanes 0..31.
        const int t = (int)threadIdx.x, lane = t & 63, w0 = t & ~63;
#pragma unroll
        for (int tx = 0; tx < T; ++tx) s_tile[tx * 256 + t] = spread_dword(tx);
        __builtin_amdgcn_wave_barrier(); // same wave writes and reads: LDS operations of a wave execute in order
        const int txr = lane >> 3, seg = lane & 7;
        if (txr < T) {
            const uint4 lo4 = *(const uint4*)&s_tile[txr * 256 + w0 + seg * 8], hi4 = *(const uint4*)&s_tile[txr * 256 + w0 + seg * 8 + 4];
            uint32_t d[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            bytes32_to_bitplanes(d);
            // first position of the wave: lane 0's cell (the wave's items are consecutive in (ty, gy, k) order)
            const int64_t cell0 = __builtin_amdgcn_readfirstlane((int)cell);
            const int64_t plane_dw = lm_stride >> 5; // lm_stride = bits per plane
            uint32_t* out = (uint32_t*)lm + (((int64_t)(ty * T + txr) * WH + cell0) >> 5) + seg;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const uint32_t e = d[bitplane_reg(o)];
                out[(int64_t)o * plane_dw] = d[bitplane_reg((o + 7) & 7)] | e | d[bitplane_reg((o + 1) & 7)];
                out[(int64_t)(8 + o) * plane_dw] = e;
            }
        }
    } else if (compact) { // one plane of spread bytes: the reader applies the response LUT for its own orientation
#pragma unroll
        for (int tx = 0; tx < T; ++tx) *(uint32_t*)(lm + (int64_t)(ty * T + tx) * WH + cell) = spread_dword(tx);
    } else {
        // 8 response planes: T * 8 dword stores per (row, 4 cells) -- 64 at T = 8, on a level with few rows.  For a
        // single frame (split > 1) the item is cut into LM_FULL_SPLIT parts (tx half = part >> 1, orientation half =
        // part & 1; the loads and ORs are repeated) so that the level is four times as many, four times shorter waves:
        // 7.7 -> 6.9 us.  A batch of frames has enough waves and only pays the repeats (23 -> 25 us): split = 1.
        if (split > 1) {
            const int txh = part >> 1, o0 = (part & 1) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h != txh) continue;
#pragma unroll
                for (int t = 0; t < T / 2; ++t) {
                    const int tx = h * (T / 2) + t;
                    const uint32_t sp = spread_dword(tx);
                    const int64_t dst = (int64_t)(ty * T + tx) * WH + cell;
#pragma unroll
                    for (int o = 0; o < 4; ++o) *(uint32_t*)(lm + (o0 + o) * lm_stride + dst) = response4(sp, o0 + o);
                }
            }
        } else {
#pragma unroll
            for (int tx = 0; tx < T; ++tx) {
                const uint32_t sp = spread_dword(tx);
                const int64_t dst = (int64_t)(ty * T + tx) * WH + cell;
#pragma unroll
                for (int o = 0; o < 8; ++o) *(uint32_t*)(lm + o * lm_stride + dst) = response4(sp, o);
            }
        }
    }
}

// Strip-interleaved spread plane of a T = 4 level, ALL FOUR ty of a grid row per thread (round 4).  build_lm_rows_item
// gives every (ty, gy) pixel row a thread of its own, which loads the T source rows ty .. ty+T-1 of its window: every
// source row is loaded T times and OR-ed T times.  Here a thread owns 4 grid cells of one GRID row gy for all ty: it loads
// the 2T-1 = 7 source rows gy*4 .. gy*4+6 once, forms the four vertical windows from suffix ORs of the first four rows and
// prefix ORs of the next three (V_ty = rows[ty..3] | rows[4..3+ty]), and then does per ty what the other form does: OR of
// T pixels forward, register transpose, LDS hop so that a wave stores one strip (runs of 256 bytes).  A workgroup = 16
// grid rows x 64 cells x 4 ty; 16 KB of LDS.  7 row loads and ~45 ORs per 16 stored dwords instead of 16 and ~80.
//
// BITS (round 4, read by k_similarity_local_bits, sbm_local_bits.h): the level is stored as BIT STRIPS instead -- per
// (sub-plane, orientation o, strip of 16 columns, grid row) one dword = "response > 0" bits of the 16 cells (spread bits
// o-1, o, o+1) in the low half, "response == 4" bits (spread bit o) in the high half:
//     dword index = ((sub * 8 + o) * (W / 16) + strip) * H + gy            (lm_bits_offset; 2 bytes per pixel, zero tail behind)
// so a feature's 16 x 16 patch is two runs of 64 bytes.  The workgroup is then 32 grid rows x 32 cells (two strips) instead
// of 16 x 64: its loads and its stores are both runs of 128 bytes.  Measured on the way (16 frames; the byte form's launch
// is 13.2 us, its refinement of 16 candidate-heavy frames 35): 16-row workgroups, 64-byte store runs 19.2 us; the four
// strips of a 64-cell group side by side, 256-byte store runs 15.2 -- but the reader's 16 rows then lie in 8 cache lines
// instead of 2 and its launch goes from 19 to 33 us; strip pairs side by side 17.0 / 23; this form 17.7 / 19.5.
// After the LDS hop a thread takes two units (sub, strip, row pair): the 32 spread bytes of two strip rows -> 8 bit-plane
// dwords (bytes32_to_bitplanes) -> 8 stores of 8 bytes.
__host__ __device__ __forceinline__ int64_t lm_bits_offset(int plane, int strip, int gy, int W, int H)
{
    return ((int64_t)plane * (W >> 4) + strip) * H + gy;
}
// dwords of a level stored as bit strips, without the zero tail
__host__ __device__ __forceinline__ int64_t lm_bits_dwords(int W, int H) { return (int64_t)128 * (W >> 4) * H; }

template <bool BITS = false>
__device__ __forceinline__ void build_lm_strip4_allty(const uint8_t* __restrict__ q, int rows, int cols, int W, int H,
                                                      uint8_t* __restrict__ lm, int64_t blk, uint32_t* s_tile4)
{
    constexpr int T = 4;
    constexpr int TR = BITS ? 32 : 16, TK = BITS ? 8 : 16; // tile: grid rows x threads per row (4 cells each)
    const int n_cb = (W + 4 * TK - 1) / (4 * TK);
    const int t = (int)threadIdx.x;
    const int tile_cb = (int)(blk % n_cb), tile_gyb = (int)(blk / n_cb);
    const int row = t / TK, kk = t % TK;
    const int gy = tile_gyb * TR + row, k = tile_cb * TK + kk;
    const bool active = gy < H && k * 4 < W;
    const int c0 = (active ? k : 0) * 16; // first pixel column of this thread's 4 cells
    uint32_t r[7][5];
#pragma unroll
    for (int d = 0; d < 7; ++d) {
        const int y = gy * T + d;
#pragma unroll
        for (int i = 0; i < 5; ++i) r[d][i] = 0;
        if (active && y < rows) { // rows past the image: zero (the clipped window of spread(), :626-627)
            const uint8_t* src = q + (size_t)y * cols + c0;
            const uint4 w = *(const uint4*)src;
            r[d][0] = w.x, r[d][1] = w.y, r[d][2] = w.z, r[d][3] = w.w;
            if (c0 + 16 < cols) r[d][4] = *(const uint32_t*)(src + 16); // right halo (zero past the last column)
        }
    }
    // suffix ORs of rows 0..3 (in place: r[d] = rows d..3), prefix ORs of rows 4..6 (r[4+d] = rows 4..4+d)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        r[2][i] |= r[3][i];
        r[1][i] |= r[2][i];
        r[0][i] |= r[1][i];
        r[5][i] |= r[4][i];
        r[6][i] |= r[5][i];
    }
#pragma unroll
    for (int ty = 0; ty < T; ++ty) {
        uint32_t v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = ty == 0 ? r[0][i] : (r[ty][i] | r[3 + ty][i]);
        // horizontal OR over 4 pixels forward: byte c |= bytes c+1 .. c+3; dword i = cell i, byte tx = sub-column tx
        uint32_t sp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            sp[i] = v[i] | __builtin_amdgcn_alignbyte(v[i + 1], v[i], 1) | __builtin_amdgcn_alignbyte(v[i + 1], v[i], 2) |
                    __builtin_amdgcn_alignbyte(v[i + 1], v[i], 3);
        // per tx the 4 cells' bytes in one dword (4 x 4 byte transpose)
#pragma unroll
        for (int tx = 0; tx < T; ++tx) {
            const uint32_t p01 = perm_b32(sp[1], sp[0], 0x0c0c0000u | ((4 + tx) << 8) | tx);
            const uint32_t p23 = perm_b32(sp[3], sp[2], 0x00000c0cu | ((4 + tx) << 24) | (tx << 16));
            s_tile4[(ty * T + tx) * 256 + (kk >> 2) * (TR * 4) + row * 4 + (kk & 3)] = p01 | p23;
        }
    }
    __syncthreads();
    if (BITS) {
        struct __attribute__((aligned(4))) u32x2_a4 { uint32_t x, y; };
        uint32_t* const lb = (uint32_t*)lm;
        const int n_strips = W >> 4;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int u = pass * 256 + t;
            const int rp = u & 15, st = (u >> 4) & 1, sub = u >> 5;
            const int S = tile_cb * 2 + st, gy2 = tile_gyb * 32 + rp * 2;
            const uint4 lo4 = *(const uint4*)&s_tile4[sub * 256 + st * 128 + rp * 8];
            const uint4 hi4 = *(const uint4*)&s_tile4[sub * 256 + st * 128 + rp * 8 + 4];
            uint32_t d[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            bytes32_to_bitplanes(d); // bit b of d[bitplane_reg(o)]: row gy2 column b (b < 16), row gy2 + 1 column b - 16
            if (S < n_strips && gy2 < H) {
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const uint32_t e = d[bitplane_reg(o)];
                    const uint32_t any = d[bitplane_reg((o + 7) & 7)] | e | d[bitplane_reg((o + 1) & 7)];
                    const uint32_t w0 = perm_b32(e, any, 0x05040100u), w1 = perm_b32(e, any, 0x07060302u);
                    uint32_t* dst = lb + lm_bits_offset(sub * 8 + o, S, gy2, W, H);
                    // non-temporal: the strips are read once, by a later launch, and 2 MB per frame of them would push the
                    // orientation maps and the coarse bit planes out of the L2s (16-frame launch 18.5 -> 16.5 us, pipelined
                    // step 94.5 -> 92.9; the refinement launch pays 1 - 2 us for reading them from HBM)
                    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
                    if (gy2 + 1 < H && (H & 1) == 0) __builtin_nontemporal_store(u32x2v{w0, w1}, (u32x2v*)dst); // 8-byte aligned
                    else if (gy2 + 1 < H) *(u32x2_a4*)dst = u32x2_a4{w0, w1};
                    else *dst = w0;
                }
            }
        }
        return;
    }
    const int S = tile_cb * 4 + (t >> 6), gy2 = tile_gyb * 16 + ((t & 63) >> 2);
    if (S < (W >> 4) && gy2 < H) {
#pragma unroll
        for (int sub = 0; sub < T * T; ++sub)
            *(uint32_t*)(lm + lm_strip_offset(sub, gy2, S * 16 + (t & 3) * 4, W, H)) = s_tile4[sub * 256 + t];
    }
}

// All pyramid levels in one launch: block ranges per level (levels whose T is 4 or 8), plus the
// reset of the per-frame counters, so the whole linear-memory stage costs one kernel boundary.
struct LmLevelArgs {
    const uint8_t* q;
    uint8_t* lm;
    int64_t lm_stride;
    int32_t rows, cols, W, H, T;
    int32_t block_begin; // first block of this level
    int64_t q_fs, lm_fs; // bytes from one frame of a batch to the next
    int32_t compact;     // 1: lm is ONE plane [T*T][W*H] of spread bytes (a level that only the refinement pass
                         // reads): 1/8 of the stores and of the HBM write-back; 2: the same plane strip-interleaved;
                         // 3: lm is the coarsest level's 16 BIT planes (lm_stride bits each; sbm_coarse_bits.h);
                         // 4 (with allty): a refinement level as bit strips (build_lm_strip4_allty, sbm_local_bits.h)
    int32_t split;       // 8-plane levels: LM_FULL_SPLIT work items per (pixel row, 4 cells) instead of 1
    int32_t allty;       // compact == 2, T == 4: one thread per (grid row, 4 cells) for all four ty (build_lm_strip4_allty)
};
struct LmArgs {
    LmLevelArgs lv[SBM_MAX_LEVELS];
    int32_t n_levels;
    int32_t* counters;  // may be null
    int32_t* out_count; // may be null
};

__global__ __launch_bounds__(256) void k_build_lm_rows(const LmArgs a)
{
    raise_wave_priority();
    __shared__ uint32_t s_tile[16 * 256]; // strip levels: the workgroup's spread dwords, [tx][strip][row][4 cells] ([ty][tx].. all-ty form)
    const size_t frame = blockIdx.y; // batch of frames: one grid row each
    if (blockIdx.x == 0 && a.counters) {
        if (threadIdx.x < CTR_STRIDE) a.counters[frame * CTR_STRIDE + threadIdx.x] = 0;
        if (threadIdx.x < 2 && a.out_count) a.out_count[frame * 2 + threadIdx.x] = 0;
    }
    // the level whose block range holds this block (the host orders the ranges heaviest blocks first)
    int l = 0, lb = -1;
#pragma unroll
    for (int i = 0; i < SBM_MAX_LEVELS; ++i)
        if (i < a.n_levels && (int)blockIdx.x >= a.lv[i].block_begin && a.lv[i].block_begin > lb) l = i, lb = a.lv[i].block_begin;
    const LmLevelArgs& p = a.lv[l];
    const int64_t item = (int64_t)((int)blockIdx.x - p.block_begin) * 256 + threadIdx.x;
    const uint8_t* q = p.q + frame * p.q_fs;
    uint8_t* lm = p.lm + frame * p.lm_fs;
    if (p.allty && p.compact == 4) build_lm_strip4_allty<true>(q, p.rows, p.cols, p.W, p.H, lm, (int64_t)((int)blockIdx.x - p.block_begin), s_tile);
    else if (p.allty) build_lm_strip4_allty<false>(q, p.rows, p.cols, p.W, p.H, lm, (int64_t)((int)blockIdx.x - p.block_begin), s_tile);
    else if (p.T == 4) build_lm_rows_item<4>(q, p.rows, p.cols, p.W, p.H, lm, p.lm_stride, item, p.compact, p.split, s_tile);
    else build_lm_rows_item<8>(q, p.rows, p.cols, p.W, p.H, lm, p.lm_stride, item, p.compact, p.split, s_tile);
}

// compact plane (spread bytes) -> the 8 response planes, for the stage entry points that hand out or read a
// full linear memory of a refinement-only level
__global__ __launch_bounds__(256) void k_expand_lm(const uint8_t* __restrict__ lmc, int64_t n_bytes,
                                                   uint8_t* __restrict__ lm, int64_t lm_stride, int strip, int W, int H)
{
    const int64_t n4 = n_bytes >> 2; // T*T*W*H is a multiple of 16
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        int64_t src = 4 * i;
        if (strip) { // i enumerates the row-major output: [sub][gy][gx .. gx+3]
            const int64_t wh = (int64_t)W * H, sub = (4 * i) / wh, rem = 4 * i - sub * wh;
            src = lm_strip_offset((int)sub, (int)(rem / W), (int)(rem % W), W, H);
        }
        const uint32_t v = *(const uint32_t*)(lmc + src);
#pragma unroll
        for (int o = 0; o < 8; ++o) *(uint32_t*)(lm + o * lm_stride + 4 * i) = response4(v, o);
    }
}

// unfused single-function kernels (stage entry points / parity tests)
__global__ __launch_bounds__(256) void k_spread(const uint8_t* __restrict__ src, int rows, int cols, int T,
                                                uint8_t* __restrict__ dst)
{
    const int64_t n = (int64_t)rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
        uint8_t v = 0;
        for (int dr = 0; dr < T && r + dr < rows; ++dr)
            for (int dc = 0; dc < T && c + dc < cols; ++dc) v |= src[(size_t)(r + dr) * cols + c + dc];
        dst[idx] = v;
    }
}

__global__ __launch_bounds__(256) void k_response(const uint8_t* __restrict__ spread, int64_t n,
                                                  uint8_t* __restrict__ maps)
{
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        uint32_t v = spread[idx];
#pragma unroll
        for (int o = 0; o < 8; ++o) maps[(int64_t)o * n + idx] = (uint8_t)response4(v, o);
    }
}

__global__ __launch_bounds__(256) void k_linearize(const uint8_t* __restrict__ map, int rows, int cols, int T,
                                                   uint8_t* __restrict__ lm)
{
    const int W = cols / T, H = rows / T;
    const int64_t n = (int64_t)rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        // idx enumerates the OUTPUT: [ty][tx][gy][gx]
        int gx = (int)(idx % W);
        int64_t t1 = idx / W;
        int gyy = (int)(t1 % H);
        int sub = (int)(t1 / H);
        int ty = sub / T, tx = sub - ty * T;
        lm[idx] = map[(size_t)(gyy * T + ty) * cols + gx * T + tx];
    }
}


} // namespace sbm
