// sbm_wave_gfx950.h — the wave-level vocabulary of the row-streaming kernels (sbm_quantize_stream.h), gfx950.
//
// A kernel written against this header is a pure dataflow program over per-lane 32-bit values `V` and lane
// predicates `P` of one 64-lane wavefront, with wave-uniform control flow only: every function below is one
// (or two) CDNA4 instructions.  tests/emu/wave_emu.h implements the same names on the CPU (64-element arrays,
// the instructions' documented semantics), which lets the CPU test-suite run the kernel's source bit for bit
// against the oracle without a GPU; the product only ever includes THIS header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wv {

typedef uint32_t V; // one dword per lane
typedef bool P;     // one predicate bit per lane (an SGPR pair once the compiler has vectorised it)

typedef unsigned short wv_us2 __attribute__((ext_vector_type(2)));
typedef short wv_ss2 __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((aligned(1))) wv_u32_unaligned;
typedef uint16_t __attribute__((aligned(1))) wv_u16_unaligned;

__device__ __forceinline__ V lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ V splat(uint32_t s) { return s; }

// a constant assigned on a rarely taken path: materialised THERE (a plain constant would be hoisted above the
// branch as a default and cost the common path one v_mov per value)
__device__ __forceinline__ V cold_splat(uint32_t s)
{
    V r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s));
    return r;
}
// a value nobody reads: an empty asm "defines" it where it is written, so the register allocator simply gives it the
// register of the value it joins -- no instruction, and nothing the optimiser could materialise as a constant in
// front of the branch (__builtin_nondeterministic_value became 12 x v_mov 0 per row on the common path)
__device__ __forceinline__ V dont_care()
{
    V r;
    asm volatile("" : "=v"(r));
    return r;
}

// lane i <- lane i-1 (v_mov_b32_dpp wave_shr:1 bound_ctrl:0); lane 0 reads 0
__device__ __forceinline__ V from_left(V x) { return (V)__builtin_amdgcn_mov_dpp((int)x, 0x138, 0xf, 0xf, true); }
// lane i <- lane i+1 (wave_shl:1 bound_ctrl:0); lane 63 reads 0
__device__ __forceinline__ V from_right(V x) { return (V)__builtin_amdgcn_mov_dpp((int)x, 0x130, 0xf, 0xf, true); }
// a wave-uniform constant the optimiser must not look into (keeps it an SGPR operand of the instruction written)
__device__ __forceinline__ uint32_t opaque(uint32_t k)
{
    asm("" : "+s"(k));
    return k;
}

// v_perm_b32: result byte i = selector byte i picks 0..3 -> lo.byte, 4..7 -> hi.byte, 0x0c -> 0x00
__device__ __forceinline__ V perm(V hi, V lo, V sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// v_alignbit_b32: ({hi, lo} >> (sh & 31)) & 0xffffffff; sh may be per lane (the instruction reads its low 5 bits)
__device__ __forceinline__ V alignbit(V hi, V lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }

// v_dot2_u32_u16 / v_dot2_i32_i16: a.lo*b.lo + a.hi*b.hi + c (32-bit wrap-around)
__device__ __forceinline__ V udot2(V a, V b, V c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(wv_us2, a), __builtin_bit_cast(wv_us2, b), c, false);
}
__device__ __forceinline__ V sdot2(V a, V b, V c)
{
    return (V)__builtin_amdgcn_sdot2(__builtin_bit_cast(wv_ss2, a), __builtin_bit_cast(wv_ss2, b), (int)c, false);
}

// a.lo*b.lo + a.hi*b.hi for operands whose sum cannot leave the int32 range (the caller guarantees it): the clamp bit
// then changes nothing, but it selects the VOP3P encoding v_dot2_i32_i16 with an inline 0 addend -- the plain form is
// matched to the VOP2 v_dot2c_i32_i16, whose accumulator is tied to the destination and costs a v_mov 0 per use
__device__ __forceinline__ V sdot2_small(V a, V b)
{
    return (V)__builtin_amdgcn_sdot2(__builtin_bit_cast(wv_ss2, a), __builtin_bit_cast(wv_ss2, b), 0, true);
}

// packed 2 x 16-bit lanes, wrap-around (v_pk_add_u16, v_pk_sub_u16, v_pk_mul_lo_u16, v_pk_mad_u16, v_pk_max_i16,
// v_pk_lshrrev_b16)
__device__ __forceinline__ V pk_add(V a, V b) { return __builtin_bit_cast(V, (wv_us2)(__builtin_bit_cast(wv_us2, a) + __builtin_bit_cast(wv_us2, b))); }
__device__ __forceinline__ V pk_sub(V a, V b) { return __builtin_bit_cast(V, (wv_us2)(__builtin_bit_cast(wv_us2, a) - __builtin_bit_cast(wv_us2, b))); }
__device__ __forceinline__ V pk_mul(V a, V b) { return __builtin_bit_cast(V, (wv_us2)(__builtin_bit_cast(wv_us2, a) * __builtin_bit_cast(wv_us2, b))); }
__device__ __forceinline__ V pk_mad(V a, V b, V c)
{
    return __builtin_bit_cast(V, (wv_us2)(__builtin_bit_cast(wv_us2, a) * __builtin_bit_cast(wv_us2, b) + __builtin_bit_cast(wv_us2, c)));
}
__device__ __forceinline__ V pk_max_i16(V a, V b)
{
    return __builtin_bit_cast(V, __builtin_elementwise_max(__builtin_bit_cast(wv_ss2, a), __builtin_bit_cast(wv_ss2, b)));
}
__device__ __forceinline__ V pk_lshr(V a, uint32_t sh)
{
    const wv_us2 s = {(unsigned short)sh, (unsigned short)sh};
    return __builtin_bit_cast(V, (wv_us2)(__builtin_bit_cast(wv_us2, a) >> s));
}

__device__ __forceinline__ V add3(V a, V b, V c) { return a + b + c; }              // v_add3_u32
__device__ __forceinline__ V lshl_or(V a, uint32_t sh, V b) { return (a << sh) | b; } // v_lshl_or_b32
__device__ __forceinline__ V ashr(V a, uint32_t sh) { return (V)((int32_t)a >> sh); }
__device__ __forceinline__ V popcount(V a) { return (V)__builtin_popcount(a); }
__device__ __forceinline__ V mul_u24(V a, V b) { return (V)__umul24(a, b); }
// v_ffbl_b32: index of the lowest set bit, 0xffffffff for 0
__device__ __forceinline__ V ffbl(V a)
{
    V r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ V clamp_i(V x, int lo, int hi)
{
    const int v = (int)x;
    return (V)(v < lo ? lo : (v > hi ? hi : v));
}

// predicates
__device__ __forceinline__ P eq(V a, V b) { return a == b; }
__device__ __forceinline__ P ne(V a, V b) { return a != b; }
__device__ __forceinline__ P lt_i(V a, V b) { return (int32_t)a < (int32_t)b; }
__device__ __forceinline__ P gt_i(V a, V b) { return (int32_t)a > (int32_t)b; }
__device__ __forceinline__ P ge_i(V a, V b) { return (int32_t)a >= (int32_t)b; }
__device__ __forceinline__ P p_and(P a, P b) { return a && b; }
__device__ __forceinline__ P p_or(P a, P b) { return a || b; }
__device__ __forceinline__ P p_not(P a) { return !a; }
__device__ __forceinline__ V select(P p, V a, V b) { return p ? a : b; }
__device__ __forceinline__ bool all(P p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; } // wave-uniform
__device__ __forceinline__ uint32_t read_first(V a) { return __builtin_amdgcn_readfirstlane(a); }

// memory: wave-uniform base pointer + per-lane byte offset
__device__ __forceinline__ V load_u32(const uint8_t* base, V off) { return *(const wv_u32_unaligned*)(base + off); }

// Stores go through a buffer resource: a lane whose offset is >= the buffer size (BUF_DROP) is discarded by the
// hardware range check, so predicated stores need no branch and the kernel keeps wave-uniform control flow.
struct Buf {
    __amdgpu_buffer_rsrc_t rsrc;
};
constexpr uint32_t BUF_DROP = 0x7ffffff0u;
__device__ __forceinline__ Buf make_buf(uint8_t* base, uint32_t bytes)
{
    Buf b;
    b.rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
    return b;
}
// address = base + voff (per lane) + soff (wave-uniform); 4- / 2-byte aligned by construction
__device__ __forceinline__ void buf_store_u32(const Buf& b, V voff, uint32_t soff, V v) { __builtin_amdgcn_raw_buffer_store_b32(v, b.rsrc, (int)voff, (int)soff, 0); }
__device__ __forceinline__ void buf_store_u16(const Buf& b, V voff, uint32_t soff, V v)
{
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, b.rsrc, (int)voff, (int)soff, 0);
}

} // namespace wv
