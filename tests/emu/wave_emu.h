// wave_emu.h — TEST INFRASTRUCTURE.  CPU emulation of shape_based_matching_amd/csrc/sbm_wave_gfx950.h: the same
// names, each implementing the documented semantics of the CDNA4 instruction it stands for, over explicit
// 64-lane arrays.  tests/emu/quantize_stream_emu.cpp compiles the row-streaming kernel's SOURCE
// (sbm_quantize_stream.h) against this header so that `pytest -m "not gpu"` can check the kernel's dataflow bit
// for bit against the oracle in a container without a GPU.  Nothing under shape_based_matching_amd/ includes it;
// it is never the thing measured or shipped.
#pragma once
#include <stdint.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#ifndef __restrict__
#define __restrict__
#endif

namespace wv {

constexpr int WAVE = 64;

struct V {
    uint32_t l[WAVE];
};
struct P {
    bool l[WAVE];
};

#define WV_FOR for (int i = 0; i < WAVE; ++i)
#define WV_BIN(OP)                                                                \
    inline V operator OP(const V& a, const V& b) { V r; WV_FOR r.l[i] = a.l[i] OP b.l[i]; return r; } \
    inline V operator OP(const V& a, uint32_t b) { V r; WV_FOR r.l[i] = a.l[i] OP b; return r; }
WV_BIN(+)
WV_BIN(-)
WV_BIN(&)
WV_BIN(|)
WV_BIN(^)
WV_BIN(*)
#undef WV_BIN
inline V operator<<(const V& a, uint32_t s) { V r; WV_FOR r.l[i] = a.l[i] << (s & 31); return r; }
inline V operator>>(const V& a, uint32_t s) { V r; WV_FOR r.l[i] = a.l[i] >> (s & 31); return r; }
// v_lshlrev_b32 / v_lshrrev_b32: the shift amount is the low 5 bits of the per-lane operand
inline V operator<<(const V& a, const V& s) { V r; WV_FOR r.l[i] = a.l[i] << (s.l[i] & 31); return r; }
inline V operator>>(const V& a, const V& s) { V r; WV_FOR r.l[i] = a.l[i] >> (s.l[i] & 31); return r; }
inline V operator<<(uint32_t a, const V& s) { V r; WV_FOR r.l[i] = a << (s.l[i] & 31); return r; }
inline V operator~(const V& a) { V r; WV_FOR r.l[i] = ~a.l[i]; return r; }

inline V lane_id() { V r; WV_FOR r.l[i] = (uint32_t)i; return r; }
inline V splat(uint32_t s) { V r; WV_FOR r.l[i] = s; return r; }
inline V from_left(const V& x) { V r; WV_FOR r.l[i] = i > 0 ? x.l[i - 1] : 0u; return r; }
inline V from_right(const V& x) { V r; WV_FOR r.l[i] = i < WAVE - 1 ? x.l[i + 1] : 0u; return r; }
inline uint32_t opaque(uint32_t k) { return k; }
inline V cold_splat(uint32_t s) { return splat(s); }
inline V dont_care() { return splat(0xdeadbeefu); }

inline uint32_t perm1(uint32_t hi, uint32_t lo, uint32_t sel)
{
    uint32_t r = 0;
    for (int b = 0; b < 4; ++b) {
        const uint32_t s = (sel >> (8 * b)) & 0xff;
        uint32_t v;
        if (s <= 3) v = (lo >> (8 * s)) & 0xff;
        else if (s <= 7) v = (hi >> (8 * (s - 4))) & 0xff;
        else if (s == 8) v = (lo & 0x8000u) ? 0xff : 0;      // sign of lo[15]
        else if (s == 9) v = (lo & 0x80000000u) ? 0xff : 0;  // sign of lo[31]
        else if (s == 10) v = (hi & 0x8000u) ? 0xff : 0;     // sign of hi[15]
        else if (s == 11) v = (hi & 0x80000000u) ? 0xff : 0; // sign of hi[31]
        else if (s == 12) v = 0;
        else v = 0xff;
        r |= v << (8 * b);
    }
    return r;
}
inline V perm(const V& hi, const V& lo, const V& sel) { V r; WV_FOR r.l[i] = perm1(hi.l[i], lo.l[i], sel.l[i]); return r; }
inline V perm(const V& hi, const V& lo, uint32_t sel) { V r; WV_FOR r.l[i] = perm1(hi.l[i], lo.l[i], sel); return r; }
inline V alignbit(const V& hi, const V& lo, uint32_t sh)
{
    V r;
    WV_FOR r.l[i] = (uint32_t)((((uint64_t)hi.l[i] << 32) | lo.l[i]) >> (sh & 31));
    return r;
}
// the shift amount in a VGPR: its low 5 bits, per lane
inline V alignbit(const V& hi, const V& lo, const V& sh)
{
    V r;
    WV_FOR r.l[i] = (uint32_t)((((uint64_t)hi.l[i] << 32) | lo.l[i]) >> (sh.l[i] & 31));
    return r;
}
inline V udot2(const V& a, uint32_t b, const V& c)
{
    V r;
    WV_FOR r.l[i] = (a.l[i] & 0xffff) * (b & 0xffff) + (a.l[i] >> 16) * (b >> 16) + c.l[i];
    return r;
}
inline V udot2(const V& a, uint32_t b, uint32_t c) { return udot2(a, b, splat(c)); }
inline int32_t sdot2_1(uint32_t a, uint32_t b, int32_t c)
{
    return (int32_t)(int16_t)(a & 0xffff) * (int32_t)(int16_t)(b & 0xffff) + (int32_t)(int16_t)(a >> 16) * (int32_t)(int16_t)(b >> 16) + c;
}
inline V sdot2(const V& a, const V& b, uint32_t c) { V r; WV_FOR r.l[i] = (uint32_t)sdot2_1(a.l[i], b.l[i], (int32_t)c); return r; }
inline V sdot2(const V& a, uint32_t b, uint32_t c) { V r; WV_FOR r.l[i] = (uint32_t)sdot2_1(a.l[i], b, (int32_t)c); return r; }
// v_dot2_i32_i16 ... 0 clamp: saturating; the int64 sum shows an operand pair that would make the clamp matter
inline V sdot2_small(const V& a, const V& b)
{
    V r;
    WV_FOR {
        int64_t s = (int64_t)(int16_t)(a.l[i] & 0xffff) * (int16_t)(b.l[i] & 0xffff) + (int64_t)(int16_t)(a.l[i] >> 16) * (int16_t)(b.l[i] >> 16);
        r.l[i] = (uint32_t)(int32_t)(s > INT32_MAX ? INT32_MAX : (s < INT32_MIN ? INT32_MIN : s));
    }
    return r;
}

#define WV_PK2(NAME, EXPR)                                                                         \
    inline uint32_t NAME##1(uint32_t a, uint32_t b)                                                \
    {                                                                                              \
        uint32_t r = 0;                                                                            \
        for (int h = 0; h < 2; ++h) {                                                              \
            const uint32_t x = (a >> (16 * h)) & 0xffff, y = (b >> (16 * h)) & 0xffff;             \
            r |= ((uint32_t)(EXPR) & 0xffff) << (16 * h);                                          \
        }                                                                                          \
        return r;                                                                                  \
    }                                                                                              \
    inline V NAME(const V& a, const V& b) { V r; WV_FOR r.l[i] = NAME##1(a.l[i], b.l[i]); return r; } \
    inline V NAME(const V& a, uint32_t b) { V r; WV_FOR r.l[i] = NAME##1(a.l[i], b); return r; }      \
    inline V NAME(uint32_t a, const V& b) { V r; WV_FOR r.l[i] = NAME##1(a, b.l[i]); return r; }
WV_PK2(pk_add, x + y)
WV_PK2(pk_sub, x - y)
WV_PK2(pk_mul, x * y)
WV_PK2(pk_max_i16, ((int16_t)x > (int16_t)y ? x : y))
#undef WV_PK2
inline V pk_mad(const V& a, uint32_t b, const V& c) { return pk_add(pk_mul(a, b), c); }
inline V pk_mad(const V& a, const V& b, const V& c) { return pk_add(pk_mul(a, b), c); }
inline V pk_lshr(const V& a, uint32_t sh)
{
    V r;
    WV_FOR r.l[i] = (((a.l[i] & 0xffff) >> sh) & 0xffff) | (((a.l[i] >> 16) >> sh) << 16);
    return r;
}

inline V add3(const V& a, const V& b, const V& c) { return a + b + c; }
inline V lshl_or(const V& a, uint32_t sh, const V& b) { return (a << sh) | b; }
inline V ashr(const V& a, uint32_t sh) { V r; WV_FOR r.l[i] = (uint32_t)((int32_t)a.l[i] >> (sh & 31)); return r; }
inline V popcount(const V& a) { V r; WV_FOR r.l[i] = (uint32_t)__builtin_popcount(a.l[i]); return r; }
inline V mul_u24(const V& a, uint32_t b) { V r; WV_FOR r.l[i] = (a.l[i] & 0xffffff) * (b & 0xffffff); return r; }
inline V ffbl(const V& a) { V r; WV_FOR r.l[i] = a.l[i] ? (uint32_t)__builtin_ctz(a.l[i]) : 0xffffffffu; return r; }
inline V clamp_i(const V& x, int lo, int hi)
{
    V r;
    WV_FOR { const int v = (int)x.l[i]; r.l[i] = (uint32_t)(v < lo ? lo : (v > hi ? hi : v)); }
    return r;
}

#define WV_CMP(NAME, EXPR)                                                                               \
    inline P NAME(const V& a, const V& b) { P r; WV_FOR { const uint32_t x = a.l[i], y = b.l[i]; r.l[i] = (EXPR); } return r; } \
    inline P NAME(const V& a, uint32_t b) { P r; WV_FOR { const uint32_t x = a.l[i], y = b; r.l[i] = (EXPR); } return r; }
WV_CMP(eq, x == y)
WV_CMP(ne, x != y)
WV_CMP(lt_i, (int32_t)x < (int32_t)y)
WV_CMP(gt_i, (int32_t)x > (int32_t)y)
WV_CMP(ge_i, (int32_t)x >= (int32_t)y)
#undef WV_CMP
inline P p_and(const P& a, const P& b) { P r; WV_FOR r.l[i] = a.l[i] && b.l[i]; return r; }
inline P p_or(const P& a, const P& b) { P r; WV_FOR r.l[i] = a.l[i] || b.l[i]; return r; }
inline P p_not(const P& a) { P r; WV_FOR r.l[i] = !a.l[i]; return r; }
inline V select(const P& p, const V& a, const V& b) { V r; WV_FOR r.l[i] = p.l[i] ? a.l[i] : b.l[i]; return r; }
inline V select(const P& p, const V& a, uint32_t b) { V r; WV_FOR r.l[i] = p.l[i] ? a.l[i] : b; return r; }
inline V select(const P& p, uint32_t a, const V& b) { V r; WV_FOR r.l[i] = p.l[i] ? a : b.l[i]; return r; }
inline V select(const P& p, uint32_t a, uint32_t b) { V r; WV_FOR r.l[i] = p.l[i] ? a : b; return r; }
inline bool all(const P& p) { WV_FOR if (!p.l[i]) return false; return true; }
inline uint32_t read_first(const V& a) { return a.l[0]; }

inline V load_u32(const uint8_t* base, const V& off)
{
    V r;
    WV_FOR memcpy(&r.l[i], base + off.l[i], 4);
    return r;
}
// buffer stores: a lane whose offset lies at or beyond the buffer size is dropped (hardware range check on the
// per-lane offset); a kept lane that would write outside the buffer is a kernel bug -> abort
struct Buf {
    uint8_t* base;
    uint32_t bytes;
};
constexpr uint32_t BUF_DROP = 0x7ffffff0u;
inline Buf make_buf(uint8_t* base, uint32_t bytes) { return Buf{base, bytes}; }
inline void buf_store_n(const Buf& b, const V& voff, uint32_t soff, const V& v, uint32_t n)
{
    WV_FOR {
        if ((uint64_t)voff.l[i] + n > b.bytes) continue;
        if ((uint64_t)voff.l[i] + soff + n > b.bytes || ((voff.l[i] + soff) & (n - 1))) __builtin_trap();
        memcpy(b.base + voff.l[i] + soff, &v.l[i], n);
    }
}
inline void buf_store_u32(const Buf& b, const V& voff, uint32_t soff, const V& v) { buf_store_n(b, voff, soff, v, 4); }
inline void buf_store_u16(const Buf& b, const V& voff, uint32_t soff, const V& v) { buf_store_n(b, voff, soff, v, 2); }
#undef WV_FOR

} // namespace wv
