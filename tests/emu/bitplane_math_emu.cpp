// TEST INFRASTRUCTURE: the host pass of shape_based_matching_amd/csrc/sbm_bitplane_math.h (the register-level arithmetic of
// the bit-plane producer and of the coarse pass on bit planes) behind a C interface, for tests/test_bitplane_math.py.
#include <stdint.h>
#include "sbm_bitplane_math.h"

// n slices (n % 8 == 0) added eight at a time to 32 counters of P planes that start at `bias`: counts[b] = final value
// mod 2^P of position b; returns the sticky OR of the carries out of the top plane
// inv: count the slices' ZERO bits through the adders' inverted-input form (what pass A of the coarse pass does)
template <int P, bool INV>
static uint32_t count(const uint32_t* x, int n, int bias, int32_t* counts)
{
    uint32_t c[P];
    for (int p = 0; p < P; ++p) c[p] = (bias >> p) & 1 ? ~0u : 0u;
    uint32_t dead = 0;
    for (int g = 0; g < n; g += 8) {
        uint32_t b8[8];
        for (int k = 0; k < 8; ++k) b8[k] = x[g + k];
        dead |= sbm::bitslice_add8<P, INV>(c, b8);
    }
    for (int b = 0; b < 32; ++b) counts[b] = sbm::bitslice_get<P>(c, b);
    return dead;
}

template <int P, bool INV>
static uint32_t count32(const uint32_t* x, int n, int bias, int32_t* counts)
{
    uint32_t c[P];
    for (int p = 0; p < P; ++p) c[p] = (bias >> p) & 1 ? ~0u : 0u;
    uint32_t dead = 0;
    for (int g = 0; g < n; g += 32) {
        uint32_t b32[32];
        for (int k = 0; k < 32; ++k) b32[k] = x[g + k];
        dead |= sbm::bitslice_add32<P, INV>(c, b32);
    }
    for (int b = 0; b < 32; ++b) counts[b] = sbm::bitslice_get<P>(c, b);
    return dead;
}

extern "C" {
// the 32-slice form (one ripple per 32 slices); n % 32 == 0
uint32_t sbm_emu_bitslice_count32(const uint32_t* x, int n, int P, int bias, int32_t* counts, int inv)
{
    switch (P) {
    case 7: return inv ? count32<7, true>(x, n, bias, counts) : count32<7, false>(x, n, bias, counts);
    case 10: return inv ? count32<10, true>(x, n, bias, counts) : count32<10, false>(x, n, bias, counts);
    case 13: return inv ? count32<13, true>(x, n, bias, counts) : count32<13, false>(x, n, bias, counts);
    default: return 0;
    }
}

// 32 spread bytes (cells 0..31) -> out[o] = bit o of every cell, cell b at bit b
void sbm_emu_bytes32_to_bitplanes(const uint8_t* cells, uint32_t* out)
{
    uint32_t d[8];
    for (int i = 0; i < 8; ++i)
        d[i] = (uint32_t)cells[4 * i] | (uint32_t)cells[4 * i + 1] << 8 | (uint32_t)cells[4 * i + 2] << 16 | (uint32_t)cells[4 * i + 3] << 24;
    sbm::bytes32_to_bitplanes(d);
    for (int o = 0; o < 8; ++o) out[o] = d[sbm::bitplane_reg(o)];
}

uint32_t sbm_emu_bitslice_count(const uint32_t* x, int n, int P, int bias, int32_t* counts, int inv)
{
    switch (P) {
    case 3: return inv ? count<3, true>(x, n, bias, counts) : count<3, false>(x, n, bias, counts);
    case 7: return inv ? count<7, true>(x, n, bias, counts) : count<7, false>(x, n, bias, counts);
    case 10: return inv ? count<10, true>(x, n, bias, counts) : count<10, false>(x, n, bias, counts);
    case 13: return inv ? count<13, true>(x, n, bias, counts) : count<13, false>(x, n, bias, counts);
    default: return 0;
    }
}
}
