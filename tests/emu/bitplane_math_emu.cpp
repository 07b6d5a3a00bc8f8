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

// The refinement pass on bit strips (sbm_local_bits.h) for ONE patch row: x[n] = per-feature dwords (any bits of the 16 columns
// | exact bits << 16).  As the kernel does it: feature j goes to slot j % 4; every slot counts its features in batches of
// 16 (a full window of 64 features) / 8 / 4 / 2; the four slot counters are added plane-wise in two steps; the row's best raw
// (3 #any + #exact) and its first column come from bitslice_row_best.
template <int P>
static uint32_t row_best(const uint32_t* x, int n, int* first)
{
    constexpr int PM = P + 4;
    uint32_t c[4][PM] = {};
    for (int g = 0; g < 4; ++g) {
        uint32_t v[8];
        uint32_t (&cp)[P] = *(uint32_t (*)[P])&c[g][0];
        // the slot's features in kernel order: windows of 64 features = 16 per slot
        for (int jb = 0; jb < n; jb += 64) {
            const int nbf = n - jb < 64 ? n - jb : 64;
            auto fetch = [&](int u, int k) { const int j = jb + u + 4 * k + g; return u + 4 * k + g < nbf && j < n ? x[j] : 0u; };
            int u = 0;
            auto run = [&](int steps) {
                for (int k = 0; k < steps; ++k) v[k] = fetch(u, k);
                if (steps == 8) { const uint32_t b8[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}; sbm::bitslice_add8<P>(cp, b8); }
                else if (steps == 4) sbm::bitslice_add4<P>(cp, v[0], v[1], v[2], v[3]);
                else sbm::bitslice_add2<P>(cp, v[0], v[1]);
            };
            if (nbf > 32) { run(8); u = 32; run(8); }
            else if (nbf > 16) run(8);
            else if (nbf > 8) run(4);
            else run(2);
        }
    }
    sbm::bitslice_add_planes<P>(c[0], c[1]);
    sbm::bitslice_add_planes<P>(c[2], c[3]);
    sbm::bitslice_add_planes<P + 1>(c[0], c[2]);
    return sbm::bitslice_row_best<P + 2>(c[0], first);
}

extern "C" {
uint32_t sbm_emu_local_row_best(const uint32_t* x, int n, int P, int* first)
{
    switch (P) {
    case 6: return row_best<6>(x, n, first);
    case 9: return row_best<9>(x, n, first);
    case 12: return row_best<12>(x, n, first);
    default: return 0;
    }
}

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
