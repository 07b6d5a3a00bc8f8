// quantize_stream_emu.cpp — TEST INFRASTRUCTURE: the SOURCE of the gfx950 row-streaming gradient kernel
// (shape_based_matching_amd/csrc/sbm_quantize_stream.h) compiled for the CPU against tests/emu/wave_emu.h, so the
// CPU test-suite can compare the kernel's dataflow with the oracle bit for bit.  Never loaded by the product.
#define SBM_WAVE_EMU 1
#define SBM_WAVE_HEADER "wave_emu.h"
#include <climits>
#include <cmath>
#include "sbm_quantize_stream.h"

extern "C" int sbm_emu_quantize_stream(const uint8_t* img, int rows, int cols, int stride, int ch, const uint8_t* mask,
                                       float weak, uint8_t* out, uint8_t* pyr, int hs)
{
    if ((ch != 1 && ch != 3) || cols < 4 || (cols & 3) || rows < 1 || hs < 2 || (hs & 1)) return -1;
    sbm::QSArgs a{};
    a.img = img;
    a.mask = mask;
    a.out = out;
    a.pyr = pyr;
    a.rows = rows;
    a.cols = cols;
    a.stride = stride;
    const float thr_sq = weak * weak;
    a.thr_i = thr_sq < 2147483000.f ? (int)floorf(thr_sq) : INT_MAX;
    a.hs = hs;
    a.n_strips = (cols + sbm::QS_USEFUL - 1) / sbm::QS_USEFUL;
    a.n_rblocks = (rows + hs - 1) / hs;
    for (int rb = 0; rb < a.n_rblocks; ++rb)
        for (int s = 0; s < a.n_strips; ++s) {
            if (ch == 3) sbm::quantize_stream_wave<3>(a, s, rb, 0);
            else sbm::quantize_stream_wave<1>(a, s, rb, 0);
        }
    return 0;
}
