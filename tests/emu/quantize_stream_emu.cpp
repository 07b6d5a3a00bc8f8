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
    a.row_lo = 0;
    a.row_hi = rows;
    a.n_strips = (cols + sbm::QS_USEFUL - 1) / sbm::QS_USEFUL;
    a.n_rblocks = (rows + hs - 1) / hs;
    a.frames = 1;
    for (int item = 0; item < sbm::quantize_stream_items(a); ++item) {
        if (ch == 3) sbm::quantize_stream_item<3>(a, item);
        else sbm::quantize_stream_item<1>(a, item);
    }
    return 0;
}

// a batch of `frames` contiguous frames, the way the engine launches it: with pack != 0 the last strip of up to
// 64 / pack_lanes frames shares a wave (quantize_stream_pack_lanes decides, as on the GPU)
extern "C" int sbm_emu_quantize_stream_batch(const uint8_t* img, int frames, int rows, int cols, int stride, int ch, const uint8_t* mask,
                                             float weak, uint8_t* out, uint8_t* pyr, int hs, int pack)
{
    if ((ch != 1 && ch != 3) || cols < 4 || (cols & 3) || rows < 1 || hs < 2 || (hs & 1) || frames < 1) return -1;
    sbm::QSArgs a{};
    a.img = img;
    a.mask = mask;
    a.out = out;
    a.pyr = pyr;
    a.img_fs = (int64_t)rows * stride;
    a.out_fs = (int64_t)rows * cols;
    a.pyr_fs = (int64_t)(rows / 2) * (cols / 2) * ch;
    a.rows = rows;
    a.cols = cols;
    a.stride = stride;
    const float thr_sq = weak * weak;
    a.thr_i = thr_sq < 2147483000.f ? (int)floorf(thr_sq) : INT_MAX;
    a.hs = hs;
    a.row_lo = 0;
    a.row_hi = rows;
    a.n_strips = (cols + sbm::QS_USEFUL - 1) / sbm::QS_USEFUL;
    a.n_rblocks = (rows + hs - 1) / hs;
    a.frames = frames;
    a.pack_lanes = pack ? sbm::quantize_stream_pack_lanes(rows, cols, ch, frames) : 0;
    a.pack_groups = a.pack_lanes ? (frames + 64 / a.pack_lanes - 1) / (64 / a.pack_lanes) : 0;
    for (int item = 0; item < sbm::quantize_stream_items(a); ++item) {
        if (ch == 3) sbm::quantize_stream_item<3>(a, item);
        else sbm::quantize_stream_item<1>(a, item);
    }
    return a.pack_lanes;
}

// one row band [row_lo, row_hi) of a level (what a rank of the build-sharded step launches): writes only that band's rows
// of `out` and the pyrDown rows [row_lo / 2, row_hi / 2) of `pyr`
extern "C" int sbm_emu_quantize_stream_band(const uint8_t* img, int rows, int cols, int stride, int ch, const uint8_t* mask,
                                            float weak, uint8_t* out, uint8_t* pyr, int hs, int row_lo, int row_hi)
{
    if ((ch != 1 && ch != 3) || cols < 4 || (cols & 3) || rows < 1 || hs < 2 || (hs & 1)) return -1;
    if (row_lo < 0 || row_hi > rows || row_lo >= row_hi || (row_lo & 1)) return -1;
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
    a.row_lo = row_lo;
    a.row_hi = row_hi;
    a.n_strips = (cols + sbm::QS_USEFUL - 1) / sbm::QS_USEFUL;
    a.n_rblocks = (row_hi - row_lo + hs - 1) / hs;
    a.frames = 1;
    for (int item = 0; item < sbm::quantize_stream_items(a); ++item) {
        if (ch == 3) sbm::quantize_stream_item<3>(a, item);
        else sbm::quantize_stream_item<1>(a, item);
    }
    return 0;
}

// the vote word (1 << 4 * label) the streaming kernel derives from a packed gradient, for n gradients: out[i] = label
// (0..7) of (gx[i], gy[i]) -- checked exhaustively against the oracle's 16-bin rule by the CPU tests
extern "C" void sbm_emu_vote_labels(const int16_t* gx, const int16_t* gy, int64_t n, uint8_t* out)
{
    for (int64_t b = 0; b < n; b += 64) {
        wv::V v;
        wv::P keep;
        for (int i = 0; i < 64; ++i) {
            const int64_t k = b + i < n ? b + i : n - 1;
            v.l[i] = ((uint32_t)(uint16_t)gx[k]) | ((uint32_t)(uint16_t)gy[k] << 16);
            keep.l[i] = v.l[i] != 0;
        }
        const wv::V w = sbm::qs_vote_word(v, keep);
        for (int i = 0; i < 64 && b + i < n; ++i) {
            int label = -1;
            for (int l = 0; l < 8; ++l)
                if (w.l[i] == (1u << (4 * l))) label = l;
            out[b + i] = (uint8_t)label;
        }
    }
}
