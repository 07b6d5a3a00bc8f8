// sbm_quantize_stream.h — the gradient stage of Detector::match as a barrier-free, LDS-free row-streaming
// kernel for gfx950 (wave64).  Replaces, for the hot path, the tile kernel k_quantize of sbm_kernels.h.
//
// Reference functions restated (ddcr/shape_based_matching):
//   quantizedOrientations        line2Dup.cpp:313-404   GaussianBlur 7x7 (BORDER_REPLICATE), Sobel 3x3, squared
//                                                       magnitude, channel of maximum magnitude, orientation
//   hysteresisGradient           line2Dup.cpp:218-311   16 -> 8 bins, border ring, 3x3 majority vote (>= 5)
//   ColorGradientPyramid::pyrDown line2Dup.cpp:424-444  cv::pyrDown of the source image (next level's input)
//   ColorGradientPyramid::quantize line2Dup.cpp:446-450 mask
//
// Work decomposition.  One WAVE owns a column strip of 256 pixels (4 per lane; 2 halo lanes on either side, so
// 240 useful columns) and a block of `hs` output rows, and marches down the rows.  Everything a row needs from
// the rows above lives in registers (the 7-row Gaussian window, two rows of Sobel partial sums, two rows of vote
// sums); everything it needs from the columns left and right comes from the neighbouring lanes through DPP
// (wave_shr / wave_shl).  There is no LDS, no barrier and no per-phase index arithmetic: the waves of a workgroup
// never talk to each other, and control flow is wave-uniform throughout.
//
// Per row and lane (4 pixels), all integer and exact:
//   load      12 (BGR) / 4 (gray) source bytes, clamped = BORDER_REPLICATE; de-interleave + widen to u16 pairs
//             E = (p0, p2), O = (p1, p3) with v_perm_b32
//   Gaussian  vertical 7-tap {8,28,56,72,56,28,8} on packed u16 (v_pk_add / v_pk_mad_u16: <= 65280, no overflow),
//             horizontal 7-tap as 18 v_dot2_u32_u16 (32-bit sums), one rounding (sum + 2^15) >> 16 -- the two
//             8.8 fixed-point passes of cv::GaussianBlur are exact, so their order does not matter
//   Sobel     horizontal parts (s[x+1]-s[x-1], s[x-1]+2s[x]+s[x+1]) packed, kept for two rows; vertical combine
//   select    gx^2+gy^2 by v_dot2_i32_i16, channel of maximum magnitude (ties: lower channel, :370-387)
//   label     vote word 1 << 4*label from four sign tests of integer linear forms (sbm_kernels.h vote_word)
//   vote      3x3 sum of vote words (eight 4-bit counters), >= 5 test as (s + 0x33333333) & mask, one-hot byte
//   pyrDown   [1 4 6 4 1]^2 of the source rows already in the window, every second row, packed u16 throughout
// BORDER_REPLICATE of cv::Sobel never reaches the output: a pixel whose 3x3 window leaves the image is on the
// 1-pixel ring that hysteresisGradient zeroes (:229-236), so only the Gaussian's replicate border is materialised.
//
// Constant-row shortcut: while the last 11 source rows of the strip are one constant colour -- or every row the work
// item has loaded so far is -- the output row is 0, the pyrDown row is that colour and the carried state is known in
// closed form; such rows skip the arithmetic.
//
// Batches: the narrow last strip of a row (cols - 240 k columns) of several frames shares one wave (QSArgs::pack_lanes),
// and the work items are ordered row block first, so that the three workgroups a CU holds come from three regions of
// the image (quantize_stream_item).
//
// Requires cols % 4 == 0 and cols >= 4 (every pyramid level of a match() frame with T in {4, 8, ...} satisfies
// it); other geometries and the float outputs (magnitude / angle of the stage API and of training) stay with
// k_quantize.
#pragma once
#ifndef SBM_WAVE_HEADER
#define SBM_WAVE_HEADER "sbm_wave_gfx950.h"
#endif
#include SBM_WAVE_HEADER

namespace sbm {

constexpr int QS_LANE_PX = 4;
constexpr int QS_HALO_LANES = 2;
constexpr int QS_USEFUL = (64 - 2 * QS_HALO_LANES) * QS_LANE_PX; // 240 useful columns per strip
constexpr int QS_PREFETCH = 4;                                   // source rows in flight ahead of the one being consumed

struct QSArgs {
    const uint8_t* img;  // frame 0, level image, `stride` bytes per row, CH interleaved channels
    const uint8_t* mask; // rows x cols, may be null
    uint8_t* out;        // rows x cols one-hot orientation bytes
    uint8_t* pyr;        // (rows/2) x (cols/2) x CH, may be null
    int64_t img_fs, out_fs, pyr_fs; // bytes from one frame of the batch to the next
    int32_t rows, cols, stride;
    int32_t thr_i;       // floor(weak^2): `mag > weak^2` in integers
    int32_t hs;          // output rows per work item (even)
    // Output rows of this launch: [row_lo, row_hi), 0 .. rows for a whole level.  A row BAND of a level is what one
    // rank of a build-sharded step computes (sbm_match_batch_device_banded): the band's rows depend on source rows
    // outside it exactly as they do in a whole-level launch (loads are clamped to the IMAGE, not to the band), so a
    // level assembled from bands equals the level computed at once.  n_rblocks = ceil((row_hi - row_lo) / hs).
    int32_t row_lo, row_hi;
    int32_t n_strips, n_rblocks;
    int32_t frames;      // frames of the batch
    // Packed last strip.  The last strip of a row covers cols - 240 (n_strips - 1) columns -- 64 of 1024, 32 of 512 --
    // and a wave of its own would leave most lanes idle.  With pack_lanes > 0 a wave takes that strip of up to
    // 64 / pack_lanes FRAMES side by side: segment g of pack_lanes lanes (2 halo + useful + 2 halo) is frame f0 + g.
    // Same rows, same columns, so every scalar decision of the row loop is unchanged; only the per-lane column and
    // the per-lane frame offset of loads and stores differ.  pack_groups = frame groups per row block.
    int32_t pack_lanes, pack_groups;
};

// Gaussian / pyrDown weights as packed pairs (lo | hi << 16)
#define QS_K(lo, hi) ((uint32_t)(lo) | ((uint32_t)(hi) << 16))

template <int CH>
struct QSState {
    wv::V win[7][CH][2];                              // source rows y-6 .. y as u16 pairs: [slot][channel][E / O]
    wv::V dx1[CH][2], dx2[CH][2], sx1[CH][2], sx2[CH][2]; // Sobel horizontal parts of smoothed rows yg-1, yg-2
    wv::V hs1[4], hs2[4];                             // horizontal 3-sums of the vote words of rows ys-1, ys-2
    wv::V sm1[4];                                     // "strong & allowed" masks (0x88888888 or 0) of row ys-1
};

// vote word 1 << 4*label of a packed gradient v = (gx & 0xffff) | gy << 16 (see vote_word in sbm_kernels.h): the
// four sign tests p*|gy| - q*|gx| of the first quadrant as v_dot2_u32_u16 over (1023 - |gx|, |gy|) with the constant
// -1023 q folded into the addend (|gx| <= 1020 < 1024, so 1023 - |gx| = |gx| ^ 1023): same 32-bit result, and the
// unsigned form is a VOP3P instruction with a free addend (the signed one is v_dot2c: tied accumulator + v_mov)
// keep: the lanes / pixels whose vote is real (a real column AND a non-zero gradient); every other pixel votes for bin 0
// (the ring and the columns outside the image by hysteresisGradient's border rule, a zero gradient because fastAtan2(0, 0)
// is 0).  One select for both conditions: the predicates are combined on the scalar unit.
__device__ __forceinline__ wv::V qs_vote_word(wv::V v, wv::P keep)
{
    using namespace wv;
    const V av = pk_max_i16(v, pk_sub(splat(0u), v)) ^ 0x000003ffu; // (1023 - |gx|, |gy|)
    const V u1 = udot2(av, QS_K(73, 367), 0u - 73u * 1023u);         // 367|gy| -  73|gx| >= 0
    const V u2 = udot2(av, QS_K(264, 395), 0u - 264u * 1023u);       // 395|gy| - 264|gx| >= 0
    const V u3 = udot2(av, QS_K(395, 264), 0u - 395u * 1023u - 1u);  // 264|gy| - 395|gx| >  0
    const V u4 = udot2(av, QS_K(367, 73), 0u - 367u * 1023u - 1u);   //  73|gy| - 367|gx| >  0
    // k = number of tests that fail = boundaries NOT passed in the first quadrant: the four sign bits added up (full-rate
    // shifts and adds; round 2 packed them with three v_alignbit and counted with v_bcnt)
    const V k = add3(u1 >> 31, u2 >> 31, u3 >> 31) + (u4 >> 31);
    // the vote word is 1 << 4*label with label = 4 - k, mirrored to (8 - label) & 7 when exactly one of gx, gy is
    // negative: a ROTATE of 1 by -4*label or +4*label -- v_alignbit(1, 1, y) is 1 rotated right by y & 31, so y = 4k - 16
    // (= -4*label) gives 1 << 4*label, and y negated gives the mirrored word; no masking, no variable shift
    const V y = (k << 2) - 16u;
    const V m = ashr((v << 16) ^ v, 31); // all ones iff exactly one of gx, gy is negative
    const V w = alignbit(splat(1u), splat(1u), (y ^ m) - m);
    return select(keep, w, 1u);
}

// One work item: strip `strip`, output rows [rb*hs, min(rb*hs+hs, rows)), frame `frame`.
// nseg == 0: the whole wave is strip `strip` of frame `frame`; nseg > 0: packed last strip of frames frame .. frame+nseg-1
template <int CH>
__device__ __forceinline__ void quantize_stream_wave(const QSArgs& a, int strip, int rb, int frame, int nseg = 0)
{
    using namespace wv;
    constexpr int ND = CH == 3 ? 3 : 1; // source dwords per lane and row
    const int rows = a.rows, cols = a.cols;
    // Every work item owns exactly min(hs, row_hi - row_lo) output rows: the last row block is moved up to end at the
    // launch's last row (it then recomputes a few rows of the block above -- the same bytes, stored twice).  All items of a launch run
    // the same number of row iterations, in whole groups of 7 (see the row loop).
    int R0 = a.row_lo + rb * a.hs;
    if (R0 + a.hs > a.row_hi) R0 = a.row_hi - a.row_lo > a.hs ? a.row_hi - a.hs : a.row_lo;
    const int R1 = R0 + a.hs < a.row_hi ? R0 + a.hs : a.row_hi;
    const int cb = strip * QS_USEFUL - QS_HALO_LANES * QS_LANE_PX; // column of lane 0, pixel 0
    const uint8_t* img = a.img + (int64_t)frame * a.img_fs;
    uint8_t* out = a.out + (int64_t)frame * a.out_fs;
    uint8_t* pyr = a.pyr ? a.pyr + (int64_t)frame * a.pyr_fs : nullptr;
    const int drows = rows >> 1, dcols = cols >> 1;

    // ---- per-lane constants ----
    const V lane = lane_id();
    V lseg = lane, segv = splat(0u); // lane inside its segment, segment (= frame - `frame`) of the lane
    const uint32_t seg_lanes = nseg ? (uint32_t)a.pack_lanes : 64u;
    if (nseg) {
        for (uint32_t k = 1; k * seg_lanes < 64u; ++k) segv = segv + select(ge_i(lane, splat(k * seg_lanes)), 1u, 0u);
        lseg = lane - segv * seg_lanes;
    }
    const P seg_ok = lt_i(segv, splat(nseg ? (uint32_t)nseg : 1u)); // lanes past the last frame of the group idle
    const V segc = select(seg_ok, segv, 0u);
    const V c0 = splat((uint32_t)cb) + (lseg << 2);       // first column of this lane (may be negative / >= cols)
    const V lcol = clamp_i(c0, 0, cols - 4);              // column actually loaded (BORDER_REPLICATE)
    const V ld_off = (CH == 3 ? lcol + (lcol << 1) : lcol) + segc * (uint32_t)a.img_fs; // byte offset from the frame's source row
    const P left_out = lt_i(c0, splat(0u)), right_out = ge_i(c0, splat((uint32_t)cols));
    const P outside = p_or(left_out, right_out);
    const bool border_strip = nseg || cb < 0 || cb + 256 > cols; // some lane replicates the first / last pixel
    // pixels allowed to be non-zero / to carry a real vote: columns 1 .. cols-2 (ring and outside vote for bin 0)
    V cm[4];
    P real_col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const V c = c0 + (uint32_t)j;
        real_col[j] = p_and(gt_i(c, splat(0u)), lt_i(c, splat((uint32_t)(cols - 1))));
        cm[j] = select(real_col[j], 0x88888888u, 0u);
    }
    const P store_ok = p_and(p_and(p_and(ge_i(lseg, splat((uint32_t)QS_HALO_LANES)), lt_i(lseg, splat(seg_lanes - QS_HALO_LANES))),
                                   p_and(ge_i(c0, splat(0u)), lt_i(c0, splat((uint32_t)cols)))),
                             seg_ok);
    // lanes that own no output (halo, past the right border, no frame) carry an offset the buffer range check discards;
    // a packed wave's buffers span its nseg frames (out_fs, pyr_fs: one frame's bytes)
    const uint32_t nfr = nseg ? (uint32_t)nseg : 1u;
    const Buf out_buf = make_buf(out, nseg ? nfr * (uint32_t)a.out_fs : (uint32_t)rows * (uint32_t)cols);
    const Buf pyr_buf = make_buf(pyr, pyr ? (nseg ? nfr * (uint32_t)a.pyr_fs : (uint32_t)drows * (uint32_t)dcols * CH) : 0u);
    const V out_off = select(store_ok, c0 + segc * (uint32_t)a.out_fs, BUF_DROP);                  // + yv * cols
    const V pyr_off = select(store_ok, (CH == 3 ? (c0 >> 1) + ((c0 >> 1) << 1) : (c0 >> 1)) + segc * (uint32_t)a.pyr_fs,
                             BUF_DROP);                                                           // (c0 / 2) * CH
    // border lanes: byte selectors that turn the loaded (clamped) group into 4 copies of its first / last pixel
    V selA = splat(0x03020100u), selB = splat(0x03020100u), selC = splat(0x07060504u);
    if (CH == 3) {
        selA = select(left_out, 0x00020100u, select(right_out, 0x05070605u, 0x03020100u));
        selB = select(left_out, 0x01000201u, 0x06050706u); // only used by outside lanes
        selC = select(left_out, 0x02010002u, select(right_out, 0x07060507u, 0x07060504u));
    } else {
        selA = select(left_out, 0x00000000u, select(right_out, 0x03030303u, 0x03020100u));
    }
    // pyrDown REFLECT_101 at the left / right image border: the lane holding column 0 / column cols-4
    const P pyr_first = eq(c0, splat(0u)), pyr_last = eq(c0, splat((uint32_t)(cols - 4)));

    // multipliers that are powers of two: kept opaque so that they stay v_pk_mad_u16 operands (one instruction)
    // instead of being strength-reduced into v_pk_lshlrev_b16 + v_pk_add_u16 (two)
    const uint32_t K2 = opaque(QS_K(2, 2)), K4 = opaque(QS_K(4, 4)), K8 = opaque(QS_K(8, 8));

    QSState<CH> st;
    V dq[7][ND]; // source rows in flight, ring indexed like the window: row i lives in dq[i % 7]
    auto issue_load = [&](int i, V (&d)[ND]) {
        int y = R0 - 5 + i;
        y = y < 0 ? 0 : (y > rows - 1 ? rows - 1 : y);
        const uint8_t* rowp = img + (int64_t)y * a.stride;
#pragma unroll
        for (int q = 0; q < ND; ++q) d[q] = load_u32(rowp + 4 * q, ld_off);
    };

    const int n_iter = R1 - R0 + 10;
#pragma unroll
    for (int p = 0; p < QS_PREFETCH; ++p) issue_load(p, dq[p]);

    int flat_cnt = 0;         // consecutive loaded rows of one constant colour (the latest included)
    uint32_t flat_key = 0;    // that colour
    bool in_run = false;      // the previous row took the constant-row shortcut (the carried state is already constant)

    // The row loop runs whole groups of 7 iterations (the window's rotation period, unrolled so that every register
    // index is static) and never leaves a group early: an iteration that is skipped hands every carried value on
    // UNCHANGED, while one that runs hands its two-row carries on EXCHANGED (x2 <- x1, x1 <- new); with both kinds of
    // edges into the same join the register allocator has to copy the whole state every row (36 v_mov per row and lane
    // measured).  Iterations past n_iter (at most 6, none when hs + 10 is a multiple of 7, which is what the host picks)
    // compute on clamped rows and store nothing.
    const int n_groups = (n_iter + 6) / 7;
    for (int grp = 0; grp < n_groups; ++grp) {
        const int i0 = grp * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = i0 + k;
            const bool store_row = i >= 10 && i < n_iter; // this iteration's output row yv is one of the item's rows
            const int y = R0 - 5 + i; // source row entering the window (slot k)
            // unconditional (rows past the end re-read the clamped last row): a conditional write would keep all
            // seven ring entries live across the loop
            issue_load(i + QS_PREFETCH, dq[(k + QS_PREFETCH) % 7]);
            V d[ND];
#pragma unroll
            for (int q = 0; q < ND; ++q) d[q] = dq[k][q];

            // ---- constant-row test on the loaded bytes (a clamped group outside the image is a group of the image, so the
            //      test sees the replicated border as well) ----
            // Round 4: the test runs on every row only while a run is being counted (flat_cnt > 0); otherwise on every fourth
            // row.  A row that is not tested counts as not constant -- the shortcut then starts up to three rows later, never
            // wrongly -- and textured content pays a quarter of the test (3 compares + the wave-wide AND per row).
            bool row_flat = false;
            uint32_t key = flat_key;
            if (flat_cnt > 0 || (i & 3) == 0) {
                if (CH == 3) {
                    key = read_first(d[0]) & 0xffffffu; // (b, g, r) of the strip's first pixel
                    const uint32_t b = key & 0xff, g = (key >> 8) & 0xff, r = key >> 16;
                    const uint32_t p0 = key | (b << 24), p1 = g | (r << 8) | (b << 16) | (g << 24), p2 = r | (b << 8) | (g << 16) | (r << 24);
                    row_flat = all(p_and(p_and(eq(d[0], splat(p0)), eq(d[1 % ND], splat(p1))), eq(d[2 % ND], splat(p2))));
                } else {
                    key = read_first(d[0]) & 0xffu;
                    row_flat = all(eq(d[0], splat(key * 0x01010101u)));
                }
            }
            flat_cnt = row_flat ? (flat_cnt > 0 && key == flat_key ? flat_cnt + 1 : 1) : 0;
            flat_key = key;
            const bool long_run = flat_cnt >= 11; // source rows y-10 .. y are one constant colour
            // ... or every row this work item has loaded so far is: nothing it will store can depend on anything else
            // (its first stored row, i = 10, sees exactly the rows 0 .. 10), so a work item that starts inside a constant
            // region skips the arithmetic of its warm-up rows too
            const bool fast = long_run || flat_cnt == i + 1;

            // ---- border replicate, de-interleave + widen into window slot k.  Skipped inside a long constant run: the
            //      slot holds row y-7, which is the same constant row (a run that began with the work item still has
            //      to fill the window) ----
            if (!long_run) {
                if (border_strip) {
                    if (CH == 3) {
                        const V n0 = perm(d[2 % ND], d[0], selA), n2 = perm(d[2 % ND], d[0], selC);
                        const V n1 = select(outside, perm(d[2 % ND], d[0], selB), d[1 % ND]);
                        d[0] = n0;
                        d[1 % ND] = n1;
                        d[2 % ND] = n2;
                    } else {
                        d[0] = perm(d[0], d[0], selA);
                    }
                }
                if (CH == 3) {
                    st.win[k][0][0] = perm(d[1 % ND], d[0], 0x0c060c00u);
                    st.win[k][1 % CH][0] = perm(d[1 % ND], d[0], 0x0c070c01u);
                    st.win[k][2 % CH][0] = perm(d[2 % ND], d[0], 0x0c040c02u);
                    st.win[k][0][1] = perm(d[2 % ND], d[0], 0x0c050c03u);
                    st.win[k][1 % CH][1] = perm(d[2 % ND], d[1 % ND], 0x0c060c00u);
                    st.win[k][2 % CH][1] = perm(d[2 % ND], d[1 % ND], 0x0c070c01u);
                } else {
                    st.win[k][0][0] = perm(d[0], d[0], 0x0c020c00u);
                    st.win[k][0][1] = perm(d[0], d[0], 0x0c030c01u);
                }
            }

            const int yg = y - 3, ys = y - 4, yv = y - 5; // rows of the Gaussian / Sobel+label / vote stages
            const uint32_t out_row = (uint32_t)(yv * cols); // byte offset of output row yv (used when yv >= 0)

            // ---- cv::pyrDown row oy (source rows 2oy-2 .. 2oy+2 = window rows y-4 .. y) ----
            if (pyr && ((y & 1) == 0) && y >= 2) {
                const int oy = (y - 2) >> 1;
                if (oy >= (R0 >> 1) && oy < (R1 >> 1) && oy < drows) {
                    const uint32_t prow = (uint32_t)(oy * dcols * CH);
                    V ob[CH]; // (out_a, out_b) per channel, bytes in the low half of each u16
                    if (fast) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) ob[c] = splat(((flat_key >> (8 * c)) & 0xffu) * 0x00010001u);
                    } else {
                        // REFLECT_101 at the top / bottom of the image folds taps onto rows inside the window
                        uint32_t wr[5] = {1, 4, 6, 4, 1};
                        if (oy == 0) {
                            wr[0] = 0, wr[1] = 0, wr[3] = 8, wr[4] = 2;
                            if (rows == 2) wr[2] = 8, wr[4] = 0; // rows -2 and 2 both fold onto row 0
                        } else if (2 * oy + 2 >= rows) {
                            wr[2] = 7, wr[4] = 0;
                        }
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            V ve = pk_mul(st.win[(k + 3) % 7][c][0], wr[0] * 0x00010001u);
                            V vo = pk_mul(st.win[(k + 3) % 7][c][1], wr[0] * 0x00010001u);
#pragma unroll
                            for (int t = 1; t < 5; ++t) {
                                ve = pk_mad(st.win[(k + 3 + t) % 7][c][0], wr[t] * 0x00010001u, ve);
                                vo = pk_mad(st.win[(k + 3 + t) % 7][c][1], wr[t] * 0x00010001u, vo);
                            }
                            V pve = from_left(ve), pvo = from_left(vo), nve = from_right(ve);
                            if (border_strip) { // columns -2, -1 fold onto 2, 1; column cols onto cols-2
                                pve = select(pyr_first, ve, pve);
                                pvo = select(pyr_first, vo << 16, pvo);
                                nve = select(pyr_last, ve >> 16, nve);
                            }
                            const V m2 = alignbit(ve, pve, 16), m1 = alignbit(vo, pvo, 16), p2 = alignbit(nve, ve, 16);
                            V s = m2 + p2; // halves stay below 2^16 (<= 65408 in total): plain 32-bit adds
                            s = pk_mad(m1 + vo, K4, s);
                            s = pk_mad(ve, QS_K(6, 6), s);
                            ob[c] = pk_lshr(s + QS_K(128, 128), 8);
                        }
                    }
                    if (CH == 3) {
                        // bytes (b0 g0) (r0 b1) (g1 r1): three aligned 16-bit stores per lane
                        buf_store_u16(pyr_buf, pyr_off, prow + 0, perm(ob[1 % CH], ob[0], 0x0c0c0400u));
                        buf_store_u16(pyr_buf, pyr_off, prow + 2, perm(ob[2 % CH], ob[0], 0x0c0c0204u));
                        buf_store_u16(pyr_buf, pyr_off, prow + 4, perm(ob[2 % CH], ob[1 % CH], 0x0c0c0602u));
                    } else {
                        buf_store_u16(pyr_buf, pyr_off, prow, perm(ob[0], ob[0], 0x0c0c0200u));
                    }
                }
            }

            if (__builtin_expect(fast, 0)) {
                // The output row is 0, and the state the next row needs is known in closed form: smoothed rows y-3, y-4
                // are the colour c (dx = 0, sx = 4c), the vote words of rows y-4, y-5 are bin 0 (3 per horizontal sum),
                // nothing is strong in row y-4.
                if (store_row) buf_store_u32(out_buf, out_off, out_row, splat(0u));
                if (!in_run) { // first row of the run
                in_run = true;
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t v4 = ((flat_key >> (8 * c)) & 0xffu) * 4u;
                        st.dx1[c][h] = st.dx2[c][h] = cold_splat(0u);
                        st.sx1[c][h] = st.sx2[c][h] = cold_splat(v4 * 0x00010001u);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    st.hs1[j] = st.hs2[j] = cold_splat(3u);
                    st.sm1[j] = cold_splat(0u);
                }
                } else {
                // Inside a run both rows of every two-row carry hold the same constants, so the row may hand them on
                // EXCHANGED -- which is what a textured row does with its names (x2 <- x1, x1 <- new): the register
                // allocator can then give "x2 after this row" the register of "x1 before it" on both paths, and the
                // textured path carries its state without a single copy (36 v_mov per row before).
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const V td = st.dx2[c][h], tsx = st.sx2[c][h];
                        st.dx2[c][h] = st.dx1[c][h];
                        st.dx1[c][h] = td;
                        st.sx2[c][h] = st.sx1[c][h];
                        st.sx1[c][h] = tsx;
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const V th = st.hs2[j];
                    st.hs2[j] = st.hs1[j];
                    st.hs1[j] = th;
                }
                }
            } else {
            in_run = false;
            // ---- Gaussian at row yg + Sobel horizontal parts (needed for smoothed rows R0-2 .. R1+1 inside the image) ----
            V dxc[CH][2], sxc[CH][2];
            const bool do_gauss = i >= 6 && yg >= 0 && yg <= rows - 1;
            if (do_gauss) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    V g[2]; // vertical pass: (g0, g2), (g1, g3), <= 65280
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        // sums of two bytes per 16-bit half: no carry between the halves, so the full-rate 32-bit add does
                        // (v_add_u32 issues every 2 cycles, the packed 16-bit forms every 4)
                        const V s06 = st.win[(k + 1) % 7][c][h] + st.win[k][c][h];
                        const V s15 = st.win[(k + 2) % 7][c][h] + st.win[(k + 6) % 7][c][h];
                        const V s24 = st.win[(k + 3) % 7][c][h] + st.win[(k + 5) % 7][c][h];
                        V acc = pk_mul(st.win[(k + 4) % 7][c][h], QS_K(72, 72));
                        acc = pk_mad(s24, QS_K(56, 56), acc);
                        acc = pk_mad(s15, QS_K(28, 28), acc);
                        g[h] = pk_mad(s06, K8, acc);
                    }
                    const V E = g[0], O = g[1];
                    const V pE = from_left(E), pO = from_left(O), nE = from_right(E), nO = from_right(O);
                    // horizontal pass, taps x-3 .. x+3, rounding constant in the first accumulate
                    V a0 = udot2(pO, QS_K(8, 56), 32768u);
                    a0 = udot2(O, QS_K(56, 8), a0);
                    a0 = udot2(pE, QS_K(0, 28), a0);
                    a0 = udot2(E, QS_K(72, 28), a0);
                    V a1 = udot2(pE, QS_K(0, 8), 32768u);
                    a1 = udot2(E, QS_K(56, 56), a1);
                    a1 = udot2(nE, QS_K(8, 0), a1);
                    a1 = udot2(pO, QS_K(0, 28), a1);
                    a1 = udot2(O, QS_K(72, 28), a1);
                    V a2 = udot2(pO, QS_K(0, 8), 32768u);
                    a2 = udot2(O, QS_K(56, 56), a2);
                    a2 = udot2(nO, QS_K(8, 0), a2);
                    a2 = udot2(E, QS_K(28, 72), a2);
                    a2 = udot2(nE, QS_K(28, 0), a2);
                    V a3 = udot2(E, QS_K(8, 56), 32768u);
                    a3 = udot2(nE, QS_K(56, 8), a3);
                    a3 = udot2(O, QS_K(28, 72), a3);
                    a3 = udot2(nO, QS_K(28, 0), a3);
                    // smoothed bytes are bits 16..23 of the sums: (s0, s2), (s1, s3) as u16 pairs
                    const V SE = perm(a2, a0, 0x0c060c02u), SO = perm(a3, a1, 0x0c060c02u);
                    const V pSO = from_left(SO), nSE = from_right(SE);
                    const V LE = alignbit(SO, pSO, 16); // (s-1, s1)
                    const V RO = alignbit(nSE, SE, 16); // (s2, s4)
                    dxc[c][0] = pk_sub(SO, LE);
                    sxc[c][0] = pk_mad(SE, K2, LE) + SO; // <= 1020 per half: plain 32-bit add
                    dxc[c][1] = pk_sub(RO, SE);
                    sxc[c][1] = pk_mad(SO, K2, SE) + RO;
                }
            } else { // never read by a row that is stored (see the stage gating above)
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        dxc[c][h] = dont_care();
                        sxc[c][h] = dont_care();
                    }
            }

            // ---- Sobel vertical combine, magnitude, channel select, vote words of row ys ----
            V hcur[4], smc[4];
            const bool real_row = i >= 8 && ys >= 1 && ys <= rows - 2;
            if (real_row) {
                V bv[4], bm[4];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    V gx[2], gy[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        gx[h] = pk_add(pk_mad(st.dx1[c][h], K2, st.dx2[c][h]), dxc[c][h]);
                        gy[h] = pk_sub(sxc[c][h], st.sx2[c][h]);
                    }
                    V v[4];
                    v[0] = perm(gy[0], gx[0], 0x05040100u);
                    v[2] = perm(gy[0], gx[0], 0x07060302u);
                    v[1] = perm(gy[1], gx[1], 0x05040100u);
                    v[3] = perm(gy[1], gx[1], 0x07060302u);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const V mg = sdot2_small(v[j], v[j]); // |gx|, |gy| <= 1020: no overflow
                        if (c == 0) {
                            bv[j] = v[j];
                            bm[j] = mg;
                        } else {
                            const P better = gt_i(mg, bm[j]); // ties keep the lower channel (:370-387)
                            bv[j] = select(better, v[j], bv[j]);
                            bm[j] = select(better, mg, bm[j]);
                        }
                    }
                }
                V w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w[j] = qs_vote_word(bv[j], p_and(real_col[j], ne(bv[j], splat(0u))));
                    smc[j] = select(gt_i(bm[j], splat((uint32_t)a.thr_i)), cm[j], 0u);
                }
                if (a.mask) { // quantize(): angle.copyTo(dst, mask)
                    const V mw = load_u32(a.mask + (int64_t)ys * cols, lcol);
#pragma unroll
                    for (int j = 0; j < 4; ++j) smc[j] = select(ne(mw & (0xffu << (8 * j)), splat(0u)), smc[j], 0u);
                }
                const V wl = from_left(w[3]), wr_ = from_right(w[0]);
                hcur[0] = add3(wl, w[0], w[1]);
                hcur[1] = add3(w[0], w[1], w[2]);
                hcur[2] = add3(w[1], w[2], w[3]);
                hcur[3] = add3(w[2], w[3], wr_);
            } else {
                // ring rows (0, rows-1) and rows outside the image vote for bin 0 and are never strong
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hcur[j] = cold_splat(3u);
                    smc[j] = cold_splat(0u);
                }
            }

            // ---- 3x3 majority vote of row yv ----
            if (store_row) {
                V packed = splat(0u);
                if (yv >= 1 && yv <= rows - 2) {
                    V val[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const V tot = add3(st.hs2[j], st.hs1[j], hcur[j]);
                        const V hit = (tot + 0x33333333u) & st.sm1[j]; // a counter >= 5 (at most one), strong centre
                        val[j] = 1u << ((ffbl(hit) >> 2) & 31u);       // no hit: bit 31, dropped below
                    }
                    packed = perm(val[1], val[0], 0x0c0c0400u) | perm(val[3], val[2], 0x04000c0cu);
                }
                buf_store_u32(out_buf, out_off, out_row, packed);
            }

            // ---- carry ----
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    st.dx2[c][h] = st.dx1[c][h];
                    st.dx1[c][h] = dxc[c][h];
                    st.sx2[c][h] = st.sx1[c][h];
                    st.sx1[c][h] = sxc[c][h];
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st.hs2[j] = st.hs1[j];
                st.hs1[j] = hcur[j];
                st.sm1[j] = smc[j];
            }
            } // !fast
        }
    }
}

// Work items of a launch: (strip, row block, frame) for the strips that get a wave per frame, then (row block, frame
// group) for a packed last strip.
__host__ __device__ inline int quantize_stream_items(const QSArgs& a)
{
    const int n_full = a.pack_lanes ? a.n_strips - 1 : a.n_strips;
    return n_full * a.n_rblocks * a.frames + (a.pack_lanes ? a.n_rblocks * a.pack_groups : 0);
}

template <int CH>
__device__ __forceinline__ void quantize_stream_item(const QSArgs& a, int item)
{
    // Order: row block slowest, then frame, then strip.  All work items are resident at once (one round) and the
    // dispatcher deals workgroups to the 256 CUs round-robin, so a CU's three workgroups are 256 and 512 workgroups
    // apart: with the frame slowest those were the SAME region of three different frames -- on a frame that is part
    // constant canvas a CU got three textured or three constant workgroups (16 BASELINE frames: 77 us, the time of a
    // fully textured batch); with the row block slowest they are a third of the image apart.
    const int n_full = a.pack_lanes ? a.n_strips - 1 : a.n_strips; // strips that get a wave per frame
    const int per_rb = n_full * a.frames, n_plain = per_rb * a.n_rblocks;
    if (item < n_plain) {
        const int rb = item / per_rb, r = item - rb * per_rb;
        quantize_stream_wave<CH>(a, r % n_full, rb, r / n_full, 0);
    } else {
        const int j = item - n_plain;
        if (!a.pack_lanes || j >= a.n_rblocks * a.pack_groups) return;
        const int rb = j / a.pack_groups, grp = j - rb * a.pack_groups, per = 64 / a.pack_lanes, f0 = grp * per;
        const int nseg = a.frames - f0 < per ? a.frames - f0 : per;
        quantize_stream_wave<CH>(a, a.n_strips - 1, rb, f0, nseg);
    }
}

// segment lanes of a packed last strip for this geometry, 0 = the last strip gets a wave per frame like the others
__host__ __device__ inline int quantize_stream_pack_lanes(int rows, int cols, int ch, int frames)
{
    const int n_strips = (cols + QS_USEFUL - 1) / QS_USEFUL;
    const int lanes = (cols - QS_USEFUL * (n_strips - 1)) / QS_LANE_PX + 2 * QS_HALO_LANES;
    const int per = 64 / lanes;
    // per-lane offsets are 32-bit: the frames of a group must lie within 2 GiB of the first
    if (frames < 2 || per < 2 || (long long)per * rows * cols * ch >= 0x7ff00000ll) return 0;
    return lanes;
}

#ifndef SBM_WAVE_EMU
// grid = ceil(work items / 4), block = 256 = four independent waves
template <int CH>
__global__ __launch_bounds__(256) void k_quantize_stream(const QSArgs a)
{
    // the wave index is the same in all 64 lanes: tell the compiler, so that everything derived from the work item
    // (row counters, stage gating, row base addresses) lives in SGPRs and branches are scalar
    const int item = (int)blockIdx.x * 4 + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    quantize_stream_item<CH>(a, item);
}
#endif

} // namespace sbm
