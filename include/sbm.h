/*
 * sbm.h — C ABI of libsbm_hip.so, the MI355X (gfx950) LINE-2D matching engine.
 *
 * This is the drop-in boundary for the match() hot path of
 * ddcr/shape_based_matching.  The reference has no FFI layer: its boundary is
 * the C++ class line2Dup::Detector (line2Dup.h:257-333).  include/line2Dup.h in
 * this repository re-declares that class and implements it on top of the
 * functions below, so a reference caller (test.cpp, test_jabil.cpp) re-links
 * against this library unchanged; other hosts bind the C ABI directly
 * (INTEGRATION.md shows both).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative sbm_status, with a message available from
 * sbm_last_error() (thread-local).  One context drives one GPU; a context is
 * not thread-safe, different contexts may be used concurrently.  "host"
 * pointers are ordinary process memory, "device" pointers are HBM addresses on
 * the context's GPU (e.g. torch tensors' data_ptr()), `stream` is a
 * hipStream_t passed as void* (NULL = the context's own stream).
 *
 * Each entry point cites the reference function (file:line) it replaces.
 */
#ifndef SBM_H
#define SBM_H

#include <stdint.h>
#include "sbm_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SBM_ABI_VERSION 1

typedef enum sbm_status {
    SBM_OK = 0,
    SBM_ERR_INVALID = -1,   /* bad argument (CV_Assert / CV_Error in the reference) */
    SBM_ERR_HIP = -2,       /* HIP runtime failure or no usable GPU */
    SBM_ERR_CAPACITY = -3,  /* candidate / match buffer too small */
    SBM_ERR_STATE = -4      /* call sequence error (no templates, no pyramid, ...) */
} sbm_status;

typedef struct sbm_ctx sbm_ctx;

/* Detector constructor arguments that matter to match()
 * (line2Dup.cpp:1056-1076: pyramid_levels, T_at_level, ColorGradient::weak_threshold). */
typedef struct sbm_config {
    int32_t n_levels;
    int32_t T[SBM_MAX_LEVELS];
    float weak_threshold;
    int32_t device_id;
    int64_t max_candidates; /* capacity of the coarse-candidate and match lists; 0 = 1<<20 */
} sbm_config;

const char* sbm_last_error(void);
int sbm_abi_version(void);

int sbm_create(const sbm_config* cfg, sbm_ctx** out);
void sbm_destroy(sbm_ctx* ctx);

/* ---- templates ----------------------------------------------------------
 * Replaces the in-memory TemplatesMap the reference walks in matchClass
 * (line2Dup.h:319-321, line2Dup.cpp:1160-1172).  levels is [n_templates][n_levels]
 * (level 0 first), features is the flat array the levels index into;
 * class_idx / template_id label the emitted matches (either may be NULL:
 * class 0 / id = position).  Templates with >= 8192 features at any level are
 * rejected with SBM_ERR_INVALID (CV_Error, line2Dup.cpp:1195, :1260). */
int sbm_upload_templates(sbm_ctx* ctx, int32_t n_templates, const sbm_template_level* levels,
                         const sbm_feature* features, int64_t n_features,
                         const int32_t* class_idx, const int32_t* template_id);

/* Restrict matching to templates whose class_idx is listed (Detector::match's
 * class_ids argument, line2Dup.cpp:1124-1140).  n == 0 selects every class.  A
 * second form selects an explicit template index range [first, first+count):
 * the template shard of one GPU (reference analogue: the OpenMP loop bounds,
 * line2Dup.cpp:1169-1170). */
int sbm_select_classes(sbm_ctx* ctx, const int32_t* class_idx, int32_t n);
int sbm_select_range(sbm_ctx* ctx, int32_t first, int32_t count);

/* An explicit list of template indices (upload order) as the active set: what sbm_select_classes / sbm_select_range
 * build internally; used to shard a class selection over several GPUs. */
int sbm_select_templates(sbm_ctx* ctx, const int32_t* template_idx, int32_t n);
/* Contiguous, work-balanced shards of a template list for a rows x cols frame: the list is template_idx[0..n) (NULL:
 * every uploaded template in upload order); shard s is list[first[s] .. first[s] + count[s]).  Work = byte-adds of the
 * coarse pass (in-bounds coarsest-level features x template_positions, line2Dup.cpp:818-837). */
int sbm_partition_templates(sbm_ctx* ctx, int32_t rows, int32_t cols, const int32_t* template_idx, int32_t n,
                            int32_t n_shards, int32_t* first, int32_t* count);

/* ---- whole hot path -----------------------------------------------------
 * Detector::match (line2Dup.cpp:1078-1150) without the final std::sort /
 * std::unique: emits the pre-dedup multiset of matches in unspecified order;
 * sbm_canonicalize() applies the epilogue.  img is 8-bit, 1 or 3 interleaved
 * channels (BGR order as cv::imread gives), row stride in bytes; mask (rows x
 * cols, non-zero = keep) may be NULL.  rows/cols must satisfy the reference's
 * preconditions at every level (rows_l % T_l == 0, cols_l % T_l == 0,
 * (rows_l * cols_l) % 16 == 0; line2Dup.cpp:639, :751-752) else SBM_ERR_INVALID. */
int sbm_match(sbm_ctx* ctx, const uint8_t* img_host, int32_t rows, int32_t cols, int32_t stride,
              int32_t channels, const uint8_t* mask_host, float threshold, sbm_match_rec* out_host,
              int64_t cap, int64_t* n_out);

/* Single-process multi-GPU form of sbm_match: one context per GPU (all holding the same templates, each with its own
 * selection -- sbm_partition_templates + sbm_select_range / sbm_select_templates), one host thread per context, the
 * frame uploaded to every GPU, the per-GPU lists concatenated on the host: the reference's OpenMP team over
 * templates with its concatenating reduction (line2Dup.cpp:1166-1170).  One process per GPU + RCCL is the other
 * multi-GPU form (sbm_comm_* below).  Several contexts on ONE GPU are allowed (tests). */
int sbm_match_sharded(sbm_ctx* const* ctxs, int32_t n_ctx, const uint8_t* img_host, int32_t rows, int32_t cols,
                      int32_t stride, int32_t channels, const uint8_t* mask_host, float threshold,
                      sbm_match_rec* out_host, int64_t cap, int64_t* n_out);

/* A batch of frames from HOST memory, pipelined (streaming API with asynchronous upload overlap; the
 * reference calls Detector::match once per frame, line2Dup.cpp:1078): the frames travel over PCIe in sub-batches of
 * `sub_batch` frames (0 = 8) on a copy stream into one of two device buffers while the kernels of the previous
 * sub-batch run; every kernel is launched once per sub-batch (as in sbm_match_batch_device); the lists arrive in a
 * pinned host block written by the last kernel.  frames[f] points to frame f (rows x stride bytes).  Results: the
 * records of frame f at out + f * cap, {n_matches, overflow} at counts + 2 * f.  _begin enqueues everything and
 * returns (uploads from memory that is not pinned are staged by the runtime and may block meanwhile); _end waits and
 * copies the lists out; one batch in flight per context.  The frames and the mask must stay mapped and unchanged
 * from _begin until _end has returned (pinned memory is read by the copy engine during that time).  SBM_ERR_CAPACITY if a frame has more than cap matches (its
 * first cap records are still returned). */
int sbm_match_batch_host_begin(sbm_ctx* ctx, const uint8_t* const* frames, int32_t n_frames, int32_t rows, int32_t cols,
                               int32_t stride, int32_t channels, const uint8_t* mask_host, float threshold, int64_t cap,
                               int32_t sub_batch);
int sbm_match_batch_host_end(sbm_ctx* ctx, sbm_match_rec* out_host, int32_t* counts);
int sbm_match_batch_host(sbm_ctx* ctx, const uint8_t* const* frames, int32_t n_frames, int32_t rows, int32_t cols,
                         int32_t stride, int32_t channels, const uint8_t* mask_host, float threshold,
                         sbm_match_rec* out_host, int64_t cap, int32_t* counts, int32_t sub_batch);

/* Optional: pin a caller-owned host buffer (hipHostRegister) so that sbm_match / sbm_build_pyramid upload frames that
 * lie inside it with one asynchronous DMA instead of the runtime's staged pageable copy (a camera loop that re-uses
 * its frame buffer: 115 instead of ~160 us per 1024 x 1024 BGR match).  Explicit by design -- the library never pins
 * memory it was merely handed: the caller must keep the range mapped until sbm_unpin_host_buffer (or sbm_destroy)
 * and must not free / re-map it while pinned.  Buffers the caller pinned itself (hipHostMalloc, hipHostRegister)
 * need neither call.  sbm_unpin_host_buffer waits for the context's stream first. */
int sbm_pin_host_buffer(sbm_ctx* ctx, const void* host_ptr, int64_t bytes);
int sbm_unpin_host_buffer(sbm_ctx* ctx, const void* host_ptr);

/* Same path with the frame already resident in HBM.  Asynchronous: enqueues
 * every kernel on `stream` and returns.  The FIRST call after a change of state (templates, template selection,
 * threshold, frame geometry, batch size) rebuilds small device tables on the context's own stream and synchronises
 * the device before it enqueues anything: warm the context up with one call before capturing `stream` into a
 * hipGraph or relying on asynchrony; steady-state calls never wait on the host and never touch another stream.  Results go to caller-provided device
 * buffers (d_out: cap records, d_count: one int32 — the number of matches,
 * which may exceed cap, in which case only cap records were stored) so the host
 * side can all-gather them over RCCL without another copy. */
int sbm_match_device(sbm_ctx* ctx, const void* d_img, int32_t rows, int32_t cols, int32_t stride,
                     int32_t channels, const void* d_mask, float threshold, void* d_out,
                     int64_t cap, void* d_count, void* stream);

/* A batch of frames of one geometry in one call (the streaming row; the reference
 * calls Detector::match once per frame, line2Dup.cpp:1078).  Frame f starts at d_imgs + f * frame_stride
 * bytes; every kernel of the path is launched once for the whole batch (the frame is a grid dimension), so
 * the per-launch cost is shared by n_frames frames.  Results of frame f: records at d_out + f * cap,
 * {n_matches, overflow} at d_counts + 2 * f (int32); a result mirror (sbm_set_result_mirror) must hold
 * n_frames * cap records and n_frames * 2 int32, laid out the same way.  The mask, if any, is shared by
 * the frames.  Needs the register-only linear-memory kernel: T in {4, 8}, level widths multiples of 16.
 * Each frame's list is what sbm_match_device returns for that frame alone. */
int sbm_match_batch_device(sbm_ctx* ctx, const void* d_imgs, int64_t frame_stride, int32_t n_frames, int32_t rows,
                           int32_t cols, int32_t stride, int32_t channels, const void* d_mask, float threshold,
                           void* d_out, int64_t cap, void* d_counts, void* stream);

/* hipGraph replay of an entry point's launches: the kernel sequence of a call is recorded once per distinct argument
 * tuple (stream capture) and replayed with one hipGraphLaunch.
 *   enabled = -1 (default, "auto"): sbm_match_batch_device and sbm_match_templates_device replay once the caller keeps
 *       several calls in flight (sbm_set_pipeline_depth >= 2) and an argument tuple comes back (a tuple is captured at
 *       its second sighting; up to 16 captures are kept; a caller whose tuples never repeat stays on stream launches).
 *       Why: with several streams in flight about one process in eight on the MI355X pool runs its stream launches in a
 *       slow mode (0.5 - 1.8 ms per 16-frame step instead of 0.11, kernel durations unchanged) that replay -- one host
 *       call per step -- does not have; otherwise the two measure the same.  sbm_match_device (one frame at a time)
 *       stays on stream launches: there the replayed graph is slower (74 - 85 us against 70 us on ROCm 7.2).
 *   enabled = 0: never (the opt-out; also what profiling uses).
 *   enabled = 1: always, sbm_match_device included (two branches: the fine levels' linear memories are built while the
 *       coarse-level chain runs).
 * Results are identical either way. */
int sbm_set_graph_mode(sbm_ctx* ctx, int32_t enabled);
/* How many captured graphs the context holds right now (a test / diagnostics accessor for the rule above). */
int sbm_graph_count(sbm_ctx* ctx, int32_t* n_graphs);

/* Which gradient kernel (quantizedOrientations + hysteresisGradient, line2Dup.cpp:313-404, :218-311) the match
 * entry points launch.  mode 0 (default): by launch size — the row-streaming kernel (one wave per 256-column
 * strip, registers + DPP only, no LDS / barriers) when a launch has thousands of waves of work (a batch of
 * frames), else the 16 x 64 tile kernel, which also serves the float outputs and widths that are not a multiple
 * of 4.  mode 1: always the tile kernel.  mode 2: the streaming kernel whenever the geometry allows it.
 * rows_per_wave > 0 overrides the streaming kernel's rows per work item (rounded up to even).  Results are
 * bit-identical either way; this is a tuning / test knob. */
int sbm_set_quantize_mode(sbm_ctx* ctx, int32_t mode, int32_t rows_per_wave);

/* Which coarse-pass kernel (similarity / similarity_64 + candidate scan, line2Dup.cpp:807-858, :924-984, :1199-1216) the
 * match entry points launch.  0 (default): by launch size -- one wave per (1024 positions, template, frame) for batches
 * and large template sets, four waves sharing an item's features for a single frame with a few hundred templates;
 * 1: always the four-wave kernel; 2: always the one-wave kernel.  Identical candidates either way; a test / tuning knob. */
int sbm_set_coarse_mode(sbm_ctx* ctx, int32_t mode);

/* Which form of a refinement level with T = 4 the match entry points build (and the template loop builds, once per pyramid,
 * beside response planes that the stage entry points made) and the refinement pass (similarityLocal(_64),
 * line2Dup.cpp:860-922, :986-1048) reads.  1: BIT STRIPS -- per (sub-plane, orientation, strip of 16 columns, grid row) one
 * dword, "response > 0" bits of the 16 cells | "response == 4" bits << 16; the reference's sum of response bytes {0, 3, 4}
 * is 3 #any + #exact, counted with bit-sliced carry-save counters, one wave per candidate.  0: one plane of spread bytes,
 * the response looked up per byte (rounds 2-3).  -1 (default): the process default (bit strips unless SBM_LOCAL_BITS=0).
 * Levels with another T, or whose grid width is not a multiple of 16 cells, use the byte form either way.  Identical
 * matches; a test / tuning knob. */
int sbm_set_refine_bits(sbm_ctx* ctx, int32_t mode);

/* Which candidate a workgroup of the refinement pass (similarityLocal(_64) + the loop at line2Dup.cpp:1221-1293) takes.
 * 0: a grid of (frames x slots), every frame's candidates walked by that frame's slots; 2: the candidates of up to 64
 * frames as ONE frame-major list walked by the whole grid (the workgroups running together stay on one or two frames,
 * whose linear memories then stay in the L2s); -1 (default): 2 once the batch's finest-level planes exceed the L2s
 * together (32 MiB), else 0.  Identical matches either way; a test / tuning knob. */
int sbm_set_refine_order(sbm_ctx* ctx, int32_t order);

/* Hint: how many batches the caller keeps in flight on this GPU at the same time (through other contexts and
 * streams; bench.py runs three).  1 (default): every launch is sized for its own latency -- all its work items resident
 * at once.  >= 2: launches are sized for throughput -- the row-streaming gradient kernel takes fewer, longer work items
 * (less warm-up work in total; the SIMDs such a launch leaves idle are used by the other batches' kernels): 7.6 % more
 * frames per second on 16 x 1024 x 1024 x 3 batches with three in flight, at 1.3x the latency of a batch alone.
 * Results are identical either way. */
int sbm_set_pipeline_depth(sbm_ctx* ctx, int32_t batches_in_flight);

/* Optional second destination for the results of sbm_match_device /
 * sbm_match_templates: every match record (up to the call's cap) and the final
 * {n_matches, overflow} pair are ALSO stored, with plain stores from the last
 * kernel, at these device-visible addresses — typically pinned host memory
 * (hipHostMalloc / torch pin_memory), so the match list reaches the host with
 * no copy-engine operation after the kernels.  Pass NULL, NULL to disable.  The
 * mirror must hold cap records / two int32 and outlive the calls using it. */
int sbm_set_result_mirror(sbm_ctx* ctx, void* mirror_out, void* mirror_count);

/* ---- multi-GPU: template shards + one RCCL exchange step -------------------
 * The reference's only parallelism is the OpenMP loop over templates whose
 * per-thread match vectors are concatenated by a reduction
 * (line2Dup.cpp:1166-1170).  Analogue: one context (process) per GPU, each with
 * a template range (sbm_select_range), and ONE collective per frame: an
 * ncclAllGather over xGMI of the per-rank lists, issued by the library on the
 * same stream as the kernels.  librccl is resolved with dlopen on first use.
 *
 *   sbm_comm_unique_id   rank 0: 128-byte id to hand to every rank (any transport)
 *   sbm_comm_init        every rank: join the communicator on the context's GPU
 *   sbm_match_device_sharded
 *        d_local    : SBM_SHARD_HEADER_BYTES + cap * 24 bytes, this rank's list:
 *                     int32 {n_matches, overflow, 0, 0} then the records
 *        d_gathered : world * (that size), all ranks' lists in rank order
 *        gathered_mirror : optional device-visible copy of d_gathered (pinned host)
 */
#define SBM_COMM_ID_BYTES 128
#define SBM_SHARD_HEADER_BYTES 16
int sbm_comm_unique_id(void* id_out);
int sbm_comm_init(sbm_ctx* ctx, int32_t world, int32_t rank, const void* id);
int sbm_comm_destroy(sbm_ctx* ctx);
/* Size of the context's communicator as RCCL reports it (ncclCommCount): what a launcher records as "ranks seen". */
int sbm_comm_count(sbm_ctx* ctx, int32_t* n_ranks);
/* The template loop alone (sbm_match_templates_device: matchClass over the selected templates on the resident pyramid,
 * line2Dup.cpp:1160-1297) followed by the same exchange step as sbm_match_device_sharded: d_local = 16-byte header
 * {n_matches, overflow, 0, 0} + cap records, d_gathered = world such shards in rank order.  The sharded form of
 * BASELINE configs 3 and 4 (template ranges over the ranks, pyramid resident on each). */
int sbm_match_templates_device_sharded(sbm_ctx* ctx, float threshold, void* d_local, int64_t cap, void* d_gathered,
                                       void* gathered_mirror, void* stream);
int sbm_match_device_sharded(sbm_ctx* ctx, const void* d_img, int32_t rows, int32_t cols, int32_t stride,
                             int32_t channels, const void* d_mask, float threshold, void* d_local,
                             int64_t cap, void* d_gathered, void* gathered_mirror, void* stream);
/* The same exchange for a batch of frames (sbm_match_batch_device + one ncclAllGather of the whole shard):
 *   d_local : header = n_frames * 8 bytes rounded up to 16 ({n_matches, overflow} int32 pairs), then
 *             n_frames blocks of cap records; d_gathered: world such shards in rank order. */
int sbm_match_batch_device_sharded(sbm_ctx* ctx, const void* d_imgs, int64_t frame_stride, int32_t n_frames,
                                   int32_t rows, int32_t cols, int32_t stride, int32_t channels,
                                   const void* d_mask, float threshold, void* d_local, int64_t cap,
                                   void* d_gathered, void* gathered_mirror, void* stream);

/* Build-sharded step (round 3): the gradient stage is sharded too.  The reference builds the pyramid serially
 * (line2Dup.cpp:1084-1120) and parallelises only the template loop (:1166-1170); with the pyramid replicated on every
 * rank a step cannot scale past (build + loop) / build.  Here rank r of a world of n computes ROW BAND r of every
 * level's orientation map (quantizedOrientations + hysteresisGradient of rows [r, r+1) * rows_l / n, plus the halo
 * rows the next level's band needs through the fused cv::pyrDown), one grouped in-place ncclAllGather over xGMI
 * assembles the maps on every rank (1 byte per pixel and level: 1.25 MiB per 1024 x 1024 frame), every rank builds
 * the linear memories from the assembled maps and matches its template range (sbm_select_range), and the per-rank
 * match lists are gathered as in sbm_match_batch_device_sharded (same d_local / d_gathered layout).
 * Needs rows_l % n == 0 with an even quotient at every level, cols % 4 == 0.  The assembled maps are bit for bit the
 * maps of sbm_match_batch_device (a band's rows depend on rows outside it exactly as in a whole-level launch).
 * n_bands: 0 or the communicator size with a multi-rank communicator.  On one GPU (no communicator, or a one-rank
 * one) the call computes all n_bands bands itself, one launch per band and level -- the rehearsal form used by the
 * tests and by bench.py --banded; d_gathered may then be NULL. */
int sbm_match_batch_device_banded(sbm_ctx* ctx, const void* d_imgs, int64_t frame_stride, int32_t n_frames,
                                  int32_t rows, int32_t cols, int32_t stride, int32_t channels, const void* d_mask,
                                  float threshold, void* d_local, int64_t cap, void* d_gathered,
                                  void* gathered_mirror, int32_t n_bands, void* stream);

/* Detector::match epilogue (line2Dup.cpp:1142-1145) in canonical form: sort by
 * (similarity desc, template_id asc, class_idx asc, y asc, x asc), drop exact
 * duplicates.  Host-side, in place; returns the new count. */
int64_t sbm_canonicalize(sbm_match_rec* recs, int64_t n);

/* ---- pyramid state ------------------------------------------------------
 * sbm_build_pyramid: the first half of match() (line2Dup.cpp:1084-1120):
 * quantizedOrientations -> [pyrDown -> quantizedOrientations]* -> per level
 * quantize(mask) -> spread -> computeResponseMaps -> linearize, leaving the
 * flat linear memories of every level resident in HBM. */
int sbm_build_pyramid(sbm_ctx* ctx, const uint8_t* img_host, int32_t rows, int32_t cols,
                      int32_t stride, int32_t channels, const uint8_t* mask_host);
/* Install a caller-made one-hot orientation map as level `level` of the
 * pyramid and build its linear memories (spread -> computeResponseMaps ->
 * linearize; line2Dup.cpp:1110-1116).  Levels must be set from 0 upwards. */
int sbm_set_quantized(sbm_ctx* ctx, int32_t level, const uint8_t* quantized_host, int32_t rows,
                      int32_t cols);
/* Read back what a level holds: its one-hot map (rows*cols) and its flat
 * linear memories ([8][lm_stride] bytes; lm_stride >= T*T*W*H, zero tail). */
int sbm_get_quantized(sbm_ctx* ctx, int32_t level, uint8_t* out_host);
int sbm_get_linear_memories(sbm_ctx* ctx, int32_t level, uint8_t* out_host, int64_t cap_bytes,
                            int64_t* lm_stride);
/* The coarsest level's linear memories as BIT planes (round 4): what the coarse pass reads instead of the reference's
 * one byte per (position, orientation) (similarity, line2Dup.cpp:843-856; values {0, 3, 4} of SIMILARITY_LUT :632-635).
 * Per frame 16 planes of lm_stride BITS each, little-endian dwords, in the flat order of the byte linear memories:
 * plane o (0..7): bit j = (LM[o][j] > 0); plane 8 + o: bit j = (LM[o][j] == 4).  frame: which frame of the last batch.
 * out_host: 16 * lm_stride / 8 bytes.  Fails with SBM_ERR_STATE when the last call did not produce them (a threshold < 0,
 * or sbm_set_coarse_mode picked a byte kernel).  A parity-test accessor. */
int sbm_get_coarse_bitplanes(sbm_ctx* ctx, int32_t frame, uint8_t* out_host, int64_t cap_bytes);
int sbm_level_dims(sbm_ctx* ctx, int32_t level, int32_t* rows, int32_t* cols);

/* Second half of match(): Detector::matchClass over the selected templates
 * (line2Dup.cpp:1160-1297) against the resident pyramid. */
int sbm_match_templates(sbm_ctx* ctx, float threshold, sbm_match_rec* out_host, int64_t cap,
                        int64_t* n_out);

/* Asynchronous form: the template loop against the resident pyramid on `stream`, results into
 * caller-provided device buffers (as sbm_match_device). */
int sbm_match_templates_device(sbm_ctx* ctx, float threshold, void* d_out, int64_t cap, void* d_count,
                               void* stream);

/* ---- single reference functions (stage entry points; host arrays) --------
 * Each runs the HIP kernel that replaces one reference function and copies the
 * result back, so it can be tested (and adopted) on its own. */

/* quantizedOrientations + hysteresisGradient (line2Dup.cpp:313-404, 218-311).
 * magnitude / angle_ori (float, rows*cols) may be NULL; angle is the one-hot map. */
int sbm_quantized_orientations(sbm_ctx* ctx, const uint8_t* img_host, int32_t rows, int32_t cols,
                               int32_t stride, int32_t channels, float weak_threshold,
                               float* magnitude, uint8_t* angle, float* angle_ori);
/* The per-pixel scan of ColorGradientPyramid::extractTemplate (line2Dup.cpp:452-539; training side): the pixels of
 * [2, rows-2) x [2, cols-2) that pass the 3x3-eroded mask (NULL: all), whose squared magnitude exceeds
 * strong_threshold^2 and that the reference's row-major `magnitude_valid` scan accepts as 5x5 local maxima -- found
 * by a data-parallel kernel (no strictly larger neighbour), ties among equal neighbours resolved in row-major order on
 * the host exactly as the sequential scan does.  xy[i] = x | y << 16, in row-major order; *n_out = their number
 * (SBM_ERR_CAPACITY if > cap).  The orientation test (:504) and the feature selection stay with the caller. */
int sbm_extract_local_maxima(sbm_ctx* ctx, const float* magnitude_host, const uint8_t* mask_host, int32_t rows, int32_t cols,
                             float strong_threshold, int32_t* xy, int64_t cap, int64_t* n_out);
/* The 16-bin quantisation of hysteresisGradient (line2Dup.cpp:225: convertTo(CV_8U, 16/360) of the
 * phase image) as the match path computes it: an integer rule on the Sobel gradient (gx, gy), exact for
 * |gx|, |gy| <= 1020.  q16[i] in 0..16. */
int sbm_orientation_bins(sbm_ctx* ctx, const int16_t* gx, const int16_t* gy, int64_t n, uint8_t* q16);
/* cv::resize(src, dst, cv::Size(), fx, fy) (INTER_LINEAR, 8-bit) as shapeInfo_producer::transform calls it to make
 * the scaled training images (line2Dup.h:379-405).  Output size = (cvRound(rows*fy), cvRound(cols*fx)), written to
 * out_rows / out_cols; out == NULL only queries the size. */
int sbm_resize_linear(sbm_ctx* ctx, const uint8_t* img_host, int32_t rows, int32_t cols, int32_t stride, int32_t channels,
                      double fx, double fy, uint8_t* out, int64_t cap_bytes, int32_t* out_rows, int32_t* out_cols);
/* cv::pyrDown as called by ColorGradientPyramid::pyrDown (line2Dup.cpp:431-433). */
int sbm_pyrdown(sbm_ctx* ctx, const uint8_t* img_host, int32_t rows, int32_t cols, int32_t stride,
                int32_t channels, uint8_t* out_host);
/* spread (line2Dup.cpp:616-630). */
int sbm_spread(sbm_ctx* ctx, const uint8_t* src_host, int32_t rows, int32_t cols, int32_t T,
               uint8_t* dst_host);
/* computeResponseMaps (line2Dup.cpp:637-747): maps_host is [8][rows*cols]. */
int sbm_compute_response_maps(sbm_ctx* ctx, const uint8_t* spread_host, int32_t rows, int32_t cols,
                              uint8_t* maps_host);
/* linearize (line2Dup.cpp:749-777): lm_host is [T*T][(rows/T)*(cols/T)]. */
int sbm_linearize(sbm_ctx* ctx, const uint8_t* map_host, int32_t rows, int32_t cols, int32_t T,
                  uint8_t* lm_host);
/* similarity / similarity_64 (line2Dup.cpp:807-858, 924-984) of uploaded
 * template `template_index` at the coarsest level of the resident pyramid:
 * dst_host is the H x W uint16 score map. */
int sbm_similarity(sbm_ctx* ctx, int32_t template_index, uint16_t* dst_host);
/* similarityLocal / similarityLocal_64 (line2Dup.cpp:860-922, 986-1048) at
 * pyramid level `level` around centre (cx, cy): dst_host is 16 x 16 uint16. */
int sbm_similarity_local(sbm_ctx* ctx, int32_t level, int32_t template_index, int32_t cx,
                         int32_t cy, uint16_t* dst_host);

/* ---- measurement ---------------------------------------------------------
 * Per-kernel HIP-event timings of the last sbm_match/sbm_build_pyramid/
 * sbm_match_templates call when profiling was enabled (adds synchronisation;
 * never enabled in the throughput path). names/ms arrays hold up to cap entries.
 * enabled = 2: the timings of successive asynchronous calls accumulate (no per-call reset) until
 * sbm_get_timings has returned all of them -- the way to time kernels while several streams are in flight. */
int sbm_set_profiling(sbm_ctx* ctx, int32_t enabled);
int sbm_get_timings(sbm_ctx* ctx, const char** names, float* ms, int32_t cap, int32_t* n);
/* Algorithmic bytes of the coarse pass for the selected templates on the
 * resident pyramid: sum over in-bounds coarsest-level features of
 * max(template_positions, 0)  (SURVEY.md 8d). */
int sbm_coarse_bytes(sbm_ctx* ctx, int64_t* bytes);

/* Counters of the last template-matching call: coarse candidates found
 * (line2Dup.cpp:1208-1214) and the bytes the refinement passes read for frame 0,
 * sum over refined candidates of nf_level * 256 on response / spread bytes (the
 * reference's figure) and nf_level * 128 on bit strips
 * (sbm_set_refine_bits); accumulated only while profiling is enabled, to keep the
 * atomic out of the throughput path.  Synchronises. */
int sbm_get_stats(sbm_ctx* ctx, int64_t* n_candidates, int64_t* refine_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SBM_H */
