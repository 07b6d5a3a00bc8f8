// sbm_capi.hip — host side of libsbm_hip.so: the C ABI of include/sbm.h on top
// of the kernels in sbm_kernels.h.  gfx950 only; there is no CPU fallback: every
// entry point fails with SBM_ERR_HIP when no GPU is usable.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sbm.h"
#include "sbm_kernels.h"
#include "sbm_quantize_stream.h"
#include "sbm_train_kernels.h"

using namespace sbm;

namespace {

thread_local std::string g_err;
void (*g_rccl_destroy_hook)(sbm_ctx*) = nullptr; // set once librccl is loaded

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(SBM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes, bool zero = false)
    {
        if (bytes <= cap && p) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes ? bytes : 16;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(SBM_ERR_HIP, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        cap = want;
        if (zero) {
            e = hipMemset(p, 0, want);
            if (e != hipSuccess) return fail(SBM_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
        }
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

struct Timing {
    const char* name;
    hipEvent_t a, b;
    bool shared_a = false; // a is the previous entry's b
};

int64_t lm_stride_for(int rows, int cols, int T)
{
    int64_t W = cols / T, H = rows / T;
    int64_t s = (int64_t)T * T * W * H + W * H + 16 * W + 80;
    return (s + 63) / 64 * 64;
}

} // namespace

static int coarse_mode_env()
{
    const char* env = getenv("SBM_COARSE");
    return env && !strcmp(env, "block") ? 1 : (env && !strcmp(env, "wave") ? 2 : 0);
}

struct sbm_ctx {
    sbm_config cfg{};
    int n_simd = 1024; // SIMDs of the device (4 per CU): 1024 on a whole MI355X, fewer on a partitioned one
    int L = 0;
    hipStream_t stream = nullptr;
    int64_t cand_cap = 0;

    // templates
    int n_templates = 0;
    int64_t n_features = 0;
    std::vector<DevTL> h_tls;          // [n_templates][L]
    std::vector<uint32_t> h_fxy;
    std::vector<int32_t> h_class, h_tid;
    std::vector<int32_t> h_active;
    DevBuf d_tls, d_fxy, d_flabel, d_flevel, d_foff, d_class, d_tid, d_active, d_rawmin, d_rawkeep;
    DevBuf d_citems, d_cfoff, d_soff, d_soffbase; // coarse pass: per active template one record, its feature offsets sorted by byte
                                                   // misalignment, the first 64 of them again by slot (k_prep_coarse_items)
    bool citems_dirty = true;
    DevBuf d_fxy_s, d_flabel_s, d_fcls; // refinement pass on the strip plane: features sorted by (x / T) & 15 per template level + 17 class offsets
    bool have_thr = false;
    float thr_cached = 0.f;

    // pyramid
    int rows[SBM_MAX_LEVELS]{}, cols[SBM_MAX_LEVELS]{};
    int channels = 0;
    int batch = 1;        // frames the per-level buffers, candidate lists and counters are allocated for
    int levels_valid = 0; // number of levels whose linear memories are resident
    int64_t lm_stride[SBM_MAX_LEVELS]{};
    DevBuf d_img[SBM_MAX_LEVELS], d_mask[SBM_MAX_LEVELS], d_quant[SBM_MAX_LEVELS], d_lm[SBM_MAX_LEVELS];
    // Levels that only the refinement pass reads (l < L-1) are built as ONE plane of spread bytes ("compact": an eighth
    // of the stores and of the HBM write-back; the refinement kernel applies the response LUT itself).  The 8-plane form
    // of such a level is materialised on demand for the stage entry points (ensure_full_lm).
    DevBuf d_lmc[SBM_MAX_LEVELS];
    bool lm_full[SBM_MAX_LEVELS]{}, lm_compact[SBM_MAX_LEVELS]{}; // which form of level l is current (frame 0 .. batch)
    bool lm_strip[SBM_MAX_LEVELS]{}; // the compact plane of level l is strip-interleaved (lm_strip_offset)
    DevBuf d_geo; // T[L], W[L], H[L] as int32 then stride[L] as int64
    bool foff_dirty = true;
    bool counters_fresh = false; // the linear-memory launch of this frame already reset the counters

    // candidates / results
    DevBuf d_cands, d_counters, d_out, d_outcount;
    void* comm = nullptr;  // ncclComm_t of this context (sbm_comm_init)
    int comm_world = 0, comm_rank = 0;
    sbm_match_rec* mirror_out = nullptr; // optional device-visible mirror of the results (sbm_set_result_mirror)
    int32_t* mirror_count = nullptr;
    // host entry point (sbm_match): pinned result buffer the last kernel writes into (no device-to-host copy, one
    // synchronisation per call).  Caller frame buffers are pinned only on request (sbm_pin_host_buffer): the upload of
    // a frame inside such a range is one asynchronous DMA; any other host pointer takes the runtime's pageable path.
    sbm_match_rec* h_res = nullptr;
    int32_t* h_res_count = nullptr;
    int64_t h_res_cap = 0;
    struct PinnedRange {
        const uint8_t* p;
        size_t bytes;
    };
    std::vector<PinnedRange> pinned; // ranges THIS context registered and has not yet unregistered
    // host batch pipeline (sbm_match_batch_host_begin / _end): frames travel over PCIe on copy_stream into one of two
    // device input buffers while the kernels of the previous sub-batch run on `stream`; every sub-batch's match lists
    // land in one pinned host block through the result mirror
    hipStream_t copy_stream = nullptr;
    DevBuf d_in[2], d_bout;
    hipEvent_t ev_up[2] = {}, ev_free[2] = {};
    uint8_t* h_bout = nullptr;
    size_t h_bout_bytes = 0;
    struct {
        bool active = false;
        int n_frames = 0;
        int64_t cap = 0;
    } pending;
    DevBuf d_scratch;

    // hipGraph cache for sbm_match_device (one captured graph per distinct argument tuple)
    struct GraphEntry {
        const void* img;
        int rows, cols, stride, ch;
        const void* mask;
        uint32_t thr_bits;
        void* out;
        int64_t cap;
        void* count;
        void* mo;
        void* mc;
        int frames;       // 0: single-frame graph (sbm_match_device), else the batch size
        int64_t frame_stride;
        hipGraph_t graph;
        hipGraphExec_t exec;
        uint64_t last_use;
    };
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    int quantize_mode = 0, quantize_hs = 0; // sbm_set_quantize_mode
    int pipeline_depth = 1;                 // sbm_set_pipeline_depth: batches the caller keeps in flight on this GPU
    int coarse_mode = coarse_mode_env();    // 0 auto, 1 four waves per item, 2 one wave per item (SBM_COARSE=block|wave: A/B knob)
    bool graph_mode = false; // measured on ROCm 7.2 / MI355X: graph replay is slower than stream launches
    hipStream_t side = nullptr;
    hipEvent_t ev_fork[SBM_MAX_LEVELS] = {};
    hipEvent_t ev_join = nullptr;
    void drop_graphs()
    {
        for (auto& g : graphs) {
            (void)hipGraphExecDestroy(g.exec);
            (void)hipGraphDestroy(g.graph);
        }
        graphs.clear();
    }

    // profiling
    bool profiling = false;
    bool profiling_keep = false; // enabled == 2: timings accumulate over calls until they are read
    std::vector<Timing> timings;
    std::vector<float> timing_ms;

    hipEvent_t chain_event = nullptr;
    hipStream_t chain_stream = nullptr;
    void clear_timings()
    {
        for (auto& t : timings) {
            if (!t.shared_a) (void)hipEventDestroy(t.a);
            (void)hipEventDestroy(t.b);
        }
        timings.clear();
        timing_ms.clear();
        chain_event = nullptr;
        chain_stream = nullptr;
    }
};

namespace {

struct Scope { // optional per-kernel HIP-event bracket on the launch stream
    sbm_ctx* c;
    hipStream_t s;
    bool on;
    Timing t{};
    Scope(sbm_ctx* c_, hipStream_t s_, const char* name) : c(c_), s(s_), on(c_->profiling)
    {
        if (!on) return;
        t.name = name;
        // consecutive launches share one event: the end of the previous kernel is the start of this
        // one, so a single event packet sits between two kernels (as in an un-instrumented stream)
        if (c->chain_event && c->chain_stream == s) {
            t.a = c->chain_event;
            t.shared_a = true;
        } else {
            (void)hipEventCreate(&t.a);
            (void)hipEventRecord(t.a, s);
        }
        (void)hipEventCreate(&t.b);
    }
    ~Scope()
    {
        if (!on) return;
        (void)hipEventRecord(t.b, s);
        c->chain_event = t.b;
        c->chain_stream = s;
        c->timings.push_back(t);
    }
};

// Launch with exact kernel begin/end timestamps when profiling: hipExtLaunchKernelGGL stamps the
// dispatch packet itself (what rocprofv3's kernel trace reads), so no event packet sits between
// kernels and the measured durations are the kernels' own.
#define SBM_LAUNCH(ctx, name_, kernel, grid, block, smem, stream, ...)                                        \
    do {                                                                                                      \
        if ((ctx)->profiling) {                                                                               \
            Timing t_;                                                                                        \
            t_.name = name_;                                                                                  \
            (void)hipEventCreate(&t_.a);                                                                      \
            (void)hipEventCreate(&t_.b);                                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, t_.a, t_.b, 0, __VA_ARGS__);             \
            (ctx)->timings.push_back(t_);                                                                     \
        } else {                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                               \
        }                                                                                                     \
    } while (0)

int check_level_dims(int rows, int cols, int T)
{
    if (rows <= 0 || cols <= 0 || T <= 0) return fail(SBM_ERR_INVALID, "bad level geometry %dx%d T=%d", rows, cols, T);
    if (rows % T || cols % T) // CV_Assert line2Dup.cpp:751-752
        return fail(SBM_ERR_INVALID, "level %dx%d is not a multiple of T=%d (linearize precondition)", rows, cols, T);
    if (((int64_t)rows * cols) % 16) // CV_Assert line2Dup.cpp:639
        return fail(SBM_ERR_INVALID, "level %dx%d: rows*cols %% 16 != 0 (computeResponseMaps precondition)", rows, cols);
    if ((int64_t)8 * lm_stride_for(rows, cols, T) >= (int64_t)INT32_MAX)
        return fail(SBM_ERR_INVALID, "level %dx%d too large for 32-bit linear-memory offsets", rows, cols);
    if (rows > 65535 || cols > 65535) return fail(SBM_ERR_INVALID, "image too large");
    return 0;
}

// (re)allocate the per-level buffers for a level-0 geometry
int ensure_geometry(sbm_ctx* c, int rows, int cols, int channels, int frames = 1)
{
    if (channels != 1 && channels != 3) return fail(SBM_ERR_INVALID, "channels must be 1 or 3, got %d", channels);
    if (frames < 1 || frames > 65535) return fail(SBM_ERR_INVALID, "batch of %d frames out of range", frames);
    bool same = c->channels == channels && c->rows[0] == rows && c->cols[0] == cols && c->d_lm[c->L - 1].p && frames <= c->batch;
    int r = rows, cc = cols;
    for (int l = 0; l < c->L; ++l) {
        if (l > 0) {
            r /= 2;
            cc /= 2;
        }
        if (int e = check_level_dims(r, cc, c->cfg.T[l])) return e;
        if (c->rows[l] != r || c->cols[l] != cc) same = false;
    }
    if (same) return 0;
    const size_t B = (size_t)std::max(frames, c->batch);
    r = rows;
    cc = cols;
    for (int l = 0; l < c->L; ++l) {
        if (l > 0) {
            r /= 2;
            cc /= 2;
        }
        c->rows[l] = r;
        c->cols[l] = cc;
        c->lm_stride[l] = lm_stride_for(r, cc, c->cfg.T[l]);
        if (int e = c->d_img[l].ensure(B * r * cc * channels)) return e;
        if (int e = c->d_mask[l].ensure((size_t)r * cc)) return e;
        if (int e = c->d_quant[l].ensure(B * r * cc)) return e;
        c->d_lm[l].release(); // fresh, zeroed: the tail past T*T*W*H must read as 0
        if (int e = c->d_lm[l].ensure(B * 8 * c->lm_stride[l], true)) return e;
        c->d_lmc[l].release();
        if (l < c->L - 1)
            if (int e = c->d_lmc[l].ensure(B * c->lm_stride[l], true)) return e;
        c->lm_full[l] = c->lm_compact[l] = c->lm_strip[l] = false;
    }
    if (B > (size_t)c->batch) { // per-frame candidate lists and counters
        if (int e = c->d_cands.ensure(B * c->cand_cap * sizeof(Cand))) return e;
        c->d_counters.release();
        if (int e = c->d_counters.ensure(B * 40 * sizeof(int32_t) + 256, true)) return e;
        c->batch = (int)B;
    }
    c->channels = channels;
    c->foff_dirty = true;
    c->levels_valid = 0;
    c->drop_graphs(); // captured launches hold the old buffer addresses
    return 0;
}

int ensure_level(sbm_ctx* c, int l, int rows, int cols)
{
    if (int e = check_level_dims(rows, cols, c->cfg.T[l])) return e;
    if (c->rows[l] == rows && c->cols[l] == cols && c->d_lm[l].p && c->d_quant[l].p) return 0;
    c->drop_graphs();
    c->rows[l] = rows;
    c->cols[l] = cols;
    c->lm_stride[l] = lm_stride_for(rows, cols, c->cfg.T[l]);
    if (int e = c->d_quant[l].ensure((size_t)rows * cols)) return e;
    c->d_lm[l].release();
    if (int e = c->d_lm[l].ensure((size_t)8 * c->lm_stride[l], true)) return e;
    c->d_lmc[l].release();
    c->lm_full[l] = c->lm_compact[l] = false;
    c->foff_dirty = true;
    // This level now holds ONE frame of a geometry the other per-level buffers (next level's image, mask, compact
    // plane, the batch's frames) were not sized for: the next ensure_geometry() must not take its `same` fast path.
    c->channels = 0;
    return 0;
}

int upload_geo(sbm_ctx* c, hipStream_t s)
{
    const int L = c->L;
    std::vector<int32_t> g(3 * SBM_MAX_LEVELS);
    std::vector<int64_t> st(SBM_MAX_LEVELS);
    for (int l = 0; l < L; ++l) {
        g[l] = c->cfg.T[l];
        g[SBM_MAX_LEVELS + l] = c->cols[l] / c->cfg.T[l];
        g[2 * SBM_MAX_LEVELS + l] = c->rows[l] / c->cfg.T[l];
        st[l] = c->lm_stride[l];
    }
    const size_t gi = g.size() * sizeof(int32_t), gs = st.size() * sizeof(int64_t);
    if (int e = c->d_geo.ensure(gi + gs)) return e;
    HIP_TRY(hipMemcpyAsync(c->d_geo.p, g.data(), gi, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync((char*)c->d_geo.p + gi, st.data(), gs, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // host vectors go out of scope
    return 0;
}

// Which gradient kernel a launch gets.  The row-streaming kernel (sbm_quantize_stream.h) is the throughput form: a wave
// walks a 256-column strip for tens of rows, so a launch needs thousands of waves to fill the chip and a single small
// frame would be a few long serial chains; the tile kernel (k_quantize) is the latency form (1024 short-lived tiles
// per Mpixel) and the only one with the float outputs and arbitrary widths.  SBM_QUANTIZE=tile|stream forces one for
// A/B runs, SBM_QS_HS sets the rows per wave.
// segment lanes of the packed last strip (0: none); SBM_QS_PACK=0 is the A/B knob
static int qs_pack_lanes(int rows, int cols, int ch, int frames, int64_t img_fs, int stride)
{
    static const bool pack_ok = !(getenv("SBM_QS_PACK") && atoi(getenv("SBM_QS_PACK")) == 0);
    const int lanes = pack_ok ? quantize_stream_pack_lanes(rows, cols, ch, frames) : 0;
    if (!lanes) return 0;
    // the frames of a group are addressed by 32-bit per-lane offsets from the group's first frame: the caller's frame
    // stride (any value, also negative or zero) must keep them within 2 GiB
    const int64_t span = (int64_t)(64 / lanes - 1) * img_fs;
    if (img_fs < 0 || span + (int64_t)rows * stride >= (int64_t)0x7ff00000) return 0;
    return lanes;
}

// rows = the image's rows; band_rows = the output rows of this launch (rows for a whole level; a row band of a
// build-sharded step otherwise, which always takes the streaming kernel: the tile kernel has no row-range form)
int quantize_stream_rows(const sbm_ctx* c, int rows, int cols, int ch, int frames, bool wf, int64_t img_fs, int stride, int band_rows = 0)
{
    const bool band = band_rows > 0 && band_rows < rows;
    const int out_rows = band ? band_rows : rows;
    static const char* env = getenv("SBM_QUANTIZE");
    // SBM_QS_HS=a[,b]: rows per work item, a for launches of >= 1 Mpixel per frame, b (default a) for smaller levels
    static const char* env_hs_s = getenv("SBM_QS_HS");
    static const int env_hs0 = env_hs_s ? atoi(env_hs_s) : 0;
    static const int env_hs1 = env_hs_s && strchr(env_hs_s, ',') ? atoi(strchr(env_hs_s, ',') + 1) : env_hs0;
    const int env_hs = (int64_t)rows * cols >= (1 << 20) ? env_hs0 : env_hs1;
    const int mode = c->quantize_mode ? c->quantize_mode : (env && !strcmp(env, "tile") ? 1 : (env && !strcmp(env, "stream") ? 2 : 0));
    const int force_hs = c->quantize_hs ? c->quantize_hs : env_hs;
    if (wf || (mode == 1 && !band) || cols < 4 || (cols & 3) || (int64_t)rows * cols >= (int64_t)0x7ff00000) return 0;
    const int64_t strips = (cols + QS_USEFUL - 1) / QS_USEFUL;
    if (force_hs > 0) return (force_hs + 1) & ~1;
    // The kernel is bound by vector-instruction issue, and a SIMD needs its full set of resident waves (3 at the
    // BGR kernel's register count, 6 for gray) to hide the scalar bookkeeping and dependency bubbles of each: a launch
    // takes about  ceil(waves / resident slots) x (hs + 10 halo rows).  Choose the rows per wave that minimise it.
    // resident waves per SIMD the launch is sized for (experiment knob SBM_QS_WAVES, with SBM_QS_LDS capping the
    // workgroups per CU to match: fewer gradient waves leave registers for the other kernels' waves)
    static const int env_w = getenv("SBM_QS_WAVES") ? atoi(getenv("SBM_QS_WAVES")) : 0;
    const int64_t slots = (int64_t)c->n_simd * (env_w > 0 ? env_w : (ch == 3 ? 3 : 6));
    // waves per row block: one per strip and frame, except that a narrow last strip is shared by several frames
    const int pack = qs_pack_lanes(rows, cols, ch, frames, img_fs, stride);
    const int64_t per_rb = pack ? (strips - 1) * frames + (frames + 64 / pack - 1) / (64 / pack) : strips * frames;
    int hs = 0;
    int64_t best = INT64_MAX;
    // a work item runs whole groups of 7 row iterations (sbm_quantize_stream.h): rows + 10 warm-up / drain, rounded up
    if (c->pipeline_depth >= 2) {
        // Throughput sizing (the caller keeps several batches in flight on other streams / contexts, sbm_set_pipeline_depth):
        // what matters is the launch's TOTAL work, waves x row iterations -- the 10 warm-up / drain rows of every work item
        // are pure overhead, so fewer, longer items -- and not that one launch alone fills every SIMD: the other batches'
        // kernels take the SIMDs this one leaves idle.  Items longer than 42 iterations stopped paying in the measurement
        // (profiles/r03_rows_per_item_sweep.txt: 16 x 1024^2 x 3, three batches in flight, us per step: 126.0 with the
        // latency sizing 24 / 10 rows, 116.4 with 32 / 32, 122.0 with 46 / 46, 137 with 60 / 60).
        for (int h = 4; h <= 32; h += 2) {
            const int64_t waves = per_rb * ((out_rows + h - 1) / h);
            const int64_t cost = waves * ((std::min(h, out_rows) + 10 + 6) / 7 * 7);
            if (cost <= best) best = cost, hs = h;
        }
    } else {
        // Latency sizing: every work item resident at once, as few row iterations as that allows
        for (int h = 4; h <= 130; h += 2) {
            const int64_t waves = per_rb * ((out_rows + h - 1) / h);
            const int64_t cost = ((waves + slots - 1) / slots) * ((std::min(h, out_rows) + 10 + 6) / 7 * 7);
            if (cost <= best) best = cost, hs = h;
        }
    }
    // small launches: the 16 x 64 tiles of k_quantize finish sooner than a few long serial chains
    if (!band && mode != 2 && (int64_t)rows * cols * frames < ((int64_t)4 << 20)) return 0;
    return hs;
}

int launch_quantize(sbm_ctx* c, hipStream_t s, const uint8_t* d_img, int rows, int cols, int stride, int ch,
                    const uint8_t* d_mask, float weak, uint8_t* d_out, float* d_mag, float* d_ori, uint8_t* d_pyr,
                    int frames = 1, int64_t img_fs = 0, int row_lo = 0, int row_hi = -1)
{
    if (row_hi < 0) row_hi = rows;
    const bool band = row_lo > 0 || row_hi < rows;
    dim3 grid((cols + QT_C - 1) / QT_C, (rows + QT_R - 1) / QT_R, frames);
    const float thr_sq = weak * weak;
    const bool wf = d_mag || d_ori;
    const int64_t out_fs = (int64_t)rows * cols, pyr_fs = (int64_t)(rows / 2) * (cols / 2) * ch; // the context's own per-frame buffers
    if (const int hs = quantize_stream_rows(c, rows, cols, ch, frames, wf, img_fs, stride, band ? row_hi - row_lo : 0)) {
        QSArgs a;
        memset(&a, 0, sizeof a);
        a.img = d_img;
        a.mask = d_mask;
        a.out = d_out;
        a.pyr = d_pyr;
        a.img_fs = img_fs;
        a.out_fs = out_fs;
        a.pyr_fs = pyr_fs;
        a.rows = rows;
        a.cols = cols;
        a.stride = stride;
        a.thr_i = thr_sq < 2147483000.f ? (int)floorf(thr_sq) : INT_MAX; // mag is an integer: mag > weak^2 <=> mag > floor(weak^2)
        a.hs = hs;
        a.row_lo = row_lo;
        a.row_hi = row_hi;
        a.n_strips = (cols + QS_USEFUL - 1) / QS_USEFUL;
        a.n_rblocks = (row_hi - row_lo + hs - 1) / hs;
        a.frames = frames;
        a.pack_lanes = qs_pack_lanes(rows, cols, ch, frames, img_fs, stride);
        a.pack_groups = a.pack_lanes ? (frames + 64 / a.pack_lanes - 1) / (64 / a.pack_lanes) : 0;
        const dim3 g((unsigned)((quantize_stream_items(a) + 3) / 4));
        // experiment knob: dynamic LDS the kernel never touches, to cap the workgroups per CU (waves per SIMD)
        static const int lds_pad = getenv("SBM_QS_LDS") ? atoi(getenv("SBM_QS_LDS")) : 0;
        if (lds_pad > 0) {
            static bool once = false;
            if (!once) {
                once = true;
                (void)hipFuncSetAttribute((const void*)k_quantize_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_pad);
                (void)hipFuncSetAttribute((const void*)k_quantize_stream<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_pad);
            }
        }
        if (ch == 1) SBM_LAUNCH(c, "k_quantize", (k_quantize_stream<1>), g, dim3(256), lds_pad, s, a);
        else SBM_LAUNCH(c, "k_quantize", (k_quantize_stream<3>), g, dim3(256), lds_pad, s, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (band) return fail(SBM_ERR_INVALID, "a row band needs the streaming gradient kernel (cols %% 4 == 0, no float outputs)");
    // many tiles per CU (a batch of frames): 512-thread blocks, four of them per CU; else 1024-thread blocks (tile latency)
    const bool many = (int64_t)grid.x * grid.y * grid.z >= 2048;
#define SBM_QUANTIZE(CH_, WF_)                                                                                              \
    do {                                                                                                                    \
        if (many)                                                                                                           \
            SBM_LAUNCH(c, "k_quantize", (k_quantize<CH_, WF_, QN_THROUGHPUT>), grid, dim3(QN_THROUGHPUT), 0, s, d_img, rows, cols, \
                       stride, d_mask, thr_sq, d_out, d_mag, d_ori, d_pyr, img_fs, out_fs, pyr_fs);                         \
        else                                                                                                                \
            SBM_LAUNCH(c, "k_quantize", (k_quantize<CH_, WF_, QN_LATENCY>), grid, dim3(QN_LATENCY), 0, s, d_img, rows, cols, stride, \
                       d_mask, thr_sq, d_out, d_mag, d_ori, d_pyr, img_fs, out_fs, pyr_fs);                                 \
    } while (0)
    if (ch == 1 && !wf) SBM_QUANTIZE(1, false);
    else if (ch == 1) SBM_QUANTIZE(1, true);
    else if (!wf) SBM_QUANTIZE(3, false);
    else SBM_QUANTIZE(3, true);
#undef SBM_QUANTIZE
    HIP_TRY(hipGetLastError());
    return 0;
}

// the register-only linear-memory kernel handles the reference's strides on 16-byte aligned rows
bool lm_rows_ok(const uint8_t* d_q, int cols, int T)
{
    return (T == 4 || T == 8) && ((cols / T) & 3) == 0 && (cols & 15) == 0 && (((uintptr_t)d_q) & 15) == 0;
}

bool use_compact_lm()
{
    static const bool on = !(getenv("SBM_FULL_LM") && atoi(getenv("SBM_FULL_LM")) != 0); // tuning / A-B knob
    return on;
}

// 8-plane linear memories of level l (frame 0) for the stage entry points, expanded from the compact plane if needed
int ensure_full_lm(sbm_ctx* c, int l, hipStream_t s)
{
    if (c->lm_full[l] || !c->lm_compact[l]) return 0;
    const int T = c->cfg.T[l];
    const int64_t n = (int64_t)T * T * (c->cols[l] / T) * (c->rows[l] / T);
    hipLaunchKernelGGL(k_expand_lm, dim3((unsigned)std::min<int64_t>((n / 4 + 255) / 256, 4096)), dim3(256), 0, s, c->d_lmc[l].as<uint8_t>(), n,
                       c->d_lm[l].as<uint8_t>(), c->lm_stride[l], c->lm_strip[l] ? 1 : 0, c->cols[l] / T, c->rows[l] / T);
    HIP_TRY(hipGetLastError());
    c->lm_full[l] = true;
    return 0;
}

int launch_build_lm(sbm_ctx* c, hipStream_t s, const uint8_t* d_q, int rows, int cols, int T, uint8_t* d_lm,
                    int64_t lm_stride)
{
    const int W = cols / T, H = rows / T;
    if (lm_rows_ok(d_q, cols, T)) {
        LmArgs a;
        memset(&a, 0, sizeof a);
        a.n_levels = 1;
        a.lv[0] = LmLevelArgs{d_q, d_lm, lm_stride, rows, cols, W, H, T, 0, 0, 0, 0, LM_FULL_SPLIT};
        const int64_t items = (int64_t)rows * (W >> 2) * LM_FULL_SPLIT;
        SBM_LAUNCH(c, "k_build_lm", k_build_lm_rows, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (T <= 8) {
        const int tw = LM_GX * T, lw = tw + T - 1, lwp = (lw + 3) & ~3, lh = 2 * T - 1;
        const size_t smem = (size_t)lh * lwp + (size_t)lh * tw + (size_t)T * T * LM_GX;
        dim3 grid((W + LM_GX - 1) / LM_GX, H);
        SBM_LAUNCH(c, "k_build_lm", k_build_lm, grid, dim3(256), smem, s, d_q, rows, cols, T, W, H, d_lm, lm_stride);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // large strides: unfused path through scratch (spread, 8 response maps)
    const int64_t n = (int64_t)rows * cols;
    if (int e = c->d_scratch.ensure((size_t)9 * n)) return e;
    uint8_t* sp = c->d_scratch.as<uint8_t>();
    uint8_t* maps = sp + n;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    Scope sc(c, s, "k_build_lm_unfused");
    hipLaunchKernelGGL(k_spread, dim3(blocks), dim3(256), 0, s, d_q, rows, cols, T, sp);
    hipLaunchKernelGGL(k_response, dim3(blocks), dim3(256), 0, s, sp, n, maps);
    for (int o = 0; o < 8; ++o)
        hipLaunchKernelGGL(k_linearize, dim3(blocks), dim3(256), 0, s, maps + o * n, rows, cols, T, d_lm + o * lm_stride);
    HIP_TRY(hipGetLastError());
    return 0;
}

// smallest raw in [0, 4nf] with score > thr (strict) / score >= thr; INT_MAX if none.
// Evaluated with the reference's own float expression (line2Dup.cpp:1206, :1273).
void raw_thresholds(int nf, float thr, int32_t* gt, int32_t* ge)
{
    *gt = *ge = INT_MAX;
    if (nf <= 0) return;
    const int hi = 4 * nf;
    auto score = [nf](int raw) { return (raw * 100.f) / (4 * nf); };
    int lo = 0, h = hi + 1; // first raw with score > thr
    while (lo < h) {
        int m = lo + (h - lo) / 2;
        if (score(m) > thr) h = m;
        else lo = m + 1;
    }
    if (lo <= hi) *gt = lo;
    lo = 0;
    h = hi + 1; // first raw with !(score < thr)
    while (lo < h) {
        int m = lo + (h - lo) / 2;
        if (!(score(m) < thr)) h = m;
        else lo = m + 1;
    }
    if (lo <= hi) *ge = lo;
}

int ensure_thresholds(sbm_ctx* c, float thr, hipStream_t s)
{
    if (c->have_thr && memcmp(&thr, &c->thr_cached, sizeof thr) == 0) return 0;
    const size_t n = (size_t)c->n_templates * c->L;
    std::vector<int32_t> gt(n), ge(n);
    for (size_t i = 0; i < n; ++i) raw_thresholds(c->h_tls[i].nf, thr, &gt[i], &ge[i]);
    if (int e = c->d_rawmin.ensure(n * 4)) return e;
    if (int e = c->d_rawkeep.ensure(n * 4)) return e;
    HIP_TRY(hipMemcpyAsync(c->d_rawmin.p, gt.data(), n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_rawkeep.p, ge.data(), n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    c->thr_cached = thr;
    c->have_thr = true;
    c->citems_dirty = true;
    return 0;
}

int ensure_foff(sbm_ctx* c, hipStream_t s)
{
    if (!c->foff_dirty || c->n_features == 0) return 0;
    if (int e = upload_geo(c, s)) return e;
    const int32_t* g = c->d_geo.as<int32_t>();
    const int64_t* st = (const int64_t*)((char*)c->d_geo.p + 3 * SBM_MAX_LEVELS * sizeof(int32_t));
    const int blocks = (int)std::min<int64_t>((c->n_features + 255) / 256, 8192);
    SBM_LAUNCH(c, "k_prep_features", k_prep_features, dim3(blocks), dim3(256), 0, s, c->d_fxy.as<uint32_t>(),
                       c->d_flabel.as<uint8_t>(), c->d_flevel.as<uint8_t>(), c->n_features, g,
                       g + SBM_MAX_LEVELS, g + 2 * SBM_MAX_LEVELS, st, c->d_foff.as<int32_t>(), c->L - 1);
    HIP_TRY(hipGetLastError());
    // The similarity kernels may be enqueued on a different stream than `s` (graph replay, a caller's stream):
    // the table must be complete before any of them can start.  Only runs after a template / geometry change.
    HIP_TRY(hipStreamSynchronize(s));
    c->foff_dirty = false;
    c->citems_dirty = true;
    return 0;
}

// per-slot records of the coarse pass (template record + threshold + first feature offsets): after any change of the
// active set, the thresholds, the templates or the geometry
int ensure_citems(sbm_ctx* c, hipStream_t s)
{
    const int n_active = (int)c->h_active.size();
    if (!c->citems_dirty || n_active == 0) return 0;
    if (c->rows[c->L - 1] <= 0) return 0; // no geometry yet: the first match call comes back here
    const int lc = c->L - 1, T = c->cfg.T[lc], W = c->cols[lc] / T, H = c->rows[lc] / T;
    if (int e = c->d_citems.ensure((size_t)n_active * sizeof(CoarseItem))) return e;
    if (int e = c->d_cfoff.ensure((size_t)n_active * 64 * 4)) return e;
    std::vector<int32_t> base((size_t)n_active);
    int64_t total = 0;
    for (int i = 0; i < n_active; ++i) {
        base[i] = (int32_t)total;
        total += c->h_tls[(size_t)c->h_active[i] * c->L + lc].nf;
    }
    if (total >= (int64_t)INT32_MAX) return fail(SBM_ERR_INVALID, "too many coarse-level features in the selection");
    if (int e = c->d_soff.ensure((size_t)std::max<int64_t>(total, 1) * 4)) return e;
    if (int e = c->d_soffbase.ensure((size_t)n_active * 4)) return e;
    HIP_TRY(hipMemcpyAsync(c->d_soffbase.p, base.data(), (size_t)n_active * 4, hipMemcpyHostToDevice, s));
    const int zero_off = (int)(7 * c->lm_stride[lc] + (int64_t)T * T * W * H);
    hipLaunchKernelGGL(k_prep_coarse_items, dim3((unsigned)((n_active + 63) / 64)), dim3(64), 0, s, c->d_active.as<int32_t>(), n_active,
                       c->d_tls.as<DevTL>(), c->L, lc, c->d_rawmin.as<int32_t>(), c->d_foff.as<int32_t>(), c->d_soffbase.as<int32_t>(), T, W, H,
                       zero_off, c->d_citems.as<CoarseItem>(), c->d_soff.as<int32_t>(), c->d_cfoff.as<int32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s)); // consumers may run on another stream (as ensure_foff)
    c->citems_dirty = false;
    return 0;
}

// Form of level l's linear memories when the one-launch builder makes them: the refinement-only levels are ONE plane of
// spread bytes (compact), strip-interleaved when the grid width allows it (the refinement pass reads 16 x 16 cells per
// feature: 2 - 4 cache lines instead of 16)
void lm_form(const sbm_ctx* c, int l, bool* compact, bool* strip)
{
    static const bool strip_ok = !(getenv("SBM_STRIP_LM") && atoi(getenv("SBM_STRIP_LM")) == 0); // A/B knob
    *compact = l < c->L - 1 && c->d_lmc[l].p && use_compact_lm();
    *strip = *compact && strip_ok && ((c->cols[l] / c->cfg.T[l]) & 15) == 0;
}

// gradient stage + linear memories for every level; d_img0 may be external
// reset_count != null: the linear-memory launch also zeroes the per-frame counters and *reset_count
// (c->counters_fresh tells enqueue_coarse to skip its own k_reset launch).
//
// Row bands (build-sharded multi-GPU step, sbm_match_batch_device_banded): with bands.n > 1 the gradient stage of
// level l is launched only for the bands [bands.first, bands.first + bands.count) of bands.n equal row bands, each
// widened by band_halo(l) rows on either side -- the rows of level l whose fused cv::pyrDown output level l+1's band
// (itself widened) reads: halo(L-1) = 0, halo(l) = 2 * (halo(l+1) + 5) (7x7 Gaussian 3 + Sobel 1 + vote 1 rows of
// the next level's image on either side, two source rows each).  bands.between runs after the gradient launches and
// before the linear memories (the all-gather of the other ranks' bands).
struct Bands {
    int n = 1, first = 0, count = 1;
    int (*between)(sbm_ctx*, hipStream_t, int frames) = nullptr;
};

int band_halo(int L, int l)
{
    int e = 0;
    for (int k = L - 2; k >= l; --k) e = 2 * (e + 5);
    return e;
}

int check_bands(const sbm_ctx* c, int n_bands)
{
    if (n_bands < 1) return fail(SBM_ERR_INVALID, "n_bands must be >= 1");
    for (int l = 0; l < c->L; ++l)
        if (c->rows[l] % n_bands || ((c->rows[l] / n_bands) & 1))
            return fail(SBM_ERR_INVALID, "level %d: %d rows do not split into %d bands of an even number of rows", l, c->rows[l], n_bands);
    return 0;
}

int enqueue_pyramid(sbm_ctx* c, hipStream_t s, const uint8_t* d_img0, int stride0, const uint8_t* d_mask0,
                    int32_t* reset_count = nullptr, int frames = 1, int64_t img0_fs = 0, const Bands* bands = nullptr)
{
    const int ch = c->channels;
    const uint8_t* img = d_img0;
    int stride = stride0;
    const uint8_t* mask = d_mask0;
    bool all_rows = true;
    for (int l = 0; l < c->L; ++l) all_rows = all_rows && lm_rows_ok(c->d_quant[l].as<uint8_t>(), c->cols[l], c->cfg.T[l]);
    c->counters_fresh = false;
    for (int l = 0; l < c->L; ++l) {
        if (l > 0) {
            const int pr = c->rows[l - 1], pc = c->cols[l - 1];
            // the image of level l was produced by level l-1's quantize launch (fused cv::pyrDown)
            if (mask) {
                const int n = c->rows[l] * c->cols[l];
                SBM_LAUNCH(c, "k_resize_mask", k_resize_mask, dim3(std::min((n + 255) / 256, 4096)), dim3(256), 0, s, mask, pr, pc,
                                   c->d_mask[l].as<uint8_t>(), c->rows[l], c->cols[l]);
                HIP_TRY(hipGetLastError());
                mask = c->d_mask[l].as<uint8_t>();
            }
            img = c->d_img[l].as<uint8_t>();
            stride = c->cols[l] * ch;
        }
        const int nb = bands && bands->n > 1 ? bands->count : 1;
        for (int b = 0; b < nb; ++b) {
            int lo = 0, hi = c->rows[l];
            if (bands && bands->n > 1) {
                const int br = c->rows[l] / bands->n, e = band_halo(c->L, l);
                lo = std::max(0, (bands->first + b) * br - e);
                hi = std::min(c->rows[l], (bands->first + b + 1) * br + e);
            }
            if (int e = launch_quantize(c, s, img, c->rows[l], c->cols[l], stride, ch, mask, c->cfg.weak_threshold,
                                        c->d_quant[l].as<uint8_t>(), nullptr, nullptr,
                                        l + 1 < c->L ? c->d_img[l + 1].as<uint8_t>() : nullptr, frames,
                                        l == 0 ? img0_fs : (int64_t)c->rows[l] * c->cols[l] * ch, lo, hi))
                return e;
        }
        if (!all_rows && frames > 1) return fail(SBM_ERR_INVALID, "batched match needs T in {4, 8} and 16-column-aligned levels");
        if (!all_rows) {
            if (int e = launch_build_lm(c, s, c->d_quant[l].as<uint8_t>(), c->rows[l], c->cols[l], c->cfg.T[l],
                                        c->d_lm[l].as<uint8_t>(), c->lm_stride[l]))
                return e;
            c->lm_full[l] = true;
            c->lm_compact[l] = false;
        }
    }
    if (bands && bands->between)
        if (int e = bands->between(c, s, frames)) return e;
    if (all_rows) { // every level's linear memories (and the counter reset) in one launch
        LmArgs a;
        memset(&a, 0, sizeof a);
        a.n_levels = c->L;
        int blocks = 0;
        // block ranges coarsest level first: its blocks (T * 8 / 4 stores per lane) are the long ones, and a launch
        // that dispatches its long blocks last ends with a few of them running alone
        for (int l = c->L - 1; l >= 0; --l) {
            const int T = c->cfg.T[l], W = c->cols[l] / T, H = c->rows[l] / T;
            bool compact, strip;
            lm_form(c, l, &compact, &strip);
            const int split = frames < 4 ? LM_FULL_SPLIT : 1; // few frames: shorter, more numerous work items
            a.lv[l] = LmLevelArgs{c->d_quant[l].as<uint8_t>(), compact ? c->d_lmc[l].as<uint8_t>() : c->d_lm[l].as<uint8_t>(), c->lm_stride[l],
                                  c->rows[l], c->cols[l], W, H, T, blocks, (int64_t)c->rows[l] * c->cols[l],
                                  (int64_t)(compact ? 1 : 8) * c->lm_stride[l], strip ? 2 : (compact ? 1 : 0), split};
            c->lm_compact[l] = compact;
            c->lm_strip[l] = strip;
            c->lm_full[l] = !compact;
            const int64_t items = strip     ? (int64_t)((W + 63) >> 6) * ((H + 15) >> 4) * T * 256
                                  : compact ? (int64_t)c->rows[l] * (W >> 2)
                                            : (int64_t)c->rows[l] * (W >> 2) * split;
            blocks += (int)((items + 255) / 256);
        }
        if (reset_count) {
            a.counters = c->d_counters.as<int32_t>();
            a.out_count = reset_count;
            c->counters_fresh = true;
        }
        SBM_LAUNCH(c, "k_build_lm", k_build_lm_rows, dim3(blocks, frames), dim3(256), 0, s, a);
        HIP_TRY(hipGetLastError());
    }
    c->levels_valid = c->L;
    return 0;
}

// host-side preparation of the template loop: validation, integer thresholds, feature offsets.
// May synchronise (only when something changed); never called inside a stream capture.
// The tables are (re)built on the CONTEXT's stream and that stream is synchronised by the host, whatever stream the
// consumers will run on: the caller's stream is neither waited on nor given work here (a host wait on
// the caller's stream fails while that stream is being captured).  Nothing is enqueued when nothing changed.
int prepare_templates(sbm_ctx* c, hipStream_t, float threshold, int64_t cap)
{
    if (c->n_templates == 0) return fail(SBM_ERR_STATE, "no templates uploaded");
    if (cap < 0 || cap > INT32_MAX) return fail(SBM_ERR_INVALID, "bad output capacity");
    if (int e = ensure_thresholds(c, threshold, c->stream)) return e;
    if (int e = ensure_foff(c, c->stream)) return e;
    return ensure_citems(c, c->stream);
}

// coarse pass over the active templates (reset + k_similarity_coarse; single-level pyramids emit here)
int enqueue_coarse(sbm_ctx* c, hipStream_t s, sbm_match_rec* d_out, int64_t cap, int32_t* d_count, int frames = 1)
{
    const int L = c->L, lc = L - 1;
    int32_t* counters = c->d_counters.as<int32_t>();
    if (!c->counters_fresh && frames > 1) return fail(SBM_ERR_STATE, "batched template loop without a batched pyramid");
    if (!c->counters_fresh) hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, s, counters, d_count);
    c->counters_fresh = false;
    const int n_active = (int)c->h_active.size();
    if (n_active > 0) {
        const int T = c->cfg.T[lc], W = c->cols[lc] / T, H = c->rows[lc] / T;
        int chunks = (W * H + COARSE_POS_PER_BLOCK - 1) / COARSE_POS_PER_BLOCK;
        if (c->thr_cached >= 0.f) {
            // every raw_min is >= 1, so positions past a template's span (score 0) are never candidates:
            // launch only the chunks some template reaches; a multiple of 8 of them lets the kernel give
            // each XCD a contiguous, equally loaded range of chunks (its L2 then holds one slice of the
            // linear memories)
            int max_npos = 0;
            for (int32_t t : c->h_active) {
                const DevTL& tl = c->h_tls[(size_t)t * L + lc];
                const int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
                max_npos = std::max(max_npos, (H - hf) * W + (W - wf) + 1);
            }
            int need = std::max(1, (std::min(std::max(max_npos, 0), W * H) + COARSE_POS_PER_BLOCK - 1) / COARSE_POS_PER_BLOCK);
            if (need >= 8) need = (need + 7) / 8 * 8;
            chunks = std::min(need, (chunks + 7) / 8 * 8);
        }
        for (int first = 0; first < n_active; first += 65535) {
            const int cnt = std::min(65535, n_active - first);
            // large launches: one wave per (chunk, template, frame); small ones (a single frame with a few hundred
            // templates): four waves share an item's features so that enough loads are in flight
            const bool per_wave = c->coarse_mode == 2 || (c->coarse_mode == 0 && (int64_t)chunks * cnt * frames >= 8192);
            if (per_wave)
                SBM_LAUNCH(c, "k_similarity_coarse", k_similarity_coarse_wave, dim3(chunks, (cnt + 3) / 4, frames), dim3(256), 0, s,
                           c->d_lm[lc].as<uint8_t>(), c->lm_stride[lc], c->rows[lc], c->cols[lc], T, W, H, L, lc, c->d_tls.as<DevTL>(),
                           c->d_soff.as<int32_t>(), c->d_citems.as<CoarseItem>() + first, c->d_cfoff.as<int32_t>() + (size_t)first * 64, cnt,
                           c->d_rawkeep.as<int32_t>(), c->d_class.as<int32_t>(), c->d_tid.as<int32_t>(),
                           c->d_cands.as<Cand>(), counters, (int)c->cand_cap, (int64_t)8 * c->lm_stride[lc]);
            else
                SBM_LAUNCH(c, "k_similarity_coarse", k_similarity_coarse, dim3(chunks, cnt, frames), dim3(256), 0, s, c->d_lm[lc].as<uint8_t>(),
                           c->lm_stride[lc], c->rows[lc], c->cols[lc], T, W, H, L, lc, c->d_tls.as<DevTL>(), c->d_fxy.as<uint32_t>(),
                           c->d_foff.as<int32_t>(), c->d_active.as<int32_t>() + first, c->d_rawmin.as<int32_t>(),
                           c->d_rawkeep.as<int32_t>(), c->d_class.as<int32_t>(), c->d_tid.as<int32_t>(),
                           c->d_cands.as<Cand>(), counters, (int)c->cand_cap, (int64_t)8 * c->lm_stride[lc]);
        }
        HIP_TRY(hipGetLastError());
    }
    if (L == 1) {
        SBM_LAUNCH(c, "k_emit_coarse", k_emit_coarse, dim3(256, frames), dim3(256), 0, s, c->d_cands.as<Cand>(), counters, (int)c->cand_cap,
                           c->d_tls.as<DevTL>(), L, lc, c->d_class.as<int32_t>(), c->d_tid.as<int32_t>(), d_out, d_count,
                           (int)cap, c->mirror_out, c->mirror_count);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// refinement passes, finest level last (emits the match records)
int enqueue_local(sbm_ctx* c, hipStream_t s, sbm_match_rec* d_out, int64_t cap, int32_t* d_count, int frames = 1)
{
    const int L = c->L;
    int32_t* counters = c->d_counters.as<int32_t>();
    for (int l = L - 2; l >= 0; --l) {
        const int T = c->cfg.T[l], W = c->cols[l] / T, H = c->rows[l] / T;
        static const int local_grid = getenv("SBM_LOCAL_GRID") ? std::max(1, atoi(getenv("SBM_LOCAL_GRID"))) : 512;
        const bool compact = c->lm_compact[l] && !c->lm_full[l];
        if (!compact && !c->lm_full[l]) return fail(SBM_ERR_STATE, "linear memories of level %d are not built", l);
        static const int local_waves = getenv("SBM_LOCAL_WAVES") ? atoi(getenv("SBM_LOCAL_WAVES")) : 0; // tuning knob: 4 or 16
        const bool small_blocks = local_waves ? local_waves == 4 : frames >= 4;
#define SBM_LOCAL(COMPACT_, LM_, FS_)                                                                                               \
        if (small_blocks) SBM_LOCAL_LW(COMPACT_, LM_, FS_, 4); else SBM_LOCAL_LW(COMPACT_, LM_, FS_, LOCAL_WAVES)
#define SBM_LOCAL_LW(COMPACT_, LM_, FS_, LW_)                                                                                       \
        SBM_LAUNCH(c, "k_similarity_local", (k_similarity_local<COMPACT_, LW_>), dim3(frames, local_grid), dim3(64 * LW_), 0, s, LM_, \
                   c->lm_stride[l], c->rows[l], c->cols[l], T, W, H, L, l, c->d_tls.as<DevTL>(),                                   \
                   (COMPACT_) == 2 ? c->d_fxy_s.as<uint32_t>() : c->d_fxy.as<uint32_t>(),                                          \
                   c->d_foff.as<int32_t>(), c->d_rawkeep.as<int32_t>(), c->d_class.as<int32_t>(), c->d_tid.as<int32_t>(),          \
                   c->d_cands.as<Cand>(), counters, (int)c->cand_cap, l == 0 ? 1 : 0, d_out, d_count, (int)cap, c->mirror_out,     \
                   c->mirror_count, c->profiling ? 1 : 0, (int64_t)(FS_) * c->lm_stride[l],                                        \
                   (COMPACT_) == 2 ? c->d_flabel_s.as<uint8_t>() : c->d_flabel.as<uint8_t>(), c->d_fcls.as<uint16_t>())
        if (compact && c->lm_strip[l]) { SBM_LOCAL(2, c->d_lmc[l].as<uint8_t>(), 1); }
        else if (compact) { SBM_LOCAL(1, c->d_lmc[l].as<uint8_t>(), 1); }
        else { SBM_LOCAL(0, c->d_lm[l].as<uint8_t>(), 8); }
#undef SBM_LOCAL
#undef SBM_LOCAL_LW
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// matchClass over the active templates; results into d_out/d_count (device)
int enqueue_templates(sbm_ctx* c, hipStream_t s, float threshold, sbm_match_rec* d_out, int64_t cap, int32_t* d_count)
{
    if (c->levels_valid < c->L) return fail(SBM_ERR_STATE, "pyramid not built (%d of %d levels)", c->levels_valid, c->L);
    if (int e = prepare_templates(c, s, threshold, cap)) return e;
    if (int e = enqueue_coarse(c, s, d_out, cap, d_count)) return e;
    return enqueue_local(c, s, d_out, cap, d_count);
}

// The whole match() as a DAG, recorded by stream capture on the context's two private streams:
//   main: quantize(0) -> quantize(1) -> ... -> quantize(L-1) -> build_lm(L-1) -> coarse -> [join] -> local(L-2..0)
//   side:          \-> build_lm(0)      \-> build_lm(1) ...                                  /
// The linear memories of the finer levels are only needed by the refinement passes, so their
// construction overlaps the coarse-level chain.
int capture_match_graph(sbm_ctx* c, const uint8_t* d_img0, int stride0, const uint8_t* d_mask0, sbm_match_rec* d_out,
                        int64_t cap, int32_t* d_count, hipGraph_t* graph)
{
    const int ch = c->channels, L = c->L;
    hipStream_t m = c->stream, sd = c->side;
    HIP_TRY(hipStreamBeginCapture(m, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    const uint8_t* img = d_img0;
    int stride = stride0;
    const uint8_t* mask = d_mask0;
    bool forked = false;
    for (int l = 0; l < L && !rc; ++l) {
        if (l > 0) {
            if (mask) {
                const int n = c->rows[l] * c->cols[l];
                hipLaunchKernelGGL(k_resize_mask, dim3(std::min((n + 255) / 256, 4096)), dim3(256), 0, m, mask, c->rows[l - 1],
                                   c->cols[l - 1], c->d_mask[l].as<uint8_t>(), c->rows[l], c->cols[l]);
                mask = c->d_mask[l].as<uint8_t>();
            }
            img = c->d_img[l].as<uint8_t>();
            stride = c->cols[l] * ch;
        }
        rc = launch_quantize(c, m, img, c->rows[l], c->cols[l], stride, ch, mask, c->cfg.weak_threshold, c->d_quant[l].as<uint8_t>(),
                             nullptr, nullptr, l + 1 < L ? c->d_img[l + 1].as<uint8_t>() : nullptr);
        if (rc) break;
        static const bool use_fork = getenv("SBM_GRAPH_FORK") ? atoi(getenv("SBM_GRAPH_FORK")) != 0 : true;
        if (l < L - 1 && use_fork) {
            if (hipEventRecord(c->ev_fork[l], m) != hipSuccess || hipStreamWaitEvent(sd, c->ev_fork[l], 0) != hipSuccess) {
                rc = fail(SBM_ERR_HIP, "graph fork failed");
                break;
            }
            forked = true;
            rc = launch_build_lm(c, sd, c->d_quant[l].as<uint8_t>(), c->rows[l], c->cols[l], c->cfg.T[l], c->d_lm[l].as<uint8_t>(), c->lm_stride[l]);
        } else {
            rc = launch_build_lm(c, m, c->d_quant[l].as<uint8_t>(), c->rows[l], c->cols[l], c->cfg.T[l], c->d_lm[l].as<uint8_t>(), c->lm_stride[l]);
        }
    }
    for (int l = 0; l < L; ++l) { // the captured build is the 8-plane form at every level
        c->lm_full[l] = true;
        c->lm_compact[l] = false;
    }
    if (!rc) rc = enqueue_coarse(c, m, d_out, cap, d_count);
    if (!rc && forked) {
        if (hipEventRecord(c->ev_join, sd) != hipSuccess || hipStreamWaitEvent(m, c->ev_join, 0) != hipSuccess)
            rc = fail(SBM_ERR_HIP, "graph join failed");
    }
    if (!rc) rc = enqueue_local(c, m, d_out, cap, d_count);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(m, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) return fail(SBM_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    *graph = g;
    return 0;
}

// The batched match loop (BASELINE config 5: "hipGraph-captured match loop") as one captured graph: the same five
// launches sbm_match_batch_device enqueues (gradient stage per level, linear memories + counter reset, coarse pass,
// refinement per level), recorded once per distinct argument tuple and replayed with one hipGraphLaunch per batch.
int capture_batch_graph(sbm_ctx* c, const uint8_t* d_imgs, int64_t frame_stride, int frames, int stride0, const uint8_t* d_mask0,
                        sbm_match_rec* d_out, int64_t cap, int32_t* d_counts, hipGraph_t* graph)
{
    hipStream_t m = c->stream;
    HIP_TRY(hipStreamBeginCapture(m, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_pyramid(c, m, d_imgs, stride0, d_mask0, d_counts, frames, frame_stride);
    if (!rc) rc = enqueue_coarse(c, m, d_out, cap, d_counts, frames);
    if (!rc) rc = enqueue_local(c, m, d_out, cap, d_counts, frames);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(m, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) return fail(SBM_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    *graph = g;
    return 0;
}

int fetch_results(sbm_ctx* c, hipStream_t s, sbm_match_rec* out_host, int64_t cap, int64_t* n_out)
{
    int32_t cnt[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(cnt, c->d_outcount.p, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (n_out) *n_out = cnt[0];
    if (cnt[1]) return fail(SBM_ERR_CAPACITY, "coarse candidate list overflowed (max_candidates=%lld)", (long long)c->cand_cap);
    if (cnt[0] > cap || cnt[0] > c->cand_cap)
        return fail(SBM_ERR_CAPACITY, "%d matches exceed the output capacity %lld", cnt[0], (long long)std::min<int64_t>(cap, c->cand_cap));
    if (cnt[0] > 0) {
        HIP_TRY(hipMemcpyAsync(out_host, c->d_out.p, (size_t)cnt[0] * sizeof(sbm_match_rec), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return 0;
}

void collect_timings(sbm_ctx* c)
{
    c->timing_ms.resize(c->timings.size());
    for (size_t i = 0; i < c->timings.size(); ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(c->timings[i].b);
        (void)hipEventElapsedTime(&ms, c->timings[i].a, c->timings[i].b);
        c->timing_ms[i] = ms;
    }
}

bool feature_in_bounds(const sbm_feature& f) { return f.x >= 0 && f.y >= 0 && f.x <= 65535 && f.y <= 65535 && f.label >= 0 && f.label < 8; }

} // namespace

// ---------------------------------------------------------------------------
extern "C" {

const char* sbm_last_error(void) { return g_err.c_str(); }
int sbm_abi_version(void) { return SBM_ABI_VERSION; }

int sbm_create(const sbm_config* cfg, sbm_ctx** out)
{
    if (!cfg || !out) return fail(SBM_ERR_INVALID, "null argument");
    if (cfg->n_levels < 1 || cfg->n_levels > SBM_MAX_LEVELS) return fail(SBM_ERR_INVALID, "n_levels %d out of range", cfg->n_levels);
    for (int l = 0; l < cfg->n_levels; ++l)
        if (cfg->T[l] < 1 || cfg->T[l] > 64) return fail(SBM_ERR_INVALID, "T[%d]=%d out of range", l, cfg->T[l]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SBM_ERR_HIP, "no HIP device available: libsbm_hip has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(SBM_ERR_INVALID, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    HIP_TRY(hipSetDevice(cfg->device_id));
    sbm_ctx* c = new sbm_ctx();
    c->cfg = *cfg;
    c->L = cfg->n_levels;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_simd = 4 * prop.multiProcessorCount;
        else (void)hipGetLastError();
    }
    c->cand_cap = cfg->max_candidates > 0 ? cfg->max_candidates : (int64_t)1 << 20;
    if (c->cand_cap > INT32_MAX / 2) c->cand_cap = INT32_MAX / 2;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(SBM_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        sbm_destroy(c);
        return fail(SBM_ERR_HIP, "stream/event creation failed");
    }
    for (int l = 0; l < SBM_MAX_LEVELS; ++l)
        if (hipEventCreateWithFlags(&c->ev_fork[l], hipEventDisableTiming) != hipSuccess) {
            sbm_destroy(c);
            return fail(SBM_ERR_HIP, "event creation failed");
        }
    int rc = 0;
    if ((rc = c->d_cands.ensure((size_t)c->cand_cap * sizeof(Cand))) || (rc = c->d_counters.ensure(256, true)) ||
        (rc = c->d_out.ensure((size_t)c->cand_cap * sizeof(sbm_match_rec))) || (rc = c->d_outcount.ensure(16, true))) {
        sbm_destroy(c);
        return rc;
    }
    *out = c;
    return 0;
}

void sbm_destroy(sbm_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device_id);
    (void)hipDeviceSynchronize();
    if (c->comm && g_rccl_destroy_hook) g_rccl_destroy_hook(c);
    for (auto& pr : c->pinned) (void)hipHostUnregister((void*)pr.p); // the caller never unpinned them: still registered by us
    c->pinned.clear();
    for (int i = 0; i < 2; ++i) {
        if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]);
        if (c->ev_free[i]) (void)hipEventDestroy(c->ev_free[i]);
        c->d_in[i].release();
    }
    c->d_bout.release();
    if (c->h_bout) (void)hipHostFree(c->h_bout);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_res_count) (void)hipHostFree(c->h_res_count);
    c->drop_graphs();
    for (int l = 0; l < SBM_MAX_LEVELS; ++l)
        if (c->ev_fork[l]) (void)hipEventDestroy(c->ev_fork[l]);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side) (void)hipStreamDestroy(c->side);
    c->clear_timings();
    DevBuf* singles[] = {&c->d_tls, &c->d_fxy, &c->d_flabel, &c->d_flevel, &c->d_foff, &c->d_class, &c->d_tid, &c->d_active, &c->d_citems, &c->d_cfoff, &c->d_soff, &c->d_soffbase, &c->d_fxy_s, &c->d_flabel_s, &c->d_fcls,
                         &c->d_rawmin, &c->d_rawkeep, &c->d_geo, &c->d_cands, &c->d_counters, &c->d_out, &c->d_outcount,
                         &c->d_scratch};
    for (DevBuf* b : singles) b->release();
    for (int l = 0; l < SBM_MAX_LEVELS; ++l) {
        c->d_img[l].release();
        c->d_mask[l].release();
        c->d_quant[l].release();
        c->d_lm[l].release();
        c->d_lmc[l].release();
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int sbm_upload_templates(sbm_ctx* c, int32_t n_templates, const sbm_template_level* levels, const sbm_feature* features,
                         int64_t n_features, const int32_t* class_idx, const int32_t* template_id)
{
    if (!c || n_templates < 0 || (n_templates && !levels) || n_features < 0 || (n_features && !features))
        return fail(SBM_ERR_INVALID, "bad template arguments");
    if (n_features >= (int64_t)INT32_MAX) return fail(SBM_ERR_INVALID, "too many features");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    const int L = c->L;
    std::vector<DevTL> tls((size_t)n_templates * L);
    std::vector<uint32_t> fxy((size_t)n_features, 0);
    std::vector<uint8_t> flabel((size_t)n_features, 0), flevel((size_t)n_features, 0);
    for (int t = 0; t < n_templates; ++t)
        for (int l = 0; l < L; ++l) {
            const sbm_template_level& s = levels[(size_t)t * L + l];
            if (s.n_features < 0 || s.n_features > SBM_MAX_FEATURES) // CV_Error line2Dup.cpp:1195, :1260
                return fail(SBM_ERR_INVALID, "template %d level %d: feature size too large (%d >= 8192)", t, l, s.n_features);
            if (s.feature_offset < 0 || s.feature_offset + s.n_features > n_features)
                return fail(SBM_ERR_INVALID, "template %d level %d: feature range out of bounds", t, l);
            for (int i = 0; i < s.n_features; ++i) {
                const sbm_feature& f = features[s.feature_offset + i];
                if (!feature_in_bounds(f)) // CV_DbgAssert(f.x >= 0 && f.y >= 0), line2Dup.cpp:788-789
                    return fail(SBM_ERR_INVALID, "template %d level %d feature %d: (%d,%d,label %d) outside the supported range", t, l, i, f.x, f.y, f.label);
                fxy[s.feature_offset + i] = (uint32_t)f.x | ((uint32_t)f.y << 16);
                flabel[s.feature_offset + i] = (uint8_t)f.label;
                flevel[s.feature_offset + i] = (uint8_t)l;
            }
            DevTL d;
            d.width = s.width;
            d.height = s.height;
            d.nf = s.n_features;
            d.feat_off = (int32_t)s.feature_offset;
            tls[(size_t)t * L + l] = d;
        }
    // class-sorted copies for the refinement pass on the strip-interleaved plane (accumulate_rows16_q): inside a
    // template level the features are ordered by (x / T) & 15 (stable), fcls holds the 17 class offsets.  A sum does not
    // care about the order of its terms; every other kernel keeps the caller's order.
    std::vector<uint32_t> fxy_s(fxy);
    std::vector<uint8_t> flabel_s(flabel);
    std::vector<uint16_t> fcls((size_t)n_templates * L * 17, 0);
    for (int t = 0; t < n_templates; ++t)
        for (int l = 0; l < L; ++l) {
            const DevTL& d = tls[(size_t)t * L + l];
            const int T = c->cfg.T[l];
            const int log2t = T == 4 ? 2 : (T == 8 ? 3 : -1);
            uint16_t* cl = &fcls[((size_t)t * L + l) * 17];
            if (log2t < 0) { // never read (the strip plane exists for T = 4 and 8 only)
                for (int k = 1; k <= 16; ++k) cl[k] = (uint16_t)d.nf;
                continue;
            }
            int cnt[17] = {0};
            for (int i = 0; i < d.nf; ++i) ++cnt[(((fxy[d.feat_off + i] & 0xffff) >> log2t) & 15) + 1];
            for (int k = 0; k < 16; ++k) cnt[k + 1] += cnt[k];
            for (int k = 0; k <= 16; ++k) cl[k] = (uint16_t)cnt[k];
            int pos[16];
            for (int k = 0; k < 16; ++k) pos[k] = cnt[k];
            for (int i = 0; i < d.nf; ++i) {
                const int k = ((fxy[d.feat_off + i] & 0xffff) >> log2t) & 15;
                fxy_s[d.feat_off + pos[k]] = fxy[d.feat_off + i];
                flabel_s[d.feat_off + pos[k]] = flabel[d.feat_off + i];
                ++pos[k];
            }
        }
    std::vector<int32_t> cls(n_templates), tid(n_templates);
    for (int t = 0; t < n_templates; ++t) {
        cls[t] = class_idx ? class_idx[t] : 0;
        tid[t] = template_id ? template_id[t] : t;
    }
    HIP_TRY(hipDeviceSynchronize()); // the buffers below may be re-allocated while frames are in flight
    int rc = 0;
    if ((rc = c->d_tls.ensure(tls.size() * sizeof(DevTL))) || (rc = c->d_fxy.ensure(fxy.size() * 4)) ||
        (rc = c->d_flabel.ensure(flabel.size())) || (rc = c->d_flevel.ensure(flevel.size())) ||
        (rc = c->d_foff.ensure(fxy.size() * 4)) || (rc = c->d_class.ensure(cls.size() * 4)) ||
        (rc = c->d_tid.ensure(tid.size() * 4)) || (rc = c->d_fxy_s.ensure(fxy_s.size() * 4)) ||
        (rc = c->d_flabel_s.ensure(flabel_s.size())) || (rc = c->d_fcls.ensure(std::max<size_t>(fcls.size(), 1) * 2)))
        return rc;
    HIP_TRY(hipDeviceSynchronize()); // frames still in flight on the caller's streams read the old tables
    if (!tls.empty()) HIP_TRY(hipMemcpy(c->d_tls.p, tls.data(), tls.size() * sizeof(DevTL), hipMemcpyHostToDevice));
    if (!fxy.empty()) {
        HIP_TRY(hipMemcpy(c->d_fxy.p, fxy.data(), fxy.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_flabel.p, flabel.data(), flabel.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_flevel.p, flevel.data(), flevel.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_fxy_s.p, fxy_s.data(), fxy_s.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_flabel_s.p, flabel_s.data(), flabel_s.size(), hipMemcpyHostToDevice));
    }
    if (!fcls.empty()) HIP_TRY(hipMemcpy(c->d_fcls.p, fcls.data(), fcls.size() * 2, hipMemcpyHostToDevice));
    if (n_templates) {
        HIP_TRY(hipMemcpy(c->d_class.p, cls.data(), cls.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_tid.p, tid.data(), tid.size() * 4, hipMemcpyHostToDevice));
    }
    c->n_templates = n_templates;
    c->n_features = n_features;
    c->h_tls.swap(tls);
    c->h_fxy.swap(fxy);
    c->h_class.swap(cls);
    c->h_tid.swap(tid);
    c->have_thr = false;
    c->foff_dirty = true;
    c->drop_graphs();
    return sbm_select_classes(c, nullptr, 0);
}

static int set_active(sbm_ctx* c, std::vector<int32_t>& act)
{
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize()); // frames in flight read the current selection
    if (int e = c->d_active.ensure(std::max<size_t>(act.size(), 1) * 4)) return e;
    if (!act.empty()) HIP_TRY(hipMemcpy(c->d_active.p, act.data(), act.size() * 4, hipMemcpyHostToDevice));
    c->h_active.swap(act);
    c->citems_dirty = true;
    c->drop_graphs(); // grid sizes depend on the active set
    return 0;
}

int sbm_select_classes(sbm_ctx* c, const int32_t* class_idx, int32_t n)
{
    if (!c || n < 0 || (n && !class_idx)) return fail(SBM_ERR_INVALID, "bad class selection");
    std::vector<int32_t> act;
    if (n == 0) {
        act.resize(c->n_templates);
        for (int t = 0; t < c->n_templates; ++t) act[t] = t;
    } else {
        // class_ids order, then template order: the order matchClass is called in (line2Dup.cpp:1134-1139)
        for (int i = 0; i < n; ++i)
            for (int t = 0; t < c->n_templates; ++t)
                if (c->h_class[t] == class_idx[i]) act.push_back(t);
    }
    return set_active(c, act);
}

int sbm_select_range(sbm_ctx* c, int32_t first, int32_t count)
{
    if (!c || first < 0 || count < 0 || first + count > c->n_templates) return fail(SBM_ERR_INVALID, "bad template range");
    std::vector<int32_t> act(count);
    for (int i = 0; i < count; ++i) act[i] = first + i;
    return set_active(c, act);
}

int sbm_select_templates(sbm_ctx* c, const int32_t* idx, int32_t n)
{
    if (!c || n < 0 || (n && !idx)) return fail(SBM_ERR_INVALID, "bad template list");
    std::vector<int32_t> act(idx, idx + n);
    for (int32_t t : act)
        if (t < 0 || t >= c->n_templates) return fail(SBM_ERR_INVALID, "template index %d out of range", t);
    return set_active(c, act);
}

int sbm_partition_templates(sbm_ctx* c, int32_t rows, int32_t cols, const int32_t* idx, int32_t n, int32_t n_shards, int32_t* first,
                            int32_t* count)
{
    if (!c || n_shards < 1 || !first || !count || n < 0 || rows <= 0 || cols <= 0) return fail(SBM_ERR_INVALID, "bad partition arguments");
    // the template list to divide: idx[0..n) if given, else every uploaded template in upload order
    std::vector<int32_t> list;
    if (idx) list.assign(idx, idx + n);
    else {
        list.resize(c->n_templates);
        for (int t = 0; t < c->n_templates; ++t) list[t] = t;
    }
    const int L = c->L, lc = L - 1, T = c->cfg.T[lc];
    const int rl = rows >> lc, cl = cols >> lc, W = cl / T, H = rl / T;
    // work of a template = byte-adds of its coarse pass (in-bounds features x template_positions, line2Dup.cpp:818-837);
    // at least 1 so that templates without work still spread out
    std::vector<double> cum(list.size() + 1, 0.0);
    for (size_t i = 0; i < list.size(); ++i) {
        const int32_t t = list[i];
        if (t < 0 || t >= c->n_templates) return fail(SBM_ERR_INVALID, "template index %d out of range", t);
        const DevTL& tl = c->h_tls[(size_t)t * L + lc];
        const int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
        const int npos = (H - hf) * W + (W - wf) + 1;
        int64_t inb = 0;
        if (npos > 0)
            for (int k = 0; k < tl.nf; ++k) {
                const uint32_t xy = c->h_fxy[tl.feat_off + k];
                if ((int)(xy & 0xffff) < cl && (int)(xy >> 16) < rl) ++inb;
            }
        cum[i + 1] = cum[i] + std::max<double>(1.0, (double)inb * std::max(npos, 0));
    }
    int prev = 0;
    for (int sh = 0; sh < n_shards; ++sh) {
        int end = (int)list.size();
        if (sh + 1 < n_shards) {
            const double target = cum.back() * (sh + 1) / n_shards;
            end = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
            end = std::min(std::max(end, prev), (int)list.size());
        }
        first[sh] = prev;
        count[sh] = end - prev;
        prev = end;
    }
    return 0;
}

int sbm_match_device(sbm_ctx* c, const void* d_img, int32_t rows, int32_t cols, int32_t stride, int32_t channels,
                     const void* d_mask, float threshold, void* d_out, int64_t cap, void* d_count, void* stream)
{
    if (!c || !d_img || !d_out || !d_count) return fail(SBM_ERR_INVALID, "null argument");
    if (stride < cols * channels) return fail(SBM_ERR_INVALID, "stride %d < cols*channels", stride);
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // anything the launches below are about to change may still be read by frames in flight
    const bool dirty = !(c->channels == channels && c->rows[0] == rows && c->cols[0] == cols && c->levels_valid == c->L) ||
                       !c->have_thr || memcmp(&threshold, &c->thr_cached, 4) != 0 || c->foff_dirty;
    if (dirty) HIP_TRY(hipDeviceSynchronize());
    if (!c->graph_mode || c->profiling) {
        if (int e = ensure_geometry(c, rows, cols, channels)) return e;
        if (c->profiling && !c->profiling_keep) c->clear_timings();
        if (int e = prepare_templates(c, s, threshold, cap)) return e;
        if (int e = enqueue_pyramid(c, s, (const uint8_t*)d_img, stride, (const uint8_t*)d_mask, (int32_t*)d_count)) return e;
        if (int e = enqueue_coarse(c, s, (sbm_match_rec*)d_out, cap, (int32_t*)d_count)) return e;
        return enqueue_local(c, s, (sbm_match_rec*)d_out, cap, (int32_t*)d_count);
    }
    // graph path: all state changes happen up front (they may synchronise), then one hipGraphLaunch
    uint32_t thr_bits;
    memcpy(&thr_bits, &threshold, 4);
    if (int e = ensure_geometry(c, rows, cols, channels)) return e;
    if (int e = prepare_templates(c, c->stream, threshold, cap)) return e;
    sbm_ctx::GraphEntry* hit = nullptr;
    for (auto& g : c->graphs)
        if (g.img == d_img && g.rows == rows && g.cols == cols && g.stride == stride && g.ch == channels && g.mask == d_mask &&
            g.thr_bits == thr_bits && g.out == d_out && g.cap == cap && g.count == d_count && g.mo == (void*)c->mirror_out &&
            g.mc == (void*)c->mirror_count && g.frames == 0)
            hit = &g;
    if (!hit) {
        if (c->graphs.size() >= 8) { // evict the least recently used capture
            size_t lru = 0;
            for (size_t i = 1; i < c->graphs.size(); ++i)
                if (c->graphs[i].last_use < c->graphs[lru].last_use) lru = i;
            HIP_TRY(hipDeviceSynchronize());
            (void)hipGraphExecDestroy(c->graphs[lru].exec);
            (void)hipGraphDestroy(c->graphs[lru].graph);
            c->graphs.erase(c->graphs.begin() + lru);
        }
        sbm_ctx::GraphEntry ge{d_img, rows, cols, stride, channels, d_mask, thr_bits, d_out, cap, d_count,
                               (void*)c->mirror_out, (void*)c->mirror_count, 0, 0, nullptr, nullptr, 0};
        if (int e = capture_match_graph(c, (const uint8_t*)d_img, stride, (const uint8_t*)d_mask, (sbm_match_rec*)d_out, cap,
                                        (int32_t*)d_count, &ge.graph))
            return e;
        hipError_t he = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
        if (he != hipSuccess) {
            (void)hipGraphDestroy(ge.graph);
            return fail(SBM_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(he));
        }
        c->graphs.push_back(ge);
        hit = &c->graphs.back();
    }
    hit->last_use = ++c->graph_clock;
    c->levels_valid = c->L;
    for (int l = 0; l < c->L; ++l) { // the captured build writes the 8-plane form at every level
        c->lm_full[l] = true;
        c->lm_compact[l] = false;
    }
    HIP_TRY(hipGraphLaunch(hit->exec, s));
    return 0;
}

int sbm_match_batch_device(sbm_ctx* c, const void* d_imgs, int64_t frame_stride, int32_t n_frames, int32_t rows, int32_t cols,
                           int32_t stride, int32_t channels, const void* d_mask, float threshold, void* d_out, int64_t cap,
                           void* d_counts, void* stream)
{
    if (!c || !d_imgs || !d_out || !d_counts) return fail(SBM_ERR_INVALID, "null argument");
    if (n_frames < 1) return fail(SBM_ERR_INVALID, "n_frames must be >= 1");
    if (stride < cols * channels) return fail(SBM_ERR_INVALID, "stride %d < cols*channels", stride);
    if (n_frames > 1 && frame_stride < (int64_t)stride * rows) return fail(SBM_ERR_INVALID, "frame_stride smaller than one frame");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const bool dirty = !(c->channels == channels && c->rows[0] == rows && c->cols[0] == cols && c->levels_valid == c->L &&
                         n_frames <= c->batch) ||
                       !c->have_thr || memcmp(&threshold, &c->thr_cached, 4) != 0 || c->foff_dirty;
    if (dirty) HIP_TRY(hipDeviceSynchronize());
    if (int e = ensure_geometry(c, rows, cols, channels, n_frames)) return e;
    if (c->profiling && !c->profiling_keep) c->clear_timings();
    if (int e = prepare_templates(c, s, threshold, cap)) return e;
    if (c->graph_mode && !c->profiling && n_frames > 1) {
        // graph path: every state change happened above (they may synchronise); now one hipGraphLaunch per batch
        uint32_t thr_bits;
        memcpy(&thr_bits, &threshold, 4);
        sbm_ctx::GraphEntry* hit = nullptr;
        for (auto& g : c->graphs)
            if (g.img == d_imgs && g.rows == rows && g.cols == cols && g.stride == stride && g.ch == channels && g.mask == d_mask &&
                g.thr_bits == thr_bits && g.out == d_out && g.cap == cap && g.count == d_counts && g.mo == (void*)c->mirror_out &&
                g.mc == (void*)c->mirror_count && g.frames == n_frames && g.frame_stride == frame_stride)
                hit = &g;
        if (!hit) {
            if (c->graphs.size() >= 8) { // evict the least recently used capture
                size_t lru = 0;
                for (size_t i = 1; i < c->graphs.size(); ++i)
                    if (c->graphs[i].last_use < c->graphs[lru].last_use) lru = i;
                HIP_TRY(hipDeviceSynchronize());
                (void)hipGraphExecDestroy(c->graphs[lru].exec);
                (void)hipGraphDestroy(c->graphs[lru].graph);
                c->graphs.erase(c->graphs.begin() + lru);
            }
            sbm_ctx::GraphEntry ge{d_imgs, rows, cols, stride, channels, d_mask, thr_bits, d_out, cap, d_counts,
                                   (void*)c->mirror_out, (void*)c->mirror_count, n_frames, frame_stride, nullptr, nullptr, 0};
            if (int e = capture_batch_graph(c, (const uint8_t*)d_imgs, frame_stride, n_frames, stride, (const uint8_t*)d_mask,
                                            (sbm_match_rec*)d_out, cap, (int32_t*)d_counts, &ge.graph))
                return e;
            hipError_t he = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
            if (he != hipSuccess) {
                (void)hipGraphDestroy(ge.graph);
                return fail(SBM_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(he));
            }
            c->graphs.push_back(ge);
            hit = &c->graphs.back();
        }
        hit->last_use = ++c->graph_clock;
        c->levels_valid = c->L;
        for (int l = 0; l < c->L; ++l) { // what the replay leaves resident (as enqueue_pyramid records it)
            bool compact, strip;
            lm_form(c, l, &compact, &strip);
            c->lm_compact[l] = compact;
            c->lm_full[l] = !compact;
            c->lm_strip[l] = strip;
        }
        HIP_TRY(hipGraphLaunch(hit->exec, s));
        return 0;
    }
    if (int e = enqueue_pyramid(c, s, (const uint8_t*)d_imgs, stride, (const uint8_t*)d_mask, (int32_t*)d_counts, n_frames, frame_stride)) return e;
    if (int e = enqueue_coarse(c, s, (sbm_match_rec*)d_out, cap, (int32_t*)d_counts, n_frames)) return e;
    return enqueue_local(c, s, (sbm_match_rec*)d_out, cap, (int32_t*)d_counts, n_frames);
}

int sbm_match_templates_device(sbm_ctx* c, float threshold, void* d_out, int64_t cap, void* d_count, void* stream)
{
    if (!c || !d_out || !d_count) return fail(SBM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (!c->have_thr || memcmp(&threshold, &c->thr_cached, 4) != 0 || c->foff_dirty) HIP_TRY(hipDeviceSynchronize());
    if (c->profiling && !c->profiling_keep) c->clear_timings();
    return enqueue_templates(c, s, threshold, (sbm_match_rec*)d_out, cap, (int32_t*)d_count);
}

int sbm_set_quantize_mode(sbm_ctx* c, int32_t mode, int32_t rows_per_wave)
{
    if (!c || mode < 0 || mode > 2 || rows_per_wave < 0 || rows_per_wave > 4096) return fail(SBM_ERR_INVALID, "bad quantize mode");
    c->quantize_mode = mode;
    c->quantize_hs = rows_per_wave;
    c->drop_graphs(); // captured launches hold the old kernel choice
    return 0;
}

int sbm_set_pipeline_depth(sbm_ctx* c, int32_t batches_in_flight)
{
    if (!c || batches_in_flight < 1) return fail(SBM_ERR_INVALID, "bad pipeline depth");
    c->pipeline_depth = batches_in_flight;
    c->drop_graphs(); // captured launches hold the old launch geometry
    return 0;
}

int sbm_set_graph_mode(sbm_ctx* c, int32_t enabled)
{
    if (!c) return fail(SBM_ERR_INVALID, "null context");
    c->graph_mode = enabled != 0;
    if (!c->graph_mode) {
        (void)hipDeviceSynchronize();
        c->drop_graphs();
    }
    return 0;
}

static int upload_image(sbm_ctx* c, const uint8_t* img, int rows, int cols, int stride, int ch, const uint8_t* mask)
{
    if (!img) return fail(SBM_ERR_INVALID, "null image");
    if (stride < cols * ch) return fail(SBM_ERR_INVALID, "stride %d < cols*channels", stride);
    if (int e = ensure_geometry(c, rows, cols, ch)) return e;
    // No implicit pinning: whether the copy below is a direct DMA (the frame lies in memory the caller pinned with
    // sbm_pin_host_buffer / hipHostMalloc / hipHostRegister) or a staged pageable copy is decided by the runtime from
    // the pointer's current attributes, never from a cache of addresses seen before.
    HIP_TRY(hipMemcpy2DAsync(c->d_img[0].p, (size_t)cols * ch, img, stride, (size_t)cols * ch, rows, hipMemcpyHostToDevice, c->stream));
    if (mask) HIP_TRY(hipMemcpyAsync(c->d_mask[0].p, mask, (size_t)rows * cols, hipMemcpyHostToDevice, c->stream));
    return 0;
}

int sbm_match_batch_host_begin(sbm_ctx* c, const uint8_t* const* frames, int32_t n_frames, int32_t rows, int32_t cols, int32_t stride,
                               int32_t channels, const uint8_t* mask, float threshold, int64_t cap, int32_t sub_batch)
{
    if (!c || !frames || n_frames < 1 || cap < 1) return fail(SBM_ERR_INVALID, "bad batch arguments");
    if (c->pending.active) return fail(SBM_ERR_STATE, "a host batch is already in flight (call sbm_match_batch_host_end)");
    if (stride < cols * channels) return fail(SBM_ERR_INVALID, "stride %d < cols*channels", stride);
    for (int f = 0; f < n_frames; ++f)
        if (!frames[f]) return fail(SBM_ERR_INVALID, "frame %d is null", f);
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    int sub = std::max(1, std::min(sub_batch > 0 ? sub_batch : 8, n_frames));
    HIP_TRY(hipDeviceSynchronize()); // geometry / template state may change below
    if (int e = ensure_geometry(c, rows, cols, channels, sub)) return e;
    // geometries the one-launch linear-memory builder does not take (level widths that are not multiples of 16, other
    // strides): one frame per "sub-batch" through the generic kernels -- the uploads still overlap the kernels
    for (int l = 0; l < c->L; ++l)
        if (!lm_rows_ok(c->d_quant[l].as<uint8_t>(), c->cols[l], c->cfg.T[l])) sub = 1;
    if (!c->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_free[i], hipEventDisableTiming));
        }
    }
    const size_t frame_bytes = (size_t)rows * cols * channels;
    for (int i = 0; i < 2; ++i)
        if (int e = c->d_in[i].ensure((size_t)sub * frame_bytes)) return e;
    // results: n_frames blocks of cap records, then n_frames {n_matches, overflow} pairs -- on the device and in pinned memory
    const size_t rec_bytes = (size_t)n_frames * (size_t)cap * sizeof(sbm_match_rec), total = rec_bytes + (size_t)n_frames * 8;
    if (int e = c->d_bout.ensure(total)) return e;
    if (total > c->h_bout_bytes) {
        if (c->h_bout) (void)hipHostFree(c->h_bout);
        c->h_bout = nullptr;
        c->h_bout_bytes = 0;
        HIP_TRY(hipHostMalloc((void**)&c->h_bout, total, hipHostMallocDefault));
        c->h_bout_bytes = total;
    }
    int32_t* h_counts = (int32_t*)(c->h_bout + rec_bytes);
    for (int f = 0; f < n_frames; ++f) h_counts[2 * f] = -1, h_counts[2 * f + 1] = 0;
    if (c->profiling && !c->profiling_keep) c->clear_timings();
    if (int e = prepare_templates(c, c->stream, threshold, cap)) return e;
    if (mask) HIP_TRY(hipMemcpyAsync(c->d_mask[0].p, mask, (size_t)rows * cols, hipMemcpyHostToDevice, c->stream));
    sbm_match_rec* const user_mo = c->mirror_out;
    int32_t* const user_mc = c->mirror_count;
    int rc = 0;
    for (int f0 = 0, k = 0; f0 < n_frames && !rc; f0 += sub, ++k) {
        const int nf = std::min(sub, n_frames - f0), buf = k & 1;
        // the kernels of sub-batch k-2 must have finished reading this input buffer
        if (k >= 2 && hipStreamWaitEvent(c->copy_stream, c->ev_free[buf], 0) != hipSuccess) rc = fail(SBM_ERR_HIP, "stream wait failed");
        for (int f = 0; f < nf && !rc; ++f) {
            uint8_t* dst = c->d_in[buf].as<uint8_t>() + (size_t)f * frame_bytes;
            // one DMA per frame when the caller's memory is pinned (sbm_pin_host_buffer / hipHostMalloc); staged by the
            // runtime otherwise -- either way the kernels of the previous sub-batch keep the GPU busy meanwhile
            hipError_t he = stride == cols * channels
                                ? hipMemcpyAsync(dst, frames[f0 + f], frame_bytes, hipMemcpyHostToDevice, c->copy_stream)
                                : hipMemcpy2DAsync(dst, (size_t)cols * channels, frames[f0 + f], stride, (size_t)cols * channels, rows,
                                                   hipMemcpyHostToDevice, c->copy_stream);
            if (he != hipSuccess) rc = fail(SBM_ERR_HIP, "frame upload failed: %s", hipGetErrorString(he));
        }
        if (rc) break;
        if (hipEventRecord(c->ev_up[buf], c->copy_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_up[buf], 0) != hipSuccess) {
            rc = fail(SBM_ERR_HIP, "event record / wait failed");
            break;
        }
        sbm_match_rec* d_out = c->d_bout.as<sbm_match_rec>() + (size_t)f0 * cap;
        int32_t* d_cnt = (int32_t*)((char*)c->d_bout.p + rec_bytes) + 2 * f0;
        c->mirror_out = (sbm_match_rec*)c->h_bout + (size_t)f0 * cap;
        c->mirror_count = h_counts + 2 * f0;
        rc = enqueue_pyramid(c, c->stream, c->d_in[buf].as<uint8_t>(), cols * channels, mask ? c->d_mask[0].as<uint8_t>() : nullptr, d_cnt, nf,
                             (int64_t)frame_bytes);
        if (!rc) rc = enqueue_coarse(c, c->stream, d_out, cap, d_cnt, nf);
        if (!rc) rc = enqueue_local(c, c->stream, d_out, cap, d_cnt, nf);
        if (!rc && hipEventRecord(c->ev_free[buf], c->stream) != hipSuccess) rc = fail(SBM_ERR_HIP, "event record failed");
    }
    c->mirror_out = user_mo;
    c->mirror_count = user_mc;
    if (rc) {
        (void)hipDeviceSynchronize();
        return rc;
    }
    c->pending.active = true;
    c->pending.n_frames = n_frames;
    c->pending.cap = cap;
    return 0;
}

int sbm_match_batch_host_end(sbm_ctx* c, sbm_match_rec* out, int32_t* counts)
{
    if (!c || !out || !counts) return fail(SBM_ERR_INVALID, "null argument");
    if (!c->pending.active) return fail(SBM_ERR_STATE, "no host batch in flight");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    c->pending.active = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int n = c->pending.n_frames;
    const int64_t cap = c->pending.cap;
    const int32_t* h_counts = (const int32_t*)(c->h_bout + (size_t)n * (size_t)cap * sizeof(sbm_match_rec));
    int bad = -1;
    for (int f = 0; f < n; ++f) {
        counts[2 * f] = h_counts[2 * f];
        counts[2 * f + 1] = h_counts[2 * f + 1];
        const int64_t k = std::min<int64_t>(std::max(h_counts[2 * f], 0), cap);
        if (k > 0) memcpy(out + (size_t)f * cap, (const sbm_match_rec*)c->h_bout + (size_t)f * cap, (size_t)k * sizeof(sbm_match_rec));
        if ((h_counts[2 * f] < 0 || h_counts[2 * f] > cap || h_counts[2 * f + 1]) && bad < 0) bad = f;
    }
    if (c->profiling) collect_timings(c);
    if (bad >= 0)
        return fail(SBM_ERR_CAPACITY, "frame %d: %d matches (overflow flag %d) exceed the per-frame capacity %lld", bad, h_counts[2 * bad],
                    h_counts[2 * bad + 1], (long long)cap);
    return 0;
}

int sbm_match_batch_host(sbm_ctx* c, const uint8_t* const* frames, int32_t n_frames, int32_t rows, int32_t cols, int32_t stride,
                         int32_t channels, const uint8_t* mask, float threshold, sbm_match_rec* out, int64_t cap, int32_t* counts,
                         int32_t sub_batch)
{
    if (!out || !counts) return fail(SBM_ERR_INVALID, "null argument");
    if (int e = sbm_match_batch_host_begin(c, frames, n_frames, rows, cols, stride, channels, mask, threshold, cap, sub_batch)) return e;
    return sbm_match_batch_host_end(c, out, counts);
}

int sbm_pin_host_buffer(sbm_ctx* c, const void* p, int64_t bytes)
{
    if (!c || !p || bytes <= 0) return fail(SBM_ERR_INVALID, "bad buffer");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    for (auto& pr : c->pinned)
        if (pr.p == (const uint8_t*)p) return fail(SBM_ERR_STATE, "buffer %p is already pinned by this context (unpin it first)", p);
    hipError_t e = hipHostRegister((void*)p, (size_t)bytes, hipHostRegisterPortable); // every device of the process may DMA from it
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(SBM_ERR_HIP, "hipHostRegister(%p, %lld) failed: %s", p, (long long)bytes, hipGetErrorString(e));
    }
    c->pinned.push_back({(const uint8_t*)p, (size_t)bytes});
    return 0;
}

int sbm_unpin_host_buffer(sbm_ctx* c, const void* p)
{
    if (!c || !p) return fail(SBM_ERR_INVALID, "bad buffer");
    for (size_t i = 0; i < c->pinned.size(); ++i)
        if (c->pinned[i].p == (const uint8_t*)p) {
            HIP_TRY(hipSetDevice(c->cfg.device_id));
            HIP_TRY(hipStreamSynchronize(c->stream)); // an upload from it may still be in flight
            hipError_t e = hipHostUnregister((void*)p);
            c->pinned.erase(c->pinned.begin() + i);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(SBM_ERR_HIP, "hipHostUnregister(%p) failed: %s", p, hipGetErrorString(e));
            }
            return 0;
        }
    return fail(SBM_ERR_STATE, "buffer %p was not pinned by this context", p);
}

int sbm_match(sbm_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t channels,
              const uint8_t* mask, float threshold, sbm_match_rec* out, int64_t cap, int64_t* n_out)
{
    if (!c || (!out && cap > 0) || !n_out) return fail(SBM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    if (c->profiling && !c->profiling_keep) c->clear_timings();
    if (int e = upload_image(c, img, rows, cols, stride, channels, mask)) return e;
    if (int e = prepare_templates(c, c->stream, threshold, c->cand_cap)) return e;
    if (int e = enqueue_pyramid(c, c->stream, c->d_img[0].as<uint8_t>(), cols * channels, mask ? c->d_mask[0].as<uint8_t>() : nullptr,
                                c->d_outcount.as<int32_t>()))
        return e;
    // results: the emitting kernel also stores every record and the final {count, overflow} pair into pinned host
    // memory, so one stream synchronisation ends the call; lists longer than the pinned buffer take the copy path
    sbm_match_rec* const user_mo = c->mirror_out;
    int32_t* const user_mc = c->mirror_count;
    bool own_mirror = false;
    if (!user_mo) {
        if (!c->h_res) {
            c->h_res_cap = 4096;
            if (hipHostMalloc((void**)&c->h_res, (size_t)c->h_res_cap * sizeof(sbm_match_rec), hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void**)&c->h_res_count, 16, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                if (c->h_res) (void)hipHostFree(c->h_res);
                c->h_res = nullptr;
                c->h_res_count = nullptr;
            }
        }
        if (c->h_res && c->h_res_count) {
            c->h_res_count[0] = -1;
            c->h_res_count[1] = 0;
            c->mirror_out = c->h_res;
            c->mirror_count = c->h_res_count;
            own_mirror = true;
        }
    }
    // with the pinned mirror the emitting kernel's record capacity is the pinned buffer's (it bounds both copies)
    const int64_t dev_cap = own_mirror ? std::min<int64_t>(c->cand_cap, c->h_res_cap) : c->cand_cap;
    int rc = enqueue_coarse(c, c->stream, c->d_out.as<sbm_match_rec>(), dev_cap, c->d_outcount.as<int32_t>());
    if (!rc) rc = enqueue_local(c, c->stream, c->d_out.as<sbm_match_rec>(), dev_cap, c->d_outcount.as<int32_t>());
    c->mirror_out = user_mo;
    c->mirror_count = user_mc;
    if (rc) return rc;
    if (own_mirror) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int n = c->h_res_count[0];
        if (n >= 0 && c->h_res_count[1] == 0 && n <= dev_cap) {
            *n_out = n;
            if (n > cap) return fail(SBM_ERR_CAPACITY, "%d matches exceed the output capacity %lld", n, (long long)cap);
            if (n > 0) memcpy(out, c->h_res, (size_t)n * sizeof(sbm_match_rec));
            if (c->profiling) collect_timings(c);
            return 0;
        }
        // more records than the pinned buffer holds (or a candidate overflow to report): run the template loop again on
        // the resident pyramid with the full device capacity and take the copy path
        if ((rc = enqueue_coarse(c, c->stream, c->d_out.as<sbm_match_rec>(), c->cand_cap, c->d_outcount.as<int32_t>()))) return rc;
        if ((rc = enqueue_local(c, c->stream, c->d_out.as<sbm_match_rec>(), c->cand_cap, c->d_outcount.as<int32_t>()))) return rc;
    }
    rc = fetch_results(c, c->stream, out, cap, n_out);
    if (c->profiling) collect_timings(c);
    return rc;
}

static int rec_cmp(const sbm_match_rec& x, const sbm_match_rec& y)
{
    if (x.similarity != y.similarity) return x.similarity > y.similarity ? -1 : 1;
    if (x.template_id != y.template_id) return x.template_id < y.template_id ? -1 : 1;
    if (x.class_idx != y.class_idx) return x.class_idx < y.class_idx ? -1 : 1;
    if (x.y != y.y) return x.y < y.y ? -1 : 1;
    if (x.x != y.x) return x.x < y.x ? -1 : 1;
    return 0;
}

int64_t sbm_canonicalize(sbm_match_rec* recs, int64_t n)
{
    if (!recs || n <= 0) return 0;
    std::sort(recs, recs + n, [](const sbm_match_rec& a, const sbm_match_rec& b) { return rec_cmp(a, b) < 0; });
    int64_t k = 1;
    for (int64_t i = 1; i < n; ++i)
        if (rec_cmp(recs[i], recs[k - 1]) != 0) recs[k++] = recs[i];
    return k;
}

int sbm_build_pyramid(sbm_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t channels,
                      const uint8_t* mask)
{
    if (!c) return fail(SBM_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    if (c->profiling && !c->profiling_keep) c->clear_timings();
    if (int e = upload_image(c, img, rows, cols, stride, channels, mask)) return e;
    if (int e = enqueue_pyramid(c, c->stream, c->d_img[0].as<uint8_t>(), cols * channels, mask ? c->d_mask[0].as<uint8_t>() : nullptr)) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->profiling) collect_timings(c);
    return 0;
}

int sbm_set_quantized(sbm_ctx* c, int32_t level, const uint8_t* q, int32_t rows, int32_t cols)
{
    if (!c || !q || level < 0 || level >= c->L) return fail(SBM_ERR_INVALID, "bad level");
    if (level > c->levels_valid) return fail(SBM_ERR_STATE, "levels must be set from 0 upwards");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    if (int e = ensure_level(c, level, rows, cols)) return e;
    if (c->profiling && !c->profiling_keep && level == 0) c->clear_timings();
    HIP_TRY(hipMemcpyAsync(c->d_quant[level].p, q, (size_t)rows * cols, hipMemcpyHostToDevice, c->stream));
    if (int e = launch_build_lm(c, c->stream, c->d_quant[level].as<uint8_t>(), rows, cols, c->cfg.T[level],
                                c->d_lm[level].as<uint8_t>(), c->lm_stride[level]))
        return e;
    c->lm_full[level] = true;
    c->lm_compact[level] = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->levels_valid = std::max(c->levels_valid, level + 1);
    if (c->profiling) collect_timings(c);
    return 0;
}

int sbm_get_quantized(sbm_ctx* c, int32_t level, uint8_t* out)
{
    if (!c || !out || level < 0 || level >= c->levels_valid) return fail(SBM_ERR_STATE, "level not resident");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_quant[level].p, (size_t)c->rows[level] * c->cols[level], hipMemcpyDeviceToHost));
    return 0;
}

int sbm_get_linear_memories(sbm_ctx* c, int32_t level, uint8_t* out, int64_t cap_bytes, int64_t* lm_stride)
{
    if (!c || level < 0 || level >= c->levels_valid) return fail(SBM_ERR_STATE, "level not resident");
    if (lm_stride) *lm_stride = c->lm_stride[level];
    if (!out) return 0;
    const int64_t need = 8 * c->lm_stride[level];
    if (cap_bytes < need) return fail(SBM_ERR_CAPACITY, "need %lld bytes", (long long)need);
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize()); // the pyramid may have been built on the caller's stream
    if (int e = ensure_full_lm(c, level, c->stream)) return e;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_lm[level].p, (size_t)need, hipMemcpyDeviceToHost));
    return 0;
}

int sbm_level_dims(sbm_ctx* c, int32_t level, int32_t* rows, int32_t* cols)
{
    if (!c || level < 0 || level >= c->levels_valid) return fail(SBM_ERR_STATE, "level not resident");
    if (rows) *rows = c->rows[level];
    if (cols) *cols = c->cols[level];
    return 0;
}

int sbm_match_templates(sbm_ctx* c, float threshold, sbm_match_rec* out, int64_t cap, int64_t* n_out)
{
    if (!c || (!out && cap > 0) || !n_out) return fail(SBM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    if (c->profiling && !c->profiling_keep) c->clear_timings();
    if (int e = enqueue_templates(c, c->stream, threshold, c->d_out.as<sbm_match_rec>(), c->cand_cap, c->d_outcount.as<int32_t>())) return e;
    int rc = fetch_results(c, c->stream, out, cap, n_out);
    if (c->profiling) collect_timings(c);
    return rc;
}

// ---- stage entry points -----------------------------------------------------
int sbm_quantized_orientations(sbm_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t ch,
                               float weak, float* magnitude, uint8_t* angle, float* angle_ori)
{
    if (!c || !img || !angle || rows < 3 || cols < 3) return fail(SBM_ERR_INVALID, "bad argument");
    if (ch != 1 && ch != 3) return fail(SBM_ERR_INVALID, "channels must be 1 or 3");
    if (stride < cols * ch) return fail(SBM_ERR_INVALID, "stride too small");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    const size_t npx = (size_t)rows * cols;
    DevBuf d_in, d_q, d_mag, d_ori;
    int rc = 0;
    if ((rc = d_in.ensure(npx * ch)) || (rc = d_q.ensure(npx)) || (magnitude && (rc = d_mag.ensure(npx * 4))) ||
        (angle_ori && (rc = d_ori.ensure(npx * 4))))
        goto done;
    if (hipMemcpy2D(d_in.p, (size_t)cols * ch, img, stride, (size_t)cols * ch, rows, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(SBM_ERR_HIP, "image upload failed");
        goto done;
    }
    if ((rc = launch_quantize(c, c->stream, d_in.as<uint8_t>(), rows, cols, cols * ch, ch, nullptr, weak, d_q.as<uint8_t>(),
                              magnitude ? d_mag.as<float>() : nullptr, angle_ori ? d_ori.as<float>() : nullptr, nullptr)))
        goto done;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(angle, d_q.p, npx, hipMemcpyDeviceToHost) != hipSuccess ||
        (magnitude && hipMemcpy(magnitude, d_mag.p, npx * 4, hipMemcpyDeviceToHost) != hipSuccess) ||
        (angle_ori && hipMemcpy(angle_ori, d_ori.p, npx * 4, hipMemcpyDeviceToHost) != hipSuccess))
        rc = fail(SBM_ERR_HIP, "quantize kernel or download failed: %s", hipGetErrorString(hipGetLastError()));
done:
    d_in.release();
    d_q.release();
    d_mag.release();
    d_ori.release();
    return rc;
}

int sbm_extract_local_maxima(sbm_ctx* c, const float* magnitude, const uint8_t* mask, int32_t rows, int32_t cols, float strong_threshold,
                             int32_t* xy, int64_t cap, int64_t* n_out)
{
    if (!c || !magnitude || !n_out || (!xy && cap > 0) || rows < 1 || cols < 1 || rows > 32767 || cols > 32767) return fail(SBM_ERR_INVALID, "bad argument");
    *n_out = 0;
    if (rows < 5 || cols < 5) return 0; // the scanned region [2, rows-2) x [2, cols-2) is empty
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    const size_t npx = (size_t)rows * cols;
    DevBuf d_mag, d_mask, d_xy, d_cnt;
    std::vector<int32_t> pts;
    int rc = 0;
    int64_t dev_cap = std::max<int64_t>(4096, std::min<int64_t>((int64_t)npx / 8, 1 << 22));
    if ((rc = d_mag.ensure(npx * 4)) || (mask && (rc = d_mask.ensure(npx))) || (rc = d_cnt.ensure(16))) goto done;
    if (hipMemcpy(d_mag.p, magnitude, npx * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (mask && hipMemcpy(d_mask.p, mask, npx, hipMemcpyHostToDevice) != hipSuccess)) {
        rc = fail(SBM_ERR_HIP, "upload failed");
        goto done;
    }
    for (;;) {
        int32_t n = 0;
        if ((rc = d_xy.ensure((size_t)dev_cap * 4))) goto done;
        if (hipMemsetAsync(d_cnt.p, 0, 16, c->stream) != hipSuccess) {
            rc = fail(SBM_ERR_HIP, "memset failed");
            goto done;
        }
        hipLaunchKernelGGL(k_local_maxima5, dim3((unsigned)std::min<size_t>((npx + 255) / 256, 8192)), dim3(256), 0, c->stream, d_mag.as<float>(),
                           mask ? d_mask.as<uint8_t>() : nullptr, rows, cols, strong_threshold * strong_threshold, d_xy.as<int32_t>(),
                           d_cnt.as<int32_t>(), (int)dev_cap);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&n, d_cnt.p, 4, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(SBM_ERR_HIP, "k_local_maxima5 failed: %s", hipGetErrorString(hipGetLastError()));
            goto done;
        }
        if (n > dev_cap) { // more maxima than the buffer holds: once more with room for all of them
            dev_cap = n;
            continue;
        }
        pts.resize((size_t)n);
        if (n && hipMemcpy(pts.data(), d_xy.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SBM_ERR_HIP, "download failed");
        break;
    }
    if (!rc) {
        // row-major order, then the reference's invalidation among equal-score neighbours: a maximum is dropped when an
        // earlier kept one lies within its 5x5 window (sbm_train_kernels.h).  y << 16 | x sorts row-major as an integer.
        std::sort(pts.begin(), pts.end());
        std::vector<int32_t> kept;
        size_t lo = 0; // first kept point that can still be within two rows of the current one
        for (int32_t p : pts) {
            const int y = p >> 16, x = p & 0xffff;
            while (lo < kept.size() && (kept[lo] >> 16) < y - 2) ++lo;
            bool ok = true;
            for (size_t i = lo; i < kept.size() && ok; ++i) ok = std::abs((kept[i] & 0xffff) - x) > 2;
            if (ok) kept.push_back(p);
        }
        *n_out = (int64_t)kept.size();
        if ((int64_t)kept.size() > cap) rc = fail(SBM_ERR_CAPACITY, "%zu local maxima exceed the capacity %lld", kept.size(), (long long)cap);
        else
            for (size_t i = 0; i < kept.size(); ++i) xy[i] = kept[i];
    }
done:
    d_mag.release();
    d_mask.release();
    d_xy.release();
    d_cnt.release();
    return rc;
}

int sbm_orientation_bins(sbm_ctx* c, const int16_t* gx, const int16_t* gy, int64_t n, uint8_t* q16)
{
    if (!c || !gx || !gy || !q16 || n < 0) return fail(SBM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    DevBuf a, b, o;
    int rc = 0;
    if ((rc = a.ensure((size_t)n * 2)) || (rc = b.ensure((size_t)n * 2)) || (rc = o.ensure((size_t)n))) goto done;
    if (hipMemcpy(a.p, gx, (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(b.p, gy, (size_t)n * 2, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(SBM_ERR_HIP, "upload failed");
        goto done;
    }
    hipLaunchKernelGGL(k_orientation_bins, dim3((unsigned)std::min<int64_t>((n + 255) / 256 + 1, 8192)), dim3(256), 0, c->stream,
                       a.as<int16_t>(), b.as<int16_t>(), n, o.as<uint8_t>());
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(q16, o.p, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "orientation_bins failed: %s", hipGetErrorString(hipGetLastError()));
done:
    a.release();
    b.release();
    o.release();
    return rc;
}

int sbm_pyrdown(sbm_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t ch, uint8_t* out)
{
    if (!c || !img || !out || rows < 2 || cols < 2 || ch < 1 || ch > 4) return fail(SBM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    DevBuf d_in, d_out;
    int rc = 0;
    const size_t nout = (size_t)(rows / 2) * (cols / 2) * ch;
    if ((rc = d_in.ensure((size_t)rows * cols * ch)) || (rc = d_out.ensure(nout))) goto done;
    if (hipMemcpy2D(d_in.p, (size_t)cols * ch, img, stride, (size_t)cols * ch, rows, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(SBM_ERR_HIP, "upload failed");
        goto done;
    }
    hipLaunchKernelGGL(k_pyrdown, dim3((unsigned)std::min<size_t>((nout / ch + 255) / 256, 4096)), dim3(256), 0, c->stream,
                       d_in.as<uint8_t>(), rows, cols, ch, cols * ch, d_out.as<uint8_t>());
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, d_out.p, nout, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "pyrdown failed: %s", hipGetErrorString(hipGetLastError()));
done:
    d_in.release();
    d_out.release();
    return rc;
}

int sbm_resize_linear(sbm_ctx* c, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride, int32_t ch, double fx, double fy,
                      uint8_t* out, int64_t cap_bytes, int32_t* out_rows, int32_t* out_cols)
{
    if (!c || !img || rows < 1 || cols < 1 || ch < 1 || ch > 4 || !(fx > 0) || !(fy > 0)) return fail(SBM_ERR_INVALID, "bad argument");
    if (stride < cols * ch) return fail(SBM_ERR_INVALID, "stride too small");
    int dr, dc;
    resize_linear_dims(rows, cols, fx, fy, &dr, &dc);
    if (out_rows) *out_rows = dr;
    if (out_cols) *out_cols = dc;
    if (!out) return 0; // size query
    if (dr < 1 || dc < 1) return fail(SBM_ERR_INVALID, "resize to an empty image");
    const size_t nout = (size_t)dr * dc * ch;
    if ((int64_t)nout > cap_bytes) return fail(SBM_ERR_CAPACITY, "need %zu bytes", nout);
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    std::vector<int32_t> xi, yi;
    std::vector<int16_t> xa, ya;
    resize_linear_table(dc, cols, 1.0 / fx, xi, xa);
    resize_linear_table(dr, rows, 1.0 / fy, yi, ya);
    DevBuf d_in, d_out, d_xi, d_xa, d_yi, d_ya;
    int rc = 0;
    if ((rc = d_in.ensure((size_t)rows * cols * ch)) || (rc = d_out.ensure(nout)) || (rc = d_xi.ensure(xi.size() * 4)) ||
        (rc = d_xa.ensure(xa.size() * 2)) || (rc = d_yi.ensure(yi.size() * 4)) || (rc = d_ya.ensure(ya.size() * 2)))
        goto done;
    if (hipMemcpy2D(d_in.p, (size_t)cols * ch, img, stride, (size_t)cols * ch, rows, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_xi.p, xi.data(), xi.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_xa.p, xa.data(), xa.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_yi.p, yi.data(), yi.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ya.p, ya.data(), ya.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(SBM_ERR_HIP, "upload failed");
        goto done;
    }
    hipLaunchKernelGGL(k_resize_linear_u8, dim3((unsigned)std::min<size_t>((nout + 255) / 256, 4096)), dim3(256), 0, c->stream,
                       d_in.as<uint8_t>(), rows, cols, ch, cols * ch, d_xi.as<int32_t>(), d_xa.as<int16_t>(), d_yi.as<int32_t>(),
                       d_ya.as<int16_t>(), d_out.as<uint8_t>(), dr, dc);
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, d_out.p, nout, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "resize failed: %s", hipGetErrorString(hipGetLastError()));
done:
    d_in.release();
    d_out.release();
    d_xi.release();
    d_xa.release();
    d_yi.release();
    d_ya.release();
    return rc;
}

int sbm_spread(sbm_ctx* c, const uint8_t* src, int32_t rows, int32_t cols, int32_t T, uint8_t* dst)
{
    if (!c || !src || !dst || rows < 1 || cols < 1 || T < 1) return fail(SBM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    const size_t n = (size_t)rows * cols;
    DevBuf a, b;
    int rc = 0;
    if ((rc = a.ensure(n)) || (rc = b.ensure(n))) goto done;
    if (hipMemcpy(a.p, src, n, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(SBM_ERR_HIP, "upload failed");
        goto done;
    }
    hipLaunchKernelGGL(k_spread, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, c->stream,
                       a.as<uint8_t>(), rows, cols, T, b.as<uint8_t>());
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(dst, b.p, n, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "spread failed: %s", hipGetErrorString(hipGetLastError()));
done:
    a.release();
    b.release();
    return rc;
}

int sbm_compute_response_maps(sbm_ctx* c, const uint8_t* spread, int32_t rows, int32_t cols, uint8_t* maps)
{
    if (!c || !spread || !maps || rows < 1 || cols < 1) return fail(SBM_ERR_INVALID, "bad argument");
    if (((int64_t)rows * cols) % 16) return fail(SBM_ERR_INVALID, "rows*cols %% 16 != 0 (line2Dup.cpp:639)");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    const size_t n = (size_t)rows * cols;
    DevBuf a, b;
    int rc = 0;
    if ((rc = a.ensure(n)) || (rc = b.ensure(8 * n))) goto done;
    if (hipMemcpy(a.p, spread, n, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(SBM_ERR_HIP, "upload failed");
        goto done;
    }
    hipLaunchKernelGGL(k_response, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, c->stream,
                       a.as<uint8_t>(), (int64_t)n, b.as<uint8_t>());
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(maps, b.p, 8 * n, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "response failed: %s", hipGetErrorString(hipGetLastError()));
done:
    a.release();
    b.release();
    return rc;
}

int sbm_linearize(sbm_ctx* c, const uint8_t* map, int32_t rows, int32_t cols, int32_t T, uint8_t* lm)
{
    if (!c || !map || !lm || rows < 1 || cols < 1 || T < 1) return fail(SBM_ERR_INVALID, "bad argument");
    if (rows % T || cols % T) return fail(SBM_ERR_INVALID, "rows/cols not a multiple of T (line2Dup.cpp:751-752)");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    const size_t n = (size_t)rows * cols;
    DevBuf a, b;
    int rc = 0;
    if ((rc = a.ensure(n)) || (rc = b.ensure(n))) goto done;
    if (hipMemcpy(a.p, map, n, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(SBM_ERR_HIP, "upload failed");
        goto done;
    }
    hipLaunchKernelGGL(k_linearize, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, c->stream,
                       a.as<uint8_t>(), rows, cols, T, b.as<uint8_t>());
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(lm, b.p, n, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "linearize failed: %s", hipGetErrorString(hipGetLastError()));
done:
    a.release();
    b.release();
    return rc;
}

int sbm_similarity(sbm_ctx* c, int32_t t, uint16_t* dst)
{
    if (!c || !dst || t < 0 || t >= c->n_templates) return fail(SBM_ERR_INVALID, "bad template index");
    if (c->levels_valid < c->L) return fail(SBM_ERR_STATE, "pyramid not built");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    if (int e = ensure_foff(c, c->stream)) return e;
    const int lc = c->L - 1, T = c->cfg.T[lc], W = c->cols[lc] / T, H = c->rows[lc] / T;
    DevBuf d;
    if (int e = d.ensure((size_t)W * H * 2)) return e;
    hipLaunchKernelGGL(k_similarity_map, dim3((W * H + SIM_POS_PER_BLOCK - 1) / SIM_POS_PER_BLOCK), dim3(256), 0, c->stream,
                       c->d_lm[lc].as<uint8_t>(), c->lm_stride[lc], c->rows[lc], c->cols[lc], T, W, H, c->h_tls[(size_t)t * c->L + lc],
                       c->d_fxy.as<uint32_t>(), c->d_foff.as<int32_t>(), d.as<uint16_t>());
    int rc = 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(dst, d.p, (size_t)W * H * 2, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "similarity failed: %s", hipGetErrorString(hipGetLastError()));
    d.release();
    return rc;
}

int sbm_similarity_local(sbm_ctx* c, int32_t level, int32_t t, int32_t cx, int32_t cy, uint16_t* dst)
{
    if (!c || !dst || t < 0 || t >= c->n_templates || level < 0 || level >= c->L) return fail(SBM_ERR_INVALID, "bad argument");
    if (c->levels_valid <= level) return fail(SBM_ERR_STATE, "level not resident");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    if (int e = ensure_foff(c, c->stream)) return e;
    HIP_TRY(hipDeviceSynchronize());
    if (int e = ensure_full_lm(c, level, c->stream)) return e;
    const int T = c->cfg.T[level], W = c->cols[level] / T, H = c->rows[level] / T;
    DevBuf d;
    if (int e = d.ensure(512)) return e;
    hipLaunchKernelGGL(k_similarity_local_patch, dim3(1), dim3(64 * LOCAL_WAVES), 0, c->stream, c->d_lm[level].as<uint8_t>(),
                       c->lm_stride[level], c->rows[level], c->cols[level], T, W, H, c->h_tls[(size_t)t * c->L + level], c->d_fxy.as<uint32_t>(),
                       c->d_foff.as<int32_t>(), cx, cy, d.as<uint16_t>());
    int rc = 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(dst, d.p, 512, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SBM_ERR_HIP, "similarity_local failed: %s", hipGetErrorString(hipGetLastError()));
    d.release();
    return rc;
}

int sbm_set_profiling(sbm_ctx* c, int32_t enabled)
{
    if (!c) return fail(SBM_ERR_INVALID, "null context");
    c->profiling = enabled != 0;
    c->profiling_keep = enabled == 2;
    c->clear_timings();
    return 0;
}

int sbm_get_timings(sbm_ctx* c, const char** names, float* ms, int32_t cap, int32_t* n)
{
    if (!c || !n) return fail(SBM_ERR_INVALID, "null argument");
    if (c->timing_ms.size() != c->timings.size()) collect_timings(c);
    *n = (int32_t)c->timings.size();
    for (int i = 0; i < *n && i < cap; ++i) {
        if (names) names[i] = c->timings[i].name;
        if (ms) ms[i] = c->timing_ms[i];
    }
    if (c->profiling_keep && cap >= *n && *n > 0) c->clear_timings(); // read out: start the next accumulation
    return 0;
}

int sbm_get_stats(sbm_ctx* c, int64_t* n_candidates, int64_t* refine_bytes)
{
    if (!c) return fail(SBM_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    HIP_TRY(hipDeviceSynchronize());
    int32_t h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(h, c->d_counters.p, sizeof h, hipMemcpyDeviceToHost));
    if (n_candidates) *n_candidates = h[0];
    if (refine_bytes) {
        uint64_t b;
        memcpy(&b, &h[2], sizeof b);
        *refine_bytes = (int64_t)b;
    }
    return 0;
}

int sbm_set_result_mirror(sbm_ctx* c, void* mirror_out, void* mirror_count)
{
    if (!c || ((mirror_out == nullptr) != (mirror_count == nullptr))) return fail(SBM_ERR_INVALID, "both mirror pointers or neither");
    c->mirror_out = (sbm_match_rec*)mirror_out;
    c->mirror_count = (int32_t*)mirror_count;
    return 0;
}

} // extern "C" (reopened below)

// ---- RCCL exchange step (resolved at run time: the library itself does not link librccl) ----
namespace {
struct Id128 { // ncclUniqueId is passed by value
    char b[128];
};
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
int rccl_load()
{
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!g_rccl.h) return fail(SBM_ERR_HIP, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.h, "ncclGetErrorString");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(g_rccl.h, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(g_rccl.h, "ncclGroupEnd");
    g_rccl_destroy_hook = [](sbm_ctx* c) {
        if (c->comm) g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    };
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        g_rccl.h = nullptr;
        return fail(SBM_ERR_HIP, "librccl lacks the expected entry points");
    }
    return 0;
}
const char* rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "nccl error"; }
} // namespace

extern "C" int sbm_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(SBM_ERR_INVALID, "null argument");
    if (int e = rccl_load()) return e;
    int rc = g_rccl.GetUniqueId(id_out);
    if (rc) return fail(SBM_ERR_HIP, "ncclGetUniqueId: %s", rccl_err(rc));
    return 0;
}

extern "C" int sbm_comm_init(sbm_ctx* c, int32_t world, int32_t rank, const void* id)
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(SBM_ERR_INVALID, "bad communicator arguments");
    if (int e = rccl_load()) return e;
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    if (c->comm) {
        g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    Id128 uid;
    memcpy(uid.b, id, 128);
    int rc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (rc) return fail(SBM_ERR_HIP, "ncclCommInitRank: %s", rccl_err(rc));
    c->comm_world = world;
    c->comm_rank = rank;
    return 0;
}

extern "C" int sbm_comm_destroy(sbm_ctx* c)
{
    if (!c) return fail(SBM_ERR_INVALID, "null context");
    if (c->comm && g_rccl.CommDestroy) {
        (void)hipDeviceSynchronize();
        g_rccl.CommDestroy(c->comm);
    }
    c->comm = nullptr;
    c->comm_world = 0;
    return 0;
}

extern "C" int sbm_match_device_sharded(sbm_ctx* c, const void* d_img, int32_t rows, int32_t cols, int32_t stride, int32_t channels,
                                        const void* d_mask, float threshold, void* d_local, int64_t cap, void* d_gathered,
                                        void* gathered_mirror, void* stream)
{
    if (!c || !d_local || !d_gathered) return fail(SBM_ERR_INVALID, "null argument");
    if (!c->comm) return fail(SBM_ERR_STATE, "sbm_comm_init has not been called on this context");
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t bytes = (size_t)SBM_SHARD_HEADER_BYTES + (size_t)cap * sizeof(sbm_match_rec);
    // this rank's shard: {n_matches, overflow, 0, 0} header followed by the records
    if (int e = sbm_match_device(c, d_img, rows, cols, stride, channels, d_mask, threshold, (char*)d_local + SBM_SHARD_HEADER_BYTES, cap,
                                 d_local, s))
        return e;
    // the one exchange step of the path: every rank's list to every rank, over xGMI, on the same stream
    int rc = g_rccl.AllGather(d_local, d_gathered, bytes, /* ncclUint8 */ 1, c->comm, s);
    if (rc) return fail(SBM_ERR_HIP, "ncclAllGather: %s", rccl_err(rc));
    if (gathered_mirror) {
        const size_t total = bytes * (size_t)c->comm_world;
        hipLaunchKernelGGL(k_copy_bytes, dim3((unsigned)std::min<size_t>((total / 16 + 255) / 256 + 1, 1024)), dim3(256), 0, s,
                           (const uint8_t*)d_gathered, (uint8_t*)gathered_mirror, total);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int sbm_match_batch_device_sharded(sbm_ctx* c, const void* d_imgs, int64_t frame_stride, int32_t n_frames, int32_t rows,
                                              int32_t cols, int32_t stride, int32_t channels, const void* d_mask, float threshold,
                                              void* d_local, int64_t cap, void* d_gathered, void* gathered_mirror, void* stream)
{
    if (!c || !d_local || !d_gathered) return fail(SBM_ERR_INVALID, "null argument");
    if (!c->comm) return fail(SBM_ERR_STATE, "sbm_comm_init has not been called on this context");
    if (n_frames < 1) return fail(SBM_ERR_INVALID, "n_frames must be >= 1");
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // this rank's shard: n_frames {n_matches, overflow} pairs (padded to 16 bytes), then n_frames blocks of cap records
    const size_t header = ((size_t)n_frames * 8 + 15) / 16 * 16;
    const size_t bytes = header + (size_t)n_frames * (size_t)cap * sizeof(sbm_match_rec);
    if (int e = sbm_match_batch_device(c, d_imgs, frame_stride, n_frames, rows, cols, stride, channels, d_mask, threshold,
                                       (char*)d_local + header, cap, d_local, s))
        return e;
    int rc = g_rccl.AllGather(d_local, d_gathered, bytes, /* ncclUint8 */ 1, c->comm, s);
    if (rc) return fail(SBM_ERR_HIP, "ncclAllGather: %s", rccl_err(rc));
    if (gathered_mirror) {
        const size_t total = bytes * (size_t)c->comm_world;
        hipLaunchKernelGGL(k_copy_bytes, dim3((unsigned)std::min<size_t>((total / 16 + 255) / 256 + 1, 1024)), dim3(256), 0, s,
                           (const uint8_t*)d_gathered, (uint8_t*)gathered_mirror, total);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// the exchange of the build-sharded step: every rank's row band of every level's orientation map, in place
// (rank r's band of frame f already sits at its final address: ncclAllGather with sendbuff = recvbuff + r * count),
// all frames and levels in one ncclGroup = one fused launch on the kernels' stream
static int gather_bands(sbm_ctx* c, hipStream_t s, int frames)
{
    int rc = g_rccl.GroupStart();
    if (rc) return fail(SBM_ERR_HIP, "ncclGroupStart: %s", rccl_err(rc));
    for (int l = 0; l < c->L && !rc; ++l) {
        const size_t fs = (size_t)c->rows[l] * c->cols[l], cnt = fs / (size_t)c->comm_world;
        for (int f = 0; f < frames && !rc; ++f) {
            uint8_t* recv = c->d_quant[l].as<uint8_t>() + (size_t)f * fs;
            rc = g_rccl.AllGather(recv + (size_t)c->comm_rank * cnt, recv, cnt, /* ncclUint8 */ 1, c->comm, s);
        }
    }
    const int rc2 = g_rccl.GroupEnd();
    if (rc || rc2) return fail(SBM_ERR_HIP, "ncclAllGather (bands): %s", rccl_err(rc ? rc : rc2));
    return 0;
}

extern "C" int sbm_match_batch_device_banded(sbm_ctx* c, const void* d_imgs, int64_t frame_stride, int32_t n_frames, int32_t rows,
                                             int32_t cols, int32_t stride, int32_t channels, const void* d_mask, float threshold,
                                             void* d_local, int64_t cap, void* d_gathered, void* gathered_mirror, int32_t n_bands,
                                             void* stream)
{
    if (!c || !d_imgs || !d_local) return fail(SBM_ERR_INVALID, "null argument");
    if (n_frames < 1) return fail(SBM_ERR_INVALID, "n_frames must be >= 1");
    if (stride < cols * channels) return fail(SBM_ERR_INVALID, "stride %d < cols*channels", stride);
    if (n_frames > 1 && frame_stride < (int64_t)stride * rows) return fail(SBM_ERR_INVALID, "frame_stride smaller than one frame");
    const bool comm = c->comm != nullptr;
    if (comm && !d_gathered) return fail(SBM_ERR_INVALID, "d_gathered is required with a communicator");
    const bool multi = comm && c->comm_world > 1;
    if (multi && n_bands != 0 && n_bands != c->comm_world)
        return fail(SBM_ERR_INVALID, "n_bands %d != communicator size %d", n_bands, c->comm_world);
    if (multi) n_bands = c->comm_world;
    if (n_bands < 1) return fail(SBM_ERR_INVALID, "n_bands must be >= 1 on a single GPU");
    HIP_TRY(hipSetDevice(c->cfg.device_id));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const bool dirty = !(c->channels == channels && c->rows[0] == rows && c->cols[0] == cols && c->levels_valid == c->L &&
                         n_frames <= c->batch) ||
                       !c->have_thr || memcmp(&threshold, &c->thr_cached, 4) != 0 || c->foff_dirty;
    if (dirty) HIP_TRY(hipDeviceSynchronize());
    if (int e = ensure_geometry(c, rows, cols, channels, n_frames)) return e;
    if (int e = check_bands(c, n_bands)) return e;
    if (c->profiling && !c->profiling_keep) c->clear_timings();
    if (int e = prepare_templates(c, s, threshold, cap)) return e;
    const size_t header = ((size_t)n_frames * 8 + 15) / 16 * 16;
    const size_t bytes = header + (size_t)n_frames * (size_t)cap * sizeof(sbm_match_rec);
    sbm_match_rec* out = (sbm_match_rec*)((char*)d_local + header);
    int32_t* counts = (int32_t*)d_local;
    Bands b;
    b.n = n_bands;
    b.first = multi ? c->comm_rank : 0;  // several ranks: this rank's band, then the exchange
    b.count = multi ? 1 : n_bands;       // one GPU: every band, one launch each (rehearsal of the band launches; with a
    b.between = comm ? gather_bands : nullptr; // one-rank communicator also of the grouped in-place all-gathers)
    // a result mirror set on the context would be written by the last kernel; the gathered mirror below replaces it
    if (int e = enqueue_pyramid(c, s, (const uint8_t*)d_imgs, stride, (const uint8_t*)d_mask, counts, n_frames, frame_stride, &b)) return e;
    if (int e = enqueue_coarse(c, s, out, cap, counts, n_frames)) return e;
    if (int e = enqueue_local(c, s, out, cap, counts, n_frames)) return e;
    const void* result = d_local;
    size_t total = bytes;
    if (comm) { // the second exchange of the step: the per-rank match lists (as sbm_match_batch_device_sharded)
        int rc = g_rccl.AllGather(d_local, d_gathered, bytes, /* ncclUint8 */ 1, c->comm, s);
        if (rc) return fail(SBM_ERR_HIP, "ncclAllGather: %s", rccl_err(rc));
        result = d_gathered;
        total = bytes * (size_t)c->comm_world;
    }
    if (gathered_mirror) {
        hipLaunchKernelGGL(k_copy_bytes, dim3((unsigned)std::min<size_t>((total / 16 + 255) / 256 + 1, 1024)), dim3(256), 0, s,
                           (const uint8_t*)result, (uint8_t*)gathered_mirror, total);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// Single-process multi-GPU match (sbm_match_sharded(ctxs[], n_gpus, ...)): the analogue of the reference's
// OpenMP team (line2Dup.cpp:1166-1170) with one host thread and one context per GPU.  Every context holds the same
// templates and its own selection (sbm_partition_templates + sbm_select_range / sbm_select_templates); the frame goes to
// every GPU, each matches its template shard, the lists are concatenated on the host (the reduction of :1168).
extern "C" int sbm_match_sharded(sbm_ctx* const* ctxs, int32_t n_ctx, const uint8_t* img, int32_t rows, int32_t cols, int32_t stride,
                                 int32_t channels, const uint8_t* mask, float threshold, sbm_match_rec* out, int64_t cap, int64_t* n_out)
{
    if (!ctxs || n_ctx < 1 || !n_out || (!out && cap > 0)) return fail(SBM_ERR_INVALID, "bad arguments");
    for (int i = 0; i < n_ctx; ++i)
        if (!ctxs[i]) return fail(SBM_ERR_INVALID, "context %d is null", i);
    if (n_ctx == 1) return sbm_match(ctxs[0], img, rows, cols, stride, channels, mask, threshold, out, cap, n_out);
    struct Shard {
        std::vector<sbm_match_rec> recs;
        int64_t n = 0;
        int rc = 0;
        std::string err;
    };
    std::vector<Shard> sh(n_ctx);
    std::vector<std::thread> th;
    for (int i = 0; i < n_ctx; ++i)
        th.emplace_back([&, i]() {
            Shard& me = sh[i];
            me.recs.resize((size_t)std::max<int64_t>(cap, 1));
            me.rc = sbm_match(ctxs[i], img, rows, cols, stride, channels, mask, threshold, me.recs.data(), (int64_t)me.recs.size(), &me.n);
            if (me.rc) me.err = sbm_last_error(); // the message is thread-local: carry it to the caller's thread
        });
    for (auto& t : th) t.join();
    int64_t total = 0;
    for (int i = 0; i < n_ctx; ++i) {
        if (sh[i].rc && sh[i].rc != SBM_ERR_CAPACITY) return fail(sh[i].rc, "shard %d: %s", i, sh[i].err.c_str());
        total += sh[i].n;
    }
    *n_out = total;
    for (int i = 0; i < n_ctx; ++i)
        if (sh[i].rc == SBM_ERR_CAPACITY) return fail(SBM_ERR_CAPACITY, "shard %d: %s", i, sh[i].err.c_str());
    if (total > cap) return fail(SBM_ERR_CAPACITY, "%lld matches exceed the output capacity %lld", (long long)total, (long long)cap);
    int64_t k = 0;
    for (int i = 0; i < n_ctx; ++i) {
        if (sh[i].n > 0) memcpy(out + k, sh[i].recs.data(), (size_t)sh[i].n * sizeof(sbm_match_rec));
        k += sh[i].n;
    }
    return 0;
}

extern "C" {

int sbm_coarse_bytes(sbm_ctx* c, int64_t* bytes)
{
    if (!c || !bytes) return fail(SBM_ERR_INVALID, "null argument");
    if (c->levels_valid < c->L) return fail(SBM_ERR_STATE, "pyramid not built");
    const int lc = c->L - 1, T = c->cfg.T[lc], W = c->cols[lc] / T, H = c->rows[lc] / T;
    int64_t total = 0;
    for (int32_t t : c->h_active) {
        const DevTL& tl = c->h_tls[(size_t)t * c->L + lc];
        const int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
        const int npos = (H - hf) * W + (W - wf) + 1;
        if (npos <= 0) continue;
        for (int i = 0; i < tl.nf; ++i) {
            const uint32_t xy = c->h_fxy[tl.feat_off + i];
            if ((int)(xy & 0xffff) < c->cols[lc] && (int)(xy >> 16) < c->rows[lc]) total += npos;
        }
    }
    *bytes = total;
    return 0;
}

} // extern "C"
