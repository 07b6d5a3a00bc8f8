// line2Dup_amd.cpp — host implementation of include/line2Dup.h on top of the C
// ABI of libsbm_hip.so (include/sbm.h).  Everything that touches pixels —
// quantizedOrientations, pyrDown, spread/response/linearize, similarity,
// similarityLocal — runs in the HIP kernels; this file is the Detector's
// bookkeeping (template containers, YAML persistence, training-side feature
// selection) written against the behaviour of the reference (citations are
// file:line in ddcr/shape_based_matching).
#include "../../include/line2Dup.h"
#include "../../include/sbm.h"

#include <algorithm>
#include <climits>
#include <condition_variable>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

using namespace cv;

namespace {

void check(int rc, const char* what)
{
    if (rc != 0) CV_Error(rc == SBM_ERR_INVALID ? Error::StsBadArg : Error::StsError, std::string(what) + ": " + sbm_last_error());
}

// A context used for the single-function stage calls of the training path
// (ColorGradientPyramid::update / pyrDown); created on first use, device 0.
sbm_ctx* util_ctx()
{
    static sbm_ctx* ctx = nullptr;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!ctx) {
        sbm_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.n_levels = 1;
        cfg.T[0] = 4;
        cfg.weak_threshold = 30.f;
        cfg.device_id = 0;
        cfg.max_candidates = 1024;
        check(sbm_create(&cfg, &ctx), "sbm_create");
    }
    return ctx;
}

int label_of(int quantized) // getLabel, line2Dup.cpp:16-40
{
    for (int i = 0; i < 8; ++i)
        if (quantized == (1 << i)) return i;
    CV_Error(Error::StsBadArg, "Invalid value of quantized parameter");
    return -1;
}

// cropTemplates, line2Dup.cpp:115-161
Rect crop_templates(std::vector<line2Dup::Template>& templates)
{
    int min_x = INT_MAX, min_y = INT_MAX, max_x = INT_MIN, max_y = INT_MIN;
    for (const auto& t : templates)
        for (const auto& f : t.features) {
            const int x = f.x << t.pyramid_level, y = f.y << t.pyramid_level;
            min_x = std::min(min_x, x);
            min_y = std::min(min_y, y);
            max_x = std::max(max_x, x);
            max_y = std::max(max_y, y);
        }
    if (min_x % 2 == 1) --min_x;
    if (min_y % 2 == 1) --min_y;
    for (auto& t : templates) {
        t.width = (max_x - min_x) >> t.pyramid_level;
        t.height = (max_y - min_y) >> t.pyramid_level;
        t.tl_x = min_x >> t.pyramid_level;
        t.tl_y = min_y >> t.pyramid_level;
        for (auto& f : t.features) {
            f.x -= t.tl_x;
            f.y -= t.tl_y;
        }
    }
    return Rect(min_x, min_y, max_x - min_x, max_y - min_y);
}

} // namespace

namespace line2Dup {

// ---- persistence of Feature / Template (line2Dup.cpp:42-113) ------------------
void Feature::read(const FileNode& fn)
{
    FileNodeIterator it = fn.begin();
    it >> x >> y >> label;
}

void Feature::write(FileStorage& fs) const { fs << "[:" << x << y << label << "]"; }

void Template::read(const FileNode& fn)
{
    width = fn["width"];
    height = fn["height"];
    tl_x = fn["tl_x"];
    tl_y = fn["tl_y"];
    sscale = fn["scale"];
    orientation = fn["orientation"];
    tagFieldID = fn["tagFieldID"];
    fn["fiducial_src"] >> fiducial_src;
    pyramid_level = fn["pyramid_level"];
    FileNode ff = fn["features"];
    features.resize(ff.size());
    size_t i = 0;
    for (FileNodeIterator it = ff.begin(); it != ff.end(); ++it, ++i) features[i].read(*it);
}

void Template::write(FileStorage& fs) const
{
    fs << "width" << width;
    fs << "height" << height;
    fs << "tl_x" << tl_x;
    fs << "tl_y" << tl_y;
    fs << "scale" << sscale;
    fs << "orientation" << orientation;
    fs << "tagFieldID" << tagFieldID;
    fs << "fiducial_src" << fiducial_src;
    fs << "pyramid_level" << pyramid_level;
    fs << "features" << "[";
    for (const auto& f : features) f.write(fs);
    fs << "]";
}

// ---- ColorGradientPyramid (line2Dup.cpp:406-539) --------------------------------
ColorGradientPyramid::ColorGradientPyramid(const Mat& _src, const Mat& _mask, float _weak_threshold, size_t _num_features,
                                           float _strong_threshold)
    : src(_src), mask(_mask), pyramid_level(0), weak_threshold(_weak_threshold), num_features(_num_features),
      strong_threshold(_strong_threshold)
{
    update();
}

// quantizedOrientations + hysteresisGradient on the GPU (line2Dup.cpp:313-404, 218-311)
void ColorGradientPyramid::update()
{
    CV_Assert(!src.empty() && src.depth() == CV_8U && (src.channels() == 1 || src.channels() == 3));
    magnitude.create(src.rows, src.cols, CV_32FC1);
    angle.create(src.rows, src.cols, CV_8UC1);
    angle_ori.create(src.rows, src.cols, CV_32FC1);
    check(sbm_quantized_orientations(util_ctx(), src.data, src.rows, src.cols, (int)src.step, src.channels(), weak_threshold,
                                     magnitude.ptr<float>(), angle.ptr<uchar>(), angle_ori.ptr<float>()),
          "sbm_quantized_orientations");
}

void ColorGradientPyramid::pyrDown()
{
    num_features /= 2; // line2Dup.cpp:427
    ++pyramid_level;
    Mat next(src.rows / 2, src.cols / 2, src.type());
    check(sbm_pyrdown(util_ctx(), src.data, src.rows, src.cols, (int)src.step, src.channels(), next.data), "sbm_pyrdown");
    if (!mask.empty()) { // resize(mask, INTER_NEAREST), line2Dup.cpp:439
        Mat next_mask(next.rows, next.cols, CV_8UC1);
        const double fx = (double)mask.cols / next.cols, fy = (double)mask.rows / next.rows;
        for (int y = 0; y < next.rows; ++y) {
            const int sy = std::min((int)std::floor(y * fy), mask.rows - 1);
            for (int x = 0; x < next.cols; ++x) next_mask.ptr(y)[x] = mask.ptr(sy)[std::min((int)std::floor(x * fx), mask.cols - 1)];
        }
        mask = next_mask;
    }
    src = next;
    update();
}

void ColorGradientPyramid::quantize(Mat& dst) const
{
    dst = Mat::zeros(angle.size(), CV_8UC1);
    angle.copyTo(dst, mask);
}

// selectScatteredFeatures, line2Dup.cpp:163-212: sweep the (score-sorted) candidates keeping
// those at least `distance` from every kept feature; grow the distance while a sweep still
// yields enough features, then shrink it (keeping what was chosen) until enough or distance < 3.
bool ColorGradientPyramid::selectScatteredFeatures(const std::vector<Candidate>& candidates, std::vector<Feature>& features,
                                                   size_t num_features, float distance)
{
    features.clear();
    if (candidates.empty()) return true;
    bool growing = true;
    for (;;) {
        const float d2 = distance * distance;
        for (const Candidate& c : candidates) {
            bool far_enough = true;
            for (size_t j = 0; j < features.size() && far_enough; ++j) {
                const int dx = c.f.x - features[j].x, dy = c.f.y - features[j].y;
                far_enough = (float)(dx * dx + dy * dy) >= d2;
            }
            if (far_enough) features.push_back(c.f);
        }
        const bool enough = features.size() >= num_features;
        if (growing) {
            if (enough) {
                features.clear();
                distance += 1.0f;
                continue;
            }
            growing = false;
        }
        distance -= 1.0f;
        if (enough || distance < 3) break;
    }
    return true;
}

// extractTemplate (line2Dup.cpp:452-539).  The reference scans the image pixel by pixel on the host; here the scan is a
// HIP kernel (csrc/sbm_train_kernels.h: every pixel above strong_threshold^2 that passes the eroded mask and has no
// larger 5x5 neighbour, found in parallel; ties between equal neighbours resolved in row-major order as the reference's
// `magnitude_valid` map does) and the host only turns the accepted maxima into candidates and spreads them out.
bool ColorGradientPyramid::extractTemplate(Template& templ) const
{
    Mat mask8;
    if (!mask.empty()) {
        CV_Assert(mask.type() == CV_8UC1 && mask.size() == magnitude.size());
        mask8 = mask.isContinuous() ? mask : mask.clone();
    }
    Mat mag = magnitude.isContinuous() ? magnitude : magnitude.clone();
    std::vector<int32_t> maxima((size_t)1 << 14);
    int64_t n = 0;
    for (;;) {
        const int rc = sbm_extract_local_maxima(util_ctx(), mag.ptr<float>(), mask8.empty() ? nullptr : mask8.data, mag.rows, mag.cols,
                                                strong_threshold, maxima.data(), (int64_t)maxima.size(), &n);
        if (rc == SBM_ERR_CAPACITY && n > (int64_t)maxima.size()) {
            maxima.resize((size_t)n);
            continue;
        }
        check(rc, "sbm_extract_local_maxima");
        break;
    }
    // accepted maxima arrive in row-major order; one with a quantised orientation is a feature candidate (:504)
    std::vector<Candidate> candidates;
    candidates.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int x = maxima[(size_t)i] & 0xffff, y = maxima[(size_t)i] >> 16;
        const uchar a = angle.ptr(y)[x];
        if (!a) continue;
        candidates.push_back(Candidate(x, y, label_of(a), magnitude.at<float>(y, x)));
        candidates.back().f.theta = angle_ori.at<float>(y, x);
    }
    if (candidates.size() <= 4 && candidates.size() < num_features) { // :507-512: five or more go on, however few
        std::cout << "extractTemplate: only " << candidates.size() << " candidate features at pyramid level " << pyramid_level
                  << ", giving up on this template" << std::endl;
        return false;
    }
    if (candidates.size() < num_features)
        std::cout << "extractTemplate: " << candidates.size() << " candidates for " << num_features << " features at pyramid level "
                  << pyramid_level << ", taking what there is" << std::endl;
    std::stable_sort(candidates.begin(), candidates.end()); // by score, ties stay in row-major order (:515)
    const float distance = static_cast<float>(candidates.size() / num_features + 1); // integer division (:519)
    if (!selectScatteredFeatures(candidates, templ.features, num_features, distance)) return false;
    templ.width = templ.height = -1; // set by cropTemplates
    templ.pyramid_level = pyramid_level;
    return true;
}

// ---- ColorGradient (line2Dup.cpp:541-578) ---------------------------------------
ColorGradient::ColorGradient() : weak_threshold(30.0f), num_features(63), strong_threshold(60.0f) {}
ColorGradient::ColorGradient(float w, size_t n, float s) : weak_threshold(w), num_features(n), strong_threshold(s) {}
std::string ColorGradient::name() const { return "ColorGradient"; }
void ColorGradient::read(const FileNode& fn)
{
    std::string type = fn["type"];
    CV_Assert(type == "ColorGradient");
    weak_threshold = fn["weak_threshold"];
    num_features = (size_t)(int)fn["num_features"];
    strong_threshold = fn["strong_threshold"];
}
void ColorGradient::write(FileStorage& fs) const
{
    fs << "type" << "ColorGradient";
    fs << "weak_threshold" << weak_threshold;
    fs << "num_features" << int(num_features);
    fs << "strong_threshold" << strong_threshold;
}

// ---- Detector (line2Dup.cpp:1054-1599) --------------------------------------------
Detector* Detector::instance = nullptr;

// ---- device side: a pool of lanes -------------------------------------------------------------------------------
// The reference's match() is const and keeps no state (line2Dup.h:272-274): two threads may call it on one Detector.
// Here a call needs an engine context (device buffers sized for the frame, uploaded templates, a stream), which is not
// thread-safe.  So the Detector keeps a pool of LANES -- a lane = one context per device + what was uploaded / selected
// on them -- and every call takes a free lane for its duration: concurrent callers never share a context, a single
// caller only ever uses lane 0.
struct Detector::Engine {
    struct Flat { // the TemplatesMap flattened for sbm_upload_templates, rebuilt when class_templates changed
        std::vector<sbm_template_level> levels;
        std::vector<sbm_feature> feats;
        std::vector<int32_t> cls, tid;
        std::vector<std::string> class_order; // class index -> id (map order = the order match() walks the classes, :1127-1129)
    };
    struct Lane {
        std::vector<sbm_ctx*> ctxs; // one per device
        std::shared_ptr<const Flat> uploaded;
        std::vector<int32_t> selected; // class selection currently active in the engine
        bool selection_valid = false;
        int selection_mode = 0; // 0: every context holds the whole selection; 1: sharded over the contexts
        int selection_rows = 0, selection_cols = 0; // geometry the shards were balanced for
        std::vector<unsigned char> recs; // match record scratch, kept between calls
        bool busy = false;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::unique_ptr<Lane>> lanes;
    int max_lanes = 4;
    std::shared_ptr<const Flat> flat; // null = class_templates changed since the last flattening
    std::map<const void*, std::pair<int, sbm_ctx*>> pins; // pinBuffer: buffer -> (lane, context it is registered with)
    struct Async { // batch in flight (matchAsync): its lane, frames per context, capacity, what a retry needs
        bool active = false;
        int lane = -1;
        std::vector<int> first, count;
        int64_t cap = 0;
        size_t n_frames = 0;
        std::vector<Mat> sources;
        Mat mask8;
        float threshold = 0.f;
        std::shared_ptr<const Flat> flat;
    } async;
};

namespace {
struct LaneLease { // releases the lane when the call ends, also by exception
    line2Dup::Detector::Engine* e;
    int idx;
    ~LaneLease();
};
} // namespace

Detector::Detector() : modality(makePtr<ColorGradient>()), pyramid_levels(2), T_at_level({4, 8}), eng_(new Engine), device_id_(0) {}
Detector::Detector(std::vector<int> T)
    : modality(makePtr<ColorGradient>()), pyramid_levels((int)T.size()), T_at_level(T), eng_(new Engine), device_id_(0)
{
}
Detector::Detector(int num_features, std::vector<int> T, float weak_thresh, float strong_thresh)
    : modality(makePtr<ColorGradient>(weak_thresh, (size_t)num_features, strong_thresh)), pyramid_levels((int)T.size()), T_at_level(T),
      eng_(new Engine), device_id_(0)
{
}
Detector::Detector(const Detector& o)
    : modality(makePtr<ColorGradient>(*o.modality)), pyramid_levels(o.pyramid_levels), T_at_level(o.T_at_level),
      class_templates(o.class_templates), eng_(new Engine), device_id_(o.device_id_), device_ids_(o.device_ids_)
{
    eng_->max_lanes = o.eng_->max_lanes;
}
Detector& Detector::operator=(const Detector& o)
{
    if (this != &o) {
        dropContext();
        modality = makePtr<ColorGradient>(*o.modality);
        pyramid_levels = o.pyramid_levels;
        T_at_level = o.T_at_level;
        class_templates = o.class_templates;
        device_id_ = o.device_id_;
        device_ids_ = o.device_ids_;
        eng_->max_lanes = o.eng_->max_lanes;
    }
    return *this;
}
Detector::~Detector()
{
    dropContext();
    delete eng_;
}

// mutators only (not thread-safe against running calls, as in the reference)
void Detector::dropContext()
{
    Engine& e = *eng_;
    std::lock_guard<std::mutex> lock(e.mu);
    e.async = Engine::Async();
    for (auto& kv : e.pins) (void)sbm_unpin_host_buffer(kv.second.second, kv.first);
    e.pins.clear();
    for (auto& l : e.lanes)
        for (sbm_ctx* c : l->ctxs) sbm_destroy(c);
    e.lanes.clear();
    e.flat.reset();
}

void Detector::setDevice(int device_id)
{
    if (device_id != device_id_ || device_ids_.size() > 1) dropContext();
    device_id_ = device_id;
    device_ids_.clear();
}

void Detector::setDevices(const std::vector<int>& device_ids)
{
    CV_Assert(!device_ids.empty());
    dropContext();
    device_ids_ = device_ids;
    device_id_ = device_ids[0];
}

void Detector::setConcurrency(int n)
{
    CV_Assert(n >= 1);
    std::lock_guard<std::mutex> lock(eng_->mu);
    eng_->max_lanes = n; // lanes beyond n that already exist stay (their memory is the price already paid); no new ones
}

LaneLease::~LaneLease()
{
    {
        std::lock_guard<std::mutex> lock(e->mu);
        e->lanes[(size_t)idx]->busy = false;
    }
    e->cv.notify_all();
}

namespace {

// take a free lane (want < 0: any; else that one), creating an empty one while the pool may grow, else wait
int acquire_lane(line2Dup::Detector::Engine& e, int want = -1)
{
    std::unique_lock<std::mutex> lock(e.mu);
    for (;;) {
        if (want >= 0) {
            if ((size_t)want < e.lanes.size() && !e.lanes[(size_t)want]->busy) {
                e.lanes[(size_t)want]->busy = true;
                return want;
            }
        } else {
            for (size_t i = 0; i < e.lanes.size(); ++i)
                if (!e.lanes[i]->busy) {
                    e.lanes[i]->busy = true;
                    return (int)i;
                }
            // a batch in flight (matchAsync ... wait) holds its lane between two calls of the SAME thread: it does not count
            // against the limit, or `matchAsync(); match(); wait();` with setConcurrency(1) would wait for itself
            if ((int)e.lanes.size() < e.max_lanes + (e.async.active && e.async.lane >= 0 ? 1 : 0)) {
                e.lanes.emplace_back(new line2Dup::Detector::Engine::Lane);
                e.lanes.back()->busy = true;
                return (int)e.lanes.size() - 1;
            }
        }
        e.cv.wait(lock);
    }
}

} // namespace

// Contexts, templates and class selection of lane `li` (held by the caller) for a call.  sharded: the selected templates
// are divided over the lane's contexts (match() with several devices); else every context holds the whole selection (one
// device, or matchBatch, which deals frames).  The engine's selection calls synchronise the device, so they are issued
// only when something changed.  Returns the flattening the lane now holds, or null when the selection is empty.
static std::shared_ptr<const Detector::Engine::Flat> prepare_lane(const Detector& det, Detector::Engine& e, int li, const std::vector<int>& devs,
                                                                  const std::map<std::string, std::vector<std::vector<Template>>>& class_templates,
                                                                  int pyramid_levels, const std::vector<int>& T_at_level, float weak_threshold,
                                                                  const std::vector<std::string>& class_ids, int rows, int cols, bool sharded)
{
    (void)det;
    typedef Detector::Engine::Flat Flat;
    Detector::Engine::Lane& lane = *e.lanes[(size_t)li];
    if (lane.ctxs.empty()) {
        CV_Assert(pyramid_levels >= 1 && pyramid_levels <= SBM_MAX_LEVELS && (int)T_at_level.size() >= pyramid_levels);
        for (int d : devs) {
            sbm_config cfg;
            memset(&cfg, 0, sizeof cfg);
            cfg.n_levels = pyramid_levels;
            for (int l = 0; l < pyramid_levels; ++l) cfg.T[l] = T_at_level[l];
            cfg.weak_threshold = weak_threshold;
            cfg.device_id = d;
            cfg.max_candidates = 0;
            sbm_ctx* c = nullptr;
            check(sbm_create(&cfg, &c), "sbm_create");
            lane.ctxs.push_back(c);
        }
        lane.uploaded.reset();
    }
    // one flattening of the TemplatesMap per change, shared by the lanes
    std::shared_ptr<const Flat> flat;
    {
        std::lock_guard<std::mutex> lock(e.mu);
        if (!e.flat) {
            std::shared_ptr<Flat> f(new Flat);
            for (const auto& kv : class_templates) {
                const int ci = (int)f->class_order.size();
                f->class_order.push_back(kv.first);
                for (size_t t = 0; t < kv.second.size(); ++t) {
                    const std::vector<Template>& tp = kv.second[t];
                    CV_Assert((int)tp.size() == pyramid_levels);
                    for (const Template& tm : tp) {
                        if (tm.features.size() >= 8192) CV_Error(Error::StsBadArg, "feature size too large"); // :1195
                        sbm_template_level lv;
                        lv.width = tm.width;
                        lv.height = tm.height;
                        lv.tl_x = tm.tl_x;
                        lv.tl_y = tm.tl_y;
                        lv.pyramid_level = tm.pyramid_level;
                        lv.n_features = (int32_t)tm.features.size();
                        lv.feature_offset = (int64_t)f->feats.size();
                        for (const Feature& ft : tm.features) f->feats.push_back(sbm_feature{ft.x, ft.y, ft.label});
                        f->levels.push_back(lv);
                    }
                    f->cls.push_back(ci);
                    f->tid.push_back((int32_t)t);
                }
            }
            e.flat = f;
        }
        flat = e.flat;
    }
    if (lane.uploaded != flat) {
        for (sbm_ctx* c : lane.ctxs)
            check(sbm_upload_templates(c, (int32_t)flat->cls.size(), flat->levels.data(), flat->feats.data(), (int64_t)flat->feats.size(),
                                       flat->cls.data(), flat->tid.data()),
                  "sbm_upload_templates");
        lane.uploaded = flat;
        lane.selection_valid = false; // sbm_upload_templates selects every class again
    }
    if (flat->cls.empty()) return nullptr;
    std::vector<int32_t> sel;
    if (!class_ids.empty()) { // unknown ids are skipped silently (:1136-1138)
        for (const std::string& id : class_ids) {
            auto it = std::find(flat->class_order.begin(), flat->class_order.end(), id);
            if (it != flat->class_order.end()) sel.push_back((int32_t)(it - flat->class_order.begin()));
        }
        if (sel.empty()) return nullptr;
    }
    const int mode = sharded && lane.ctxs.size() > 1 ? 1 : 0;
    if (lane.selection_valid && sel == lane.selected && mode == lane.selection_mode &&
        (mode == 0 || (rows == lane.selection_rows && cols == lane.selection_cols)))
        return flat;
    if (mode == 0) {
        for (sbm_ctx* c : lane.ctxs)
            check(sbm_select_classes(c, sel.empty() ? nullptr : sel.data(), (int32_t)sel.size()), "sbm_select_classes");
    } else {
        // the list matchClass would walk (class_ids order, then template order, :1134-1139), cut into work-balanced pieces
        std::vector<int32_t> act;
        if (sel.empty()) {
            act.resize(flat->cls.size());
            for (size_t t = 0; t < act.size(); ++t) act[t] = (int32_t)t;
        } else {
            for (int32_t ci : sel)
                for (size_t t = 0; t < flat->cls.size(); ++t)
                    if (flat->cls[t] == ci) act.push_back((int32_t)t);
        }
        const int D = (int)lane.ctxs.size();
        std::vector<int32_t> first(D), count(D);
        check(sbm_partition_templates(lane.ctxs[0], rows, cols, act.data(), (int32_t)act.size(), D, first.data(), count.data()),
              "sbm_partition_templates");
        for (int d = 0; d < D; ++d) check(sbm_select_templates(lane.ctxs[d], act.data() + first[d], count[d]), "sbm_select_templates");
    }
    lane.selected = sel;
    lane.selection_mode = mode;
    lane.selection_rows = rows;
    lane.selection_cols = cols;
    lane.selection_valid = true;
    return flat;
}

// epilogue (:1142-1145): canonical sort, exact-duplicate removal, then the reference's own adjacent std::unique (its
// operator== ignores template_id)
static std::vector<Match> to_matches(const Detector::Engine::Flat& flat, const void* recs_in, int64_t n)
{
    std::vector<sbm_match_rec> recs((const sbm_match_rec*)recs_in, (const sbm_match_rec*)recs_in + n);
    n = sbm_canonicalize(recs.data(), n);
    std::vector<Match> matches;
    matches.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        matches.push_back(Match(recs[i].x, recs[i].y, recs[i].similarity, flat.class_order[(size_t)recs[i].class_idx], recs[i].template_id));
    matches.erase(std::unique(matches.begin(), matches.end()), matches.end());
    return matches;
}

// one frame on the lane's context(s); grows the lane's record scratch until the list fits
static int64_t match_on_lane(Detector::Engine::Lane& lane, const Mat& source, const Mat& mask8, float threshold)
{
    if (lane.recs.size() < (size_t)4096 * sizeof(sbm_match_rec)) lane.recs.resize((size_t)4096 * sizeof(sbm_match_rec));
    int64_t n = 0;
    for (;;) {
        const int64_t cap = (int64_t)(lane.recs.size() / sizeof(sbm_match_rec));
        // several devices: one host thread and context per device, each with its shard of the templates, lists
        // concatenated (the OpenMP team of :1166-1170); one device: the plain call
        int rc = lane.ctxs.size() > 1
                     ? sbm_match_sharded(lane.ctxs.data(), (int32_t)lane.ctxs.size(), source.data, source.rows, source.cols, (int)source.step,
                                         source.channels(), mask8.empty() ? nullptr : mask8.data, threshold, (sbm_match_rec*)lane.recs.data(), cap, &n)
                     : sbm_match(lane.ctxs[0], source.data, source.rows, source.cols, (int)source.step, source.channels(),
                                 mask8.empty() ? nullptr : mask8.data, threshold, (sbm_match_rec*)lane.recs.data(), cap, &n);
        if (rc == SBM_ERR_CAPACITY && n > cap) {
            lane.recs.resize((size_t)n * sizeof(sbm_match_rec));
            continue;
        }
        check(rc, "sbm_match");
        return n;
    }
}

std::vector<Match> Detector::match(Mat source, float threshold, const std::vector<std::string>& class_ids, const Mat mask) const
{
    CV_Assert(mask.empty() || mask.size() == source.size()); // :1086
    CV_Assert(!source.empty() && source.depth() == CV_8U && (source.channels() == 1 || source.channels() == 3));
    Mat mask8;
    if (!mask.empty()) {
        CV_Assert(mask.type() == CV_8UC1);
        mask8 = mask.isContinuous() ? mask : mask.clone();
    }
    Engine& e = *eng_;
    LaneLease lease{&e, acquire_lane(e)};
    const std::vector<int> devs = device_ids_.empty() ? std::vector<int>{device_id_} : device_ids_;
    const std::shared_ptr<const Engine::Flat> flat = prepare_lane(*this, e, lease.idx, devs, class_templates, pyramid_levels, T_at_level,
                                                                  modality->weak_threshold, class_ids, source.rows, source.cols, true);
    if (!flat) return std::vector<Match>();
    Engine::Lane& lane = *e.lanes[(size_t)lease.idx];
    const int64_t n = match_on_lane(lane, source, mask8, threshold);
    return to_matches(*flat, lane.recs.data(), n);
}

// ---- throughput path: batches of frames from host memory, uploads overlapped with the kernels ----------------------
void Detector::matchAsync(const std::vector<Mat>& sources, float threshold, const std::vector<std::string>& class_ids, const Mat mask) const
{
    CV_Assert(!sources.empty());
    const Mat& s0 = sources[0];
    for (const Mat& m : sources) {
        CV_Assert(!m.empty() && m.depth() == CV_8U && (m.channels() == 1 || m.channels() == 3));
        CV_Assert(m.rows == s0.rows && m.cols == s0.cols && m.channels() == s0.channels() && m.step == s0.step);
    }
    CV_Assert(mask.empty() || (mask.size() == s0.size() && mask.type() == CV_8UC1));
    Engine& e = *eng_;
    {
        std::lock_guard<std::mutex> lock(e.mu);
        CV_Assert(!e.async.active); // one batch in flight per detector
        e.async = Engine::Async();
        e.async.active = true; // claimed; filled in below by this thread only
    }
    Engine::Async& as = e.async;
    const int li = acquire_lane(e);
    as.lane = li;
    auto give_up = [&]() { // nothing in flight after all: free the lane and the claim
        LaneLease drop{&e, li};
        std::lock_guard<std::mutex> lock(e.mu);
        as.active = false;
    };
    try {
        as.n_frames = sources.size();
        as.threshold = threshold;
        const std::vector<int> devs = device_ids_.empty() ? std::vector<int>{device_id_} : device_ids_;
        as.flat = prepare_lane(*this, e, li, devs, class_templates, pyramid_levels, T_at_level, modality->weak_threshold, class_ids, s0.rows,
                               s0.cols, false);
        if (!as.flat) { // nothing selected: wait() returns empty lists
            as.cap = 0;
            return;
        }
        Engine::Lane& lane = *e.lanes[(size_t)li];
        // the frames (Mat headers: the pixels are the caller's and must stay unchanged until wait()) and the mask (a copy
        // when it is not continuous) are kept until wait(): the uploads enqueued below read them after this call returns
        as.sources = sources;
        if (!mask.empty()) as.mask8 = mask.isContinuous() ? mask : mask.clone();
        // frames dealt over the devices in contiguous groups (frames are independent: Detector::match keeps no state)
        const int D = (int)lane.ctxs.size(), n = (int)sources.size();
        as.cap = 1024;
        for (int d = 0; d < D; ++d) {
            as.first.push_back((int)((int64_t)n * d / D));
            as.count.push_back((int)((int64_t)n * (d + 1) / D) - as.first.back());
        }
        for (int d = 0; d < D; ++d) {
            if (!as.count[d]) continue;
            std::vector<const uint8_t*> ptrs;
            for (int f = 0; f < as.count[d]; ++f) ptrs.push_back(sources[as.first[d] + f].data);
            const int rc = sbm_match_batch_host_begin(lane.ctxs[d], ptrs.data(), (int32_t)ptrs.size(), s0.rows, s0.cols, (int)s0.step, s0.channels(),
                                                      as.mask8.empty() ? nullptr : as.mask8.data, threshold, as.cap, 0);
            if (rc) {
                const std::string msg = sbm_last_error();
                for (int k = 0; k < d; ++k) { // drain what was started
                    std::vector<sbm_match_rec> tmp((size_t)as.count[k] * as.cap);
                    std::vector<int32_t> cnt((size_t)as.count[k] * 2);
                    if (as.count[k]) (void)sbm_match_batch_host_end(lane.ctxs[k], tmp.data(), cnt.data());
                }
                CV_Error(rc == SBM_ERR_INVALID ? Error::StsBadArg : Error::StsError, "sbm_match_batch_host_begin: " + msg);
            }
        }
    } catch (...) {
        give_up();
        throw;
    }
}

std::vector<std::vector<Match>> Detector::wait() const
{
    Engine& e = *eng_;
    Engine::Async& as = e.async;
    {
        std::lock_guard<std::mutex> lock(e.mu);
        CV_Assert(as.active && as.lane >= 0);
    }
    LaneLease lease{&e, as.lane}; // the lane matchAsync took is released when this call ends
    struct Done {
        Detector::Engine& e;
        ~Done()
        {
            std::lock_guard<std::mutex> lock(e.mu);
            e.async = Detector::Engine::Async();
        }
    } done{e};
    std::vector<std::vector<Match>> out(as.n_frames);
    if (as.cap == 0) return out;
    Engine::Lane& lane = *e.lanes[(size_t)as.lane];
    int bad = 0;
    std::string msg;
    std::vector<size_t> redo; // frames whose list did not fit the batch's per-frame capacity
    for (size_t d = 0; d < lane.ctxs.size(); ++d) {
        const int nf = as.count[d];
        if (!nf) continue;
        std::vector<sbm_match_rec> recs((size_t)nf * as.cap);
        std::vector<int32_t> cnt((size_t)nf * 2);
        const int rc = sbm_match_batch_host_end(lane.ctxs[d], recs.data(), cnt.data());
        if (rc && rc != SBM_ERR_CAPACITY && !bad) {
            bad = rc;
            msg = sbm_last_error();
        }
        if (rc && rc != SBM_ERR_CAPACITY) continue;
        // SBM_ERR_CAPACITY: some frame of this context produced more raw records than the per-frame capacity of the batch
        // (its count says how many, or its overflow flag is set).  The other frames' lists are complete: keep them, and
        // run the affected frames again one at a time, where match() grows its buffer -- the promise is
        // matchBatch(frames)[f] == match(frames[f]) whatever the lists' sizes.
        for (int f = 0; f < nf; ++f) {
            if (cnt[2 * f] < 0 || cnt[2 * f] > as.cap || cnt[2 * f + 1] != 0) redo.push_back((size_t)(as.first[d] + f));
            else out[(size_t)(as.first[d] + f)] = to_matches(*as.flat, recs.data() + (size_t)f * as.cap, cnt[2 * f]);
        }
    }
    if (bad) CV_Error(bad == SBM_ERR_INVALID ? Error::StsBadArg : Error::StsError, "sbm_match_batch_host_end: " + msg);
    if (!redo.empty()) {
        // the lane's selection is the batch's (whole selection on every context); one context is enough for a frame
        Engine::Lane single;
        single.ctxs.assign(1, lane.ctxs[0]);
        for (size_t f : redo) {
            const int64_t n = match_on_lane(single, as.sources[f], as.mask8, as.threshold);
            out[f] = to_matches(*as.flat, single.recs.data(), n);
        }
    }
    return out;
}

std::vector<std::vector<Match>> Detector::matchBatch(const std::vector<Mat>& sources, float threshold, const std::vector<std::string>& class_ids,
                                                     const Mat mask) const
{
    if (sources.empty()) return std::vector<std::vector<Match>>();
    // not matchAsync + wait: several threads may run batches on one detector at once, each on a lane of its own, while
    // matchAsync / wait is the detector's ONE batch in flight
    const Mat& s0 = sources[0];
    for (const Mat& m : sources) {
        CV_Assert(!m.empty() && m.depth() == CV_8U && (m.channels() == 1 || m.channels() == 3));
        CV_Assert(m.rows == s0.rows && m.cols == s0.cols && m.channels() == s0.channels() && m.step == s0.step);
    }
    CV_Assert(mask.empty() || (mask.size() == s0.size() && mask.type() == CV_8UC1));
    Engine& e = *eng_;
    LaneLease lease{&e, acquire_lane(e)};
    const std::vector<int> devs = device_ids_.empty() ? std::vector<int>{device_id_} : device_ids_;
    const std::shared_ptr<const Engine::Flat> flat = prepare_lane(*this, e, lease.idx, devs, class_templates, pyramid_levels, T_at_level,
                                                                  modality->weak_threshold, class_ids, s0.rows, s0.cols, false);
    std::vector<std::vector<Match>> out(sources.size());
    if (!flat) return out;
    Engine::Lane& lane = *e.lanes[(size_t)lease.idx];
    Mat mask8;
    if (!mask.empty()) mask8 = mask.isContinuous() ? mask : mask.clone();
    const int D = (int)lane.ctxs.size(), n = (int)sources.size();
    const int64_t cap = 1024;
    std::vector<int> first, count;
    for (int d = 0; d < D; ++d) {
        first.push_back((int)((int64_t)n * d / D));
        count.push_back((int)((int64_t)n * (d + 1) / D) - first.back());
    }
    int begun = 0, bad = 0;
    std::string msg;
    for (int d = 0; d < D && !bad; ++d, ++begun) {
        if (!count[d]) continue;
        std::vector<const uint8_t*> ptrs;
        for (int f = 0; f < count[d]; ++f) ptrs.push_back(sources[first[d] + f].data);
        const int rc = sbm_match_batch_host_begin(lane.ctxs[d], ptrs.data(), (int32_t)ptrs.size(), s0.rows, s0.cols, (int)s0.step, s0.channels(),
                                                  mask8.empty() ? nullptr : mask8.data, threshold, cap, 0);
        if (rc) {
            bad = rc;
            msg = std::string("sbm_match_batch_host_begin: ") + sbm_last_error();
            break;
        }
    }
    std::vector<size_t> redo;
    for (int d = 0; d < begun; ++d) { // every context that began must end, whatever happened since
        if (!count[d]) continue;
        std::vector<sbm_match_rec> recs((size_t)count[d] * cap);
        std::vector<int32_t> cnt((size_t)count[d] * 2);
        const int rc = sbm_match_batch_host_end(lane.ctxs[d], recs.data(), cnt.data());
        if (rc && rc != SBM_ERR_CAPACITY) {
            if (!bad) {
                bad = rc;
                msg = std::string("sbm_match_batch_host_end: ") + sbm_last_error();
            }
            continue;
        }
        if (bad) continue;
        for (int f = 0; f < count[d]; ++f) { // see wait(): frames whose list did not fit are matched again one at a time
            if (cnt[2 * f] < 0 || cnt[2 * f] > cap || cnt[2 * f + 1] != 0) redo.push_back((size_t)(first[d] + f));
            else out[(size_t)(first[d] + f)] = to_matches(*flat, recs.data() + (size_t)f * cap, cnt[2 * f]);
        }
    }
    if (bad) CV_Error(bad == SBM_ERR_INVALID ? Error::StsBadArg : Error::StsError, msg);
    if (!redo.empty()) {
        Engine::Lane single;
        single.ctxs.assign(1, lane.ctxs[0]);
        for (size_t f : redo) {
            const int64_t k = match_on_lane(single, sources[f], mask8, threshold);
            out[f] = to_matches(*flat, single.recs.data(), k);
        }
    }
    return out;
}

void Detector::pinBuffer(const Mat& frame) const
{
    CV_Assert(!frame.empty());
    Engine& e = *eng_;
    LaneLease lease{&e, acquire_lane(e)};
    Engine::Lane& lane = *e.lanes[(size_t)lease.idx];
    if (lane.ctxs.empty()) { // a lane nobody matched on yet: give it its contexts (templates follow with the first call)
        const std::vector<int> devs = device_ids_.empty() ? std::vector<int>{device_id_} : device_ids_;
        prepare_lane(*this, e, lease.idx, devs, class_templates, pyramid_levels, T_at_level, modality->weak_threshold, std::vector<std::string>(), 0,
                     0, false);
    }
    // page-locking is process-wide (hipHostRegister): every lane's uploads from the buffer are direct DMAs; the context
    // only keeps the registration so that unpinBuffer can drop it
    check(sbm_pin_host_buffer(lane.ctxs[0], frame.data, (int64_t)frame.step * frame.rows), "sbm_pin_host_buffer");
    std::lock_guard<std::mutex> lock(e.mu);
    e.pins[frame.data] = std::make_pair(lease.idx, lane.ctxs[0]);
}

void Detector::unpinBuffer(const Mat& frame) const
{
    CV_Assert(!frame.empty());
    Engine& e = *eng_;
    int li = -1;
    {
        std::lock_guard<std::mutex> lock(e.mu);
        auto it = e.pins.find(frame.data);
        CV_Assert(it != e.pins.end());
        li = it->second.first;
    }
    LaneLease lease{&e, acquire_lane(e, li)}; // the lane whose context holds the registration, once it is free
    check(sbm_unpin_host_buffer(e.lanes[(size_t)li]->ctxs[0], frame.data), "sbm_unpin_host_buffer");
    std::lock_guard<std::mutex> lock(e.mu);
    e.pins.erase(frame.data);
}

int Detector::addTemplate(const Mat source, const std::string& class_id, const Mat& object_mask, float sscale, float orientation,
                          int tagFieldID, std::string fiducial_src, int num_features)
{
    std::vector<TemplatePyramid>& template_pyramids = class_templates[class_id];
    const int template_id = static_cast<int>(template_pyramids.size());
    TemplatePyramid tp(pyramid_levels);
    {
        Ptr<ColorGradientPyramid> qp = modality->process(source, object_mask);
        if (num_features > 0) qp->num_features = num_features;
        for (int l = 0; l < pyramid_levels; ++l) {
            if (l > 0) qp->pyrDown();
            const bool ok = qp->extractTemplate(tp[l]);
            tp[l].sscale = sscale;
            tp[l].orientation = orientation;
            tp[l].tagFieldID = tagFieldID;
            tp[l].fiducial_src = fiducial_src;
            if (!ok) return -1;
        }
    }
    crop_templates(tp);
    template_pyramids.push_back(tp);
    {
        std::lock_guard<std::mutex> lock(eng_->mu);
        eng_->flat.reset(); // class_templates changed: the next call flattens and uploads again
    }
    return template_id;
}

// addTemplate_rotate, line2Dup.cpp:1395-1451
int Detector::addTemplate_rotate(const std::string& class_id, int zero_id, float theta, Point2f center)
{
    std::vector<TemplatePyramid>& template_pyramids = class_templates[class_id];
    const int template_id = static_cast<int>(template_pyramids.size());
    CV_Assert(zero_id >= 0 && zero_id < template_id);
    const TemplatePyramid base = template_pyramids[zero_id];
    TemplatePyramid tp(pyramid_levels);
    const double ang = -theta / 180 * CV_PI;
    for (int l = 0; l < pyramid_levels; ++l) {
        if (l > 0) center /= 2;
        for (const Feature& f : base[l].features) {
            const Point2f p((float)(f.x + base[l].tl_x), (float)(f.y + base[l].tl_y));
            const Point2f q = p - center;
            Point2f r;
            r.x = (float)(std::cos(ang) * q.x - std::sin(ang) * q.y);
            r.y = (float)(std::sin(ang) * q.x + std::cos(ang) * q.y);
            r = r + center;
            Feature g;
            g.x = int(r.x + 0.5f);
            g.y = int(r.y + 0.5f);
            g.theta = f.theta - theta;
            while (g.theta > 360) g.theta -= 360;
            while (g.theta < 0) g.theta += 360;
            g.label = int(g.theta * 16 / 360 + 0.5f);
            g.label &= 7;
            tp[l].features.push_back(g);
        }
        tp[l].pyramid_level = l;
    }
    crop_templates(tp);
    template_pyramids.push_back(tp);
    {
        std::lock_guard<std::mutex> lock(eng_->mu);
        eng_->flat.reset(); // class_templates changed: the next call flattens and uploads again
    }
    return template_id;
}

const std::vector<Template>& Detector::getTemplates(const std::string& class_id, int template_id) const
{
    auto it = class_templates.find(class_id);
    CV_Assert(it != class_templates.end());
    CV_Assert(it->second.size() > size_t(template_id));
    return it->second[template_id];
}

int Detector::numTemplates() const
{
    int n = 0;
    for (const auto& kv : class_templates) n += (int)kv.second.size();
    return n;
}

int Detector::numTemplates(const std::string& class_id) const
{
    auto it = class_templates.find(class_id);
    return it == class_templates.end() ? 0 : (int)it->second.size();
}

std::vector<std::string> Detector::classIds() const
{
    std::vector<std::string> ids;
    for (const auto& kv : class_templates) ids.push_back(kv.first);
    return ids;
}

void Detector::read(const FileNode& fn)
{
    class_templates.clear();
    pyramid_levels = fn["pyramid_levels"];
    fn["T"] >> T_at_level;
    modality = makePtr<ColorGradient>();
    modality->read(fn);
    dropContext();
}

void Detector::write(FileStorage& fs) const
{
    fs << "pyramid_levels" << pyramid_levels;
    fs << "T" << T_at_level;
    modality->write(fs);
}

std::string Detector::readClass(const FileNode& fn, const std::string& class_id_override)
{
    std::string class_id = class_id_override;
    if (class_id.empty()) {
        class_id = (std::string)fn["class_id"];
        CV_Assert(class_templates.find(class_id) == class_templates.end()); // :1514
    }
    std::vector<TemplatePyramid> tps;
    FileNode tps_fn = fn["template_pyramids"];
    tps.resize(tps_fn.size());
    int expected_id = 0;
    for (FileNodeIterator it = tps_fn.begin(); it != tps_fn.end(); ++it, ++expected_id) {
        const int template_id = (*it)["template_id"];
        CV_Assert(template_id == expected_id); // :1532
        FileNode templates_fn = (*it)["templates"];
        tps[template_id].resize(templates_fn.size());
        int idx = 0;
        for (FileNodeIterator jt = templates_fn.begin(); jt != templates_fn.end(); ++jt) tps[template_id][idx++].read(*jt);
    }
    class_templates[class_id] = tps;
    {
        std::lock_guard<std::mutex> lock(eng_->mu);
        eng_->flat.reset(); // class_templates changed: the next call flattens and uploads again
    }
    return class_id;
}

void Detector::writeClass(const std::string& class_id, FileStorage& fs) const
{
    auto it = class_templates.find(class_id);
    CV_Assert(it != class_templates.end());
    fs << "class_id" << it->first;
    fs << "pyramid_levels" << pyramid_levels;
    fs << "template_pyramids" << "[";
    for (size_t i = 0; i < it->second.size(); ++i) {
        fs << "{";
        fs << "template_id" << int(i);
        fs << "templates" << "[";
        for (const Template& t : it->second[i]) {
            fs << "{";
            t.write(fs);
            fs << "}";
        }
        fs << "]";
        fs << "}";
    }
    fs << "]";
}

void Detector::readClasses(const std::vector<std::string>& class_ids, const std::string& format)
{
    for (const std::string& id : class_ids) {
        const std::string filename = cv::format(format.c_str(), id.c_str());
        FileStorage fs(filename, FileStorage::READ);
        if (!fs.isOpened()) CV_Error(Error::StsBadArg, "cannot open template file " + filename);
        readClass(fs.root());
    }
}

void Detector::writeClasses(const std::string& format) const
{
    for (const auto& kv : class_templates) {
        const std::string filename = cv::format(format.c_str(), kv.first.c_str());
        FileStorage fs(filename, FileStorage::WRITE);
        writeClass(kv.first, fs);
    }
}

// getInstance, line2Dup.cpp:1355-1393: detector config + "classes" + "templates_dir" from one YAML
Detector* Detector::getInstance() { return getInstance("model_images/detector_linemod.yaml"); }

Detector* Detector::getInstance(std::string path)
{
    if (Detector::instance) return Detector::instance;
    FileStorage fs(path, FileStorage::READ);
    if (!fs.isOpened()) {
        std::cout << "LINEMOD configuration file (" << path << ") not found!" << std::endl;
        throw std::runtime_error("LINEMOD configuration file not found");
    }
    Detector* d = new Detector();
    d->read(fs.root());
    std::vector<std::string> ids;
    FileNode classes_fn = fs["classes"];
    for (FileNodeIterator it = classes_fn.begin(); it != classes_fn.end(); ++it) ids.push_back((std::string)*it);
    const std::string dir = (std::string)fs["templates_dir"];
    d->readClasses(ids, dir + "/%s.yaml.gz"); // line2Dup.cpp:1390
    Detector::instance = d;
    return d;
}

} // namespace line2Dup

// ---- shapeInfo_producer (line2Dup.h:344-458) ------------------------------------------
namespace shape_based_matching {

shapeInfo_producer::shapeInfo_producer(cv::Mat src_, cv::Mat mask_)
{
    src = src_;
    mask = mask_.empty() ? cv::Mat(src.size(), CV_8UC1, cv::Scalar(255)) : mask_;
}

// line2Dup.h:379-405: exactly 90 / 180 / 270 degrees rotate, then resize; every other angle is resize only (the fork
// removed the warpAffine of upstream and silently ignores such angles)
cv::Mat shapeInfo_producer::transform(cv::Mat src, float angle, float scale)
{
    cv::Mat dst;
    if (std::abs(angle - 90.0) < ANGLE_TOLERANCE) {
        cv::rotate(src, dst, cv::ROTATE_90_CLOCKWISE);
        cv::resize(dst, dst, cv::Size(), scale, scale);
    } else if (std::abs(angle - 180.0) < ANGLE_TOLERANCE) {
        cv::rotate(src, dst, cv::ROTATE_180);
        cv::resize(dst, dst, cv::Size(), scale, scale);
    } else if (std::abs(angle - 270.0) < ANGLE_TOLERANCE) {
        cv::rotate(src, dst, cv::ROTATE_90_COUNTERCLOCKWISE);
        cv::resize(dst, dst, cv::Size(), scale, scale);
    } else {
        cv::resize(src, dst, cv::Size(), scale, scale);
    }
    return dst;
}

cv::Mat shapeInfo_producer::mask_of(const Info& info)
{
    cv::Mat m = transform(mask, info.angle, info.scale);
    for (int r = 0; r < m.rows; ++r)
        for (int c = 0; c < m.cols; ++c) m.ptr(r)[c] = m.ptr(r)[c] > 0 ? 255 : 0;
    return m;
}

void shapeInfo_producer::produce_infos()
{
    infos.clear();
    CV_Assert(angle_range.size() <= 2 && scale_range.size() <= 2);
    CV_Assert(angle_step > eps * 10 && scale_step > eps * 10);
    if (angle_range.empty()) angle_range.push_back(0);
    if (scale_range.empty()) scale_range.push_back(1);
    const bool a2 = angle_range.size() == 2, s2 = scale_range.size() == 2;
    if (s2) CV_Assert(scale_range[1] > scale_range[0]);
    if (a2) CV_Assert(angle_range[1] > angle_range[0]);
    // float accumulation on purpose: the reference's loops are `for (v = lo; v <= hi + eps; v += step)`
    for (float scale = scale_range[0]; scale <= (s2 ? scale_range[1] : scale_range[0]) + eps; scale += scale_step) {
        for (float angle = angle_range[0]; angle <= (a2 ? angle_range[1] : angle_range[0]) + eps; angle += angle_step) {
            infos.emplace_back(angle, scale);
            if (!a2) break;
        }
        if (!s2) break;
    }
}

void shapeInfo_producer::save_infos(std::vector<Info>& infos, std::string path)
{
    cv::FileStorage fs(path, cv::FileStorage::WRITE);
    fs << "infos" << "[";
    for (const Info& i : infos) {
        fs << "{";
        fs << "angle" << i.angle;
        fs << "scale" << i.scale;
        fs << "}";
    }
    fs << "]";
}

std::vector<shapeInfo_producer::Info> shapeInfo_producer::load_infos(std::string path)
{
    cv::FileStorage fs(path, cv::FileStorage::READ);
    std::vector<Info> infos;
    cv::FileNode n = fs["infos"];
    for (cv::FileNodeIterator it = n.begin(); it != n.end(); ++it) infos.emplace_back((float)(*it)["angle"], (float)(*it)["scale"]);
    return infos;
}

} // namespace shape_based_matching
