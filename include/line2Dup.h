/*
 * line2Dup.h — drop-in declaration of the reference's public API
 * (ddcr/shape_based_matching line2Dup.h:113-458) implemented on the MI355X
 * engine: every call that touches pixels runs hand-written HIP kernels through
 * the C ABI of include/sbm.h; this header and its implementation
 * (shape_based_matching_amd/facade/line2Dup_amd.cpp) are host C++ only.
 *
 * A reference caller (test.cpp, test_jabil.cpp) compiles against this header
 * unchanged and links libsbm_facade + libsbm_hip instead of the reference's
 * line2Dup.cpp.  With OpenCV available define SBM_USE_OPENCV; otherwise the
 * bundled cv:: subset (sbm_cvlite.h) provides Mat / FileStorage / imread(PNM).
 *
 * Differences from the reference header, all deliberate:
 *   - no global Timer class and no csv.hpp include (test.cpp defines its own Timer);
 *   - shapeInfo_producer::save_infos / load_infos are present (upstream API that
 *     test.cpp:200,339 calls and the fork deleted);
 *   - Match lists come back in a deterministic canonical order (one of the
 *     orders the reference's comparator allows; the reference's own order
 *     depends on its OpenMP thread count).
 */
#ifndef CXXLINEMOD_H
#define CXXLINEMOD_H

#ifdef SBM_USE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/highgui/highgui.hpp>
#include <opencv2/imgproc.hpp>
#else
#include "sbm_cvlite.h"
#endif

#include <cfloat>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#define ANGLE_TOLERANCE FLT_EPSILON
#define ANGLE_MULTIPLE_90 0

struct sbm_ctx; /* include/sbm.h */

/* The reference's header pulls in its CPU SIMD wrapper (MIPP/mipp.h, line2Dup.h:10) and its MIPP_test demo prints
 * these constants (test.cpp:530-536).  This engine has no CPU SIMD layer — the hot loops are HIP kernels — so the
 * names exist only to keep such callers compiling; they describe the device the work runs on. */
namespace mipp {
static const std::string InstructionType = "HIP";
static const std::string InstructionFullType = "HIP gfx950 (CDNA4), wave64";
static const std::string InstructionVersion = "1";
static const int RegisterSizeBit = 64 * 32; /* one VGPR of a 64-lane wavefront */
static const int Lanes = 1;
static const bool Support64Bit = true;
static const bool SupportByteWord = true;
} // namespace mipp
#define has_max_int8_t 1   /* v_pk_max / SWAR byte maxima in the kernels */
#define has_shuff_int8_t 1 /* v_perm_b32 */

namespace line2Dup {

/* line2Dup.h:116-129 */
struct Feature {
    int x;
    int y;
    int label;
    float theta;

    void read(const cv::FileNode& fn);
    void write(cv::FileStorage& fs) const;

    Feature() : x(0), y(0), label(0), theta(0.f) {}
    Feature(int x, int y, int label);
};
inline Feature::Feature(int _x, int _y, int _label) : x(_x), y(_y), label(_label), theta(0.f) {}

/* line2Dup.h:131-153 */
struct Template {
    int width;
    int height;
    int tl_x;
    int tl_y;
    int pyramid_level;
    std::vector<Feature> features;

    float sscale;
    float orientation;
    int tagFieldID;
    std::string fiducial_src;

    Template() : width(0), height(0), tl_x(0), tl_y(0), pyramid_level(0), sscale(0.f), orientation(0.f), tagFieldID(0) {}
    void read(const cv::FileNode& fn);
    void write(cv::FileStorage& fs) const;
};

/* line2Dup.h:155-200.  update() and pyrDown() run the HIP gradient kernels. */
class ColorGradientPyramid {
public:
    ColorGradientPyramid(const cv::Mat& src, const cv::Mat& mask, float weak_threshold, size_t num_features,
                         float strong_threshold);

    void quantize(cv::Mat& dst) const;
    bool extractTemplate(Template& templ) const;
    void pyrDown();
    void update();

    struct Candidate {
        Candidate(int x, int y, int label, float score);
        bool operator<(const Candidate& rhs) const { return score > rhs.score; }
        Feature f;
        float score;
    };

    cv::Mat src;
    cv::Mat mask;

    int pyramid_level;
    cv::Mat angle;
    cv::Mat magnitude;
    cv::Mat angle_ori;

    float weak_threshold;
    size_t num_features;
    float strong_threshold;
    static bool selectScatteredFeatures(const std::vector<Candidate>& candidates, std::vector<Feature>& features,
                                        size_t num_features, float distance);
};
inline ColorGradientPyramid::Candidate::Candidate(int x, int y, int label, float _score) : f(x, y, label), score(_score) {}

/* line2Dup.h:202-220 */
class ColorGradient {
public:
    ColorGradient();
    ColorGradient(float weak_threshold, size_t num_features, float strong_threshold);

    std::string name() const;

    float weak_threshold;
    size_t num_features;
    float strong_threshold;
    void read(const cv::FileNode& fn);
    void write(cv::FileStorage& fs) const;

    cv::Ptr<ColorGradientPyramid> process(const cv::Mat src, const cv::Mat& mask = cv::Mat()) const
    {
        return cv::makePtr<ColorGradientPyramid>(src, mask, weak_threshold, num_features, strong_threshold);
    }
};

/* line2Dup.h:222-255 */
struct Match {
    Match() : x(0), y(0), similarity(0.f), template_id(0) {}
    Match(int x, int y, float similarity, const std::string& class_id, int template_id);

    bool operator<(const Match& rhs) const
    {
        if (similarity != rhs.similarity) return similarity > rhs.similarity;
        return template_id < rhs.template_id;
    }
    bool operator==(const Match& rhs) const
    {
        return x == rhs.x && y == rhs.y && similarity == rhs.similarity && class_id == rhs.class_id;
    }

    int x;
    int y;
    float similarity;
    std::string class_id;
    int template_id;
};
inline Match::Match(int _x, int _y, float _similarity, const std::string& _class_id, int _template_id)
    : x(_x), y(_y), similarity(_similarity), class_id(_class_id), template_id(_template_id)
{
}

/* line2Dup.h:257-333 */
class Detector {
public:
    Detector();
    Detector(std::vector<int> T);
    Detector(int num_features, std::vector<int> T, float weak_thresh = 30.0f, float strong_thresh = 60.0f);
    Detector(const Detector& other);
    Detector& operator=(const Detector& other);
    ~Detector();

    static Detector* getInstance(std::string path);
    static Detector* getInstance();

    std::vector<Match> match(cv::Mat sources, float threshold,
                             const std::vector<std::string>& class_ids = std::vector<std::string>(),
                             const cv::Mat masks = cv::Mat()) const;

    int addTemplate(const cv::Mat sources, const std::string& class_id, const cv::Mat& object_mask, float sscale = -1.0,
                    float orientation = -1.0, int tagFieldID = 0, std::string fiducial_src = "none", int num_features = 0);

    int addTemplate_rotate(const std::string& class_id, int zero_id, float theta, cv::Point2f center);

    const cv::Ptr<ColorGradient>& getModalities() const { return modality; }
    int getT(int pyramid_level) const { return T_at_level[pyramid_level]; }
    int pyramidLevels() const { return pyramid_levels; }
    const std::vector<Template>& getTemplates(const std::string& class_id, int template_id) const;

    int numTemplates() const;
    int numTemplates(const std::string& class_id) const;
    int numClasses() const { return static_cast<int>(class_templates.size()); }
    std::vector<std::string> classIds() const;

    void read(const cv::FileNode& fn);
    void write(cv::FileStorage& fs) const;

    std::string readClass(const cv::FileNode& fn, const std::string& class_id_override = "");
    void writeClass(const std::string& class_id, cv::FileStorage& fs) const;

    void readClasses(const std::vector<std::string>& class_ids, const std::string& format = "templates_%s.yml.gz");
    void writeClasses(const std::string& format = "templates_%s.yml.gz") const;

    /* ---- MI355X extensions (not in the reference) ----
     * setDevice: GPU ordinal of this detector's engine context.
     * setDevices: several GPUs in ONE process -- one engine context and one host thread per listed device (an ordinal
     *   may be listed more than once).  match() then shards the selected templates over them (work-balanced contiguous
     *   ranges) and concatenates the lists, as the reference's OpenMP team over templates does (line2Dup.cpp:1166-1170);
     *   matchBatch() deals the frames over them instead.
     * matchBatch: a batch of frames of one size and type; element f is exactly what match(sources[f], ...) returns.  The
     *   frames are uploaded in sub-batches while the kernels of the previous sub-batch run (sbm_match_batch_host).
     * matchAsync / wait: the same, split: matchAsync enqueues uploads and kernels and returns, wait() returns the lists.
     *   The frames must stay alive and unchanged until wait(); one batch in flight per detector.
     * pinBuffer / unpinBuffer: page-lock a frame buffer the caller re-uses (a camera loop), so that its uploads are
     *   direct DMAs; explicit because the detector cannot know the buffer's lifetime (sbm_pin_host_buffer). */
    void setDevice(int device_id);
    void setDevices(const std::vector<int>& device_ids);
    std::vector<std::vector<Match>> matchBatch(const std::vector<cv::Mat>& sources, float threshold,
                                               const std::vector<std::string>& class_ids = std::vector<std::string>(),
                                               const cv::Mat mask = cv::Mat()) const;
    void matchAsync(const std::vector<cv::Mat>& sources, float threshold,
                    const std::vector<std::string>& class_ids = std::vector<std::string>(), const cv::Mat mask = cv::Mat()) const;
    std::vector<std::vector<Match>> wait() const;
    void pinBuffer(const cv::Mat& frame) const;
    void unpinBuffer(const cv::Mat& frame) const;
    /* Threading contract.  As in the reference (line2Dup.h:272-274: match() is const and keeps no state), match() and
     * matchBatch() may be called from several host threads on ONE Detector at the same time; every call returns exactly
     * what it would return alone.  Each concurrent caller works on its own engine context(s) ("lane": device buffers +
     * uploaded templates, created on first need and kept); setConcurrency(n) bounds how many exist (default 4; callers
     * beyond that wait for a free one; 1 = calls are serialised).  matchAsync()/wait() is one batch in flight per
     * detector.  addTemplate*, read*, setDevice(s) and setConcurrency mutate the detector and must not run concurrently
     * with anything else on it -- also as in the reference. */
    void setConcurrency(int max_concurrent_calls);
    struct Engine; /* opaque: the lane pool, defined in facade/line2Dup_amd.cpp */

protected:
    cv::Ptr<ColorGradient> modality;
    int pyramid_levels;
    std::vector<int> T_at_level;

    typedef std::vector<Template> TemplatePyramid;
    typedef std::map<std::string, std::vector<TemplatePyramid>> TemplatesMap;
    TemplatesMap class_templates;

    static Detector* instance;

private:
    /* device side (facade/line2Dup_amd.cpp): a pool of LANES, each one engine context per device with the templates
     * uploaded and its own class selection.  A match() call takes a free lane for its duration -- creating one, up to
     * setConcurrency(), when all are busy, else waiting -- so concurrent callers never share a context. */
    mutable Engine* eng_;
    int device_id_;
    std::vector<int> device_ids_;
    void dropContext();
};

} // namespace line2Dup

namespace shape_based_matching {

/* line2Dup.h:344-458 plus the upstream save_infos / load_infos */
class shapeInfo_producer {
public:
    cv::Mat src;
    cv::Mat mask;

    std::vector<float> angle_range;
    std::vector<float> scale_range;

    float angle_step = 15;
    float scale_step = 0.5;
    float eps = 0.00001f;

    class Info {
    public:
        float angle;
        float scale;
        Info(float angle_, float scale_) : angle(angle_), scale(scale_) {}
    };
    std::vector<Info> infos;

    shapeInfo_producer(cv::Mat src, cv::Mat mask = cv::Mat());

    /* As in the fork (line2Dup.h:379-405): rotations by multiples of 90 degrees
     * and scale 1 are exact pixel moves; anything else is rejected (the fork
     * removed warpAffine and this engine does not bring image warping back). */
    static cv::Mat transform(cv::Mat src, float angle, float scale);

    void produce_infos();
    cv::Mat src_of(const Info& info) { return transform(src, info.angle, info.scale); }
    cv::Mat mask_of(const Info& info);

    static void save_infos(std::vector<Info>& infos, std::string path = "infos.yaml");
    static std::vector<Info> load_infos(std::string path = "info.yaml");
};

} // namespace shape_based_matching

#endif
