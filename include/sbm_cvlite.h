/*
 * sbm_cvlite.h — the small part of the cv:: vocabulary that line2Dup.h's API
 * is written in (Mat, Size, Point, Rect, Scalar, Ptr, String, Exception,
 * FileStorage/FileNode for the OpenCV YAML subset the template files use, and a
 * PNM imread/imwrite), for hosts without OpenCV.  When the real OpenCV is
 * present, compile with -DSBM_USE_OPENCV and include/line2Dup.h uses it
 * instead; nothing in libsbm_hip.so depends on either.
 *
 * This is host plumbing for the drop-in Detector facade: a 2-D byte/float
 * container plus file I/O.  No image processing lives here: the gradient stage
 * runs in the HIP kernels behind include/sbm.h.
 */
#ifndef SBM_CVLITE_H
#define SBM_CVLITE_H

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_MAT_DEPTH(t) ((t) & 7)
#define CV_MAT_CN(t) ((((t) >> CV_CN_SHIFT) & 63) + 1)
#define CV_PI 3.1415926535897932384626433832795

namespace cv {

typedef unsigned char uchar;
typedef unsigned short ushort;
typedef std::string String;

template <class T>
using Ptr = std::shared_ptr<T>;
template <class T, class... A>
Ptr<T> makePtr(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }

namespace Error {
enum Code { StsOk = 0, StsError = -2, StsBadArg = -5, StsAssert = -215 };
}

class Exception : public std::exception {
public:
    Exception(int c, const String& e, const String& fn, const String& fl, int ln) : code(c), err(e), func(fn), file(fl), line(ln)
    {
        std::ostringstream o;
        o << "sbm(cvlite) " << file << ":" << line << ": error: (" << code << ") " << err << " in function '" << func << "'";
        msg = o.str();
    }
    const char* what() const noexcept override { return msg.c_str(); }
    int code;
    String err, func, file, msg;
    int line;
};
inline void error(int code, const String& err, const char* func, const char* file, int line) { throw Exception(code, err, func, file, line); }
inline String format(const char* fmt, ...)
{
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return buf;
}
} // namespace cv
#define CV_Error(code, msg) cv::error(code, msg, __func__, __FILE__, __LINE__)
#define CV_Assert(expr) do { if (!(expr)) cv::error(cv::Error::StsAssert, #expr, __func__, __FILE__, __LINE__); } while (0)
#define CV_DbgAssert(expr) CV_Assert(expr)

namespace cv {

template <class T>
struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
    bool operator==(const Size_& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size_& o) const { return !(*this == o); }
    T area() const { return width * height; }
};
typedef Size_<int> Size;
typedef Size_<float> Size2f;

template <class T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    template <class U> Point_(const Point_<U>& o) : x((T)o.x), y((T)o.y) {}
    Point_ operator-(const Point_& o) const { return Point_(x - o.x, y - o.y); }
    Point_ operator+(const Point_& o) const { return Point_(x + o.x, y + o.y); }
    Point_& operator/=(T d) { x /= d; y /= d; return *this; }
};
typedef Point_<int> Point;
typedef Point_<float> Point2f;

template <class T>
struct Rect_ {
    T x, y, width, height;
    Rect_() : x(0), y(0), width(0), height(0) {}
    Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
    T area() const { return width * height; }
    Rect_ operator&(const Rect_& o) const
    {
        T x1 = std::max(x, o.x), y1 = std::max(y, o.y);
        T x2 = std::min(x + width, o.x + o.width), y2 = std::min(y + height, o.y + o.height);
        return (x2 <= x1 || y2 <= y1) ? Rect_() : Rect_(x1, y1, x2 - x1, y2 - y1);
    }
};
typedef Rect_<int> Rect;

template <class T, int N>
struct Vec {
    T val[N];
    Vec() { for (int i = 0; i < N; ++i) val[i] = T(); }
    T& operator[](int i) { return val[i]; }
    const T& operator[](int i) const { return val[i]; }
};
typedef Vec<uchar, 3> Vec3b;

struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
    template <class T, int N> Scalar(const Vec<T, N>& v) { for (int i = 0; i < 4; ++i) val[i] = i < N ? (double)v.val[i] : 0.0; }
    static Scalar all(double v) { return Scalar(v, v, v, v); }
    double& operator[](int i) { return val[i]; }
    const double& operator[](int i) const { return val[i]; }
};

/* cv::RotatedRect (the demos draw the matched template's outline with it, test.cpp:402-409) */
struct RotatedRect {
    Point2f center;
    Size2f size;
    float angle;
    RotatedRect() : angle(0) {}
    RotatedRect(const Point2f& c, const Size2f& s, float a) : center(c), size(s), angle(a) {}
    void points(Point2f pts[]) const
    {
        const double ang = angle * 3.14159265358979323846 / 180.;
        const float b = (float)std::cos(ang) * 0.5f, a = (float)std::sin(ang) * 0.5f;
        pts[0].x = center.x - a * size.height - b * size.width;
        pts[0].y = center.y + b * size.height - a * size.width;
        pts[1].x = center.x + a * size.height - b * size.width;
        pts[1].y = center.y - b * size.height - a * size.width;
        pts[2].x = 2 * center.x - pts[0].x;
        pts[2].y = 2 * center.y - pts[0].y;
        pts[3].x = 2 * center.x - pts[1].x;
        pts[3].y = 2 * center.y - pts[1].y;
    }
};

/* 2-D container with reference-counted storage; ROI views share the buffer. */
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar* data = nullptr;

    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(int r, int c, int type, const Scalar& v) { create(r, c, type); setTo(v); }
    Mat(Size s, int type, const Scalar& v) { create(s.height, s.width, type); setTo(v); }
    Mat(int r, int c, int type, void* ext, size_t stp = 0) : rows(r), cols(c), data((uchar*)ext), type_(type)
    {
        step = stp ? stp : (size_t)c * elemSize();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type, Scalar::all(0)); }
    static Mat zeros(Size s, int type) { return Mat(s, type, Scalar::all(0)); }

    void create(int r, int c, int type)
    {
        if (data && r == rows && c == cols && type == type_ && isContinuous()) return;
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * elemSize();
        size_t n = step * (size_t)r;
        owner_.reset(new uchar[n ? n : 1], std::default_delete<uchar[]>());
        data = owner_.get();
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Size size() const { return Size(cols, rows); }
    int type() const { return type_; }
    int depth() const { return CV_MAT_DEPTH(type_); }
    int channels() const { return CV_MAT_CN(type_); }
    size_t elemSize1() const { static const size_t s[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return s[depth()]; }
    size_t elemSize() const { return elemSize1() * channels(); }
    size_t step1() const { return step / elemSize1(); }
    size_t total() const { return (size_t)rows * cols; }
    bool isContinuous() const { return step == (size_t)cols * elemSize() || rows <= 1; }
    uchar* ptr(int r = 0) { return data + (size_t)r * step; }
    const uchar* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return ((T*)(data + (size_t)r * step))[c]; }
    template <class T> const T& at(int r, int c) const { return ((const T*)(data + (size_t)r * step))[c]; }

    Mat& setTo(const Scalar& v)
    {
        const int cn = channels();
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c)
                for (int k = 0; k < cn; ++k) {
                    const double x = v.val[k < 4 ? k : 3];
                    switch (depth()) {
                    case CV_8U: ptr<uchar>(r)[c * cn + k] = (uchar)x; break;
                    case CV_16U: ptr<ushort>(r)[c * cn + k] = (ushort)x; break;
                    case CV_16S: ptr<short>(r)[c * cn + k] = (short)x; break;
                    case CV_32S: ptr<int>(r)[c * cn + k] = (int)x; break;
                    case CV_32F: ptr<float>(r)[c * cn + k] = (float)x; break;
                    default: CV_Error(Error::StsBadArg, "unsupported depth");
                    }
                }
        return *this;
    }
    Mat clone() const
    {
        Mat m;
        copyTo(m);
        return m;
    }
    void copyTo(Mat& dst) const
    {
        if (empty()) { dst = Mat(); return; }
        if (dst.data == data && dst.rows == rows && dst.cols == cols) return;
        if (!(dst.rows == rows && dst.cols == cols && dst.type_ == type_ && dst.data)) dst.create(rows, cols, type_);
        for (int r = 0; r < rows; ++r) memcpy(dst.ptr(r), ptr(r), (size_t)cols * elemSize());
    }
    /* a temporary ROI view as destination, as cv::OutputArray allows: img.copyTo(padded(Rect(...))) (test.cpp:276) */
    void copyTo(Mat&& dst) const { Mat& d = dst; copyTo(d); }
    void copyTo(Mat&& dst, const Mat& mask) const { Mat& d = dst; copyTo(d, mask); }
    void copyTo(Mat& dst, const Mat& mask) const
    {
        if (mask.empty()) { copyTo(dst); return; }
        CV_Assert(mask.rows == rows && mask.cols == cols && mask.type() == CV_8UC1);
        if (!(dst.rows == rows && dst.cols == cols && dst.type_ == type_ && dst.data)) { dst.create(rows, cols, type_); dst.setTo(Scalar::all(0)); }
        const size_t es = elemSize();
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c)
                if (mask.ptr(r)[c]) memcpy(dst.ptr(r) + c * es, ptr(r) + c * es, es);
    }
    Mat operator()(const Rect& roi) const
    {
        CV_Assert(roi.x >= 0 && roi.y >= 0 && roi.x + roi.width <= cols && roi.y + roi.height <= rows);
        Mat m;
        m.rows = roi.height;
        m.cols = roi.width;
        m.type_ = type_;
        m.step = step;
        m.owner_ = owner_;
        m.data = data + (size_t)roi.y * step + (size_t)roi.x * elemSize();
        return m;
    }

private:
    int type_ = 0;
    std::shared_ptr<uchar> owner_;
};

/* ---------------------------------------------------------------------- */
/* FileStorage / FileNode: the OpenCV YAML 1.0 subset of the template files  */
/* (line2Dup.cpp:42-113, 1489-1599).                                        */
/* ---------------------------------------------------------------------- */
namespace detail {
struct YNode {
    enum Kind { NONE, SCALAR, SEQ, MAP } kind = NONE;
    std::string scalar;
    std::vector<YNode> seq;
    std::vector<std::pair<std::string, YNode>> map;
    const YNode* find(const std::string& k) const
    {
        for (auto& kv : map) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
YNode parse_yaml(const std::string& text);
std::string read_text_file(const std::string& path, bool* ok);
bool write_text_file(const std::string& path, const std::string& text);
} // namespace detail

class FileNode;
class FileNodeIterator {
public:
    FileNodeIterator(const detail::YNode* n = nullptr, size_t i = 0) : n_(n), i_(i) {}
    FileNode operator*() const;
    FileNodeIterator& operator++() { ++i_; return *this; }
    bool operator!=(const FileNodeIterator& o) const { return n_ != o.n_ || i_ != o.i_; }
    bool operator==(const FileNodeIterator& o) const { return !(*this != o); }
    FileNodeIterator& operator>>(int& v);
    FileNodeIterator& operator>>(float& v);
    FileNodeIterator& operator>>(std::string& v);

private:
    const detail::YNode* n_;
    size_t i_;
};

class FileNode {
public:
    FileNode(const detail::YNode* n = nullptr) : n_(n) {}
    FileNode operator[](const std::string& key) const { return FileNode(n_ && n_->kind == detail::YNode::MAP ? n_->find(key) : nullptr); }
    FileNode operator[](const char* key) const { return (*this)[std::string(key)]; }
    bool empty() const { return n_ == nullptr || n_->kind == detail::YNode::NONE; }
    size_t size() const { return !n_ ? 0 : n_->kind == detail::YNode::SEQ ? n_->seq.size() : n_->kind == detail::YNode::MAP ? n_->map.size() : (n_->kind == detail::YNode::SCALAR ? 1 : 0); }
    FileNodeIterator begin() const { return FileNodeIterator(n_, 0); }
    FileNodeIterator end() const { return FileNodeIterator(n_, n_ && n_->kind == detail::YNode::SEQ ? n_->seq.size() : 0); }
    /* missing keys read as 0 / "" exactly as cv::FileNode does */
    operator int() const { return n_ && n_->kind == detail::YNode::SCALAR ? (int)std::strtod(n_->scalar.c_str(), nullptr) : 0; }
    operator float() const { return n_ && n_->kind == detail::YNode::SCALAR ? (float)std::strtod(n_->scalar.c_str(), nullptr) : 0.f; }
    operator double() const { return n_ && n_->kind == detail::YNode::SCALAR ? std::strtod(n_->scalar.c_str(), nullptr) : 0.0; }
    operator std::string() const { return n_ && n_->kind == detail::YNode::SCALAR ? n_->scalar : std::string(); }
    const detail::YNode* raw() const { return n_; }

private:
    const detail::YNode* n_;
};
inline FileNode FileNodeIterator::operator*() const { return FileNode(n_ && i_ < n_->seq.size() ? &n_->seq[i_] : nullptr); }
inline FileNodeIterator& FileNodeIterator::operator>>(int& v) { v = (int)**this; ++i_; return *this; }
inline FileNodeIterator& FileNodeIterator::operator>>(float& v) { v = (float)**this; ++i_; return *this; }
inline FileNodeIterator& FileNodeIterator::operator>>(std::string& v) { v = (std::string) * *this; ++i_; return *this; }
inline void operator>>(const FileNode& n, std::string& v) { v = (std::string)n; }
inline void operator>>(const FileNode& n, int& v) { v = (int)n; }
inline void operator>>(const FileNode& n, float& v) { v = (float)n; }
inline void operator>>(const FileNode& n, std::vector<int>& v)
{
    v.clear();
    for (FileNodeIterator it = n.begin(); it != n.end(); ++it) v.push_back((int)*it);
}

class FileStorage {
public:
    enum Mode { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string& filename, int mode) { open(filename, mode); }
    ~FileStorage() { release(); }
    bool open(const std::string& filename, int mode);
    bool isOpened() const { return opened_; }
    void release();
    FileNode root() const { return FileNode(&doc_); }
    FileNode operator[](const std::string& key) const { return root()[key]; }
    FileNode operator[](const char* key) const { return root()[std::string(key)]; }
    /* writer: the "key" << value / "[" ... "]" / "{" ... "}" / "[:" protocol of cv::FileStorage */
    FileStorage& put(const std::string& token);
    FileStorage& putScalar(const std::string& text, bool quote);

private:
    struct Frame { char kind; bool flow; int count; };
    bool opened_ = false, writing_ = false, expect_key_ = true;
    std::string path_, pending_key_;
    std::ostringstream out_;
    std::vector<Frame> stack_;
    detail::YNode doc_;
    void begin_value();
    void indent();
};
inline FileStorage& operator<<(FileStorage& fs, const char* s) { return fs.put(s); }
inline FileStorage& operator<<(FileStorage& fs, const std::string& s) { return fs.put(s); }
inline FileStorage& operator<<(FileStorage& fs, int v) { return fs.putScalar(std::to_string(v), false); }
inline FileStorage& operator<<(FileStorage& fs, float v)
{
    char b[64];
    snprintf(b, sizeof b, "%.8g", (double)v);
    std::string t(b);
    if (t.find_first_of(".eEn") == std::string::npos) t += ".";
    return fs.putScalar(t, false);
}
inline FileStorage& operator<<(FileStorage& fs, double v) { return fs << (float)v; }
inline FileStorage& operator<<(FileStorage& fs, const std::vector<int>& v)
{
    fs.put("[:");
    for (int x : v) fs << x;
    return fs.put("]");
}

/* imgproc / highgui names the reference's demo drivers use (test.cpp).  cvtColor is real; the drawing and window
 * calls are accepted and do nothing: this build has no display (link the real OpenCV with -DSBM_USE_OPENCV to see
 * pictures). */
enum { COLOR_BGR2GRAY = 6, COLOR_GRAY2BGR = 8, FONT_HERSHEY_PLAIN = 1 };
inline void cvtColor(const Mat& src, Mat& dst, int code)
{
    if (code == COLOR_GRAY2BGR) {
        CV_Assert(src.type() == CV_8UC1);
        Mat out(src.rows, src.cols, CV_8UC3);
        for (int r = 0; r < src.rows; ++r)
            for (int c = 0; c < src.cols; ++c) out.ptr(r)[3 * c] = out.ptr(r)[3 * c + 1] = out.ptr(r)[3 * c + 2] = src.ptr(r)[c];
        dst = out;
    } else if (code == COLOR_BGR2GRAY) {
        CV_Assert(src.type() == CV_8UC3);
        Mat out(src.rows, src.cols, CV_8UC1);
        for (int r = 0; r < src.rows; ++r)
            for (int c = 0; c < src.cols; ++c) {
                const uchar* p = src.ptr(r) + 3 * c; /* cv::cvtColor 8-bit: (B*1868 + G*9617 + R*4899 + 2^13) >> 14 */
                out.ptr(r)[c] = (uchar)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
            }
        dst = out;
    } else {
        CV_Error(Error::StsBadArg, "cvtColor: unsupported conversion");
    }
}
enum { INTER_NEAREST = 0, INTER_LINEAR = 1 };
enum { ROTATE_90_CLOCKWISE = 0, ROTATE_180 = 1, ROTATE_90_COUNTERCLOCKWISE = 2 };
/* cv::resize for 8-bit images, INTER_LINEAR (OpenCV's two 11-bit fixed-point passes) and INTER_NEAREST; dsize empty =>
 * (cvRound(cols*fx), cvRound(rows*fy)).  facade/cvlite.cpp */
void resize(const Mat& src, Mat& dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void rotate(const Mat& src, Mat& dst, int rotateCode);
inline void circle(Mat&, Point, int, const Scalar&, int = 1, int = 8, int = 0) {}
inline void line(Mat&, Point, Point, const Scalar&, int = 1, int = 8, int = 0) {}
inline void rectangle(Mat&, Point, Point, const Scalar&, int = 1, int = 8, int = 0) {}
inline void rectangle(Mat&, Rect, const Scalar&, int = 1, int = 8, int = 0) {}
inline void putText(Mat&, const String&, Point, int, double, const Scalar&, int = 1, int = 8, bool = false) {}
inline void imshow(const String&, const Mat&) {}
inline int waitKey(int = 0) { return -1; }

/* PNM (P5 / P6) only: enough to feed frames to the demos without libpng. */
enum { IMREAD_COLOR = 1, IMREAD_GRAYSCALE = 0, IMREAD_UNCHANGED = -1 };
Mat imread(const std::string& path, int flags = IMREAD_COLOR);
bool imwrite(const std::string& path, const Mat& img);

} // namespace cv
#endif /* SBM_CVLITE_H */
