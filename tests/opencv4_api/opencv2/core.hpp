/* Declarations of the part of OpenCV 4's PUBLIC API (modules core, imgproc, imgcodecs/highgui) that include/line2Dup.h,
 * include/nms.hpp and shape_based_matching_amd/facade/ use -- written from the OpenCV 4 reference documentation, with
 * OpenCV's own signatures, NOT an implementation and not part of the product.
 *
 * Purpose (test infrastructure, tests/test_facade_opencv_mode.py): this image has no OpenCV, so the -DSBM_USE_OPENCV
 * build of the facade (the build a host WITH OpenCV uses; /root/reference/CMakeLists.txt:36 links OpenCV 4) could never
 * be compiled.  `g++ -fsyntax-only -DSBM_USE_OPENCV -I tests/opencv4_api` type-checks the facade against these
 * declarations: it catches every use of something only the bundled cv:: subset (include/sbm_cvlite.h) offers.  It
 * proves nothing about linking or behaviour; nothing here has a body except what OpenCV itself defines inline.
 */
#ifndef SBM_TEST_OPENCV4_API_CORE_HPP
#define SBM_TEST_OPENCV4_API_CORE_HPP

#include <cstddef>
#include <exception>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#define CV_VERSION_MAJOR 4
#define CV_CN_SHIFT 3
#define CV_DEPTH_MAX (1 << CV_CN_SHIFT)
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_MAT_DEPTH_MASK (CV_DEPTH_MAX - 1)
#define CV_MAT_DEPTH(flags) ((flags) & CV_MAT_DEPTH_MASK)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_MAT_CN(flags) ((((flags) & ((512 - 1) << CV_CN_SHIFT)) >> CV_CN_SHIFT) + 1)
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_16SC1 CV_MAKETYPE(CV_16S, 1)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_PI 3.1415926535897932384626433832795

namespace cv {

typedef unsigned char uchar;
typedef unsigned short ushort;
typedef std::string String;

template <typename T>
struct Ptr : public std::shared_ptr<T> {
    inline Ptr() noexcept : std::shared_ptr<T>() {}
    inline Ptr(std::nullptr_t) noexcept : std::shared_ptr<T>(nullptr) {}
    template <typename Y> inline Ptr(Y* p) : std::shared_ptr<T>(p) {}
    inline Ptr(const std::shared_ptr<T>& o) noexcept : std::shared_ptr<T>(o) {}
    inline Ptr(std::shared_ptr<T>&& o) noexcept : std::shared_ptr<T>(std::move(o)) {}
    inline bool empty() const { return std::shared_ptr<T>::get() == nullptr; }
};
template <typename T, typename... A1>
static inline Ptr<T> makePtr(const A1&... a1) { return std::make_shared<T>(a1...); }

namespace Error {
enum Code { StsOk = 0, StsBackTrace = -1, StsError = -2, StsInternal = -3, StsNoMem = -4, StsBadArg = -5, StsBadSize = -201,
            StsOutOfRange = -211, StsParseError = -212, StsNotImplemented = -213, StsAssert = -215 };
}

class Exception : public std::exception {
public:
    Exception();
    Exception(int _code, const String& _err, const String& _func, const String& _file, int _line);
    virtual ~Exception() throw();
    virtual const char* what() const throw() override;
    void formatMessage();
    String msg;
    int code;
    String err;
    String func;
    String file;
    int line;
};
[[noreturn]] void error(int _code, const String& _err, const char* _func, const char* _file, int _line);
String format(const char* fmt, ...);
int cvRound(double value);
int cvFloor(double value);
int cvCeil(double value);
template <typename T> T saturate_cast(int v);
template <typename T> T saturate_cast(float v);
template <typename T> T saturate_cast(double v);

} // namespace cv
#define CV_Error(code, msg) cv::error(code, msg, __func__, __FILE__, __LINE__)
#define CV_Assert(expr) do { if (!!(expr)) ; else cv::error(cv::Error::StsAssert, #expr, __func__, __FILE__, __LINE__); } while (0)
#define CV_DbgAssert(expr) CV_Assert(expr)

namespace cv {

template <typename _Tp> class Point_;
template <typename _Tp>
class Size_ {
public:
    typedef _Tp value_type;
    Size_();
    Size_(_Tp _width, _Tp _height);
    Size_(const Point_<_Tp>& pt);
    _Tp area() const;
    double aspectRatio() const;
    bool empty() const;
    template <typename _Tp2> operator Size_<_Tp2>() const;
    _Tp width, height;
};
template <typename _Tp> bool operator==(const Size_<_Tp>& a, const Size_<_Tp>& b);
template <typename _Tp> bool operator!=(const Size_<_Tp>& a, const Size_<_Tp>& b);
typedef Size_<int> Size2i;
typedef Size_<float> Size2f;
typedef Size2i Size;

template <typename _Tp>
class Point_ {
public:
    typedef _Tp value_type;
    Point_();
    Point_(_Tp _x, _Tp _y);
    Point_(const Size_<_Tp>& sz);
    template <typename _Tp2> operator Point_<_Tp2>() const;
    _Tp dot(const Point_& pt) const;
    _Tp x, y;
};
template <typename _Tp> Point_<_Tp> operator+(const Point_<_Tp>& a, const Point_<_Tp>& b);
template <typename _Tp> Point_<_Tp> operator-(const Point_<_Tp>& a, const Point_<_Tp>& b);
template <typename _Tp> Point_<_Tp>& operator+=(Point_<_Tp>& a, const Point_<_Tp>& b);
template <typename _Tp> Point_<_Tp>& operator-=(Point_<_Tp>& a, const Point_<_Tp>& b);
template <typename _Tp> Point_<_Tp>& operator/=(Point_<_Tp>& a, int b);
template <typename _Tp> Point_<_Tp>& operator/=(Point_<_Tp>& a, float b);
template <typename _Tp> Point_<_Tp>& operator/=(Point_<_Tp>& a, double b);
template <typename _Tp> Point_<_Tp> operator*(const Point_<_Tp>& a, double b);
template <typename _Tp> bool operator==(const Point_<_Tp>& a, const Point_<_Tp>& b);
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
typedef Point2i Point;

template <typename _Tp>
class Rect_ {
public:
    typedef _Tp value_type;
    Rect_();
    Rect_(_Tp _x, _Tp _y, _Tp _width, _Tp _height);
    Rect_(const Point_<_Tp>& org, const Size_<_Tp>& sz);
    Rect_(const Point_<_Tp>& pt1, const Point_<_Tp>& pt2);
    Point_<_Tp> tl() const;
    Point_<_Tp> br() const;
    Size_<_Tp> size() const;
    _Tp area() const;
    bool empty() const;
    bool contains(const Point_<_Tp>& pt) const;
    _Tp x, y, width, height;
};
template <typename _Tp> Rect_<_Tp> operator&(const Rect_<_Tp>& a, const Rect_<_Tp>& b);
template <typename _Tp> Rect_<_Tp> operator|(const Rect_<_Tp>& a, const Rect_<_Tp>& b);
template <typename _Tp> Rect_<_Tp>& operator&=(Rect_<_Tp>& a, const Rect_<_Tp>& b);
typedef Rect_<int> Rect2i;
typedef Rect_<float> Rect2f;
typedef Rect2i Rect;

template <typename _Tp, int cn>
class Vec {
public:
    typedef _Tp value_type;
    enum { channels = cn };
    Vec();
    Vec(_Tp v0);
    Vec(_Tp v0, _Tp v1);
    Vec(_Tp v0, _Tp v1, _Tp v2);
    Vec(_Tp v0, _Tp v1, _Tp v2, _Tp v3);
    const _Tp& operator[](int i) const;
    _Tp& operator[](int i);
    _Tp val[cn];
};
typedef Vec<uchar, 3> Vec3b;
typedef Vec<float, 3> Vec3f;

template <typename _Tp>
class Scalar_ : public Vec<_Tp, 4> {
public:
    Scalar_();
    Scalar_(_Tp v0, _Tp v1, _Tp v2 = 0, _Tp v3 = 0);
    Scalar_(_Tp v0);
    static Scalar_<_Tp> all(_Tp v0);
};
typedef Scalar_<double> Scalar;

class Mat;
class MatExpr;
class _InputArray {
public:
    _InputArray();
    _InputArray(const Mat& m);
    _InputArray(const MatExpr& expr);
    _InputArray(const double& val);
    template <typename _Tp> _InputArray(const std::vector<_Tp>& vec);
    template <typename _Tp, int cn> _InputArray(const Vec<_Tp, cn>& vec);
    _InputArray(const std::vector<Mat>& vec);
};
class _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat& m);
    _OutputArray(const Mat& m); /* fixed-size destination, e.g. src.copyTo(dst(roi)) */
    template <typename _Tp> _OutputArray(std::vector<_Tp>& vec);
    _OutputArray(std::vector<Mat>& vec);
};
class _InputOutputArray : public _OutputArray {
public:
    _InputOutputArray();
    _InputOutputArray(Mat& m);
};
typedef const _InputArray& InputArray;
typedef InputArray InputArrayOfArrays;
typedef const _OutputArray& OutputArray;
typedef OutputArray OutputArrayOfArrays;
typedef const _InputOutputArray& InputOutputArray;
InputOutputArray noArray();

struct MatSize {
    explicit MatSize(int* _p);
    int dims() const;
    Size operator()() const;
    const int& operator[](int i) const;
    int& operator[](int i);
    int* p;
};
struct MatStep {
    MatStep();
    explicit MatStep(size_t s);
    const size_t& operator[](int i) const;
    size_t& operator[](int i);
    operator size_t() const;
    MatStep& operator=(size_t s);
    size_t* p;
    size_t buf[2];
};

class Mat {
public:
    enum { AUTO_STEP = 0 };
    Mat();
    Mat(int rows, int cols, int type);
    Mat(Size size, int type);
    Mat(int rows, int cols, int type, const Scalar& s);
    Mat(Size size, int type, const Scalar& s);
    Mat(const Mat& m);
    Mat(int rows, int cols, int type, void* data, size_t step = AUTO_STEP);
    Mat(Size size, int type, void* data, size_t step = AUTO_STEP);
    Mat(const Mat& m, const Rect& roi);
    Mat(Mat&& m);
    ~Mat();
    Mat& operator=(const Mat& m);
    Mat& operator=(Mat&& m);
    Mat& operator=(const MatExpr& expr);
    Mat& operator=(const Scalar& s);
    Mat row(int y) const;
    Mat col(int x) const;
    Mat rowRange(int startrow, int endrow) const;
    Mat colRange(int startcol, int endcol) const;
    Mat clone() const;
    void copyTo(OutputArray m) const;
    void copyTo(OutputArray m, InputArray mask) const;
    void convertTo(OutputArray m, int rtype, double alpha = 1, double beta = 0) const;
    Mat& setTo(InputArray value, InputArray mask = noArray());
    Mat reshape(int cn, int rows = 0) const;
    static MatExpr zeros(int rows, int cols, int type);
    static MatExpr zeros(Size size, int type);
    static MatExpr ones(int rows, int cols, int type);
    void create(int rows, int cols, int type);
    void create(Size size, int type);
    void release();
    Mat operator()(const Rect& roi) const;
    bool isContinuous() const;
    bool isSubmatrix() const;
    size_t elemSize() const;
    size_t elemSize1() const;
    int type() const;
    int depth() const;
    int channels() const;
    size_t step1(int i = 0) const;
    bool empty() const;
    size_t total() const;
    uchar* ptr(int i0 = 0);
    const uchar* ptr(int i0 = 0) const;
    uchar* ptr(int row, int col);
    const uchar* ptr(int row, int col) const;
    template <typename _Tp> _Tp* ptr(int i0 = 0);
    template <typename _Tp> const _Tp* ptr(int i0 = 0) const;
    template <typename _Tp> _Tp& at(int row, int col);
    template <typename _Tp> const _Tp& at(int row, int col) const;
    template <typename _Tp> _Tp& at(Point pt);
    template <typename _Tp> const _Tp& at(Point pt) const;
    int flags;
    int dims;
    int rows, cols;
    uchar* data;
    MatSize size;
    MatStep step;
};
class MatExpr {
public:
    MatExpr();
    explicit MatExpr(const Mat& m);
    operator Mat() const;
    Size size() const;
    int type() const;
};
template <typename _Tp>
class Mat_ : public Mat {
public:
    Mat_();
    Mat_(int _rows, int _cols);
    Mat_(const Mat& m);
    _Tp& operator()(int row, int col);
    const _Tp& operator()(int row, int col) const;
};
MatExpr operator&(const Mat& a, const Mat& b);
MatExpr operator|(const Mat& a, const Mat& b);
MatExpr operator>(const Mat& a, double s);
MatExpr operator+(const Mat& a, const Mat& b);
MatExpr operator-(const Mat& a, const Mat& b);
MatExpr operator*(const Mat& a, double s);

enum RotateFlags { ROTATE_90_CLOCKWISE = 0, ROTATE_180 = 1, ROTATE_90_COUNTERCLOCKWISE = 2 };
void rotate(InputArray src, OutputArray dst, int rotateCode);
void flip(InputArray src, OutputArray dst, int flipCode);
void bitwise_and(InputArray src1, InputArray src2, OutputArray dst, InputArray mask = noArray());
int countNonZero(InputArray src);
void split(InputArray m, OutputArrayOfArrays mv);
void merge(InputArrayOfArrays mv, OutputArray dst);

/* ---- persistence ---- */
class FileNode;
class FileNodeIterator;
class FileStorage {
public:
    enum Mode { READ = 0, WRITE = 1, APPEND = 2, MEMORY = 4, FORMAT_MASK = (7 << 3), FORMAT_AUTO = 0, FORMAT_XML = (1 << 3),
                FORMAT_YAML = (2 << 3), FORMAT_JSON = (3 << 3), BASE64 = 64, WRITE_BASE64 = BASE64 | WRITE };
    FileStorage();
    FileStorage(const String& filename, int flags, const String& encoding = String());
    virtual ~FileStorage();
    virtual bool open(const String& filename, int flags, const String& encoding = String());
    virtual bool isOpened() const;
    virtual void release();
    virtual String releaseAndGetString();
    FileNode getFirstTopLevelNode() const;
    FileNode root(int streamidx = 0) const;
    FileNode operator[](const String& nodename) const;
    FileNode operator[](const char* nodename) const;
    void write(const String& name, int val);
    void write(const String& name, double val);
    void write(const String& name, const String& val);
    void startWriteStruct(const String& name, int flags, const String& typeName = String());
    void endWriteStruct();
};
class FileNode {
public:
    enum { NONE = 0, INT = 1, REAL = 2, FLOAT = REAL, STR = 3, STRING = STR, SEQ = 4, MAP = 5, TYPE_MASK = 7 };
    FileNode();
    FileNode(const FileNode& node);
    FileNode& operator=(const FileNode& node);
    FileNode operator[](const String& nodename) const;
    FileNode operator[](const char* nodename) const;
    FileNode operator[](int i) const;
    std::vector<String> keys() const;
    int type() const;
    bool empty() const;
    bool isNone() const;
    bool isSeq() const;
    bool isMap() const;
    bool isInt() const;
    bool isReal() const;
    bool isString() const;
    bool isNamed() const;
    std::string name() const;
    size_t size() const;
    operator int() const;
    operator float() const;
    operator double() const;
    operator std::string() const;
    FileNodeIterator begin() const;
    FileNodeIterator end() const;
    double real() const;
    std::string string() const;
};
class FileNodeIterator {
public:
    FileNodeIterator();
    FileNodeIterator(const FileNodeIterator& it);
    FileNodeIterator& operator=(const FileNodeIterator& it);
    FileNode operator*() const;
    FileNodeIterator& operator++();
    FileNodeIterator operator++(int);
    FileNodeIterator& operator+=(int ofs);
    size_t remaining() const;
    bool equalTo(const FileNodeIterator& it) const;
};
bool operator==(const FileNodeIterator& it1, const FileNodeIterator& it2);
bool operator!=(const FileNodeIterator& it1, const FileNodeIterator& it2);
template <typename _Tp> FileStorage& operator<<(FileStorage& fs, const _Tp& value);
FileStorage& operator<<(FileStorage& fs, const String& str);
FileStorage& operator<<(FileStorage& fs, const char* str);
FileStorage& operator<<(FileStorage& fs, char* value);
template <typename _Tp> void operator>>(const FileNode& n, _Tp& value);
template <typename _Tp> void operator>>(const FileNode& n, std::vector<_Tp>& vec);
template <typename _Tp> FileNodeIterator& operator>>(FileNodeIterator& it, _Tp& value);
void read(const FileNode& node, int& value, int default_value);
void read(const FileNode& node, float& value, float default_value);
void read(const FileNode& node, std::string& value, const std::string& default_value);

} // namespace cv
#endif
