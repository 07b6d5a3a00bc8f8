// cvlite.cpp — implementation of include/sbm_cvlite.h: OpenCV-FileStorage YAML
// subset (reader + writer) and PNM image I/O.  Host plumbing only.
#include "../csrc/sbm_resize_table.h"
#include "../../include/sbm_cvlite.h"

#include <cctype>

#include <zlib.h>

namespace cv {
namespace detail {

namespace {
struct Line {
    int indent;
    std::string text;
};

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

std::string unquote(const std::string& s)
{
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) {
        std::string o;
        for (size_t i = 1; i + 1 < s.size(); ++i) {
            if (s[i] == '\\' && i + 2 < s.size()) {
                ++i;
                o += s[i] == 'n' ? '\n' : s[i] == 't' ? '\t' : s[i];
            } else
                o += s[i];
        }
        return o;
    }
    return s;
}

// split a flow collection body at top-level commas
std::vector<std::string> split_flow(const std::string& body)
{
    std::vector<std::string> parts;
    int depth = 0;
    char quote = 0;
    std::string cur;
    for (char ch : body) {
        if (quote) {
            cur += ch;
            if (ch == quote) quote = 0;
            continue;
        }
        if (ch == '"' || ch == '\'') { quote = ch; cur += ch; continue; }
        if (ch == '[' || ch == '{') ++depth;
        if (ch == ']' || ch == '}') --depth;
        if (ch == ',' && depth == 0) { parts.push_back(trim(cur)); cur.clear(); continue; }
        cur += ch;
    }
    if (!trim(cur).empty()) parts.push_back(trim(cur));
    return parts;
}

size_t find_key_colon(const std::string& s)
{
    char quote = 0;
    for (size_t i = 0; i < s.size(); ++i) {
        char ch = s[i];
        if (quote) { if (ch == quote) quote = 0; continue; }
        if (ch == '"' || ch == '\'') { quote = ch; continue; }
        if (ch == '[' || ch == '{') return std::string::npos;
        if (ch == ':' && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
    }
    return std::string::npos;
}

YNode parse_inline(const std::string& s0)
{
    const std::string s = trim(s0);
    YNode n;
    if (s.empty()) return n;
    if (s.front() == '[' && s.back() == ']') {
        n.kind = YNode::SEQ;
        for (auto& p : split_flow(s.substr(1, s.size() - 2))) n.seq.push_back(parse_inline(p));
        return n;
    }
    if (s.front() == '{' && s.back() == '}') {
        n.kind = YNode::MAP;
        for (auto& p : split_flow(s.substr(1, s.size() - 2))) {
            size_t c = p.find(':');
            if (c == std::string::npos) continue;
            n.map.emplace_back(unquote(trim(p.substr(0, c))), parse_inline(p.substr(c + 1)));
        }
        return n;
    }
    n.kind = YNode::SCALAR;
    n.scalar = unquote(s);
    return n;
}

bool is_dash(const std::string& t) { return !t.empty() && t[0] == '-' && (t.size() == 1 || t[1] == ' '); }

YNode parse_block(std::vector<Line>& L, size_t& pos, int indent)
{
    YNode n;
    if (pos >= L.size()) return n;
    if (is_dash(L[pos].text)) {
        n.kind = YNode::SEQ;
        while (pos < L.size() && L[pos].indent == indent && is_dash(L[pos].text)) {
            std::string rest = trim(L[pos].text.substr(1));
            if (rest.empty()) {
                ++pos;
                if (pos < L.size() && L[pos].indent > indent) n.seq.push_back(parse_block(L, pos, L[pos].indent));
                else n.seq.push_back(YNode());
            } else if (find_key_colon(rest) != std::string::npos) { // "- key: value" compact mapping
                L[pos].indent = indent + 2;
                L[pos].text = rest;
                n.seq.push_back(parse_block(L, pos, indent + 2));
            } else {
                n.seq.push_back(parse_inline(rest));
                ++pos;
            }
        }
        return n;
    }
    n.kind = YNode::MAP;
    while (pos < L.size() && L[pos].indent == indent && !is_dash(L[pos].text)) {
        const std::string& t = L[pos].text;
        size_t c = find_key_colon(t);
        if (c == std::string::npos) { ++pos; continue; }
        std::string key = unquote(trim(t.substr(0, c)));
        std::string val = trim(t.substr(c + 1));
        if (val.empty()) {
            ++pos;
            if (pos < L.size() && (L[pos].indent > indent || (L[pos].indent == indent && is_dash(L[pos].text))))
                n.map.emplace_back(key, parse_block(L, pos, L[pos].indent));
            else
                n.map.emplace_back(key, YNode());
        } else {
            n.map.emplace_back(key, parse_inline(val));
            ++pos;
        }
    }
    return n;
}
} // namespace

YNode parse_yaml(const std::string& text)
{
    std::vector<Line> lines;
    std::istringstream in(text);
    std::string raw;
    while (std::getline(in, raw)) {
        if (!raw.empty() && raw.back() == '\r') raw.pop_back();
        size_t ind = 0;
        while (ind < raw.size() && raw[ind] == ' ') ++ind;
        std::string t = trim(raw);
        if (t.empty() || t[0] == '#' || t[0] == '%' || t == "---" || t == "...") continue;
        lines.push_back({(int)ind, t});
    }
    size_t pos = 0;
    if (lines.empty()) return YNode();
    return parse_block(lines, pos, lines[0].indent);
}

static bool ends_with_gz(const std::string& p) { return p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0; }

// plain or gzip-compressed text (the reference writes templates as "%s.yaml.gz", test_jabil.cpp:114)
std::string read_text_file(const std::string& path, bool* ok)
{
    if (ends_with_gz(path)) {
        gzFile g = gzopen(path.c_str(), "rb");
        if (ok) *ok = g != nullptr;
        std::string out;
        if (!g) return out;
        char buf[1 << 16];
        int n;
        while ((n = gzread(g, buf, sizeof buf)) > 0) out.append(buf, (size_t)n);
        gzclose(g);
        return out;
    }
    std::ifstream f(path, std::ios::binary);
    if (ok) *ok = (bool)f;
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

bool write_text_file(const std::string& path, const std::string& text)
{
    if (ends_with_gz(path)) {
        gzFile g = gzopen(path.c_str(), "wb");
        if (!g) return false;
        const bool ok = gzwrite(g, text.data(), (unsigned)text.size()) == (int)text.size();
        gzclose(g);
        return ok;
    }
    std::ofstream f(path, std::ios::binary);
    f << text;
    return (bool)f;
}
} // namespace detail

// ---------------------------------------------------------------------------
bool FileStorage::open(const std::string& filename, int mode)
{
    release();
    path_ = filename;
    writing_ = (mode & 1) != 0;
    if (writing_) {
        out_.str("");
        out_ << "%YAML:1.0\n---\n";
        stack_.clear();
        stack_.push_back({'m', false, 0});
        expect_key_ = true;
        opened_ = true;
    } else {
        bool ok = false;
        std::string text = detail::read_text_file(filename, &ok);
        if (!ok) { opened_ = false; doc_ = detail::YNode(); return false; }
        doc_ = detail::parse_yaml(text);
        opened_ = true;
    }
    return opened_;
}

void FileStorage::release()
{
    if (opened_ && writing_) detail::write_text_file(path_, out_.str());
    opened_ = false;
    writing_ = false;
}

void FileStorage::indent()
{
    int depth = 0;
    for (size_t i = 1; i < stack_.size(); ++i) depth += 3;
    out_ << std::string(depth, ' ');
}

// emit whatever precedes a value in the current context (key / dash / comma)
void FileStorage::begin_value()
{
    Frame& f = stack_.back();
    if (f.flow) {
        out_ << (f.count ? ", " : " ");
        if (f.kind == 'm') out_ << pending_key_ << ":";
    } else {
        indent();
        if (f.kind == 'm') out_ << pending_key_ << ":";
        else out_ << "-";
    }
    f.count++;
}

FileStorage& FileStorage::putScalar(const std::string& text, bool quote)
{
    CV_Assert(opened_ && writing_);
    Frame& f = stack_.back();
    if (f.kind == 'm' && expect_key_) CV_Error(Error::StsBadArg, "FileStorage: a key (string) was expected");
    begin_value();
    std::string t = quote ? "\"" + text + "\"" : text;
    if (stack_.back().flow) out_ << (stack_.back().kind == 'm' ? "" : "") << t;
    else out_ << " " << t << "\n";
    expect_key_ = true;
    return *this;
}

FileStorage& FileStorage::put(const std::string& tok)
{
    CV_Assert(opened_ && writing_);
    Frame& f = stack_.back();
    const bool is_open = tok == "[" || tok == "{" || tok == "[:" || tok == "{:";
    const bool is_close = tok == "]" || tok == "}";
    if (is_close) {
        Frame done = stack_.back();
        stack_.pop_back();
        if (done.flow) {
            out_ << (done.kind == 's' ? " ]" : " }");
            if (!stack_.back().flow) out_ << "\n";
        }
        expect_key_ = true;
        return *this;
    }
    if (f.kind == 'm' && expect_key_ && !is_open) {
        pending_key_ = tok;
        expect_key_ = false;
        return *this;
    }
    if (is_open) {
        begin_value();
        const bool flow = tok.size() == 2 || stack_.back().flow;
        const char kind = tok[0] == '[' ? 's' : 'm';
        if (flow) out_ << (stack_.back().flow ? "" : " ") << (kind == 's' ? "[" : "{");
        else out_ << "\n";
        stack_.push_back({kind, flow, 0});
        expect_key_ = true;
        return *this;
    }
    // a string value
    bool plain = !tok.empty();
    for (char ch : tok) plain = plain && (isalnum((unsigned char)ch) || ch == '_' || ch == '.' || ch == '/' || ch == '-');
    if (!tok.empty() && (isdigit((unsigned char)tok[0]) || tok[0] == '-' || tok[0] == '.')) plain = false;
    return putScalar(tok, !plain);
}

// ---------------------------------------------------------------------------
static bool pnm_token(std::istream& f, std::string& tok)
{
    tok.clear();
    int ch;
    while ((ch = f.get()) != EOF) {
        if (ch == '#') { while ((ch = f.get()) != EOF && ch != '\n') {} continue; }
        if (isspace(ch)) { if (!tok.empty()) return true; continue; }
        tok += (char)ch;
    }
    return !tok.empty();
}

void resize(const Mat& src, Mat& dst, Size dsize, double fx, double fy, int interpolation)
{
    CV_Assert(!src.empty() && src.depth() == CV_8U);
    const int ch = src.channels();
    int dr, dc;
    if (dsize.width > 0 && dsize.height > 0) {
        dc = dsize.width;
        dr = dsize.height;
        fx = (double)dc / src.cols;
        fy = (double)dr / src.rows;
    } else {
        CV_Assert(fx > 0 && fy > 0);
        sbm::resize_linear_dims(src.rows, src.cols, fx, fy, &dr, &dc);
    }
    CV_Assert(dr > 0 && dc > 0);
    Mat out(dr, dc, src.type());
    if (interpolation == INTER_NEAREST) {
        for (int y = 0; y < dr; ++y) {
            const int sy = std::min((int)std::floor(y / fy), src.rows - 1);
            for (int x = 0; x < dc; ++x) {
                const int sx = std::min((int)std::floor(x / fx), src.cols - 1);
                memcpy(out.ptr(y) + (size_t)x * ch, src.ptr(sy) + (size_t)sx * ch, ch);
            }
        }
    } else {
        CV_Assert(interpolation == INTER_LINEAR);
        std::vector<int32_t> xi, yi;
        std::vector<int16_t> xa, ya;
        sbm::resize_linear_table(dc, src.cols, 1.0 / fx, xi, xa);
        sbm::resize_linear_table(dr, src.rows, 1.0 / fy, yi, ya);
        for (int y = 0; y < dr; ++y) {
            const uchar* r0 = src.ptr(yi[y]);
            const uchar* r1 = src.ptr(std::min(yi[y] + 1, src.rows - 1));
            for (int x = 0; x < dc; ++x) {
                const int x0 = xi[x], x1 = std::min(x0 + 1, src.cols - 1);
                for (int k = 0; k < ch; ++k)
                    out.ptr(y)[x * ch + k] = sbm::resize_linear_sample(r0[x0 * ch + k], r0[x1 * ch + k], r1[x0 * ch + k], r1[x1 * ch + k],
                                                                       xa[2 * x], xa[2 * x + 1], ya[2 * y], ya[2 * y + 1]);
            }
        }
    }
    dst = out;
}

void rotate(const Mat& src, Mat& dst, int code)
{
    CV_Assert(!src.empty());
    const size_t es = src.elemSize();
    const bool swap = code != ROTATE_180;
    Mat out(swap ? src.cols : src.rows, swap ? src.rows : src.cols, src.type());
    for (int r = 0; r < src.rows; ++r)
        for (int c = 0; c < src.cols; ++c) {
            int rr, cc;
            if (code == ROTATE_90_CLOCKWISE) { rr = c; cc = src.rows - 1 - r; }
            else if (code == ROTATE_180) { rr = src.rows - 1 - r; cc = src.cols - 1 - c; }
            else { rr = src.cols - 1 - c; cc = r; }
            memcpy(out.ptr(rr) + cc * es, src.ptr(r) + c * es, es);
        }
    dst = out;
}

Mat imread(const std::string& path, int flags)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return Mat();
    std::string magic, w, h, mx;
    if (!pnm_token(f, magic) || (magic != "P5" && magic != "P6")) return Mat(); // PNM only (no libpng/libjpeg here)
    if (!pnm_token(f, w) || !pnm_token(f, h) || !pnm_token(f, mx)) return Mat();
    const int cols = atoi(w.c_str()), rows = atoi(h.c_str()), cn = magic == "P6" ? 3 : 1;
    if (rows <= 0 || cols <= 0 || atoi(mx.c_str()) != 255) return Mat();
    std::vector<uchar> buf((size_t)rows * cols * cn);
    f.read((char*)buf.data(), (std::streamsize)buf.size());
    if ((size_t)f.gcount() != buf.size()) return Mat();
    const bool want_color = flags == IMREAD_COLOR || (flags == IMREAD_UNCHANGED && cn == 3);
    Mat m(rows, cols, want_color ? CV_8UC3 : CV_8UC1);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const uchar* p = &buf[((size_t)r * cols + c) * cn];
            if (want_color) { // cv::imread returns BGR
                uchar* d = m.ptr(r) + c * 3;
                if (cn == 3) { d[0] = p[2]; d[1] = p[1]; d[2] = p[0]; }
                else d[0] = d[1] = d[2] = p[0];
            } else {
                m.ptr(r)[c] = cn == 1 ? p[0] : (uchar)((p[0] * 299 + p[1] * 587 + p[2] * 114 + 500) / 1000);
            }
        }
    return m;
}

bool imwrite(const std::string& path, const Mat& img)
{
    if (img.empty() || img.depth() != CV_8U || (img.channels() != 1 && img.channels() != 3)) return false;
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const int cn = img.channels();
    f << (cn == 3 ? "P6" : "P5") << "\n" << img.cols << " " << img.rows << "\n255\n";
    for (int r = 0; r < img.rows; ++r)
        for (int c = 0; c < img.cols; ++c) {
            const uchar* p = img.ptr(r) + c * cn;
            if (cn == 3) { char rgb[3] = {(char)p[2], (char)p[1], (char)p[0]}; f.write(rgb, 3); }
            else f.put((char)p[0]);
        }
    return (bool)f;
}

} // namespace cv
