// cv_compat.h -- the handful of OpenCV types that appear in the signatures of the hot-path API, for builds without
// OpenCV (this image has none).  Layouts match OpenCV 4.x: cv::DMatch 16 B {int,int,int,float}, cv::KeyPoint 28 B
// {Point2f pt; float size, angle, response; int octave, class_id}.  Define MLPL_WITH_OPENCV to compile the facade
// against the real headers instead (then cv::Mat / InputArray / OutputArray are OpenCV's own; the facade only uses the member API
// both provide).  A Mat here shares its buffer on copy, like cv::Mat (getMat() returns a header onto the same data).
#pragma once

#ifdef MLPL_WITH_OPENCV
#include <opencv2/core.hpp>
#else

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6

namespace cv {

struct Exception : public std::runtime_error {
    explicit Exception(const std::string &m) : std::runtime_error(m) {}
};
#define CV_Assert(expr) \
    do {                \
        if (!(expr)) throw cv::Exception(std::string("CV_Assert failed: ") + #expr); \
    } while (0)

struct Point2f {
    float x = 0, y = 0;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
struct DMatch {
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 3.402823466e+38f;
};
static_assert(sizeof(DMatch) == 16, "cv::DMatch layout");
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// Minimal dense 2-D single-channel matrix with OpenCV's row/step semantics (rows may be strided views).
class Mat {
   public:
    int rows = 0, cols = 0;
    size_t step = 0;  // bytes per row
    unsigned char *data = nullptr;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t step_bytes = 0) : rows(r), cols(c), data((unsigned char *)ext), type_(type) {
        step = step_bytes ? step_bytes : (size_t)c * elemSize();
    }
    void create(int r, int c, int type) {
        if (rows == r && cols == c && type_ == type && data) return;
        rows = r, cols = c, type_ = type;
        step = (size_t)c * elemSize();
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * step);
        data = buf_->data();
    }
    static Mat zeros(int r, int c, int type) {
        Mat m(r, c, type);
        if (m.data) std::memset(m.data, 0, (size_t)r * m.step);
        return m;
    }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    size_t elemSize() const { return type_ == CV_8U ? 1 : (type_ == CV_32F ? 4 : 8); }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <typename T>
    T &at(int r, int c = 0) { return *reinterpret_cast<T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T>
    const T &at(int r, int c = 0) const { return *reinterpret_cast<const T *>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T>
    T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * step); }
    template <typename T>
    const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * step); }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * elemSize());
        return m;
    }

   private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};

// Proxy argument types with the member API of OpenCV's cv::_InputArray / _OutputArray / _InputOutputArray (core/mat.hpp) that the
// facade uses -- getMat(), empty(), needed(), create(), type(), rows(), cols(), total() -- so that facade.cpp is written once against
// that API and compiles unchanged against the real classes (MLPL_WITH_OPENCV).  Same class names and the same
// `typedef const _InputArray& InputArray` shape as OpenCV, so the facade's signatures read exactly like the reference's.
class _InputArray {
   public:
    _InputArray() = default;
    _InputArray(const Mat &m) : m_(const_cast<Mat *>(&m)) {}  // NOLINT: implicit, as in OpenCV
    Mat getMat(int = -1) const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
    int type(int = -1) const { return m_ ? m_->type() : 0; }
    int rows(int = -1) const { return m_ ? m_->rows : 0; }
    int cols(int = -1) const { return m_ ? m_->cols : 0; }
    size_t total(int = -1) const { return m_ ? (size_t)m_->rows * (size_t)m_->cols : 0; }

   protected:
    Mat *m_ = nullptr;  // null = cv::noArray()
};
class _OutputArray : public _InputArray {
   public:
    _OutputArray() = default;
    _OutputArray(Mat &m) : _InputArray(m) {}  // NOLINT
    bool needed() const { return m_ != nullptr; }
    void create(int rows, int cols, int type) const {
        if (m_) m_->create(rows, cols, type);
    }
    Mat &getMatRef(int = -1) const { return *m_; }
    void release() const {
        if (m_) *m_ = Mat();
    }
};
class _InputOutputArray : public _OutputArray {
   public:
    _InputOutputArray() = default;
    _InputOutputArray(Mat &m) : _OutputArray(m) {}  // NOLINT
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
typedef const _InputOutputArray &InputOutputArray;
inline const _InputOutputArray &noArray() {
    static const _InputOutputArray none;
    return none;
}

}  // namespace cv
#endif  // MLPL_WITH_OPENCV
