// cv_compat.h -- the handful of OpenCV types that appear in the signatures of the hot-path API, for builds without
// OpenCV (this image has none).  Layouts match OpenCV 4.x: cv::DMatch 16 B {int,int,int,float}, cv::KeyPoint 28 B
// {Point2f pt; float size, angle, response; int octave, class_id}.  Define MLPL_WITH_OPENCV to compile the facade
// against the real headers instead (then cv::Mat / InputArray / OutputArray are OpenCV's own).
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

typedef const Mat &InputArray;
typedef Mat &OutputArray;
typedef Mat &InputOutputArray;
// cv::noArray(): a sentinel whose address marks "not requested"
inline Mat &noArray() {
    static thread_local Mat none;
    return none;
}
inline bool needed(const Mat &m) { return &m != &noArray(); }

}  // namespace cv
#endif  // MLPL_WITH_OPENCV
