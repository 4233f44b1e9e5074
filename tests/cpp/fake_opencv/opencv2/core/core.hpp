// Declarations only: the handful of OpenCV 2.4 / 3.x core types and members that this repository's headers touch in
// their ORBGPU_WITH_OPENCV branch, with OpenCV's own signatures (cv::Mat::create takes a type, cv::Mat::step converts
// to size_t, InputArray / OutputArray are const references to proxy classes).  OpenCV is not installed in this image;
// this file lets the test suite at least type-check that branch (tests/test_abi_and_host.py).  Nothing here is linked.
#ifndef ORBGPU_TEST_FAKE_OPENCV_CORE_HPP
#define ORBGPU_TEST_FAKE_OPENCV_CORE_HPP
#include <cstddef>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(expr) do { if (!(expr)) cv::error_stub(#expr); } while (0)

namespace cv {
void error_stub(const char *what);

template <typename T> struct Point_ { T x, y; };
typedef Point_<float> Point2f;

class KeyPoint {
public:
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

class Mat {
public:
    struct MStep {
        size_t buf[2];
        operator size_t() const;
    };
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void *data, size_t step = 0);
    void create(int rows, int cols, int type);
    void release();
    bool empty() const;
    int type() const;
    unsigned char *ptr(int row = 0);
    const unsigned char *ptr(int row = 0) const;
    template <typename T> T &at(int row, int col);
    template <typename T> const T &at(int row, int col) const;
    Mat row(int y) const;
    int flags, dims, rows, cols;
    unsigned char *data;
    MStep step;
};

class _InputArray {
public:
    _InputArray(const Mat &m);
    Mat getMat(int idx = -1) const;
    bool empty() const;
};
typedef const _InputArray &InputArray;

class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat &m);
    void create(int rows, int cols, int type, int i = -1, bool allowTransposed = false, int fixedDepthMask = 0) const;
    void release() const;
};
typedef const _OutputArray &OutputArray;
}  // namespace cv
#endif
