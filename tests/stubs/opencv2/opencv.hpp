// TESTS ONLY — declaration stubs for the handful of cv:: names that the reference header
// (include/loop_closing.hpp) and adapters/opencv/loop_closing.cpp mention.  OpenCV is not in this image; these
// declarations exist so that `g++ -fsyntax-only` can check the adapter's syntax and the layout static_asserts against
// the reference's real header (tests/test_adapter_syntax.py).  Nothing is defined, nothing is linked, nothing runs:
// this pins NO behaviour and says nothing about parity.  Written from OpenCV's public API names, not from its sources.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_8UC1 0
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert: " #expr); } while (0)

namespace cv {
struct _InputArray {};
_InputArray noArray();
class Mat {
public:
    int rows = 0, cols = 0;
    bool empty() const;
    int type() const;
    bool isContinuous() const;
    Mat clone() const;
    template <typename T> T* ptr(int row = 0);
    template <typename T> const T* ptr(int row = 0) const;
};
template <typename T> class Mat_ : public Mat { public: Mat_(int rows, int cols); };
template <typename T> struct MatCommaInitializer_ {
    template <typename T2> MatCommaInitializer_& operator,(T2 v);
    operator Mat() const;
};
template <typename T, typename T2> MatCommaInitializer_<T> operator<<(const Mat_<T>& m, T2 v);
struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };
struct Point3f { float x, y, z; };
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
template <typename T> using Ptr = std::shared_ptr<T>;
class ORB {
public:
    static Ptr<ORB> create(int nfeatures = 500);
    void detectAndCompute(const Mat& image, _InputArray mask, std::vector<KeyPoint>& keypoints, Mat& descriptors);
};
class BFMatcher {};
}  // namespace cv
