// TEST STUB -- not OpenCV and not part of the product.  The smallest cv:: surface that include/sadbm.hpp's cv::Mat
// adapter touches, so that tests/test_abi.py can COMPILE that adapter (OpenCV itself is not installed here; without this
// a typo in the adapter would never be seen).  Nothing links or runs against it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#define CV_8U 0
#define CV_8UC1 0
#define CV_MAT_DEPTH(t) ((t) & 7)
namespace cv {
struct Scalar { double v[4]; Scalar(double a = 0) : v{a, 0, 0, 0} {} };
struct MatStep { size_t p[2]; size_t operator[](int i) const { return p[i]; } };
struct Mat {
    unsigned char *data = nullptr;
    int rows = 0, cols = 0, flags = 0, cn = 1;
    MatStep step{{0, 0}};
    Mat() {}
    Mat(int r, int c, int type, const Scalar &) : rows(r), cols(c), flags(type) {}
    int depth() const { return CV_MAT_DEPTH(flags); }
    int channels() const { return cn; }
};
class Exception : public std::runtime_error { public: using std::runtime_error::runtime_error; };
namespace Error { enum { StsError = -2 }; }
}  // namespace cv
#define CV_Assert(e) do { if (!(e)) throw cv::Exception(#e); } while (0)
#define CV_Error(code, msg) throw cv::Exception(std::string(msg))
