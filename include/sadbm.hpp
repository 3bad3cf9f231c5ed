// sadbm.hpp -- header-only C++ drop-in over the C ABI of sadbm.h.
//
// Mirrors the call surface of /root/reference/SAD.cpp:24  `Mat getDisparity(Mat &left, Mat &right)`
// with the two file-scope parameters (SAD.cpp:9 winSize, SAD.cpp:11 DSR) made explicit arguments that
// default to the reference's values.  Two forms:
//   * sadbm::MatView / sadbm::Image -- a minimal owning 8-bit image for builds without OpenCV;
//   * cv::Mat sadbm::getDisparity(cv::Mat&, cv::Mat&, int winSize, int DSR) -- compiled only when <opencv2/core.hpp>
//     is on the include path (OpenCV is not in this image: tests compile it against a stub header, nothing runs it).
// Errors: the reference surfaces bad inputs as cv::Exception thrown from inside OpenCV; here a
// negative sadbm code becomes sadbm::Error (or cv::Exception in the cv::Mat overload).
#ifndef SADBM_HPP
#define SADBM_HPP

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "sadbm.h"

namespace sadbm {

struct Error : std::runtime_error {
    int code;
    explicit Error(int c) : std::runtime_error(std::string("sadbm: ") + sadbm_strerror(c) +
                                               (c == SADBM_E_HIP ? std::string(" (") + sadbm_last_hip_error() + ")" : "")), code(c) {}
};

constexpr int kDefaultWinSize = 7;    // SAD.cpp:9
constexpr int kDefaultDSR = 30;       // SAD.cpp:11

// Non-owning view with cv::Mat's fields (data, rows, cols, channels(), step[0]).
struct MatView {
    const uint8_t *data = nullptr;
    int rows = 0, cols = 0, channels = 1;
    int64_t step = 0;
};

// Owning CV_8UC{1,3,4} image.
struct Image {
    int rows = 0, cols = 0, channels = 1;
    std::vector<uint8_t> pix;
    Image() = default;
    Image(int r, int c, int ch = 1) : rows(r), cols(c), channels(ch), pix((size_t)r * c * ch, 0) {}
    MatView view() const { return MatView{pix.data(), rows, cols, channels, (int64_t)cols * channels}; }
    uint8_t *ptr(int r) { return pix.data() + (size_t)r * cols * channels; }
    const uint8_t *ptr(int r) const { return pix.data() + (size_t)r * cols * channels; }
};

// getDisparity on views: returns a rows x cols CV_8UC1 image of (loc*16)&255 (SAD.cpp:57), zero outside
// [0,rows-winSize) x [0,cols-winSize) (SAD.cpp:31,34,36).
inline Image getDisparity(const MatView &left, const MatView &right, int winSize = kDefaultWinSize,
                          int DSR = kDefaultDSR, uint32_t flags = 0)
{
    Image out(left.rows, left.cols, 1);
    sadbm_image l{left.data, left.rows, left.cols, left.channels, left.step};
    sadbm_image r{right.data, right.rows, right.cols, right.channels, right.step};
    sadbm_plane o{out.pix.data(), out.rows, out.cols, (int64_t)out.cols};
    int rc = sadbm_get_disparity(&l, &r, winSize, DSR, flags, &o);
    if (rc != SADBM_OK) throw Error(rc);
    return out;
}

inline Image getDisparity(const Image &left, const Image &right, int winSize = kDefaultWinSize, int DSR = kDefaultDSR)
{
    return getDisparity(left.view(), right.view(), winSize, DSR);
}

}  // namespace sadbm

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
namespace sadbm {
// The reference function on cv::Mat with winSize / DSR as arguments.  It lives in namespace sadbm and has NO default
// arguments on purpose: SAD.cpp keeps its own two-argument `getDisparity(Mat&, Mat&)` (now a one-line forwarder, see
// INTEGRATION.md), and a global overload with defaults would make the call at SAD.cpp:76 ambiguous.
inline cv::Mat getDisparity(cv::Mat &left, cv::Mat &right, int winSize, int DSR)
{
    CV_Assert(left.depth() == CV_8U && right.depth() == CV_8U);
    cv::Mat disparity(left.rows, left.cols, CV_8UC1, cv::Scalar(0));            // SAD.cpp:31
    sadbm_image l{left.data, left.rows, left.cols, left.channels(), (int64_t)left.step[0]};
    sadbm_image r{right.data, right.rows, right.cols, right.channels(), (int64_t)right.step[0]};
    sadbm_plane o{disparity.data, disparity.rows, disparity.cols, (int64_t)disparity.step[0]};
    int rc = sadbm_get_disparity(&l, &r, winSize, DSR, 0, &o);
    if (rc != SADBM_OK) CV_Error(cv::Error::StsError, sadbm::Error(rc).what());
    return disparity;
}
}  // namespace sadbm
#endif
#endif

#endif  // SADBM_HPP
