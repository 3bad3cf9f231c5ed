// hostcopy.hpp -- host-side row copies of the host API (sadbm_get_disparity, sadbm_compute_multi, the streaming pipeline):
// plain C++ (no HIP), so the sanitizer build compiles it on its own (make asan -> hostcopy_asan_selftest).
#ifndef SADBM_HOSTCOPY_HPP
#define SADBM_HOSTCOPY_HPP
#include <stdint.h>
#include <string.h>
#include <condition_variable>
#include <functional>
#include <immintrin.h>
#include <mutex>
#include <thread>
#include <vector>

namespace sadbm_hostcopy {

// Row-wise host copy between a caller's (pageable, strided) image and pinned staging.  Large frames are split over
// a few threads: one core copies ~10 GB/s, which would otherwise bound the host API (a 4K pair is 25 MB per call).
// Channel 0 of `cols` interleaved pixels (ch = 3: BGR as cv::imread hands the reference, SAD.cpp:65-66; ch = 4: BGRA).  The
// matcher reads channel 0 only (SAD.cpp:19-20), so the host API gathers that plane while it copies the caller's rows into
// pinned memory anyway: a third / a quarter of the bytes then cross PCIe.  SSSE3 byte shuffles, 16 pixels per step.
__attribute__((target("ssse3"))) inline void extract_ch0_ssse3(uint8_t *dst, const uint8_t *src, int cols, int ch)
{
    int x = 0;
    if (ch == 3) {
        const __m128i m0 = _mm_setr_epi8(0, 3, 6, 9, 12, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
        const __m128i m1 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, 2, 5, 8, 11, 14, -1, -1, -1, -1, -1);
        const __m128i m2 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 1, 4, 7, 10, 13);
        for (; x + 16 <= cols; x += 16) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(src + 3 * x)), b = _mm_loadu_si128((const __m128i *)(src + 3 * x + 16)),
                          c = _mm_loadu_si128((const __m128i *)(src + 3 * x + 32));
            _mm_storeu_si128((__m128i *)(dst + x), _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(a, m0), _mm_shuffle_epi8(b, m1)), _mm_shuffle_epi8(c, m2)));
        }
    } else if (ch == 4) {
        const __m128i m = _mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
        for (; x + 16 <= cols; x += 16) {
            const __m128i a = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(src + 4 * x)), m);
            const __m128i b = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(src + 4 * x + 16)), m);
            const __m128i c = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(src + 4 * x + 32)), m);
            const __m128i d = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(src + 4 * x + 48)), m);
            _mm_storeu_si128((__m128i *)(dst + x), _mm_unpacklo_epi64(_mm_unpacklo_epi32(a, b), _mm_unpacklo_epi32(c, d)));
        }
    }
    for (; x < cols; ++x) dst[x] = src[(size_t)x * ch];
}
// AVX2 form for BGR: two 48-byte groups per step, each half of a 256-bit register doing what the SSSE3 form does.
__attribute__((target("avx2"))) inline void extract_ch0_bgr_avx2(uint8_t *dst, const uint8_t *src, int cols)
{
    const __m128i m0 = _mm_setr_epi8(0, 3, 6, 9, 12, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m128i m1 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, 2, 5, 8, 11, 14, -1, -1, -1, -1, -1);
    const __m128i m2 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 1, 4, 7, 10, 13);
    const __m256i M0 = _mm256_broadcastsi128_si256(m0), M1 = _mm256_broadcastsi128_si256(m1), M2 = _mm256_broadcastsi128_si256(m2);
    int x = 0;
    for (; x + 32 <= cols; x += 32) {
        const uint8_t *p = src + 3 * x;
        const __m256i a = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)p)), _mm_loadu_si128((const __m128i *)(p + 48)), 1);
        const __m256i b = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)(p + 16))), _mm_loadu_si128((const __m128i *)(p + 64)), 1);
        const __m256i c = _mm256_inserti128_si256(_mm256_castsi128_si256(_mm_loadu_si128((const __m128i *)(p + 32))), _mm_loadu_si128((const __m128i *)(p + 80)), 1);
        _mm256_storeu_si256((__m256i *)(dst + x), _mm256_or_si256(_mm256_or_si256(_mm256_shuffle_epi8(a, M0), _mm256_shuffle_epi8(b, M1)), _mm256_shuffle_epi8(c, M2)));
    }
    if (x < cols) extract_ch0_ssse3(dst + x, src + 3 * (size_t)x, cols - x, 3);
}
inline void extract_ch0(uint8_t *dst, const uint8_t *src, int cols, int ch)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && ch == 3) { extract_ch0_bgr_avx2(dst, src, cols); return; }
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    if (ssse3) { extract_ch0_ssse3(dst, src, cols, ch); return; }
    for (int x = 0; x < cols; ++x) dst[x] = src[(size_t)x * ch];
}

// A few persistent worker threads for the host-side row copies of the host API (one core copies ~10 GB/s; a 4K pair is
// 25 MB per call, and starting threads per call costs more than copying 1 MB).  Created at the first large copy, never
// destroyed (the workers sleep on a condition variable; the process's exit ends them).
class CopyPool {
public:
    static CopyPool &get() { static CopyPool *p = new CopyPool(); return *p; }
    int workers() const { return (int)th_.size(); }
    // runs fn(part) for part = 0 .. parts-1: part 0 on the caller, the others on the workers; returns when all are done
    void run(int parts, const std::function<void(int)> &fn)
    {
        if (parts <= 1 || th_.empty()) { for (int i = 0; i < parts; ++i) fn(i); return; }
        std::unique_lock<std::mutex> call(call_mu_);             // one parallel copy at a time (callers on other threads queue up)
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; next_ = 1; parts_ = parts; pending_ = parts - 1; ++gen_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    CopyPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw > 1 ? (int)hw - 1 : 0;
        if (n > 7) n = 7;
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); });
        for (auto &t : th_) t.detach();
    }
    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            cv_.wait(g, [&] { return gen_ != seen && fn_ && next_ < parts_; });
            while (fn_ && next_ < parts_) {
                const int part = next_++;
                const std::function<void(int)> *f = fn_;
                g.unlock();
                (*f)(part);
                g.lock();
                if (--pending_ == 0) done_.notify_one();
            }
            seen = gen_;
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, parts_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
};

// Row-wise host copy between a caller's (pageable, strided) image and pinned staging; ch > 1: channel 0 of `rowb` pixels
// instead of `rowb` bytes.  Copies of 512 KB and more are split over the pool: one part per 128 KB, at most 8.
inline void copy_rows(uint8_t *dst, int64_t dstep, const uint8_t *src, int64_t sstep, size_t rowb, int rows, int ch = 1)
{
    auto body = [=](int y0, int y1) {
        for (int y = y0; y < y1; ++y) {
            if (ch == 1) memcpy(dst + (int64_t)y * dstep, src + (int64_t)y * sstep, rowb);
            else extract_ch0(dst + (int64_t)y * dstep, src + (int64_t)y * sstep, (int)rowb, ch);
        }
    };
    const size_t bytes = rowb * (size_t)rows * (size_t)ch;
    int nt = bytes >= ((size_t)1 << 19) ? (int)(bytes >> 17) : 1;
    if (nt > 1) { const int w = CopyPool::get().workers() + 1; if (nt > w) nt = w; if (nt > 8) nt = 8; if (nt > rows) nt = rows; }
    if (nt <= 1) { body(0, rows); return; }
    const int per = (rows + nt - 1) / nt;
    CopyPool::get().run(nt, [&](int part) {
        const int y0 = part * per, y1 = y0 + per < rows ? y0 + per : rows;
        if (y0 < y1) body(y0, y1);
    });
}

}  // namespace sadbm_hostcopy
#endif
