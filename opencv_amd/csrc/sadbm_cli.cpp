// sadbm_cli -- headless replacement for main() of /root/reference/SAD.cpp:63-82.
//   sadbm_cli left.{png,ppm,pgm} right.{png,ppm,pgm} [--win 7] [--dsr 30] [--raw] [-o disparity.{pgm,png}]
//   sadbm_cli --convert in.{png,ppm,pgm,bmp} out.{png,ppm,pgm,bmp}     (file I/O only, touches no GPU)
//   sadbm_cli --gaus src.bmp [--size 35] [--sigma 3] [-o low.bmp] [--high high.bmp]
//        main() of gaus_filter.cpp:42-70: Gaussian low-pass (filter2D) and high-pass (addWeighted) of one picture
//   sadbm_cli --hybrid dog.bmp cat.bmp [--size 31] [--sigma 10] [--add 12] [--alpha 0.6] [-o hybrid.bmp] [--low f] [--high f]
//        main() of hybrid_image.cpp:102-143: low-pass of the first + high-pass of the second picture, blended
// The reference loads left.png / right.png from the working directory with cv::imread (BGR, SAD.cpp:65-66),
// calls getDisparity (SAD.cpp:76) and shows three HighGUI windows (SAD.cpp:71-79).  Here the two paths are
// arguments, the files are decoded by imageio.hpp (channel 0 = blue, as imread gives), the disparity map is
// written to a file, and there is no GUI.  Defaults are the reference's winSize = 7, DSR = 30 (SAD.cpp:9,11).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "imageio.hpp"
#include "../../include/sadbm_filters.h"

static int usage(const char *argv0)
{
    fprintf(stderr, "usage: %s left right [--win 7] [--dsr 30] [--raw] [-o disparity.pgm|.png]\n"
                    "       %s --convert in out\n"
                    "       %s --gaus src [--size 35] [--sigma 3] [-o low.bmp] [--high high.bmp]\n"
                    "       %s --hybrid img1 img2 [--size 31] [--sigma 10] [--add 12] [--alpha 0.6] [-o hybrid.bmp] [--low f] [--high f]\n",
            argv0, argv0, argv0, argv0);
    return 2;
}

int main(int argc, char **argv)
{
    const char *lp = nullptr, *rp = nullptr, *op = nullptr, *lowp = nullptr, *highp = nullptr;
    int win = sadbm::kDefaultWinSize, dsr = sadbm::kDefaultDSR;
    int fsize = 0, fsigma = 0, fadd = 12;                     // hybrid_image.cpp:131
    double alpha = 0.6;                                       // hybrid_image.cpp:138
    bool convert = false, raw = false, gaus = false, hybrid = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--win") && i + 1 < argc) win = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dsr") && i + 1 < argc) dsr = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) op = argv[++i];
        else if (!strcmp(argv[i], "--convert")) convert = true;
        else if (!strcmp(argv[i], "--gaus")) gaus = true;
        else if (!strcmp(argv[i], "--hybrid")) hybrid = true;
        else if (!strcmp(argv[i], "--size") && i + 1 < argc) fsize = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--sigma") && i + 1 < argc) fsigma = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--add") && i + 1 < argc) fadd = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--alpha") && i + 1 < argc) alpha = atof(argv[++i]);
        else if (!strcmp(argv[i], "--low") && i + 1 < argc) lowp = argv[++i];
        else if (!strcmp(argv[i], "--high") && i + 1 < argc) highp = argv[++i];
        else if (!strcmp(argv[i], "--raw")) raw = true;
        else if (argv[i][0] == '-' && argv[i][1]) return usage(argv[0]);
        else if (!lp) lp = argv[i];
        else if (!rp) rp = argv[i];
        else return usage(argv[0]);
    }
    if (!lp || (!rp && !gaus)) return usage(argv[0]);
    std::string err;
    sadbm::Image L, R;
    if (!sadbm_io::read_image(lp, L, err)) { fprintf(stderr, "%s: %s\n", lp, err.c_str()); return 1; }
    if (gaus || hybrid) {
        if (hybrid && !sadbm_io::read_image(rp, R, err)) { fprintf(stderr, "%s: %s\n", rp, err.c_str()); return 1; }
        if (!fsize) fsize = gaus ? 35 : 31;                   // gaus_filter.cpp:10, hybrid_image.cpp:10
        if (!fsigma) fsigma = gaus ? 3 : 10;                  // gaus_filter.cpp:11, hybrid_image.cpp:11
        sadbm::Image low(L.rows, L.cols, L.channels), high(L.rows, L.cols, L.channels), mix(L.rows, L.cols, L.channels);
        const int64_t step = (int64_t)L.cols * L.channels;
        sadbm_image a{L.pix.data(), L.rows, L.cols, L.channels, step}, b{R.pix.data(), R.rows, R.cols, R.channels, (int64_t)R.cols * R.channels};
        const int rc = gaus ? sadbm_gaus_filter_host(&a, fsize, fsigma, low.pix.data(), step, high.pix.data(), step)
                            : sadbm_hybrid_image_host(&a, &b, fsize, fsigma, fadd, alpha, low.pix.data(), step, high.pix.data(), step, mix.pix.data(), step);
        if (rc) { fprintf(stderr, "%s\n", sadbm::Error(rc).what()); return 1; }
        const char *main_out = op ? op : (gaus ? "low_pass.bmp" : "hybrid_image.bmp");
        if (!sadbm_io::write_image(main_out, gaus ? low : mix, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (!gaus && lowp && !sadbm_io::write_image(lowp, low, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        if (highp && !sadbm_io::write_image(highp, high, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        printf("%s: %dx%d x%d, %dx%d Gaussian sigma %d\n", main_out, L.cols, L.rows, L.channels, fsize, fsize, fsigma);
        return 0;
    }
    if (!op) op = "disparity.pgm";
    if (convert) {
        if (!sadbm_io::write_image(rp, L, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        printf("%s: %dx%d, %d channel(s)\n", rp, L.cols, L.rows, L.channels);
        return 0;
    }
    if (!sadbm_io::read_image(rp, R, err)) { fprintf(stderr, "%s: %s\n", rp, err.c_str()); return 1; }
    try {
        sadbm::Image d = sadbm::getDisparity(L.view(), R.view(), win, dsr, raw ? SADBM_RAW_LOC : 0u);
        if (!sadbm_io::write_image(op, d, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        printf("%s: %dx%d winSize=%d DSR=%d\n", op, d.cols, d.rows, win, dsr);
    } catch (const sadbm::Error &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
