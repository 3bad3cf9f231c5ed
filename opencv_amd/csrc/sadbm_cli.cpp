// sadbm_cli -- headless replacement for main() of /root/reference/SAD.cpp:63-82.
//   sadbm_cli left.{png,ppm,pgm} right.{png,ppm,pgm} [--win 7] [--dsr 30] [--raw] [-o disparity.{pgm,png}]
//   sadbm_cli --convert in.{png,ppm,pgm} out.{png,ppm,pgm}       (file I/O only, touches no GPU)
// The reference loads left.png / right.png from the working directory with cv::imread (BGR, SAD.cpp:65-66),
// calls getDisparity (SAD.cpp:76) and shows three HighGUI windows (SAD.cpp:71-79).  Here the two paths are
// arguments, the files are decoded by imageio.hpp (channel 0 = blue, as imread gives), the disparity map is
// written to a file, and there is no GUI.  Defaults are the reference's winSize = 7, DSR = 30 (SAD.cpp:9,11).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "imageio.hpp"

static int usage(const char *argv0)
{
    fprintf(stderr, "usage: %s left right [--win 7] [--dsr 30] [--raw] [-o disparity.pgm|.png]\n"
                    "       %s --convert in out\n", argv0, argv0);
    return 2;
}

int main(int argc, char **argv)
{
    const char *lp = nullptr, *rp = nullptr, *op = "disparity.pgm";
    int win = sadbm::kDefaultWinSize, dsr = sadbm::kDefaultDSR;
    bool convert = false, raw = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--win") && i + 1 < argc) win = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dsr") && i + 1 < argc) dsr = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) op = argv[++i];
        else if (!strcmp(argv[i], "--convert")) convert = true;
        else if (!strcmp(argv[i], "--raw")) raw = true;
        else if (argv[i][0] == '-' && argv[i][1]) return usage(argv[0]);
        else if (!lp) lp = argv[i];
        else if (!rp) rp = argv[i];
        else return usage(argv[0]);
    }
    if (!lp || !rp) return usage(argv[0]);
    std::string err;
    sadbm::Image L, R;
    if (!sadbm_io::read_image(lp, L, err)) { fprintf(stderr, "%s: %s\n", lp, err.c_str()); return 1; }
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
