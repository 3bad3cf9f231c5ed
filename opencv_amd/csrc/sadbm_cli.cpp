// sadbm_cli -- headless replacement for main() of /root/reference/SAD.cpp:63-82.
//   sadbm_cli left.{ppm,pgm} right.{ppm,pgm} [--win 7] [--dsr 30] [-o disp.pgm]
// Reads binary PGM (P5) / PPM (P6) files.  A PPM is stored RGB; cv::imread (SAD.cpp:65-66) returns BGR
// and the matcher uses channel 0, so the planes are swapped to BGR on load -- channel 0 = blue.
// No GUI (the reference's imshow/waitKey, SAD.cpp:71-79, are out of scope).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/sadbm.hpp"

static bool read_pnm(const char *path, sadbm::Image &img)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[3] = {0};
    int w = 0, h = 0, maxv = 0;
    auto skip = [&]() { int c; while ((c = fgetc(f)) != EOF) { if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} } else if (!isspace(c)) { ungetc(c, f); break; } } };
    if (fscanf(f, "%2s", magic) != 1) { fclose(f); return false; }
    skip(); if (fscanf(f, "%d", &w) != 1) { fclose(f); return false; }
    skip(); if (fscanf(f, "%d", &h) != 1) { fclose(f); return false; }
    skip(); if (fscanf(f, "%d", &maxv) != 1) { fclose(f); return false; }
    fgetc(f);
    int ch = !strcmp(magic, "P5") ? 1 : !strcmp(magic, "P6") ? 3 : 0;
    if (!ch || maxv != 255 || w <= 0 || h <= 0) { fclose(f); return false; }
    img = sadbm::Image(h, w, ch);
    bool ok = fread(img.pix.data(), 1, img.pix.size(), f) == img.pix.size();
    fclose(f);
    if (ok && ch == 3)
        for (size_t i = 0; i + 2 < img.pix.size(); i += 3) { uint8_t t = img.pix[i]; img.pix[i] = img.pix[i + 2]; img.pix[i + 2] = t; }   // RGB -> BGR
    return ok;
}

int main(int argc, char **argv)
{
    const char *lp = nullptr, *rp = nullptr, *op = "disparity.pgm";
    int win = sadbm::kDefaultWinSize, dsr = sadbm::kDefaultDSR;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--win") && i + 1 < argc) win = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dsr") && i + 1 < argc) dsr = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) op = argv[++i];
        else if (!lp) lp = argv[i];
        else if (!rp) rp = argv[i];
    }
    if (!lp || !rp) { fprintf(stderr, "usage: %s left.ppm right.ppm [--win 7] [--dsr 30] [-o disparity.pgm]\n", argv[0]); return 2; }
    sadbm::Image L, R;
    if (!read_pnm(lp, L) || !read_pnm(rp, R)) { fprintf(stderr, "cannot read %s / %s (binary PGM/PPM, maxval 255)\n", lp, rp); return 1; }
    try {
        sadbm::Image d = sadbm::getDisparity(L, R, win, dsr);
        FILE *f = fopen(op, "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", op); return 1; }
        fprintf(f, "P5\n%d %d\n255\n", d.cols, d.rows);
        fwrite(d.pix.data(), 1, d.pix.size(), f);
        fclose(f);
        printf("%s: %dx%d winSize=%d DSR=%d\n", op, d.cols, d.rows, win, dsr);
    } catch (const sadbm::Error &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
