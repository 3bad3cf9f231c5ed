// sadbm_convert -- the file-I/O half of sadbm_cli (`--convert`) as a program of its own that links nothing: it is what
// `make asan` builds with -fsanitize=address,undefined, so the hand-written PNG / DEFLATE / BMP / PNM parsers of
// imageio.hpp -- code that reads untrusted files -- run under the sanitizers on valid, truncated and corrupted inputs
// (tests/test_sanitizers.py).  Exit status: 0 converted, 1 rejected with a message, anything else = a sanitizer report.
//   sadbm_convert in.{png,bmp,ppm,pgm} out.{png,bmp,ppm,pgm}
#include <cstdio>
#include <string>

#include "imageio.hpp"

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in out\n", argv[0]); return 2; }
    std::string err;
    sadbm::Image img;
    if (!sadbm_io::read_image(argv[1], img, err)) { fprintf(stderr, "%s: %s\n", argv[1], err.c_str()); return 1; }
    if (!sadbm_io::write_image(argv[2], img, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    printf("%s: %dx%d, %d channel(s)\n", argv[2], img.cols, img.rows, img.channels);
    return 0;
}
