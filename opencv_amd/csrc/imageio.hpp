// imageio.hpp -- dependency-free image file I/O for the headless sadbm tools.
//
// Stands in for cv::imread(path) as main() of /root/reference/SAD.cpp:65-66 calls it (default flag
// IMREAD_COLOR): whatever the file holds, the caller gets 8-bit pixels with channel 0 = BLUE for colour
// files (BGR order) -- the matcher reads channel 0 only (SAD.cpp:19-20).  Grey files stay 1-channel
// (imread would replicate grey into B, G and R; channel 0 is the same plane either way).
//   read : binary PGM (P5) / PPM (P6); PNG, non-interlaced, colour types 0/2/3/4/6, bit depths 1..16
//          (16-bit samples keep their high byte, alpha is dropped -- as imread does without ANYDEPTH/UNCHANGED)
//          BMP (what gaus_filter.cpp:44 / hybrid_image.cpp:104-105 load): BITMAPINFOHEADER family, uncompressed, 8-bit
//          paletted / 24-bit / 32-bit, bottom-up or top-down; BMP stores BGR, which is what imread returns
//   write: binary PGM / PPM; PNG (8-bit grey or RGB, stored deflate blocks); 24-bit BMP
// Written from the PNG (ISO/IEC 15948) and DEFLATE (RFC 1951) / zlib (RFC 1950) specifications.
#ifndef SADBM_IMAGEIO_HPP
#define SADBM_IMAGEIO_HPP

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sadbm.hpp"

namespace sadbm_io {

typedef std::vector<uint8_t> Bytes;

inline bool read_file(const char *path, Bytes &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    uint8_t buf[65536];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}

// ---------------------------------------------------------------------------------------------
// DEFLATE (RFC 1951) decoder
// ---------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool ok = true;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    uint32_t bits(int k)                       // k <= 16, LSB first
    {
        while (n < k) {
            if (p >= end) { ok = false; return 0; }
            acc |= (uint32_t)*p++ << n;
            n += 8;
        }
        const uint32_t v = acc & ((1u << k) - 1);
        acc >>= k; n -= k;
        return v;
    }
    void align() { acc = 0; n = 0; }
};

struct Huffman {
    uint16_t count[16], symbol[320];
    bool build(const uint8_t *len, int nsym)
    {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < nsym; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        int left = 1;
        for (int l = 1; l < 16; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < nsym; ++i) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const              // canonical code, bit by bit
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

inline bool inflate_raw(const uint8_t *src, size_t n, Bytes &out)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    BitReader br(src, src + n);
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) return false;
            const uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xFFFF) != nlen || (size_t)(br.end - br.p) < len) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lens[i] = 8;
                for (int i = 144; i < 256; ++i) lens[i] = 9;
                for (int i = 256; i < 280; ++i) lens[i] = 7;
                for (int i = 280; i < 288; ++i) lens[i] = 8;
                lit.build(lens, 288);
                for (int i = 0; i < 30; ++i) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (!br.ok || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huffman ch;
                if (!ch.build(cl, 19)) return false;
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = ch.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                    int rep; uint8_t v = 0;
                    if (sym == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)br.bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (!br.ok || i + rep > nlen + ndist) return false;
                    while (rep--) lens[i++] = v;
                }
                if (!lit.build(lens, nlen) || !dist.build(lens + nlen, ndist)) return false;
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) { out.push_back((uint8_t)sym); continue; }
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) return false;
                const int len = lbase[sym] + (int)br.bits(lext[sym]);
                const int ds = dist.decode(br);
                if (ds < 0 || ds >= 30) return false;
                const size_t d = dbase[ds] + br.bits(dext[ds]);
                if (!br.ok || d > out.size()) return false;
                const size_t from = out.size() - d;
                for (int k = 0; k < len; ++k) out.push_back(out[from + k]);
            }
        } else {
            return false;
        }
        if (last) return true;
    }
}

inline uint32_t adler32(const uint8_t *p, size_t n)
{
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; ++i) { a = (a + p[i]) % 65521; b = (b + a) % 65521; }
    return (b << 16) | a;
}

inline bool zlib_inflate(const Bytes &z, Bytes &out)
{
    if (z.size() < 6 || (z[0] & 0x0F) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) return false;
    if (!inflate_raw(z.data() + 2, z.size() - 6, out)) return false;
    const uint8_t *t = z.data() + z.size() - 4;
    const uint32_t want = ((uint32_t)t[0] << 24) | (t[1] << 16) | (t[2] << 8) | t[3];
    return adler32(out.data(), out.size()) == want;
}

inline uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

// ---------------------------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------------------------
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

inline bool decode_png(const Bytes &file, sadbm::Image &img, std::string &err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || memcmp(file.data(), sig, 8)) { err = "not a PNG file"; return false; }
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = -1, interlace = 0;
    Bytes idat, plte;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { err = "truncated PNG chunk"; return false; }
        const uint8_t *data = &file[pos + 8];
        if (crc32(type, 4 + (size_t)len) != be32(data + len)) { err = "PNG chunk CRC mismatch"; return false; }
        if (!memcmp(type, "IHDR", 4) && len == 13) {
            W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { err = "unknown PNG compression/filter method"; return false; }
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || W == 0 || H == 0 || W > 65535 || H > 65535) { err = "bad PNG header"; return false; }
    if (interlace) { err = "interlaced PNG is not supported"; return false; }
    const int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!samples || !depth_ok) { err = "unsupported PNG colour type / bit depth"; return false; }
    if (ctype == 3 && plte.size() < 3) { err = "palette PNG without PLTE"; return false; }
    Bytes raw;
    if (!zlib_inflate(idat, raw)) { err = "PNG zlib stream is corrupt"; return false; }
    const size_t bpp_bits = (size_t)samples * depth, stride = (W * bpp_bits + 7) / 8, bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    if (raw.size() < (stride + 1) * H) { err = "PNG image data too short"; return false; }
    // unfilter in place (filter types 0..4)
    Bytes prev(stride, 0);
    for (uint32_t y = 0; y < H; ++y) {
        uint8_t *row = &raw[(stride + 1) * y + 1];
        const int ft = raw[(stride + 1) * y];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? row[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) { err = "bad PNG filter type"; return false; }
            row[i] = (uint8_t)(row[i] + pred);
        }
        memcpy(prev.data(), row, stride);
    }
    const bool colour = (ctype == 2 || ctype == 6 || ctype == 3);
    img = sadbm::Image((int)H, (int)W, colour ? 3 : 1);
    const int step = depth == 16 ? 2 : 1;               // 16-bit: keep the high (first) byte
    for (uint32_t y = 0; y < H; ++y) {
        const uint8_t *row = &raw[(stride + 1) * y + 1];
        uint8_t *o = img.ptr((int)y);
        for (uint32_t x = 0; x < W; ++x) {
            if (ctype == 0 || ctype == 3) {
                uint32_t v;
                if (depth >= 8) v = row[x * step];
                else { const int per = 8 / depth; v = (row[x / per] >> ((per - 1 - x % per) * depth)) & ((1u << depth) - 1); }
                if (ctype == 3) {
                    if (3 * v + 2 >= plte.size()) v = 0;
                    o[3 * x] = plte[3 * v + 2]; o[3 * x + 1] = plte[3 * v + 1]; o[3 * x + 2] = plte[3 * v];     // BGR
                } else {
                    o[x] = depth < 8 ? (uint8_t)(v * 255 / ((1u << depth) - 1)) : (uint8_t)v;
                }
            } else if (ctype == 4) {
                o[x] = row[x * 2 * step];
            } else {
                const uint8_t *s = row + (size_t)x * samples * step;
                o[3 * x] = s[2 * step]; o[3 * x + 1] = s[step]; o[3 * x + 2] = s[0];                            // BGR
            }
        }
    }
    return true;
}

inline void put_be32(Bytes &b, uint32_t v) { b.push_back(v >> 24); b.push_back(v >> 16); b.push_back(v >> 8); b.push_back(v); }

inline void png_chunk(Bytes &out, const char *type, const Bytes &data)
{
    put_be32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32(&out[start], out.size() - start));
}

// 8-bit grey (1 channel) or colour (3 channels, BGR in memory -> RGB in the file); stored deflate blocks.
inline Bytes encode_png(const sadbm::Image &img)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    Bytes out(sig, sig + 8), ihdr, raw;
    put_be32(ihdr, (uint32_t)img.cols); put_be32(ihdr, (uint32_t)img.rows);
    ihdr.push_back(8); ihdr.push_back(img.channels == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    png_chunk(out, "IHDR", ihdr);
    for (int y = 0; y < img.rows; ++y) {
        raw.push_back(0);
        const uint8_t *s = img.ptr(y);
        if (img.channels == 1) raw.insert(raw.end(), s, s + img.cols);
        else for (int x = 0; x < img.cols; ++x) { const uint8_t *q = s + (size_t)x * img.channels; raw.push_back(q[2]); raw.push_back(q[1]); raw.push_back(q[0]); }
    }
    Bytes z;
    z.push_back(0x78); z.push_back(0x01);
    size_t pos = 0;
    do {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    put_be32(z, adler32(raw.data(), raw.size()));
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", Bytes());
    return out;
}

// ---------------------------------------------------------------------------------------------
// PNM
// ---------------------------------------------------------------------------------------------
inline bool decode_pnm(const Bytes &file, sadbm::Image &img, std::string &err)
{
    size_t pos = 0;
    auto token = [&](std::string &t) {
        t.clear();
        for (;;) {
            while (pos < file.size() && isspace(file[pos])) ++pos;
            if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') ++pos; continue; }
            break;
        }
        while (pos < file.size() && !isspace(file[pos])) t.push_back((char)file[pos++]);
        return !t.empty();
    };
    std::string magic, sw, sh, sm;
    if (!token(magic) || !token(sw) || !token(sh) || !token(sm)) { err = "truncated PNM header"; return false; }
    const int ch = magic == "P5" ? 1 : magic == "P6" ? 3 : 0;
    const int w = atoi(sw.c_str()), h = atoi(sh.c_str()), maxv = atoi(sm.c_str());
    if (!ch || maxv != 255 || w <= 0 || h <= 0) { err = "only binary PGM/PPM with maxval 255 is supported"; return false; }
    ++pos;                                              // the single whitespace byte after maxval
    const size_t need = (size_t)w * h * ch;
    if (file.size() < pos + need) { err = "PNM pixel data too short"; return false; }
    img = sadbm::Image(h, w, ch);
    memcpy(img.pix.data(), &file[pos], need);
    if (ch == 3)
        for (size_t i = 0; i + 2 < need; i += 3) { const uint8_t t = img.pix[i]; img.pix[i] = img.pix[i + 2]; img.pix[i + 2] = t; }   // RGB -> BGR
    return true;
}

inline Bytes encode_pnm(const sadbm::Image &img)
{
    char hdr[64];
    const int n = snprintf(hdr, sizeof(hdr), "P%d\n%d %d\n255\n", img.channels == 1 ? 5 : 6, img.cols, img.rows);
    Bytes out(hdr, hdr + n);
    if (img.channels == 1) out.insert(out.end(), img.pix.begin(), img.pix.end());
    else for (int y = 0; y < img.rows; ++y)
        for (int x = 0; x < img.cols; ++x) { const uint8_t *q = img.ptr(y) + (size_t)x * img.channels; out.push_back(q[2]); out.push_back(q[1]); out.push_back(q[0]); }
    return out;
}

// ---------------------------------------------------------------------------------------------
// BMP (uncompressed): the input format of the reference's filter programs (gaus_filter.cpp:44, hybrid_image.cpp:104-105)
// ---------------------------------------------------------------------------------------------
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

inline bool decode_bmp(const Bytes &file, sadbm::Image &img, std::string &err)
{
    if (file.size() < 54 || file[0] != 'B' || file[1] != 'M') { err = "not a BMP file"; return false; }
    const uint32_t off = le32(&file[10]), hdr = le32(&file[14]);
    if (hdr < 40 || (size_t)14 + hdr > file.size()) { err = "unsupported BMP header"; return false; }
    const int32_t w = (int32_t)le32(&file[18]), hs = (int32_t)le32(&file[22]);
    const uint32_t planes = le16(&file[26]), bpp = le16(&file[28]), comp = le32(&file[30]);
    uint32_t ncol = le32(&file[46]);
    const bool top_down = hs < 0;
    const int64_t h = top_down ? -(int64_t)hs : hs;
    if (planes != 1 || w <= 0 || h <= 0 || w > 65535 || h > 65535) { err = "bad BMP dimensions"; return false; }
    if (comp != 0 && !(comp == 3 && bpp == 32)) { err = "compressed BMP is not supported"; return false; }
    if (bpp != 8 && bpp != 24 && bpp != 32) { err = "only 8-, 24- and 32-bit BMP is supported"; return false; }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if (off > file.size() || (file.size() - off) / stride < (size_t)h) { err = "BMP pixel data too short"; return false; }
    const uint8_t *pal = nullptr;
    bool grey_palette = true;
    if (bpp == 8) {
        if (ncol == 0) ncol = 256;
        if (ncol > 256 || (size_t)14 + hdr + (size_t)ncol * 4 > file.size()) { err = "bad BMP palette"; return false; }
        pal = &file[14 + hdr];
        for (uint32_t i = 0; i < ncol; ++i) grey_palette = grey_palette && pal[4 * i] == pal[4 * i + 1] && pal[4 * i] == pal[4 * i + 2];
    }
    // imread(IMREAD_COLOR) always returns 3 channels; a grey palette stays 1-channel here (channel 0 is the same plane)
    const int ch = (bpp == 8 && grey_palette) ? 1 : 3;
    img = sadbm::Image((int)h, w, ch);
    for (int64_t y = 0; y < h; ++y) {
        const uint8_t *src = &file[off + (size_t)(top_down ? y : h - 1 - y) * stride];
        uint8_t *dst = img.ptr((int)y);
        if (bpp == 24) memcpy(dst, src, (size_t)w * 3);                                   // already B, G, R
        else if (bpp == 32) for (int x = 0; x < w; ++x) { dst[3 * x] = src[4 * x]; dst[3 * x + 1] = src[4 * x + 1]; dst[3 * x + 2] = src[4 * x + 2]; }
        else for (int x = 0; x < w; ++x) {
            const uint32_t idx = src[x] < ncol ? src[x] : 0;
            if (ch == 1) dst[x] = pal[4 * idx];
            else { dst[3 * x] = pal[4 * idx]; dst[3 * x + 1] = pal[4 * idx + 1]; dst[3 * x + 2] = pal[4 * idx + 2]; }
        }
    }
    return true;
}

inline Bytes encode_bmp(const sadbm::Image &img)
{
    const size_t stride = (((size_t)img.cols * 24 + 31) / 32) * 4, data = stride * (size_t)img.rows;
    Bytes out(54 + data, 0);
    auto put32 = [&](size_t at, uint32_t v) { out[at] = (uint8_t)v; out[at + 1] = (uint8_t)(v >> 8); out[at + 2] = (uint8_t)(v >> 16); out[at + 3] = (uint8_t)(v >> 24); };
    out[0] = 'B'; out[1] = 'M';
    put32(2, (uint32_t)out.size()); put32(10, 54); put32(14, 40); put32(18, (uint32_t)img.cols); put32(22, (uint32_t)img.rows);
    out[26] = 1; out[28] = 24; put32(34, (uint32_t)data); put32(38, 2834); put32(42, 2834);
    for (int y = 0; y < img.rows; ++y) {
        uint8_t *dst = &out[54 + (size_t)(img.rows - 1 - y) * stride];
        const uint8_t *src = img.ptr(y);
        if (img.channels == 3) memcpy(dst, src, (size_t)img.cols * 3);
        else for (int x = 0; x < img.cols; ++x) dst[3 * x] = dst[3 * x + 1] = dst[3 * x + 2] = src[x];
    }
    return out;
}

// imread(path) of SAD.cpp:65-66: false (and a message) instead of an empty Mat.
inline bool read_image(const char *path, sadbm::Image &img, std::string &err)
{
    Bytes file;
    if (!read_file(path, file)) { err = std::string("cannot open ") + path; return false; }
    if (file.size() >= 2 && file[0] == 'P' && (file[1] == '5' || file[1] == '6')) return decode_pnm(file, img, err);
    if (file.size() >= 2 && file[0] == 'B' && file[1] == 'M') return decode_bmp(file, img, err);
    return decode_png(file, img, err);
}

inline bool write_image(const char *path, const sadbm::Image &img, std::string &err)
{
    const size_t n = strlen(path);
    const bool png = n >= 4 && !strcmp(path + n - 4, ".png");
    if (img.channels != 1 && img.channels != 3) { err = "only 1- and 3-channel images can be written"; return false; }
    const bool bmp = n >= 4 && !strcmp(path + n - 4, ".bmp");
    const Bytes data = png ? encode_png(img) : bmp ? encode_bmp(img) : encode_pnm(img);
    FILE *f = fopen(path, "wb");
    if (!f) { err = std::string("cannot write ") + path; return false; }
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    if (!ok) err = std::string("short write to ") + path;
    return ok;
}

}  // namespace sadbm_io

#endif  // SADBM_IMAGEIO_HPP
