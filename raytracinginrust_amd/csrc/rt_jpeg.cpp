// csrc/rt_jpeg.cpp — host-side asset ingest: baseline JPEG -> RGB8, the job `image::open(path).to_rgb8()` does for the
// reference (src/main.rs:248,491; crates image 0.24.5 / jpeg-decoder 0.3.0, not vendored under /root/reference).
//
// Scope: what the reference's asset needs and a little more — baseline sequential DCT (SOF0), 8-bit, Huffman, 1 (grey)
// or 3 (YCbCr) components, all components sampled 1x1 (earthmap.jpg is 4:4:4), optional restart intervals.  Anything
// else (progressive, subsampled chroma, 12-bit, arithmetic coding, CMYK) returns an error.
// Arithmetic: the 12-bit fixed-point integer IDCT of stb_image (which jpeg-decoder's idct.rs is derived from) and the
// 20-bit fixed-point BT.601 YCbCr -> RGB of jpeg-decoder's `ycbcr_to_rgb`.  Decoders legitimately differ by +-1 LSB
// (SURVEY.md §8(c)); tests compare against an independent decoder (Pillow / libjpeg) with that tolerance.
// Off the hot path: runs once at scene build.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace rt {

namespace {

const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
    }
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int nbits = 0; bool hit_marker = false;
    BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
    void fill() {
        while (nbits <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) { p += 2; }
                    else { hit_marker = true; byte = 0; }          // a marker: feed zeros, leave p at the 0xFF
                } else p++;
            }
            acc |= byte << (24 - nbits);
            nbits += 8;
        }
    }
    int bit() { if (nbits < 1) fill(); int b = (int)(acc >> 31); acc <<= 1; nbits--; return b; }
    int bits(int n) { if (n == 0) return 0; if (nbits < n) fill(); int v = (int)(acc >> (32 - n)); acc <<= n; nbits -= n; return v; }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const Huff& h, bool& ok) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    ok = false;
    return 0;
}
inline int extend(int v, int t) { return (t == 0) ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

// The IDCT runs in 64-bit integers: for every legal 8-bit stream the values are those of the 32-bit original, and a
// hostile stream (DC predictors run up to the cap below, 15-bit AC magnitudes times 8-bit quantisers: |coef| < 2^24,
// pass 1 < 2^41, pass 2 < 2^48) cannot overflow.
typedef long long idct_t;
inline idct_t f2f(double x) { return (idct_t)(x * 4096 + 0.5); }
inline idct_t fsh(idct_t x) { return x * 4096; }
inline uint8_t clamp8(idct_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                               \
    idct_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                 \
    p2 = s2; p3 = s6;                                                                         \
    p1 = (p2 + p3) * f2f(0.5411961);                                                          \
    t2 = p1 + p3 * (-f2f(1.847759065));                                                       \
    t3 = p1 + p2 * f2f(0.765366865);                                                          \
    p2 = s0; p3 = s4;                                                                         \
    t0 = fsh(p2 + p3); t1 = fsh(p2 - p3);                                                     \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                   \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                       \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                   \
    p5 = (p3 + p4) * f2f(1.175875602);                                                        \
    t0 = t0 * f2f(0.298631336); t1 = t1 * f2f(2.053119869); t2 = t2 * f2f(3.072711026); t3 = t3 * f2f(1.501321110); \
    p1 = p5 + p1 * (-f2f(0.899976223)); p2 = p5 + p2 * (-f2f(2.562915447));                   \
    p3 = p3 * (-f2f(1.961570560)); p4 = p4 * (-f2f(0.390180644));                             \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(const int* coef /* natural order, dequantised */, uint8_t* out, int stride) {
    idct_t val[64];
    for (int i = 0; i < 8; i++) {                       // columns
        const int* d = coef + i; idct_t* v = val + i;
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            idct_t dc = (idct_t)d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {                       // rows
        const idct_t* v = val + i * 8; uint8_t* o = out + i * stride;
        IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

const uint64_t MAX_PIXELS = 64ull << 20;      // the header's 16-bit dimensions alone would allow 4 Gpixel allocations

inline int cfix(double x) { return (int)(x * (double)(1 << 20) + 0.5); }
inline uint8_t clamp_fixed(int v) { int r = v >> 20; return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); }

} // namespace

// Decodes `data` into interleaved RGB8 (grey images are replicated into three channels, as `.to_rgb8()` does).
bool decode_jpeg_rgb8(const uint8_t* data, size_t size, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height, std::string& err) {
    auto fail = [&](const char* m) { err = m; return false; };
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG (no SOI)");
    uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    struct Comp { int id, h, v, tq, td, ta, pred; std::vector<uint8_t> plane; } comp[3];
    int ncomp = 0, restart_interval = 0;
    bool have_sof = false;
    size_t i = 2;
    while (i + 4 <= size) {
        if (data[i] != 0xFF) return fail("corrupt JPEG: marker expected");
        while (i + 1 < size && data[i + 1] == 0xFF) i++;       // fill bytes
        if (i + 2 > size) return fail("truncated JPEG");
        uint8_t m = data[i + 1];
        i += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return fail("JPEG ends before a scan");
        if (i + 2 > size) return fail("truncated JPEG");
        size_t L = ((size_t)data[i] << 8) | data[i + 1];
        if (L < 2 || i + L > size) return fail("truncated JPEG segment");
        const uint8_t* seg = data + i + 2; size_t n = L - 2;
        if (m == 0xDB) {                                                    // DQT
            size_t k = 0;
            while (k < n) {
                int pq = seg[k] >> 4, tq = seg[k] & 15; k++;
                if (tq > 3) return fail("bad quantisation table id");
                if (pq != 0) return fail("only 8-bit quantisation tables are supported (baseline JPEG)");
                if (k + 64 > n) return fail("truncated DQT");
                for (int z = 0; z < 64; z++) qt[tq][z] = seg[k++];
                qt_ok[tq] = true;
            }
        } else if (m == 0xC4) {                                             // DHT
            size_t k = 0;
            while (k + 17 <= n) {
                int tc = seg[k] >> 4, th = seg[k] & 15; k++;
                if (tc > 1 || th > 3) return fail("bad Huffman table id");
                Huff& h = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = seg[k++]; total += h.bits[l]; }
                if (total > 256 || k + (size_t)total > n) return fail("truncated DHT");
                std::memcpy(h.vals, seg + k, (size_t)total); k += (size_t)total;
                h.present = true; h.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {                                // SOF0 / SOF1 (Huffman sequential)
            if (n < 6) return fail("truncated SOF");
            if (seg[0] != 8) return fail("only 8-bit JPEG is supported");
            height = ((uint32_t)seg[1] << 8) | seg[2]; width = ((uint32_t)seg[3] << 8) | seg[4];
            ncomp = seg[5];
            if (ncomp != 1 && ncomp != 3) return fail("only grey or YCbCr JPEG is supported");
            if (n < 6 + (size_t)ncomp * 3 || width == 0 || height == 0) return fail("bad SOF");
            if ((uint64_t)width * height > MAX_PIXELS) return fail("JPEG frame larger than 64 Mpixel");
            for (int c = 0; c < ncomp; c++) {
                comp[c].id = seg[6 + c * 3]; comp[c].h = seg[7 + c * 3] >> 4; comp[c].v = seg[7 + c * 3] & 15; comp[c].tq = seg[8 + c * 3];
                if (comp[c].h != 1 || comp[c].v != 1) return fail("subsampled JPEG components are not supported (only 1x1 sampling)");
                if (comp[c].tq > 3) return fail("bad quantisation table id");
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return fail("only baseline sequential (SOF0) JPEG is supported");
        } else if (m == 0xDD) {                                             // DRI
            if (n < 2) return fail("truncated DRI");
            restart_interval = (seg[0] << 8) | seg[1];
        } else if (m == 0xDA) {                                             // SOS: decode the (single) scan
            if (!have_sof) return fail("SOS before SOF");
            if (n < 1 || seg[0] != ncomp || n < 1 + (size_t)ncomp * 2 + 3) return fail("only single-scan interleaved JPEG is supported");
            for (int c = 0; c < ncomp; c++) {
                int cid = seg[1 + c * 2], tbl = seg[2 + c * 2];
                if (cid != comp[c].id) return fail("scan component order differs from the frame");
                comp[c].td = tbl >> 4; comp[c].ta = tbl & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !hdc[comp[c].td].present || !hac[comp[c].ta].present) return fail("scan refers to a missing Huffman table");
                if (!qt_ok[comp[c].tq]) return fail("frame refers to a missing quantisation table");
                comp[c].pred = 0;
            }
            const uint32_t bw = (width + 7) / 8, bh = (height + 7) / 8;
            const int stride = (int)bw * 8;
            for (int c = 0; c < ncomp; c++) comp[c].plane.assign((size_t)stride * bh * 8, 0);
            BitReader br(data + i + L, data + size);
            bool ok = true;
            uint32_t mcu = 0; int next_rst = 0;
            for (uint32_t by = 0; by < bh; by++) for (uint32_t bx = 0; bx < bw; bx++) {
                if (restart_interval && mcu && (mcu % (uint32_t)restart_interval) == 0) {
                    const uint8_t* p = br.p;                              // the bit reader stops at markers: expect RSTn here
                    while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) p++;
                    if (p + 1 >= br.end || p[1] != 0xD0 + next_rst) return fail("missing restart marker");
                    br.p = p + 2; br.reset(); next_rst = (next_rst + 1) & 7;
                    for (int c = 0; c < ncomp; c++) comp[c].pred = 0;
                }
                for (int c = 0; c < ncomp; c++) {
                    int coef[64] = {0};
                    int t = decode_symbol(br, hdc[comp[c].td], ok);
                    if (!ok || t > 11) return fail("corrupt JPEG entropy data (DC)");
                    int diff = extend(br.bits(t), t);
                    comp[c].pred += diff;
                    if (comp[c].pred < -65536 || comp[c].pred > 65535) return fail("corrupt JPEG entropy data (DC out of range)");
                    coef[0] = comp[c].pred * qt[comp[c].tq][0];
                    for (int k = 1; k < 64;) {
                        int rs = decode_symbol(br, hac[comp[c].ta], ok);
                        if (!ok) return fail("corrupt JPEG entropy data (AC)");
                        int r = rs >> 4, s = rs & 15;
                        if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                        k += r;
                        if (k > 63) return fail("corrupt JPEG entropy data (run)");
                        coef[ZIGZAG[k]] = extend(br.bits(s), s) * qt[comp[c].tq][k];
                        k++;
                    }
                    idct_block(coef, comp[c].plane.data() + (size_t)by * 8 * stride + bx * 8, stride);
                }
                mcu++;
            }
            rgb.resize((size_t)width * height * 3);
            for (uint32_t y = 0; y < height; y++) for (uint32_t x = 0; x < width; x++) {
                size_t s = (size_t)y * stride + x, o = ((size_t)y * width + x) * 3;
                if (ncomp == 1) { rgb[o] = rgb[o + 1] = rgb[o + 2] = comp[0].plane[s]; continue; }
                int yy = (int)comp[0].plane[s] * (1 << 20) + (1 << 19);                // jpeg-decoder's ycbcr_to_rgb: BT.601, 20-bit fixed point
                int cb = (int)comp[1].plane[s] - 128, cr = (int)comp[2].plane[s] - 128;
                rgb[o] = clamp_fixed(yy + cfix(1.40200) * cr);
                rgb[o + 1] = clamp_fixed(yy - cfix(0.34414) * cb - cfix(0.71414) * cr);
                rgb[o + 2] = clamp_fixed(yy + cfix(1.77200) * cb);
            }
            return true;
        }
        i += L;
    }
    return fail("no scan found in JPEG");
}

} // namespace rt
