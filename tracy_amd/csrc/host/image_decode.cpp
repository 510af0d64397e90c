// image_decode.cpp -- texture file decoders for hosts that run outside the Tracy tree.
//
// Inside Tracy, Scene::Init decodes textures with the vendored third-party stb_image
// (src/scene.cpp:23-48: stbi_loadf for *.hdr, stbi_load(..., req_comp = 4) otherwise) and hands
// the module finished RGBA32F texels; the render module never sees a file. These decoders make
// the standalone scene loader (include/tracy_scene.h) and the headless CLI self-sufficient:
//
//   * baseline sequential JPEG (ITU-T T.81: SOF0/SOF1, 8-bit, Huffman, 1 or 3 components, any
//     sampling factors up to 2x2 get the smooth "triangle" chroma up-sampling, larger factors are
//     replicated, restart intervals), integer 8x8 inverse DCT after Loeffler-Ligtenberg-Moschytz
//     with 12-bit constants, JFIF YCbCr -> RGB in 20-bit fixed point, grey -> r = g = b, alpha 255:
//     the arithmetic stb_image documents for its scalar path, written from the published
//     algorithms. Progressive JPEG (SOF2) is refused with a clear error.
//   * Radiance RGBE *.hdr ("#?RADIANCE" / "#?RGBE", FORMAT=32-bit_rle_rgbe, "-Y h +X w"), flat and
//     new-style run-length scanlines; texel = mantissa * 2^(e - 136), alpha 1.
//
// PINNED against stb_image itself: oracle/ref_dump.cpp runs the reference's own ext/stb_image.h (build container
// only) and tests/test_reference_pins.py compares these decoders with its output BIT FOR BIT on every JPEG the
// reference ships and on two Radiance files (fixtures tests/golden/ref_stb_*.npz). tests/test_image_decode.py keeps
// the independent cross-check against libjpeg-turbo (through PIL) within a stated tolerance.
#include "image_decode.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace imgdec {
namespace {

// ------------------------------------------------------------------------------------ JPEG
const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    // canonical code description (T.81 annex C / F.2.2.3)
    int32_t mincode[17] = {0}, maxcode[18] = {0}, valptr[17] = {0};
    uint8_t values[256] = {0};
    uint16_t lookup[512] = {0};   // 9-bit fast path: (length << 8) | symbol, 0 = not decodable in 9 bits
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w2 = 0, h2 = 0;          // padded plane size in samples
    int dc_pred = 0;
    std::vector<uint8_t> plane;
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void refill() {
        while (nbits <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;      // stuffed zero
                    else { hit_marker = true; byte = 0; }          // a marker: feed zeros from here on
                } else {
                    ++p;
                }
            }
            acc |= byte << (24 - nbits);
            nbits += 8;
        }
    }
    uint32_t peek(int n) { if (nbits < n) refill(); return acc >> (32 - n); }
    void drop(int n) { acc <<= n; nbits -= n; }
    int get(int n) {
        if (n == 0) return 0;
        const uint32_t v = peek(n);
        drop(n);
        return (int)v;
    }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

bool build_huffman(HuffTable& t, const uint8_t counts[16], const uint8_t* vals, int nvals) {
    memcpy(t.values, vals, (size_t)nvals);
    int code = 0, k = 0;
    memset(t.lookup, 0, sizeof(t.lookup));
    for (int len = 1; len <= 16; ++len) {
        t.valptr[len] = k;
        t.mincode[len] = code;
        for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
            if (len <= 9) {
                const int shift = 9 - len;
                for (int f = 0; f < (1 << shift); ++f) t.lookup[(code << shift) | f] = (uint16_t)((len << 8) | vals[k]);
            }
        }
        t.maxcode[len] = counts[len - 1] ? code - 1 : -1;
        if (code > (1 << len)) return false;
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    t.present = true;
    return k == nvals;
}

int decode_symbol(BitReader& br, const HuffTable& t) {
    const uint16_t fast = t.lookup[br.peek(9)];
    if (fast) {
        br.drop(fast >> 8);
        return fast & 255;
    }
    int code = (int)br.peek(16);
    for (int len = 10; len <= 16; ++len) {
        const int c = code >> (16 - len);
        if (t.maxcode[len] >= 0 && c <= t.maxcode[len] && c >= t.mincode[len]) {
            br.drop(len);
            return t.values[t.valptr[len] + c - t.mincode[len]];
        }
    }
    return -1;
}

inline int extend(int v, int nbits) {   // T.81 F.2.2.1
    return (nbits && v < (1 << (nbits - 1))) ? v - (1 << nbits) + 1 : v;
}

inline uint8_t clamp_u8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// One 1-D pass of the Loeffler-Ligtenberg-Moschytz inverse DCT, constants in 12-bit fixed point.
// The pictures must come out with stb_image's bits (the reference's textures are what stb decodes), so this pass and
// idct_block follow stb's scalar arithmetic (STBI__IDCT_1D / stbi__idct_block, ext/stb_image.h -- public domain / MIT,
// itself IJG's jidctint) term by term: same constants, same association, same rounding biases. That includes how the
// constants are rounded: (int)(x * 4096 + 0.5) truncates toward zero, so the NEGATIVE constants end up one unit
// closer to zero than -FIX(|x|) (e.g. -1.847759065 -> -7567, not -7568). tests/test_reference_pins.py holds the
// bit-for-bit comparison with stb_image on every JPEG the reference ships.
#define IMGDEC_FIX(x) ((int)((x) * 4096 + 0.5))
struct Idct1D { int x0, x1, x2, x3, t0, t1, t2, t3; };
inline Idct1D idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
    Idct1D r;
    int p1 = (s2 + s6) * IMGDEC_FIX(0.5411961f);
    int t2 = p1 + s6 * IMGDEC_FIX(-1.847759065f);
    int t3 = p1 + s2 * IMGDEC_FIX(0.765366865f);
    int t0 = (s0 + s4) * 4096;
    int t1 = (s0 - s4) * 4096;
    r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;
    int p3 = t0 + t2, p4 = t1 + t3;
    p1 = t0 + t3;
    int p2 = t1 + t2;
    const int p5 = (p3 + p4) * IMGDEC_FIX(1.175875602f);
    t0 = t0 * IMGDEC_FIX(0.298631336f);
    t1 = t1 * IMGDEC_FIX(2.053119869f);
    t2 = t2 * IMGDEC_FIX(3.072711026f);
    t3 = t3 * IMGDEC_FIX(1.501321110f);
    p1 = p5 + p1 * IMGDEC_FIX(-0.899976223f);
    p2 = p5 + p2 * IMGDEC_FIX(-2.562915447f);
    p3 = p3 * IMGDEC_FIX(-1.961570560f);
    p4 = p4 * IMGDEC_FIX(-0.390180644f);
    r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
    return r;
}

void idct_block(uint8_t* out, int stride, const short d[64]) {
    int v[64];
    for (int c = 0; c < 8; ++c) {   // columns; 2 extra bits of precision are kept
        const short* s = d + c;
        if (!(s[8] | s[16] | s[24] | s[32] | s[40] | s[48] | s[56])) {
            const int dc = s[0] * 4;
            for (int r = 0; r < 8; ++r) v[r * 8 + c] = dc;
            continue;
        }
        Idct1D k = idct_1d(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
        k.x0 += 512; k.x1 += 512; k.x2 += 512; k.x3 += 512;
        v[0 * 8 + c] = (k.x0 + k.t3) >> 10; v[7 * 8 + c] = (k.x0 - k.t3) >> 10;
        v[1 * 8 + c] = (k.x1 + k.t2) >> 10; v[6 * 8 + c] = (k.x1 - k.t2) >> 10;
        v[2 * 8 + c] = (k.x2 + k.t1) >> 10; v[5 * 8 + c] = (k.x2 - k.t1) >> 10;
        v[3 * 8 + c] = (k.x3 + k.t0) >> 10; v[4 * 8 + c] = (k.x3 - k.t0) >> 10;
    }
    for (int r = 0; r < 8; ++r) {   // rows; +128 level shift and rounding folded into the bias
        const int* s = v + r * 8;
        Idct1D k = idct_1d(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
        const int bias = 65536 + (128 << 17);
        k.x0 += bias; k.x1 += bias; k.x2 += bias; k.x3 += bias;
        uint8_t* o = out + r * stride;
        o[0] = clamp_u8((k.x0 + k.t3) >> 17); o[7] = clamp_u8((k.x0 - k.t3) >> 17);
        o[1] = clamp_u8((k.x1 + k.t2) >> 17); o[6] = clamp_u8((k.x1 - k.t2) >> 17);
        o[2] = clamp_u8((k.x2 + k.t1) >> 17); o[5] = clamp_u8((k.x2 - k.t1) >> 17);
        o[3] = clamp_u8((k.x3 + k.t0) >> 17); o[4] = clamp_u8((k.x3 - k.t0) >> 17);
    }
}

struct Jpeg {
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, restart_interval = 0;
    uint16_t quant[4][64] = {{0}};
    bool quant_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[3];
    bool adobe_rgb = false;   // APP14 transform 0 with 3 components: samples are RGB already
    bool jfif = false;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

bool decode_block(BitReader& br, Jpeg& j, Component& c, short blk[64], std::string& err) {
    memset(blk, 0, 64 * sizeof(short));
    const int s = decode_symbol(br, j.dc[c.td]);
    if (s < 0 || s > 15) { err = "bad DC huffman code"; return false; }
    const int diff = s ? extend(br.get(s), s) : 0;
    c.dc_pred += diff;
    const uint16_t* q = j.quant[c.tq];
    blk[0] = (short)(c.dc_pred * q[0]);
    for (int k = 1; k < 64;) {
        const int rs = decode_symbol(br, j.ac[c.ta]);
        if (rs < 0) { err = "bad AC huffman code"; return false; }
        const int r = rs >> 4, sz = rs & 15;
        if (sz == 0) {
            if (r != 15) break;   // EOB
            k += 16;
            continue;
        }
        k += r;
        if (k > 63) { err = "AC run past the block"; return false; }
        blk[kZigZag[k]] = (short)(extend(br.get(sz), sz) * q[k]);
        ++k;
    }
    return true;
}

bool decode_scan(Jpeg& j, const int* order, int n_in_scan, const uint8_t*& p, const uint8_t* end, std::string& err) {
    BitReader br{p, end};
    for (int i = 0; i < j.ncomp; ++i) j.comp[i].dc_pred = 0;
    short blk[64];
    int todo = j.restart_interval ? j.restart_interval : 0x7fffffff;
    auto restart = [&]() -> bool {
        // byte-align, expect RSTn
        const uint8_t* q = br.p;
        while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
            if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) return false;   // some other marker: end of data
            ++q;
        }
        if (q + 1 >= end) return false;
        br.p = q + 2;
        br.reset();
        for (int i = 0; i < j.ncomp; ++i) j.comp[i].dc_pred = 0;
        todo = j.restart_interval;
        return true;
    };
    if (n_in_scan == 1) {   // non-interleaved: the component's own 8x8 blocks, no MCU padding
        Component& c = j.comp[order[0]];
        const int bw = (((j.width * c.h + j.hmax - 1) / j.hmax) + 7) >> 3;
        const int bh = (((j.height * c.v + j.vmax - 1) / j.vmax) + 7) >> 3;
        for (int by = 0; by < bh; ++by)
            for (int bx = 0; bx < bw; ++bx) {
                if (!decode_block(br, j, c, blk, err)) return false;
                idct_block(&c.plane[(size_t)by * 8 * c.w2 + bx * 8], c.w2, blk);
                if (--todo <= 0) {
                    if (by == bh - 1 && bx == bw - 1) break;
                    if (!restart()) { p = br.p; return true; }
                }
            }
    } else {
        const int mcu_w = 8 * j.hmax, mcu_h = 8 * j.vmax;
        const int mx = (j.width + mcu_w - 1) / mcu_w, my = (j.height + mcu_h - 1) / mcu_h;
        for (int y = 0; y < my; ++y)
            for (int x = 0; x < mx; ++x) {
                for (int k = 0; k < n_in_scan; ++k) {
                    Component& c = j.comp[order[k]];
                    for (int v = 0; v < c.v; ++v)
                        for (int h = 0; h < c.h; ++h) {
                            if (!decode_block(br, j, c, blk, err)) return false;
                            const int px = (x * c.h + h) * 8, py = (y * c.v + v) * 8;
                            idct_block(&c.plane[(size_t)py * c.w2 + px], c.w2, blk);
                        }
                }
                if (--todo <= 0) {
                    if (y == my - 1 && x == mx - 1) break;
                    if (!restart()) { p = br.p; return true; }
                }
            }
    }
    p = br.p;
    return true;
}

// ---- chroma up-sampling: one output row of `w_out` samples from the component's rows
// "triangle" filter: every output sample is 3/4 the nearer input sample + 1/4 the farther one
// `rows` = the component's rows that belong to the picture, ceil(height * v / vmax): the rows below them in the plane
// are MCU padding and never take part (stb_image advances its second line only while ypos < img_comp.y).
void upsample_row(uint8_t* out, const Component& c, int hs, int vs, int y_out, int w_out, int rows, std::vector<int>& tmp) {
    const int w_in = (w_out + hs - 1) / hs;
    const int y_in = y_out / vs;
    const uint8_t* near = &c.plane[(size_t)std::min(y_in, rows - 1) * c.w2];
    if (hs == 1 && vs == 1) { memcpy(out, near, (size_t)w_out); return; }
    if (vs == 2 && hs <= 2) {
        const bool lower = (y_out & 1) != 0;                 // odd output rows lean on the next input row
        int yf = lower ? y_in + 1 : y_in - 1;
        yf = std::max(0, std::min(yf, std::min(rows, c.h2) - 1));
        const uint8_t* far = &c.plane[(size_t)yf * c.w2];
        if (hs == 1) {
            for (int i = 0; i < w_out; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
            return;
        }
        tmp.resize((size_t)w_in);
        for (int i = 0; i < w_in; ++i) tmp[i] = 3 * near[i] + far[i];
        if (w_in == 1) {
            out[0] = (uint8_t)((tmp[0] + 2) >> 2);
            if (w_out > 1) out[1] = out[0];
            return;
        }
        std::vector<uint8_t> wide((size_t)w_in * 2);
        wide[0] = (uint8_t)((tmp[0] + 2) >> 2);
        for (int i = 1; i < w_in; ++i) {
            wide[i * 2 - 1] = (uint8_t)((3 * tmp[i - 1] + tmp[i] + 8) >> 4);
            wide[i * 2] = (uint8_t)((3 * tmp[i] + tmp[i - 1] + 8) >> 4);
        }
        wide[w_in * 2 - 1] = (uint8_t)((tmp[w_in - 1] + 2) >> 2);
        memcpy(out, wide.data(), (size_t)w_out);
        return;
    }
    if (hs == 2 && vs == 1) {
        std::vector<uint8_t> wide((size_t)w_in * 2);
        if (w_in == 1) {
            wide[0] = wide[1] = near[0];
        } else {
            wide[0] = near[0];
            wide[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
            int i;
            for (i = 1; i < w_in - 1; ++i) {
                const int n = 3 * near[i] + 2;
                wide[i * 2] = (uint8_t)((n + near[i - 1]) >> 2);
                wide[i * 2 + 1] = (uint8_t)((n + near[i + 1]) >> 2);
            }
            // stb_image weighs in[w-2] by 3 here (ext/stb_image.h:3462), not the nearer in[w-1]; the
            // reference's textures are what stb produces, so the quirk is kept
            wide[i * 2] = (uint8_t)((near[w_in - 2] * 3 + near[w_in - 1] + 2) >> 2);
            wide[i * 2 + 1] = near[w_in - 1];
        }
        memcpy(out, wide.data(), (size_t)w_out);
        return;
    }
    for (int i = 0; i < w_out; ++i) out[i] = near[std::min(i / hs, c.w2 - 1)];   // other factors: replicate
}

}  // namespace

bool decode_jpeg(const uint8_t* data, size_t size, Image8& img, std::string& err) {
    img = Image8{};
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) { err = "not a JPEG (no SOI)"; return false; }
    Jpeg j;
    const uint8_t* p = data + 2;
    const uint8_t* end = data + size;
    bool have_frame = false, decoded = false;
    int adobe_transform = -1;
    while (p + 4 <= end) {
        if (p[0] != 0xFF) { ++p; continue; }
        while (p < end && *p == 0xFF) ++p;   // fill bytes
        if (p >= end) break;
        const int m = *p++;
        if (m == 0xD9) break;                                   // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;    // standalone markers
        if (p + 2 > end) break;
        const int len = be16(p);
        if (len < 2 || p + len > end) { err = "truncated JPEG segment"; return false; }
        const uint8_t* s = p + 2;
        const uint8_t* se = p + len;
        p += len;
        if (m == 0xDB) {                                        // DQT
            while (s < se) {
                const int pq = s[0] >> 4, tq = s[0] & 15;
                ++s;
                if (tq > 3 || (pq != 0 && pq != 1)) { err = "bad DQT"; return false; }
                if (s + (pq ? 128 : 64) > se) { err = "truncated DQT"; return false; }
                for (int i = 0; i < 64; ++i) { j.quant[tq][i] = pq ? (uint16_t)be16(s + 2 * i) : s[i]; }
                s += pq ? 128 : 64;
                j.quant_present[tq] = true;
            }
        } else if (m == 0xC4) {                                 // DHT
            while (s + 17 <= se) {
                const int tc = s[0] >> 4, th = s[0] & 15;
                if (tc > 1 || th > 3) { err = "bad DHT"; return false; }
                int n = 0;
                for (int i = 0; i < 16; ++i) n += s[1 + i];
                if (n > 256 || s + 17 + n > se) { err = "truncated DHT"; return false; }
                if (!build_huffman(tc ? j.ac[th] : j.dc[th], s + 1, s + 17, n)) { err = "inconsistent DHT"; return false; }
                s += 17 + n;
            }
        } else if (m == 0xC0 || m == 0xC1) {                    // SOF0 / SOF1: baseline / extended sequential, Huffman
            if (se - s < 6) { err = "truncated SOF"; return false; }
            if (s[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
            j.height = be16(s + 1);
            j.width = be16(s + 3);
            j.ncomp = s[5];
            if (j.width <= 0 || j.height <= 0) { err = "empty JPEG frame"; return false; }
            if (j.ncomp != 1 && j.ncomp != 3) { err = "JPEG with " + std::to_string(j.ncomp) + " components is not supported"; return false; }
            if (se - s < 6 + 3 * j.ncomp) { err = "truncated SOF"; return false; }
            for (int i = 0; i < j.ncomp; ++i) {
                Component& c = j.comp[i];
                c.id = s[6 + 3 * i];
                c.h = s[7 + 3 * i] >> 4;
                c.v = s[7 + 3 * i] & 15;
                c.tq = s[8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) { err = "bad SOF component"; return false; }
                j.hmax = std::max(j.hmax, c.h);
                j.vmax = std::max(j.vmax, c.v);
            }
            const int mx = (j.width + 8 * j.hmax - 1) / (8 * j.hmax), my = (j.height + 8 * j.vmax - 1) / (8 * j.vmax);
            for (int i = 0; i < j.ncomp; ++i) {
                Component& c = j.comp[i];
                c.w2 = mx * c.h * 8;
                c.h2 = my * c.v * 8;
                c.plane.assign((size_t)c.w2 * c.h2, 0);
            }
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            err = m == 0xC2 ? "progressive JPEG is not supported (baseline sequential only)" : "unsupported JPEG process (SOF" + std::to_string(m - 0xC0) + ")";
            return false;
        } else if (m == 0xDD) {                                 // DRI
            if (se - s < 2) { err = "truncated DRI"; return false; }
            j.restart_interval = be16(s);
        } else if (m == 0xE0) {
            if (se - s >= 5 && memcmp(s, "JFIF", 5) == 0) j.jfif = true;
        } else if (m == 0xEE) {
            if (se - s >= 12 && memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
        } else if (m == 0xDA) {                                 // SOS
            if (!have_frame) { err = "SOS before SOF"; return false; }
            if (se - s < 1) { err = "truncated SOS"; return false; }
            const int n = s[0];
            if (n < 1 || n > j.ncomp || se - s < 1 + 2 * n + 3) { err = "bad SOS"; return false; }
            int order[3];
            for (int k = 0; k < n; ++k) {
                int which = -1;
                for (int i = 0; i < j.ncomp; ++i) if (j.comp[i].id == s[1 + 2 * k]) which = i;
                if (which < 0) { err = "SOS names an unknown component"; return false; }
                Component& c = j.comp[which];
                c.td = s[2 + 2 * k] >> 4;
                c.ta = s[2 + 2 * k] & 15;
                if (c.td > 3 || c.ta > 3 || !j.dc[c.td].present || !j.ac[c.ta].present || !j.quant_present[c.tq]) { err = "SOS uses an undefined table"; return false; }
                order[k] = which;
            }
            if (!decode_scan(j, order, n, p, end, err)) return false;
            decoded = true;
        }
    }
    if (!decoded) { if (err.empty()) err = "JPEG has no scan"; return false; }

    // ---- planes -> RGBA8 (req_comp = 4: grey replicated, alpha 255)
    img.width = (uint32_t)j.width;
    img.height = (uint32_t)j.height;
    img.rgba.resize((size_t)j.width * j.height * 4);
    std::vector<uint8_t> rows[3];
    std::vector<int> tmp;
    const bool is_rgb = j.ncomp == 3 && (adobe_transform == 0 || (!j.jfif && adobe_transform < 0 && j.comp[0].id == 'R' && j.comp[1].id == 'G' && j.comp[2].id == 'B'));
    for (int i = 0; i < j.ncomp; ++i) rows[i].resize((size_t)j.width + 16);
    for (int y = 0; y < j.height; ++y) {
        for (int i = 0; i < j.ncomp; ++i) upsample_row(rows[i].data(), j.comp[i], j.hmax / j.comp[i].h, j.vmax / j.comp[i].v, y, j.width,
                                                         (j.height * j.comp[i].v + j.vmax - 1) / j.vmax, tmp);
        uint8_t* o = &img.rgba[(size_t)y * j.width * 4];
        if (j.ncomp == 1) {
            for (int x = 0; x < j.width; ++x) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = rows[0][x]; o[4 * x + 3] = 255; }
        } else if (is_rgb) {
            for (int x = 0; x < j.width; ++x) { o[4 * x] = rows[0][x]; o[4 * x + 1] = rows[1][x]; o[4 * x + 2] = rows[2][x]; o[4 * x + 3] = 255; }
        } else {
            // JFIF YCbCr -> RGB, 20-bit fixed point (constants rounded to 12 bits, then << 8)
            const int kCrR = IMGDEC_FIX(1.40200f) << 8, kCrG = -(IMGDEC_FIX(0.71414f) << 8);
            const int kCbG = -(IMGDEC_FIX(0.34414f) << 8), kCbB = IMGDEC_FIX(1.77200f) << 8;
            for (int x = 0; x < j.width; ++x) {
                const int yf = (rows[0][x] << 20) + (1 << 19);
                const int cb = rows[1][x] - 128, cr = rows[2][x] - 128;
                const int r = (yf + cr * kCrR) >> 20;
                const int g = (yf + cr * kCrG + (int)((uint32_t)(cb * kCbG) & 0xffff0000u)) >> 20;
                const int b = (yf + cb * kCbB) >> 20;
                o[4 * x] = clamp_u8(r); o[4 * x + 1] = clamp_u8(g); o[4 * x + 2] = clamp_u8(b); o[4 * x + 3] = 255;
            }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------ Radiance HDR
namespace {

bool read_line(const uint8_t*& p, const uint8_t* end, std::string& line) {
    line.clear();
    if (p >= end) return false;
    while (p < end && *p != '\n') line.push_back((char)*p++);
    if (p < end) ++p;
    return true;
}

inline void rgbe_to_float(const uint8_t e[4], float* out) {
    if (e[3] != 0) {
        const float f = std::ldexp(1.0f, (int)e[3] - (128 + 8));
        out[0] = e[0] * f; out[1] = e[1] * f; out[2] = e[2] * f;
    } else {
        out[0] = out[1] = out[2] = 0.f;
    }
    out[3] = 1.f;
}

}  // namespace

bool decode_hdr(const uint8_t* data, size_t size, ImageF& img, std::string& err) {
    img = ImageF{};
    const uint8_t* p = data;
    const uint8_t* end = data + size;
    std::string line;
    if (!read_line(p, end, line) || (line != "#?RADIANCE" && line != "#?RGBE")) { err = "not a Radiance HDR file"; return false; }
    bool format_ok = false;
    for (;;) {
        if (!read_line(p, end, line)) { err = "truncated HDR header"; return false; }
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) { err = "unsupported HDR format (need FORMAT=32-bit_rle_rgbe)"; return false; }
    if (!read_line(p, end, line)) { err = "truncated HDR header"; return false; }
    int h = 0, w = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || h <= 0 || w <= 0) { err = "unsupported HDR orientation (need '-Y h +X w')"; return false; }
    if ((uint64_t)w * (uint64_t)h > (1ull << 28)) { err = "HDR image too large"; return false; }
    img.width = (uint32_t)w;
    img.height = (uint32_t)h;
    img.rgba.resize((size_t)w * h * 4);
    std::vector<uint8_t> scan((size_t)w * 4);
    auto flat_from = [&](int row, int col0, const uint8_t first[4], bool have_first) -> bool {
        // rest of the file is uncompressed RGBE quadruples
        size_t idx = (size_t)row * w + col0;
        const size_t total = (size_t)w * h;
        if (have_first) { rgbe_to_float(first, &img.rgba[idx * 4]); ++idx; }
        for (; idx < total; ++idx) {
            if (p + 4 > end) { err = "truncated HDR pixel data"; return false; }
            rgbe_to_float(p, &img.rgba[idx * 4]);
            p += 4;
        }
        return true;
    };
    if (w < 8 || w >= 32768) return flat_from(0, 0, nullptr, false);
    for (int y = 0; y < h; ++y) {
        if (p + 4 > end) { err = "truncated HDR scanline"; return false; }
        const uint8_t c1 = p[0], c2 = p[1], len_hi = p[2];
        if (c1 != 2 || c2 != 2 || (len_hi & 0x80)) {
            // not run-length encoded: this quadruple is the first pixel of flat data (only legal on the first row)
            if (y != 0) { err = "corrupt HDR: flat data inside an RLE file"; return false; }
            uint8_t first[4] = {p[0], p[1], p[2], p[3]};
            p += 4;
            return flat_from(0, 0, first, true);
        }
        if (((len_hi << 8) | p[3]) != w) { err = "corrupt HDR: scanline width mismatch"; return false; }
        p += 4;
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < w) {
                if (p >= end) { err = "truncated HDR run"; return false; }
                int count = *p++;
                if (count > 128) {      // run
                    count -= 128;
                    if (count == 0 || i + count > w || p >= end) { err = "corrupt HDR run"; return false; }
                    const uint8_t v = *p++;
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = v;
                } else {                // literal
                    if (count == 0 || i + count > w || p + count > end) { err = "corrupt HDR literal"; return false; }
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = *p++;
                }
            }
        }
        for (int x = 0; x < w; ++x) rgbe_to_float(&scan[(size_t)x * 4], &img.rgba[((size_t)y * w + x) * 4]);
    }
    return true;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    return ok;
}

}  // namespace imgdec
