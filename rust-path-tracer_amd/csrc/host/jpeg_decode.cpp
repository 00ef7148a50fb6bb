/*
 * jpeg_decode.cpp — JPEG textures and skyboxes, as the reference's loader sees them.
 *
 * The reference decodes images with the `image` crate 0.24.6 (src/asset.rs:27-44, 238-262), whose JPEG path is
 * jpeg-decoder 0.3.0 (Cargo.lock:1248-1251) followed by DynamicImage::to_rgba8().  Neither crate is vendored under
 * /root/reference; this file restates jpeg-decoder's published pipeline:
 *   - baseline, extended-sequential and progressive Huffman JPEG, 8-bit samples, 1 (grey) or 3 (YCbCr) components,
 *     restart intervals; arithmetic coding, 12-bit, lossless and 4-component (CMYK) files are refused like the crate does
 *     for the cases it cannot produce RGB8 from
 *   - the integer IDCT with dequantisation of stb_image (stbi__idct_block), which jpeg-decoder's idct.rs ports:
 *     12-bit fixed-point constants, columns first (>> 10 after + 512), then rows (+ 65536 + (128 << 17), >> 17), clamp
 *   - chroma up-sampling: the triangle filters of its upsampler.rs for h2v1 / h1v2 / h2v2 (3:1 weights, + 2 >> 2 and
 *     + 8 >> 4), nearest for any other ratio
 *   - YCbCr -> RGB in 20-bit fixed point (1.40200, 0.34414, 0.71414, 1.77200; + half before the shift, clamp)
 * A texel decoded here can differ from the reference's by 1 LSB if a detail of those steps is recalled wrongly (there is
 * no Rust toolchain to check against); against libjpeg-turbo the decoder stays within the tolerance two conforming
 * decoders have (tests/test_textures.py).  Input is untrusted: every length, index and table id is checked, nothing
 * allocates from a header field without a bound, and a truncated scan decodes as far as it goes (zeros afterwards).
 */
#include <algorithm>
#include <cstring>

#include "host_internal.h"

namespace rpth {
namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t fast_len[512];      /* 9-bit prefix -> code length (0 = longer) */
    uint8_t fast_sym[512];
    int32_t maxcode[18];        /* first code of the next length, left-aligned to 16 bits */
    int32_t delta[17];          /* index of the first symbol of a length - its first code */
    uint8_t sym[256];
    uint16_t n_sym = 0;
    bool build(const uint8_t counts[16], const uint8_t *symbols, int n) {
        if (n > 256) return false;
        n_sym = (uint16_t)n;
        memcpy(sym, symbols, (size_t)n);
        memset(fast_len, 0, sizeof(fast_len));
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            delta[len] = k - code;
            if (code + counts[len - 1] > (1 << len)) return false;           /* over-subscribed */
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
                if (len <= 9) {
                    const int first = code << (9 - len);
                    for (int j = 0; j < (1 << (9 - len)); ++j) { fast_len[first + j] = (uint8_t)len; fast_sym[first + j] = symbols[k]; }
                }
            }
            maxcode[len] = code << (16 - len);
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;                  /* tables of the current scan */
    int blocks_w = 0, blocks_h = 0;      /* allocated blocks (MCU padded) */
    int dc_pred = 0;
    std::vector<int16_t> coef;           /* blocks_w * blocks_h * 64, natural order */
    std::vector<uint8_t> plane;          /* blocks_w * 8 x blocks_h * 8 samples */
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    int marker = 0;
    void fill() {
        while (bits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xff) {
                    int c = p < end ? *p : 0xd9;
                    while (c == 0xff && p + 1 < end) { ++p; c = *p; }       /* fill bytes */
                    if (c == 0) { ++p; }
                    else { hit_marker = true; marker = c; if (p < end) ++p; b = 0; }
                }
            }
            acc |= (uint32_t)b << (24 - bits);
            bits += 8;
        }
    }
    int get_bits(int n) {                /* n <= 16 */
        if (n == 0) return 0;
        if (bits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        bits -= n;
        return v;
    }
    int get_bit() { return get_bits(1); }
    void reset() { acc = 0; bits = 0; hit_marker = false; marker = 0; }
};

int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

int decode_symbol(BitReader &br, const Huff &h) {
    if (br.bits < 16) br.fill();
    const int look = (int)(br.acc >> 23);
    int len = h.fast_len[look];
    if (len) {
        br.acc <<= len;
        br.bits -= len;
        return h.fast_sym[look];
    }
    const int code16 = (int)(br.acc >> 16);
    for (len = 10; len <= 16; ++len)
        if (code16 < h.maxcode[len]) break;
    if (len > 16) return -1;
    const int idx = (code16 >> (16 - len)) + h.delta[len];
    br.acc <<= len;
    br.bits -= len;
    if (idx < 0 || idx >= h.n_sym) return -1;
    return h.sym[idx];
}

int stbi_f2f(double x) { return (int)(x * 4096 + 0.5); }
uint8_t clamp_u8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

/* jpeg-decoder idct.rs dequantize_and_idct_block_8x8 (a port of stb_image's stbi__idct_block); i32 wrapping arithmetic */
void idct_block(const int16_t *coef, const uint16_t *q, uint8_t *out, int stride) {
    auto w = [](int64_t x) { return (int32_t)(uint32_t)x; };                 /* wrapping, as the crate's wrapping_* ops */
    int32_t tmp[64];
#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                        \
    int32_t p2 = s2, p3 = s6;                                                          \
    int32_t p1 = w((int64_t)w((int64_t)p2 + p3) * stbi_f2f(0.5411961));                \
    int32_t t2 = w((int64_t)p1 + w((int64_t)p3 * stbi_f2f(-1.847759065)));             \
    int32_t t3 = w((int64_t)p1 + w((int64_t)p2 * stbi_f2f(0.765366865)));              \
    p2 = s0; p3 = s4;                                                                  \
    int32_t t0 = w(((int64_t)p2 + p3) * 4096), t1 = w(((int64_t)p2 - p3) * 4096);      \
    int32_t x0 = w((int64_t)t0 + t3), x3 = w((int64_t)t0 - t3), x1 = w((int64_t)t1 + t2), x2 = w((int64_t)t1 - t2); \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                \
    p3 = w((int64_t)t0 + t2);                                                          \
    int32_t p4 = w((int64_t)t1 + t3);                                                  \
    p1 = w((int64_t)t0 + t3);                                                          \
    p2 = w((int64_t)t1 + t2);                                                          \
    int32_t p5 = w((int64_t)w((int64_t)p3 + p4) * stbi_f2f(1.175875602));              \
    t0 = w((int64_t)t0 * stbi_f2f(0.298631336));                                       \
    t1 = w((int64_t)t1 * stbi_f2f(2.053119869));                                       \
    t2 = w((int64_t)t2 * stbi_f2f(3.072711026));                                       \
    t3 = w((int64_t)t3 * stbi_f2f(1.501321110));                                       \
    p1 = w((int64_t)p5 + w((int64_t)p1 * stbi_f2f(-0.899976223)));                     \
    p2 = w((int64_t)p5 + w((int64_t)p2 * stbi_f2f(-2.562915447)));                     \
    p3 = w((int64_t)p3 * stbi_f2f(-1.961570560));                                      \
    p4 = w((int64_t)p4 * stbi_f2f(-0.390180644));                                      \
    t3 = w((int64_t)t3 + p1 + p4);                                                     \
    t2 = w((int64_t)t2 + p2 + p3);                                                     \
    t1 = w((int64_t)t1 + p2 + p4);                                                     \
    t0 = w((int64_t)t0 + p1 + p3);
    for (int i = 0; i < 8; ++i) {                                            /* columns */
        if (coef[i + 8] == 0 && coef[i + 16] == 0 && coef[i + 24] == 0 && coef[i + 32] == 0 && coef[i + 40] == 0 && coef[i + 48] == 0 &&
            coef[i + 56] == 0) {
            const int32_t dc = w((int64_t)coef[i] * q[i] * 4);
            for (int k = 0; k < 8; ++k) tmp[i + 8 * k] = dc;
            continue;
        }
        const int32_t s0 = w((int64_t)coef[i] * q[i]), s1 = w((int64_t)coef[i + 8] * q[i + 8]), s2 = w((int64_t)coef[i + 16] * q[i + 16]),
                      s3 = w((int64_t)coef[i + 24] * q[i + 24]), s4 = w((int64_t)coef[i + 32] * q[i + 32]), s5 = w((int64_t)coef[i + 40] * q[i + 40]),
                      s6 = w((int64_t)coef[i + 48] * q[i + 48]), s7 = w((int64_t)coef[i + 56] * q[i + 56]);
        IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)
        x0 = w((int64_t)x0 + 512); x1 = w((int64_t)x1 + 512); x2 = w((int64_t)x2 + 512); x3 = w((int64_t)x3 + 512);
        tmp[i] = w((int64_t)x0 + t3) >> 10;
        tmp[i + 56] = w((int64_t)x0 - t3) >> 10;
        tmp[i + 8] = w((int64_t)x1 + t2) >> 10;
        tmp[i + 48] = w((int64_t)x1 - t2) >> 10;
        tmp[i + 16] = w((int64_t)x2 + t1) >> 10;
        tmp[i + 40] = w((int64_t)x2 - t1) >> 10;
        tmp[i + 24] = w((int64_t)x3 + t0) >> 10;
        tmp[i + 32] = w((int64_t)x3 - t0) >> 10;
    }
    for (int i = 0; i < 8; ++i) {                                            /* rows */
        const int32_t *s = tmp + 8 * i;
        IDCT_1D(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7])
        const int32_t bias = 65536 + (128 << 17);
        x0 = w((int64_t)x0 + bias); x1 = w((int64_t)x1 + bias); x2 = w((int64_t)x2 + bias); x3 = w((int64_t)x3 + bias);
        uint8_t *o = out + i * stride;
        o[0] = clamp_u8(w((int64_t)x0 + t3) >> 17);
        o[7] = clamp_u8(w((int64_t)x0 - t3) >> 17);
        o[1] = clamp_u8(w((int64_t)x1 + t2) >> 17);
        o[6] = clamp_u8(w((int64_t)x1 - t2) >> 17);
        o[2] = clamp_u8(w((int64_t)x2 + t1) >> 17);
        o[5] = clamp_u8(w((int64_t)x2 - t1) >> 17);
        o[3] = clamp_u8(w((int64_t)x3 + t0) >> 17);
        o[4] = clamp_u8(w((int64_t)x3 - t0) >> 17);
    }
#undef IDCT_1D
}

struct Decoder {
    const uint8_t *data;
    size_t size, pos = 0;
    int width = 0, height = 0, n_comp = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    Component comp[3];
    int adobe_transform = -1;
    uint32_t eobrun = 0;

    bool fail(const char *why) { set_error(std::string("JPEG: ") + why); return false; }
    int u8() { return pos < size ? data[pos++] : -1; }
    int u16() { if (pos + 2 > size) { pos = size; return -1; } int v = (data[pos] << 8) | data[pos + 1]; pos += 2; return v; }

    bool read_dqt(size_t end) {
        while (pos < end) {
            const int pq_tq = u8();
            const int pq = pq_tq >> 4, tq = pq_tq & 15;
            if (pq > 1 || tq > 3) return fail("bad quantisation table");
            if (pos + (size_t)(pq ? 128 : 64) > end) return fail("truncated quantisation table");
            for (int i = 0; i < 64; ++i) qt[tq][ZIGZAG[i]] = (uint16_t)(pq ? u16() : u8());
            qt_present[tq] = true;
        }
        return true;
    }
    bool read_dht(size_t end) {
        while (pos < end) {
            const int tc_th = u8();
            const int tc = tc_th >> 4, th = tc_th & 15;
            if (tc > 1 || th > 3 || pos + 16 > end) return fail("bad Huffman table");
            uint8_t counts[16];
            int n = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)u8(); n += counts[i]; }
            if (n > 256 || pos + (size_t)n > end) return fail("bad Huffman table");
            if (!(tc ? hac[th] : hdc[th]).build(counts, data + pos, n)) return fail("invalid Huffman code lengths");
            pos += (size_t)n;
        }
        return true;
    }
    bool read_sof(size_t end, int marker) {
        if (have_frame) return fail("more than one frame");
        if (pos + 6 > end) return fail("truncated frame header");
        const int precision = u8();
        height = u16();
        width = u16();
        n_comp = u8();
        if (precision != 8) return fail("only 8-bit samples are supported");
        if (width <= 0 || height <= 0 || width > 16384 || height > 16384) return fail("unsupported image size");
        if (n_comp != 1 && n_comp != 3) return fail("only greyscale and YCbCr images are supported (1 or 3 components)");
        if (pos + (size_t)3 * n_comp > end) return fail("truncated frame header");
        for (int i = 0; i < n_comp; ++i) {
            comp[i].id = u8();
            const int hv = u8();
            comp[i].h = hv >> 4; comp[i].v = hv & 15; comp[i].tq = u8();
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return fail("bad sampling factors");
            hmax = std::max(hmax, comp[i].h); vmax = std::max(vmax, comp[i].v);
        }
        if (n_comp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }      /* a single component is never interleaved */
        const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < n_comp; ++i) {
            comp[i].blocks_w = mcus_x * comp[i].h;
            comp[i].blocks_h = mcus_y * comp[i].v;
            const size_t n_blocks = (size_t)comp[i].blocks_w * comp[i].blocks_h;
            if (n_blocks > (size_t)1 << 23) return fail("image too large");
            comp[i].coef.assign(n_blocks * 64, 0);
        }
        progressive = marker == 0xc2;
        have_frame = true;
        return true;
    }

    /* one block of a sequential scan */
    bool block_baseline(BitReader &br, Component &c, int16_t *b) {
        const Huff &dc = hdc[c.td], &ac = hac[c.ta];
        const int t = decode_symbol(br, dc);
        if (t < 0 || t > 15) return false;
        const int diff = t ? extend(br.get_bits(t), t) : 0;
        c.dc_pred = (int)((uint32_t)c.dc_pred + (uint32_t)diff);      /* wraps, as jpeg-decoder's wrapping_add: a hostile file must not be UB */
        b[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(br, ac);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return false;
            b[ZIGZAG[k]] = (int16_t)extend(br.get_bits(s), s);
            ++k;
        }
        return true;
    }
    bool block_dc_first(BitReader &br, Component &c, int16_t *b, int al) {
        const int t = decode_symbol(br, hdc[c.td]);
        if (t < 0 || t > 15) return false;
        const int diff = t ? extend(br.get_bits(t), t) : 0;
        c.dc_pred = (int)((uint32_t)c.dc_pred + (uint32_t)diff);
        b[0] = (int16_t)((uint32_t)c.dc_pred << al);
        return true;
    }
    static void block_dc_refine(BitReader &br, int16_t *b, int al) {
        if (br.get_bit()) b[0] = (int16_t)(b[0] | (1 << al));
    }
    bool block_ac_first(BitReader &br, Component &c, int16_t *b, int ss, int se, int al) {
        if (eobrun > 0) { --eobrun; return true; }
        const Huff &ac = hac[c.ta];
        for (int k = ss; k <= se;) {
            const int rs = decode_symbol(br, ac);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1u << r) - 1;
                    if (r) eobrun += (uint32_t)br.get_bits(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return false;
            b[ZIGZAG[k]] = (int16_t)(extend(br.get_bits(s), s) * (1 << al));
            ++k;
        }
        return true;
    }
    bool block_ac_refine(BitReader &br, Component &c, int16_t *b, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -1 * (1 << al);
        const Huff &ac = hac[c.ta];
        int k = ss;
        if (eobrun == 0) {
            while (k <= se) {
                const int rs = decode_symbol(br, ac);
                if (rs < 0) return false;
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = (1u << r);
                        if (r) eobrun += (uint32_t)br.get_bits(r);
                        break;
                    }
                } else {
                    if (s != 1) return false;
                    value = br.get_bit() ? p1 : m1;
                }
                while (k <= se) {
                    int16_t &coef = b[ZIGZAG[k]];
                    if (coef != 0) {
                        if (br.get_bit() && (coef & p1) == 0) coef = (int16_t)(coef >= 0 ? coef + p1 : coef + m1);
                    } else {
                        if (r == 0) {
                            if (value) coef = (int16_t)value;
                            ++k;
                            break;
                        }
                        --r;
                    }
                    ++k;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t &coef = b[ZIGZAG[k]];
                if (coef != 0 && br.get_bit() && (coef & p1) == 0) coef = (int16_t)(coef >= 0 ? coef + p1 : coef + m1);
            }
            --eobrun;
        }
        return true;
    }

    bool read_sos(size_t end) {
        if (!have_frame) return fail("scan before frame header");
        const int ns = u8();
        if (ns < 1 || ns > n_comp || pos + (size_t)2 * ns + 3 > end) return fail("bad scan header");
        Component *sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = u8(), tt = u8();
            sc[i] = nullptr;
            for (int j = 0; j < n_comp; ++j)
                if (comp[j].id == id) sc[i] = &comp[j];
            if (!sc[i]) return fail("scan names an unknown component");
            for (int j = 0; j < i; ++j)
                if (sc[j] == sc[i]) return fail("scan names a component twice");
            sc[i]->td = tt >> 4; sc[i]->ta = tt & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table selector");
        }
        const int ss = u8(), se = u8(), ahl = u8();
        const int ah = ahl >> 4, al = ahl & 15;
        if (progressive) {
            if (ss > se || se > 63 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) return fail("bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
            return fail("bad sequential scan parameters");
        }
        const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
        for (int i = 0; i < ns; ++i) {
            if (need_dc && !(progressive && ah != 0) && !hdc[sc[i]->td].present) return fail("scan uses an undefined DC table");
            if (need_ac && !hac[sc[i]->ta].present) return fail("scan uses an undefined AC table");
        }
        pos = end;
        BitReader br{data + pos, data + size};
        for (int i = 0; i < n_comp; ++i) comp[i].dc_pred = 0;
        eobrun = 0;
        const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        /* a non-interleaved scan covers only the blocks that hold image samples */
        const int single_w = ns == 1 ? ((width * sc[0]->h + hmax - 1) / hmax + 7) / 8 : 0;
        const int single_h = ns == 1 ? ((height * sc[0]->v + vmax - 1) / vmax + 7) / 8 : 0;
        const long total = ns == 1 ? (long)single_w * single_h : (long)mcus_x * mcus_y;
        int next_rst = 0;
        bool ok = true;
        auto one = [&](Component &c, int bx, int by) -> bool {
            if (bx >= c.blocks_w || by >= c.blocks_h) return false;
            int16_t *b = c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64;
            if (!progressive) return block_baseline(br, c, b);
            if (ss == 0) { if (ah == 0) return block_dc_first(br, c, b, al); block_dc_refine(br, b, al); return true; }
            return ah == 0 ? block_ac_first(br, c, b, ss, se, al) : block_ac_refine(br, c, b, ss, se, al);
        };
        for (long m = 0; m < total && ok; ++m) {
            if (restart_interval && m && m % restart_interval == 0) {
                /* RSTn: byte-align, expect the marker, reset predictions */
                br.bits = 0; br.acc = 0;
                if (!br.hit_marker) br.fill();
                if (br.hit_marker && br.marker >= 0xd0 && br.marker <= 0xd7) {
                    next_rst = (next_rst + 1) & 7;
                    const uint8_t *resume = br.p;
                    br.reset();
                    br.p = resume;
                } else {
                    break;                                   /* damaged or truncated: keep what was decoded */
                }
                for (int i = 0; i < n_comp; ++i) comp[i].dc_pred = 0;
                eobrun = 0;
            }
            if (ns == 1) {
                ok = one(*sc[0], (int)(m % single_w), (int)(m / single_w));
            } else {
                const int mx = (int)(m % mcus_x), my = (int)(m / mcus_x);
                for (int i = 0; i < ns && ok; ++i)
                    for (int v = 0; v < sc[i]->v && ok; ++v)
                        for (int h = 0; h < sc[i]->h && ok; ++h) ok = one(*sc[i], mx * sc[i]->h + h, my * sc[i]->v + v);
            }
            if (br.hit_marker && br.bits <= 0 && !(br.marker >= 0xd0 && br.marker <= 0xd7)) break;      /* ran into the next segment */
        }
        /* continue parsing at the marker that ended the entropy-coded data */
        if (br.hit_marker) {
            pos = (size_t)(br.p - data) - 2;
        } else {
            size_t q = (size_t)(br.p - data);
            while (q + 1 < size && !(data[q] == 0xff && data[q + 1] != 0 && data[q + 1] != 0xff && !(data[q + 1] >= 0xd0 && data[q + 1] <= 0xd7))) ++q;
            pos = q;
        }
        return true;                                         /* a damaged scan is not fatal: the crate, too, returns what it has */
    }

    bool parse() {
        if (size < 4 || data[0] != 0xff || data[1] != 0xd8) return fail("not a JPEG file");
        pos = 2;
        bool seen_scan = false;
        while (pos + 4 <= size) {
            if (data[pos] != 0xff) { ++pos; continue; }
            int marker = data[pos + 1];
            if (marker == 0xff) { ++pos; continue; }
            pos += 2;
            if (marker == 0xd9) break;                       /* EOI */
            if (marker == 0x01 || (marker >= 0xd0 && marker <= 0xd7) || marker == 0x00) continue;
            const int len = u16();
            if (len < 2 || pos + (size_t)(len - 2) > size) { if (seen_scan) break; return fail("truncated segment"); }
            const size_t end = pos + (size_t)(len - 2);
            switch (marker) {
                case 0xdb: if (!read_dqt(end)) return false; break;
                case 0xc4: if (!read_dht(end)) return false; break;
                case 0xc0: case 0xc1: case 0xc2: if (!read_sof(end, marker)) return false; break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    return fail("lossless, hierarchical and arithmetic-coded JPEG are not supported");
                case 0xdd: if (end - pos < 2) return fail("bad restart interval"); restart_interval = u16(); break;
                case 0xee:
                    if (end - pos >= 12 && !memcmp(data + pos, "Adobe", 5)) adobe_transform = data[pos + 11];
                    break;
                case 0xda:
                    if (!read_sos(end)) return false;
                    seen_scan = true;
                    continue;                                /* read_sos placed pos at the next marker */
                default: break;
            }
            pos = end;
        }
        if (!have_frame || !seen_scan) return fail("no image data");
        return true;
    }

    bool reconstruct(Image8 &out) {
        for (int i = 0; i < n_comp; ++i) {
            Component &c = comp[i];
            if (!qt_present[c.tq]) return fail("frame uses an undefined quantisation table");
            const int pw = c.blocks_w * 8;
            c.plane.assign((size_t)pw * c.blocks_h * 8, 0);
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx)
                    idct_block(c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64, qt[c.tq], c.plane.data() + (size_t)by * 8 * pw + bx * 8, pw);
        }
        out.w = (uint32_t)width; out.h = (uint32_t)height;
        out.rgba.assign((size_t)width * height * 4, 255);
        std::vector<uint8_t> line[3];
        for (int i = 0; i < n_comp; ++i) line[i].assign((size_t)width + 2 * hmax + 16, 0);
        for (int y = 0; y < height; ++y) {
            for (int i = 0; i < n_comp; ++i) {
                const Component &c = comp[i];
                const int pw = c.blocks_w * 8;
                /* component size in samples (jpeg-decoder: ceil(width * h / hmax)) */
                const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
                const uint8_t *plane = c.plane.data();
                uint8_t *o = line[i].data();
                const bool h2 = c.h * 2 == hmax, v2 = c.v * 2 == vmax, h1 = c.h == hmax, v1 = c.v == vmax;
                if (h1 && v1) {
                    memcpy(o, plane + (size_t)y * pw, (size_t)width);
                } else if (h2 && v1) {                                       /* UpsamplerH2V1 */
                    const uint8_t *in = plane + (size_t)y * pw;
                    if (cw == 1) { o[0] = o[1] = in[0]; continue; }
                    o[0] = in[0];
                    o[1] = (uint8_t)((in[0] * 3u + in[1] + 2u) >> 2);
                    for (int x = 1; x < cw - 1; ++x) {
                        const uint32_t s = 3u * in[x] + 2u;
                        o[2 * x] = (uint8_t)((s + in[x - 1]) >> 2);
                        o[2 * x + 1] = (uint8_t)((s + in[x + 1]) >> 2);
                    }
                    o[2 * (cw - 1)] = (uint8_t)((in[cw - 1] * 3u + in[cw - 2] + 2u) >> 2);
                    o[2 * (cw - 1) + 1] = in[cw - 1];
                } else if ((h1 || h2) && v2) {                               /* UpsamplerH1V2 / UpsamplerH2V2 */
                    const float row_near = (float)y / 2.0f;
                    float row_far = row_near + (row_near - (float)(int)row_near) * 3.0f - 0.25f;
                    if (row_far > (float)(ch - 1)) row_far = (float)(ch - 1);
                    const int rn = (int)row_near, rf = row_far > 0.0f ? (int)row_far : 0;
                    const uint8_t *near = plane + (size_t)std::min(rn, ch - 1) * pw, *far = plane + (size_t)rf * pw;
                    if (h1) {
                        for (int x = 0; x < width; ++x) o[x] = (uint8_t)((3u * near[x] + far[x] + 2u) >> 2);
                    } else if (cw == 1) {
                        o[0] = o[1] = (uint8_t)((3u * near[0] + far[0] + 2u) >> 2);
                    } else {
                        uint32_t t0 = 3u * near[0] + far[0], t1 = 3u * near[1] + far[1];
                        o[0] = (uint8_t)((t0 + 2u) >> 2);
                        o[1] = (uint8_t)((3u * t0 + t1 + 8u) >> 4);
                        for (int x = 2; x < cw; ++x) {
                            const uint32_t t2 = 3u * near[x] + far[x];
                            o[2 * x - 2] = (uint8_t)((3u * t1 + t0 + 8u) >> 4);
                            o[2 * x - 1] = (uint8_t)((3u * t1 + t2 + 8u) >> 4);
                            t0 = t1; t1 = t2;
                        }
                        o[2 * cw - 2] = (uint8_t)((3u * t1 + t0 + 8u) >> 4);
                        o[2 * cw - 1] = (uint8_t)((t1 + 2u) >> 2);
                    }
                } else {                                                     /* UpsamplerGeneric: nearest */
                    const int sy = std::min(y * c.v / vmax, ch - 1);
                    const uint8_t *in = plane + (size_t)sy * pw;
                    for (int x = 0; x < width; ++x) o[x] = in[std::min(x * c.h / hmax, cw - 1)];
                }
            }
            uint8_t *px = out.rgba.data() + (size_t)y * width * 4;
            if (n_comp == 1) {
                for (int x = 0; x < width; ++x) { px[4 * x] = px[4 * x + 1] = px[4 * x + 2] = line[0][x]; }
            } else if (adobe_transform == 0) {                               /* Adobe marker: the three components ARE r, g, b */
                for (int x = 0; x < width; ++x) { px[4 * x] = line[0][x]; px[4 * x + 1] = line[1][x]; px[4 * x + 2] = line[2][x]; }
            } else {
                const int SH = 20, HALF = (1 << SH) / 2;
                static const int32_t c_r = (int32_t)(1.40200f * (float)(1 << SH) + 0.5f), c_gb = (int32_t)(0.34414f * (float)(1 << SH) + 0.5f),
                                     c_gr = (int32_t)(0.71414f * (float)(1 << SH) + 0.5f), c_b = (int32_t)(1.77200f * (float)(1 << SH) + 0.5f);
                for (int x = 0; x < width; ++x) {
                    const int32_t yy = (int32_t)line[0][x] * (1 << SH) + HALF, cb = (int32_t)line[1][x] - 128, cr = (int32_t)line[2][x] - 128;
                    px[4 * x] = clamp_u8((yy + c_r * cr) >> SH);
                    px[4 * x + 1] = clamp_u8((yy - c_gb * cb - c_gr * cr) >> SH);
                    px[4 * x + 2] = clamp_u8((yy + c_b * cb) >> SH);
                }
            }
        }
        return true;
    }
};

}  // namespace

bool decode_jpeg(const uint8_t *data, size_t size, Image8 &out) {
    Decoder d{data, size};
    return d.parse() && d.reconstruct(out);
}

/* PNG or JPEG by signature (image::load_from_memory guesses the format the same way) */
bool decode_image(const uint8_t *data, size_t size, Image8 &out) {
    if (size >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff) return decode_jpeg(data, size, out);
    return decode_png(data, size, out);
}

}  // namespace rpth
