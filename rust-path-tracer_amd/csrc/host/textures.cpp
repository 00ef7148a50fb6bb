/*
 * textures.cpp — image decoding, the texture atlas and the skybox buffer of the host-side scene preparation.
 *
 * Restates (reference file:line):
 *   src/asset.rs:27-44     convert_texture / load_texture: an embedded image -> DynamicImage
 *   src/asset.rs:140-148   albedo textures: into_rgb8, every channel ((p / 255)^2.2 * 255) as u8  (gamma -> linear, truncating)
 *   src/atlas.rs:8-24      PackingRect::to_uvst — y is divided by the atlas WIDTH (harmless while the atlas is square)
 *   src/atlas.rs:26-90     pack_textures: breadth-first quad-tree split until there are more leaves than textures, leaves
 *                          sorted by width (stable, descending), texture i resized to leaf i, flipped vertically, copied
 *   src/asset.rs:238-255   load_dynamic_image (skybox file)
 *   src/asset.rs:266-273   dynamic_image_to_cpu_buffer: into_rgb8 then (r, g, b, 255) / 255 — the CPU path sees the skybox
 *                          QUANTISED to 8 bits, whatever the file held; parity follows the CPU path
 *
 * Not reproducible here: the reference resizes with fast_image_resize 2.x (Lanczos3 convolution, u8x4 fixed point), a crate
 * that is not under /root/reference.  The resize below is the same filter in the Pillow-style 22-bit fixed point that
 * crate descends from; a texture whose size equals its leaf's is copied exactly either way (the Lanczos3 weights of a 1:1
 * resize are 1 at the centre and 0 at every other integer), anything else may differ from the reference in the last bit
 * of a texel.  Decoders: PNG (this file, zlib); Radiance .hdr (RGBE, RLE and flat); JPEG (jpeg_decode.cpp).
 */
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>

#include "../rpt_math.h"
#include "host_internal.h"

namespace rpth {

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* PNG -> RGBA8 the way image 0.24's DynamicImage::into_rgba8 does: grey replicated, 16 bit narrowed as (v + 128) / 257,
 * alpha 255 where the file has none.  Non-interlaced, 8/16 bit grey / grey-alpha / RGB / RGBA and 8-bit palette. */
bool decode_png(const uint8_t *data, size_t size, Image8 &out) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 || memcmp(data, sig, 8)) { set_error("not a PNG"); return false; }
    uint32_t w = 0, h = 0, bit_depth = 0, color_type = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t at = 8;
    while (at + 12 <= size) {
        uint32_t len = be32(data + at);
        const uint8_t *type = data + at + 4;
        if (len > size || at + 12 + len > size) break;
        const uint8_t *body = data + at + 8;
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(body); h = be32(body + 4);
            bit_depth = body[8]; color_type = body[9]; interlace = body[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        at += 12 + (size_t)len;
    }
    int channels = color_type == 0 ? 1 : color_type == 2 ? 3 : color_type == 3 ? 1 : color_type == 4 ? 2 : color_type == 6 ? 4 : 0;
    /* (the pixel-count bound comes before anything is sized from the header: a 100-byte file may claim 16384 x 16384 x 8 bytes) */
    if (!w || !h || w > 16384 || h > 16384 || (uint64_t)w * h > (1ull << 26) || !channels || interlace || (bit_depth != 8 && bit_depth != 16) ||
        (color_type == 3 && (bit_depth != 8 || plte.size() < 3))) {
        set_error("unsupported PNG layout");
        return false;
    }
    size_t bpp = (size_t)channels * bit_depth / 8;
    size_t stride = (size_t)w * bpp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) {
        set_error("PNG inflate failed");
        return false;
    }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = &raw[(stride + 1) * y];
        uint8_t ft = src[0];
        uint8_t *dst = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? dst[i - bpp] : 0;
            int b = up ? up[i] : 0;
            int c = (up && i >= bpp) ? up[i - bpp] : 0;
            int x = src[1 + i], r;
            switch (ft) {
                case 0: r = x; break;
                case 1: r = x + a; break;
                case 2: r = x + b; break;
                case 3: r = x + ((a + b) >> 1); break;
                case 4: {
                    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                    int pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    r = x + pr;
                    break;
                }
                default: set_error("bad PNG filter"); return false;
            }
            dst[i] = (uint8_t)r;
        }
    }
    out.w = w; out.h = h;
    out.rgba.resize((size_t)w * h * 4);
    auto sample = [&](const uint8_t *p, int ch) -> uint8_t {
        if (bit_depth == 16) {
            uint32_t v16 = ((uint32_t)p[2 * ch] << 8) | p[2 * ch + 1];
            return (uint8_t)((v16 + 128u) / 257u);
        }
        return p[ch];
    };
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t *p = &img[stride * y + (size_t)x * bpp];
            uint8_t *o = &out.rgba[((size_t)y * w + x) * 4];
            switch (color_type) {
                case 0: o[0] = o[1] = o[2] = sample(p, 0); o[3] = 255; break;
                case 2: o[0] = sample(p, 0); o[1] = sample(p, 1); o[2] = sample(p, 2); o[3] = 255; break;
                case 3: {
                    size_t k = p[0];
                    if (3 * k + 2 >= plte.size()) k = 0;
                    o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2];
                    o[3] = k < trns.size() ? trns[k] : 255;
                    break;
                }
                case 4: o[0] = o[1] = o[2] = sample(p, 0); o[3] = sample(p, 1); break;
                default: o[0] = sample(p, 0); o[1] = sample(p, 1); o[2] = sample(p, 2); o[3] = sample(p, 3); break;
            }
        }
    return true;
}

bool read_file(const char *path, std::vector<uint8_t> &data) {
    FILE *f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return false; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    data.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
    fclose(f);
    if (got != data.size()) { set_error("short read"); return false; }
    return true;
}

/* Radiance .hdr (what image::codecs::hdr::HdrDecoder::read_image_hdr returns: Rgb<f32>, rgb = mantissa * 2^(e - 136)) */
static bool decode_hdr(const std::vector<uint8_t> &d, std::vector<float> &rgb, uint32_t &w, uint32_t &h) {
    size_t at = 0;
    auto line = [&](std::string &s) {
        s.clear();
        while (at < d.size() && d[at] != '\n') s.push_back((char)d[at++]);
        if (at < d.size()) ++at;
        return !s.empty() || at < d.size();
    };
    std::string s;
    if (!line(s) || (s.rfind("#?RADIANCE", 0) != 0 && s.rfind("#?RGBE", 0) != 0)) { set_error("not a Radiance .hdr file"); return false; }
    bool fmt = false;
    while (line(s) && !s.empty())
        if (s.rfind("FORMAT=32-bit_rle_rgbe", 0) == 0) fmt = true;
    if (!fmt) { set_error(".hdr: only FORMAT=32-bit_rle_rgbe is supported"); return false; }
    if (!line(s)) { set_error(".hdr: missing resolution line"); return false; }
    int hh = 0, ww = 0;
    if (sscanf(s.c_str(), "-Y %d +X %d", &hh, &ww) != 2 || hh <= 0 || ww <= 0 || hh > 32768 || ww > 32768 || (uint64_t)hh * (uint64_t)ww > (1ull << 26)) { set_error(".hdr: only -Y h +X w orientation, at most 2^26 pixels, is supported"); return false; }
    w = (uint32_t)ww; h = (uint32_t)hh;
    std::vector<uint8_t> px((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *row = &px[(size_t)y * w * 4];
        if (at + 4 <= d.size() && d[at] == 2 && d[at + 1] == 2 && (((uint32_t)d[at + 2] << 8) | d[at + 3]) == w && w >= 8 && w < 32768) {
            at += 4;                                           /* adaptive RLE: the four components separately */
            for (int c = 0; c < 4; ++c) {
                uint32_t x = 0;
                while (x < w) {
                    if (at >= d.size()) { set_error(".hdr: truncated"); return false; }
                    uint8_t n = d[at++];
                    if (n > 128) {
                        n -= 128;
                        if (at >= d.size() || x + n > w) { set_error(".hdr: bad run"); return false; }
                        uint8_t v = d[at++];
                        for (uint8_t k = 0; k < n; ++k) row[4 * (x++) + c] = v;
                    } else {
                        if (n == 0 || at + n > d.size() || x + n > w) { set_error(".hdr: bad run"); return false; }
                        for (uint8_t k = 0; k < n; ++k) row[4 * (x++) + c] = d[at++];
                    }
                }
            }
        } else {
            if (at + (size_t)w * 4 > d.size()) { set_error(".hdr: truncated"); return false; }
            memcpy(row, &d[at], (size_t)w * 4);
            at += (size_t)w * 4;
        }
    }
    rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t *p = &px[4 * i];
        if (p[3] == 0) { rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = 0.0f; continue; }
        float scale = std::ldexp(1.0f, (int)p[3] - 136);      /* image::codecs::hdr: Rgbe8Pixel::to_hdr, exact power of two */
        rgb[3 * i] = (float)p[0] * scale; rgb[3 * i + 1] = (float)p[1] * scale; rgb[3 * i + 2] = (float)p[2] * scale;
    }
    return true;
}

/* load_dynamic_image + dynamic_image_to_cpu_buffer (src/asset.rs:238-273): the file as the CPU path sees it — 8 bits per
 * channel, alpha 1.  f32 -> u8 is image 0.24's conversion: clamp to [0, 1], * 255, round half away from zero. */
bool load_skybox_file(const char *path, std::vector<float> &rgba, uint32_t &w, uint32_t &h) {
    std::vector<uint8_t> data;
    if (!read_file(path, data)) return false;
    std::string p(path);
    std::vector<uint8_t> rgb8;
    if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".hdr") == 0) {
        std::vector<float> rgb;
        if (!decode_hdr(data, rgb, w, h)) return false;
        rgb8.resize(rgb.size());
        for (size_t i = 0; i < rgb.size(); ++i) {
            float v = rgb[i];
            v = v != v ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
            rgb8[i] = (uint8_t)std::lround(v * 255.0f);
        }
    } else {
        Image8 img;
        if (!decode_image(data.data(), data.size(), img)) return false;      /* .png / .jpg */
        w = img.w; h = img.h;
        rgb8.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; ++i)
            for (int c = 0; c < 3; ++c) rgb8[3 * i + c] = img.rgba[4 * i + c];
    }
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        rgba[4 * i + 0] = (float)rgb8[3 * i + 0] / 255.0f;      /* Vec4(r, g, b, 255) / 255.0 */
        rgba[4 * i + 1] = (float)rgb8[3 * i + 1] / 255.0f;
        rgba[4 * i + 2] = (float)rgb8[3 * i + 2] / 255.0f;
        rgba[4 * i + 3] = 255.0f / 255.0f;
    }
    return true;
}

/* asset.rs:143-146 */
void albedo_gamma_to_linear(Image8 &img) {
    uint8_t lut[256];
    for (int v = 0; v < 256; ++v) {
        float f = rptm::powr((float)v / 255.0f, 2.2f) * 255.0f;
        lut[v] = (uint8_t)rptm::f2u32_sat(f);                   /* `as u8`: truncation */
    }
    for (size_t i = 0; i < (size_t)img.w * img.h; ++i) {
        for (int c = 0; c < 3; ++c) img.rgba[4 * i + c] = lut[img.rgba[4 * i + c]];
        img.rgba[4 * i + 3] = 255;                              /* into_rgb8 dropped alpha; to_rgba8 restores 255 */
    }
}

/* ---- Lanczos3 resize, 8-bit RGBA, two passes in 22-bit fixed point --------------------------------------- */
namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;
double lanczos3(double x) {
    if (x == 0.0) return 1.0;
    if (x <= -3.0 || x >= 3.0) return 0.0;
    double px = M_PI * x;
    return (std::sin(px) / px) * (std::sin(px / 3.0) / (px / 3.0));
}
struct Coeffs { std::vector<int> bounds; std::vector<int> k; int ksize; };
Coeffs coeffs(uint32_t in, uint32_t out) {
    Coeffs c;
    double scale = (double)in / out, filterscale = scale < 1.0 ? 1.0 : scale;
    double support = 3.0 * filterscale;
    c.ksize = (int)std::ceil(support) * 2 + 1;
    c.bounds.resize(2 * out);
    c.k.assign((size_t)out * c.ksize, 0);
    std::vector<double> kk(c.ksize);
    for (uint32_t xx = 0; xx < out; ++xx) {
        double center = (xx + 0.5) * scale, ww = 0.0, ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5); if (xmax > (int)in) xmax = (int)in;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) { kk[x] = lanczos3((x + xmin - center + 0.5) * ss); ww += kk[x]; }
        for (int x = 0; x < xmax; ++x) {
            double v = ww != 0.0 ? kk[x] / ww : 0.0;
            c.k[(size_t)xx * c.ksize + x] = (int)(v < 0 ? -0.5 + v * (1 << PRECISION_BITS) : 0.5 + v * (1 << PRECISION_BITS));
        }
        c.bounds[2 * xx] = xmin; c.bounds[2 * xx + 1] = xmax;
    }
    return c;
}
uint8_t clip8(int v) { v >>= PRECISION_BITS; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
}  // namespace

void resize_lanczos3(const Image8 &src, uint32_t dw, uint32_t dh, Image8 &dst) {
    if (src.w == dw && src.h == dh) { dst = src; return; }
    Coeffs cx = coeffs(src.w, dw), cy = coeffs(src.h, dh);
    std::vector<uint8_t> tmp((size_t)dw * src.h * 4);
    for (uint32_t y = 0; y < src.h; ++y)
        for (uint32_t x = 0; x < dw; ++x) {
            int xmin = cx.bounds[2 * x], n = cx.bounds[2 * x + 1];
            const int *k = &cx.k[(size_t)x * cx.ksize];
            int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
            for (int i = 0; i < n; ++i) {
                const uint8_t *p = &src.rgba[((size_t)y * src.w + xmin + i) * 4];
                for (int c = 0; c < 4; ++c) acc[c] += p[c] * k[i];
            }
            for (int c = 0; c < 4; ++c) tmp[((size_t)y * dw + x) * 4 + c] = clip8(acc[c]);
        }
    dst.w = dw; dst.h = dh;
    dst.rgba.resize((size_t)dw * dh * 4);
    for (uint32_t y = 0; y < dh; ++y) {
        int ymin = cy.bounds[2 * y], n = cy.bounds[2 * y + 1];
        const int *k = &cy.k[(size_t)y * cy.ksize];
        for (uint32_t x = 0; x < dw; ++x) {
            int acc[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
            for (int i = 0; i < n; ++i) {
                const uint8_t *p = &tmp[((size_t)(ymin + i) * dw + x) * 4];
                for (int c = 0; c < 4; ++c) acc[c] += p[c] * k[i];
            }
            for (int c = 0; c < 4; ++c) dst.rgba[((size_t)y * dw + x) * 4 + c] = clip8(acc[c]);
        }
    }
}

/* atlas.rs:26-90.  Returns the uvst of texture i (x / W, y / W, w / W, h / H — sic) and the RGBA8 atlas. */
void pack_textures(const std::vector<Image8> &textures, uint32_t atlas_w, uint32_t atlas_h, std::vector<uint8_t> &atlas,
                   std::vector<std::array<float, 4>> &sts) {
    struct Rect { uint32_t x, y, w, h; };
    std::deque<Rect> queue{Rect{0, 0, atlas_w, atlas_h}};
    while (queue.size() <= textures.size()) {
        Rect n = queue.front();
        queue.pop_front();
        uint32_t hw = n.w / 2, hh = n.h / 2;
        queue.push_back(Rect{n.x, n.y, hw, hh});
        queue.push_back(Rect{n.x + hw, n.y, hw, hh});
        queue.push_back(Rect{n.x, n.y + hh, hw, hh});
        queue.push_back(Rect{n.x + hw, n.y + hh, hw, hh});
    }
    std::vector<Rect> leafs(queue.begin(), queue.end());
    std::stable_sort(leafs.begin(), leafs.end(), [](const Rect &a, const Rect &b) { return a.w > b.w; });
    leafs.resize(textures.size());
    atlas.assign((size_t)atlas_w * atlas_h * 4, 0);
    sts.clear();
    for (size_t i = 0; i < leafs.size(); ++i) {
        const Rect &leaf = leafs[i];
        Image8 resized;
        resize_lanczos3(textures[i], leaf.w, leaf.h, resized);
        for (uint32_t y = 0; y < leaf.h; ++y)                   /* flipv, then copy_from at (leaf.x, leaf.y) */
            memcpy(&atlas[(((size_t)leaf.y + y) * atlas_w + leaf.x) * 4], &resized.rgba[(size_t)(leaf.h - 1 - y) * leaf.w * 4], (size_t)leaf.w * 4);
        sts.push_back({(float)leaf.x / (float)atlas_w, (float)leaf.y / (float)atlas_w, (float)leaf.w / (float)atlas_w,
                       (float)leaf.h / (float)atlas_h});
    }
}

}  // namespace rpth
