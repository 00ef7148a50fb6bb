/*
 * bluenoise.cpp — per-pixel low-discrepancy offsets from bluenoise.png.
 *
 * Restates the seed setup of the reference's render dispatch
 * (src/trace.rs:5 BLUE_TEXTURE = decode().into_rgba8(); :150-157 and :246-253):
 *   pixel = BLUE_TEXTURE.get_pixel(x % w, y % h)[0] as f32 / 255.0
 *   rng[i] = (0, (pixel * 4294967295.0) as u32)
 * The PNG is 16-bit grayscale; the `image` crate (0.24.6) narrows u16 -> u8 as
 * (v + 128) / 257 in integer arithmetic.
 * Decoder: PNG chunks + zlib inflate + scanline unfilter, non-interlaced,
 * 8/16-bit gray / gray-alpha / RGB / RGBA (channel 0 is what the reference reads).
 */
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <dlfcn.h>

#include "host_internal.h"

namespace rpth {

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool load_blue_noise(const char *path, std::vector<uint8_t> &tile, uint32_t &w, uint32_t &h) {
    FILE *f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return false; }
    std::vector<uint8_t> data;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    data.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), f);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (got < 8 || memcmp(data.data(), sig, 8)) { set_error("not a PNG"); return false; }

    uint32_t bit_depth = 0, color_type = 0, interlace = 0;
    std::vector<uint8_t> idat;
    size_t at = 8;
    w = h = 0;
    while (at + 12 <= data.size()) {
        uint32_t len = be32(&data[at]);
        const uint8_t *type = &data[at + 4];
        if (at + 12 + len > data.size()) break;
        const uint8_t *body = &data[at + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(body); h = be32(body + 4);
            bit_depth = body[8]; color_type = body[9]; interlace = body[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        at += 12 + len;
    }
    int channels = color_type == 0 ? 1 : color_type == 2 ? 3 : color_type == 4 ? 2 : color_type == 6 ? 4 : 0;
    if (!w || !h || !channels || interlace || (bit_depth != 8 && bit_depth != 16)) {
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
    /* unfilter */
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
    tile.resize((size_t)w * h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint8_t *p = &img[stride * y + (size_t)x * bpp];
            if (bit_depth == 16) {
                uint32_t v16 = ((uint32_t)p[0] << 8) | p[1];
                tile[(size_t)y * w + x] = (uint8_t)((v16 + 128u) / 257u);
            } else {
                tile[(size_t)y * w + x] = p[0];
            }
        }
    return true;
}

std::string default_fixture_path(const char *name) {
    /* <repo>/rust-path-tracer_amd/lib/librpt_host.so -> <repo>/fixtures/<name> */
    Dl_info info;
    std::string dir = ".";
    if (dladdr((void *)&default_fixture_path, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t s = p.find_last_of('/');
        dir = (s == std::string::npos) ? "." : p.substr(0, s);
    }
    return dir + "/../../fixtures/" + name;
}

}  // namespace rpth
