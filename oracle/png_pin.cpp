// oracle/png_pin.cpp -- TEST INFRASTRUCTURE ONLY (golden-CRC pin of the oracle).
//
// Turns an iteration buffer into the exact PNG file bytes the reference's headless render writes, and
// CRC-64s them, so the oracle chain (view -> orbit -> LA -> CPU render) can be checked against the 12 golden
// CRCs in FractalSharkTest/TestRenderGoldens.cpp:84-97.
//
// Restated here (this file's own code):
//   default palette       FractalPalette::InitializeAllPalettes / PalTransition  FractalPalette.cpp:27-46,147-174
//   iteration -> colour   PngParallelSave::Run                                   PngParallelSave.cpp:137-189
//   CRC-64/ECMA-182       FractalSharkTest/Crc64.h:14-23,37-55
// NOT restated: the PNG encoder.  The byte stream depends on lodepng's deflate, so this file is compiled
// against the reference's own FractalSharkLib/WPngImage/{WPngImage.cc,lodepng.cpp} where they lie under
// /root/reference (oracle/Makefile target `_ref`); output goes to oracle/_ref/libpngpin.so only.
#include <cstdint>
#include <string>
#include <vector>

#include "WPngImage.hh"

namespace {

struct Color16 {
    uint16_t r, g, b, a;
};

void PalTransition(std::vector<Color16> &pal, int length, int r, int g, int b)
{
    int curR = 0, curG = 0, curB = 0;
    if (!pal.empty()) {
        curR = pal.back().r;
        curG = pal.back().g;
        curB = pal.back().b;
    }
    const double deltaR = (double)(r - curR) / length;
    const double deltaG = (double)(g - curG) / length;
    const double deltaB = (double)(b - curB) / length;
    for (int i = 0; i < length; i++) {
        Color16 c;
        c.r = static_cast<uint16_t>(curR + deltaR * (i + 1));
        c.g = static_cast<uint16_t>(curG + deltaG * (i + 1));
        c.b = static_cast<uint16_t>(curB + deltaB * (i + 1));
        c.a = 0;
        pal.push_back(c);
    }
}

std::vector<Color16> DefaultPalette(int depth)
{
    std::vector<Color16> pal;
    const int depth_total = 1 << depth;
    const int max_val = 65535;
    PalTransition(pal, depth_total, max_val, 0, 0);
    PalTransition(pal, depth_total, max_val, max_val, 0);
    PalTransition(pal, depth_total, 0, max_val, 0);
    PalTransition(pal, depth_total, 0, max_val, max_val);
    PalTransition(pal, depth_total, 0, 0, max_val);
    PalTransition(pal, depth_total, max_val, 0, max_val);
    PalTransition(pal, depth_total, 0, 0, 0);
    return pal;
}

uint64_t Crc64(const uint8_t *p, size_t len)
{
    static uint64_t table[256];
    static bool init = false;
    if (!init) {
        const uint64_t kPoly = 0x42F0E1EBA9EA3693ULL;
        for (uint32_t i = 0; i < 256; ++i) {
            uint64_t c = (uint64_t)i << 56;
            for (int k = 0; k < 8; ++k)
                c = (c & (1ULL << 63)) ? (c << 1) ^ kPoly : (c << 1);
            table[i] = c;
        }
        init = true;
    }
    uint64_t crc = 0;
    for (size_t i = 0; i < len; ++i)
        crc = table[(uint8_t)(crc >> 56) ^ p[i]] ^ (crc << 8);
    return crc;
}

} // namespace

extern "C" {

// Default palette of the given bit depth as RGBA16 entries; returns the entry count (7 << depth).
uint32_t pin_default_palette(int depth, uint16_t *out_rgba, uint32_t capacity)
{
    const auto pal = DefaultPalette(depth);
    if (out_rgba) {
        for (size_t i = 0; i < pal.size() && i < capacity; i++) {
            out_rgba[4 * i + 0] = pal[i].r;
            out_rgba[4 * i + 1] = pal[i].g;
            out_rgba[4 * i + 2] = pal[i].b;
            out_rgba[4 * i + 3] = pal[i].a;
        }
    }
    return (uint32_t)pal.size();
}

// iters: (height*aa) rows of `stride` uint32, row-major (ItersMemoryContainer layout).
// Palette Default, depth index 2 (8 bits), aux depth 0, rotation 0 (Fractal.cpp:536-538).
// Returns the CRC-64 of the PNG file bytes; optionally writes the file.
uint64_t pin_png_crc64(const uint32_t *iters, uint32_t stride, uint32_t width, uint32_t height, uint32_t aa,
                       uint64_t num_iterations, uint64_t max_possible_iters, const char *save_path)
{
    const auto pal = DefaultPalette(8);
    const uint32_t palIters = (uint32_t)pal.size();
    const uint64_t paletteRotate = 0;
    const int auxDepth = 0;
    WPngImage image((int)width, (int)height, WPngImage::Pixel16(0, 0, 0));
    for (size_t oy = 0; oy < height; oy++) {
        for (size_t ox = 0; ox < width; ox++) {
            double acc_r = 0, acc_g = 0, acc_b = 0;
            for (size_t ix = ox * aa; ix < (ox + 1) * aa; ix++) {
                for (size_t iy = oy * aa; iy < (oy + 1) * aa; iy++) {
                    size_t numIters = iters[iy * stride + ix];
                    if (numIters < num_iterations) {
                        numIters += paletteRotate;
                        if (numIters >= max_possible_iters)
                            numIters = max_possible_iters - 1;
                        const auto shifted = (numIters >> auxDepth);
                        const auto palIndex = shifted % palIters;
                        acc_r += pal[palIndex].r;
                        acc_g += pal[palIndex].g;
                        acc_b += pal[palIndex].b;
                    }
                }
            }
            acc_r /= aa * aa;
            acc_g /= aa * aa;
            acc_b /= aa * aa;
            image.set((int)ox, (int)oy, WPngImage::Pixel16((uint16_t)acc_r, (uint16_t)acc_g, (uint16_t)acc_b));
        }
    }
    std::vector<unsigned char> bytes;
    image.saveImageToRAM(bytes, WPngImage::kPngFileFormat_RGBA16);
    if (save_path && save_path[0])
        image.saveImage(std::string(save_path), WPngImage::kPngFileFormat_RGBA16);
    return Crc64(bytes.data(), bytes.size());
}

// The same PNG bytes from an RGBA16 colour buffer that is already antialiased and palette-mapped (the Color16 buffer
// GPURenderer::RenderCurrent hands back, row stride = `stride` pixels, GPU_Render.cu:1759-1805): pins the colour half of
// the path (antialiasing_kernel + palette lookup) against the same golden CRCs.  Alpha is taken as Pixel16's default
// (opaque), like the pixels written above.
uint64_t pin_png_crc64_rgba16(const uint16_t *rgba, uint32_t stride, uint32_t width, uint32_t height)
{
    WPngImage image((int)width, (int)height, WPngImage::Pixel16(0, 0, 0));
    for (size_t oy = 0; oy < height; oy++)
        for (size_t ox = 0; ox < width; ox++) {
            const uint16_t *px = rgba + 4 * (oy * stride + ox);
            image.set((int)ox, (int)oy, WPngImage::Pixel16(px[0], px[1], px[2]));
        }
    std::vector<unsigned char> bytes;
    image.saveImageToRAM(bytes, WPngImage::kPngFileFormat_RGBA16);
    return Crc64(bytes.data(), bytes.size());
}
}
