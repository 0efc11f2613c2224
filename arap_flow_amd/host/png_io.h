// png_io.h -- minimal PNG reader/writer on zlib for the C++ host programs (arap_deform, warp_image).
// The reference reads/writes PNGs with LodePNG from the vendored mLib (ARAP/external/mLib, CC BY-NC-SA: not
// copied); this is an independent implementation of the subset the pipeline needs:
//   read : non-interlaced PNG, colour types 0/2/3/4/6, bit depths 1/2/4/8/16 -> RGB8 (alpha dropped, like the
//          conversion to ColorImageR8G8B8 at ARAP/deformation/src/main.cpp:116-138)
//   write: RGB8, or 1-bit greyscale for the 0/255 warped mask (LodePNG's automatic choice)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace arapio {

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgb;      // [h][w][3]
};

// false + message in `err` on failure
bool read_png_rgb(const std::string& path, Image& out, std::string& err);
bool write_png_rgb(const std::string& path, int w, int h, const uint8_t* rgb, std::string& err);
bool write_png_mask1(const std::string& path, int w, int h, const uint8_t* mask /* 0 / non-zero */, std::string& err);

}  // namespace arapio
