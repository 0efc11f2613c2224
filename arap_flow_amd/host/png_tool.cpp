// png_tool -- test utility for png_io: `png_tool in.png out_rgb.png out_mask1.png` decodes a PNG to RGB8,
// writes it back as RGB8 and writes (red channel != 0) as a 1-bit mask.  Used by tests/test_host_cpp.py.
#include <cstdio>
#include <string>
#include <vector>

#include "png_io.h"

int main(int argc, const char* argv[])
{
    if (argc != 4) { printf("usage: png_tool in.png out_rgb.png out_mask1.png\n"); return 2; }
    arapio::Image im;
    std::string err;
    if (!arapio::read_png_rgb(argv[1], im, err)) { printf("%s\n", err.c_str()); return 1; }
    std::vector<unsigned char> m((size_t)im.w * im.h);
    for (size_t i = 0; i < m.size(); ++i) m[i] = im.rgb[3 * i];
    if (!arapio::write_png_rgb(argv[2], im.w, im.h, im.rgb.data(), err) ||
        !arapio::write_png_mask1(argv[3], im.w, im.h, m.data(), err)) { printf("%s\n", err.c_str()); return 1; }
    printf("%d %d\n", im.w, im.h);
    return 0;
}
