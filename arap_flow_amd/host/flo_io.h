// flo_io.h -- Middlebury .flo files.  Writer: ARAP/deformation/src/main.cpp:53-75; reader with its checks:
// ARAP/warping/src/main.cpp:228-274.  "PIEH", int32 W, int32 H, rows of interleaved (u,v) float32.
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace arapio {

static const float FLO_TAG_FLOAT = 202021.25f;   // main.h:7
static const char FLO_TAG_STRING[] = "PIEH";     // main.h:8

inline bool write_flo(const std::string& path, const float* flow, int width, int height)
{
    FILE* stream = fopen(path.c_str(), "wb");
    if (!stream) { printf("WriteFlowFile(%s): could not open\n", path.c_str()); return false; }
    bool ok = fwrite(FLO_TAG_STRING, 1, 4, stream) == 4;
    ok = ok && fwrite(&width, sizeof(int), 1, stream) == 1 && fwrite(&height, sizeof(int), 1, stream) == 1;
    if (!ok) printf("WriteFlowFile(%s): problem writing header\n", path.c_str());
    const size_t n = 2 * (size_t)width;
    for (int y = 0; y < height && ok; ++y)
        if (fwrite(flow + n * y, sizeof(float), n, stream) != n) {
            printf("WriteFlowFile(%s): problem writing data", path.c_str());
            ok = false;
        }
    fclose(stream);
    return ok;
}

inline bool read_flo(const std::string& path, std::vector<float>& img, int& width, int& height)
{
    const char* dot = strrchr(path.c_str(), '.');
    if (!dot || strcmp(dot, ".flo") != 0) printf("ReadFlowFile (%s): extension .flo expected", path.c_str());
    FILE* stream = fopen(path.c_str(), "rb");
    if (!stream) { printf("ReadFlowFile: could not open %s\n", path.c_str()); return false; }
    float tag;
    if (fread(&tag, sizeof(float), 1, stream) != 1 || fread(&width, sizeof(int), 1, stream) != 1 ||
        fread(&height, sizeof(int), 1, stream) != 1) {
        printf("ReadFlowFile: problem reading file %s\n", path.c_str());
        fclose(stream);
        return false;
    }
    bool ok = true;
    if (tag != FLO_TAG_FLOAT) { printf("ReadFlowFile(%s): wrong tag (possibly due to big-endian machine?)\n", path.c_str()); ok = false; }
    if (width < 1 || width > 99999) { printf("ReadFlowFile(%s): illegal width %d\n", path.c_str(), width); ok = false; }
    if (height < 1 || height > 99999) { printf("ReadFlowFile(%s): illegal height %d\n", path.c_str(), height); ok = false; }
    if (!ok) { fclose(stream); return false; }
    img.resize((size_t)width * height * 2);
    const size_t n = 2 * (size_t)width;
    for (int y = 0; y < height; ++y)
        if (fread(&img[n * y], sizeof(float), n, stream) != n) {
            printf("ReadFlowFile(%s): file is too short\n", path.c_str());
            fclose(stream);
            return false;
        }
    if (fgetc(stream) != EOF) { printf("ReadFlowFile(%s): file is too long\n", path.c_str()); fclose(stream); return false; }
    fclose(stream);
    return true;
}

}  // namespace arapio
