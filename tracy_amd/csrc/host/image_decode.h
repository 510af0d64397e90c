// image_decode.h -- JPEG (baseline) and Radiance HDR decoders of the standalone scene loader.
// See image_decode.cpp for scope and parity status.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace imgdec {

struct Image8 {   // what stbi_load(file, &w, &h, &n, 4) returns (src/scene.cpp:39-43): RGBA8, top row first
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;
};
struct ImageF {   // what stbi_loadf(file, &w, &h, &n, 4) returns (src/scene.cpp:30-35): RGBA32F, top row first
    uint32_t width = 0, height = 0;
    std::vector<float> rgba;
};

bool decode_jpeg(const uint8_t* data, size_t size, Image8& img, std::string& err);
bool decode_hdr(const uint8_t* data, size_t size, ImageF& img, std::string& err);
bool read_file(const std::string& path, std::vector<uint8_t>& out);

}  // namespace imgdec
