// half_round.h -- IEEE binary16 conversions with directed rounding, shared by the host BVH builder
// (bvh_builder.cpp, g++ or hipcc host pass) and the device builder (lbvh_build.hip): node bounds are
// stored as half floats rounded OUTWARD, so a quantised box always contains the fp32 box.
#pragma once
#include <cstdint>
#include <cstring>

#ifdef __HIPCC__
#define TRH_HD __host__ __device__ inline
#else
#define TRH_HD inline
#endif

namespace bvh {

TRH_HD float bits_to_float(uint32_t b) { float v; memcpy(&v, &b, 4); return v; }
TRH_HD uint32_t float_to_bits(float v) { uint32_t b; memcpy(&b, &v, 4); return b; }

TRH_HD float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t em = h & 0x7fffu;
    if (em >= 0x7c00u) return bits_to_float(sign | 0x7f800000u | ((em & 0x3ffu) << 13));   // inf / nan
    // exponent still biased by 15: scale by 2^(127-15); exact, and turns half subnormals into the right floats
    const float v = bits_to_float(em << 13) * 0x1p112f;
    return sign ? -v : v;
}
TRH_HD uint16_t half_next_down(uint16_t h) {   // next representable value toward -inf (h finite or +inf)
    if (h == 0x0000u) return 0x8001u;
    return (h & 0x8000u) ? (uint16_t)(h + 1u) : (uint16_t)(h - 1u);
}
TRH_HD uint16_t half_next_up(uint16_t h) {     // toward +inf (h finite or -inf)
    if (h == 0x8000u) return 0x0001u;
    return (h & 0x8000u) ? (uint16_t)(h - 1u) : (uint16_t)(h + 1u);
}
TRH_HD uint16_t half_truncate(float f) {       // |result| <= |f| (round toward zero); |f| < 65520 assumed
    const uint32_t bits = float_to_bits(f);
    const uint16_t sign = (uint16_t)((bits >> 16) & 0x8000u);
    const uint32_t a = bits & 0x7fffffffu;
    if (a >= 0x38800000u) return sign | (uint16_t)((a >> 13) - (112u << 10));   // >= 2^-14: normal half
    return sign | (uint16_t)(uint32_t)(bits_to_float(a) * 0x1p24f);             // subnormal half: units of 2^-24
}
TRH_HD uint16_t half_floor(double v) {         // largest half <= v
    if (v >= 65504.0) return 0x7BFFu;
    if (v < -65504.0) return 0xFC00u;
    uint16_t h = half_truncate((float)v);
    while ((double)half_to_float(h) > v) h = half_next_down(h);
    for (;;) {                          // tighten: (float)v or the truncation may have stepped too far
        const uint16_t up = half_next_up(h);
        if (up == 0x7C00u || (double)half_to_float(up) > v) break;
        h = up;
    }
    return h;
}
TRH_HD uint16_t half_ceil(double v) {          // smallest half >= v
    if (v <= -65504.0) return 0xFBFFu;
    if (v > 65504.0) return 0x7C00u;
    uint16_t h = half_truncate((float)v);
    while ((double)half_to_float(h) < v) h = half_next_up(h);
    for (;;) {
        const uint16_t dn = half_next_down(h);
        if (dn == 0xFC00u || (double)half_to_float(dn) < v) break;
        h = dn;
    }
    return h;
}

}  // namespace bvh
