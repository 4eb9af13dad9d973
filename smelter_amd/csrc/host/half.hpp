// half.hpp — IEEE binary16 <-> binary32 on the host (the role vImageConvert_Planar16FtoPlanarF /
// PlanarFtoPlanar16F play in /root/reference/Sources/Smelter/Float16.swift:17-77): exact widening,
// round-to-nearest-even narrowing, overflow -> inf, gradual underflow.
#pragma once
#include <stdint.h>
#include <string.h>

namespace smelter {

inline float half_bits_to_float(uint16_t h) {
    // scale trick: place the 15 magnitude bits at the top of an fp32 mantissa/exponent and rescale by 2^112
    uint32_t mag = (uint32_t)(h & 0x7FFFu) << 13;
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    float f;
    if ((h & 0x7C00u) == 0x7C00u) {            // inf / nan
        uint32_t b = sign | 0x7F800000u | mag; memcpy(&f, &b, 4); return f;
    }
    memcpy(&f, &mag, 4);
    const uint32_t scale_bits = (uint32_t)(127 + 112) << 23; float scale; memcpy(&scale, &scale_bits, 4);
    f *= scale;                                 // exact: also normalises half subnormals
    uint32_t b; memcpy(&b, &f, 4); b |= sign; memcpy(&f, &b, 4);
    return f;
}

inline uint16_t float_to_half_bits(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x > 0x7F800000u) return (uint16_t)(sign | 0x7E00u | ((x >> 13) & 0x1FFu));   // nan (quiet)
    if (x >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);                          // >= 65536 -> inf (65520..65535.99 handled below)
    if (x < 0x38800000u) {
        // result is a half subnormal (or zero): add 0.5f so the fp32 adder performs the RNE shift
        float a; memcpy(&a, &x, 4);
        a += 0.5f;
        uint32_t b; memcpy(&b, &a, 4);
        return (uint16_t)(sign | (b - 0x3F000000u));
    }
    // normal: rebias exponent, round mantissa to nearest even
    uint32_t odd = (x >> 13) & 1u;
    x += 0xC8000FFFu + odd;                      // ((15-127)<<23) + 0xFFF + odd
    return (uint16_t)(sign | (x >> 13));         // a mantissa carry rolls into the exponent; 65520+ becomes 0x7C00
}

}  // namespace smelter
