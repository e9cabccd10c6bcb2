// Host-side generation of the integer tables OpenCV's 8-bit colour conversions use
// (imgproc/src/color_lab.cpp initLabTabs / RGB2Lab_b ctor, color_hsv.simd.hpp RGB2HSV_b).
// OpenCV evaluates them with softfloat: binary32 for x = i/255 and the cube root result,
// binary64 inside pow() and the cube-root polynomial, round-half-even everywhere.  Ordinary
// IEEE float/double reproduce that as long as nothing is contracted into an FMA, hence
// -ffp-contract=off for this file (see build.py).
#include <cmath>
#include <cstdint>
#include <cstring>

static int round_even(double v) { return (int)std::nearbyint(v); }
static int round_even(float v) { return (int)std::nearbyintf(v); }

// cv::cubeRoot for binary32: cube root of the exponent by integer division, mantissa by a
// quartic rational minimax (evaluated in binary64, rounded once to binary32).
static float cube_root32(float x)
{
    uint32_t bits;
    std::memcpy(&bits, &x, 4);
    if ((bits << 1) == 0) return 0.0f;
    const uint32_t sign = bits & 0x80000000u;
    const uint32_t mag = bits & 0x7fffffffu;
    int e = (int)(mag >> 23) - 127;
    int rem = e % 3;
    if (rem >= 0) rem -= 3;              // rem in {-3,-2,-1}
    const int e3 = (e - rem) / 3;
    uint32_t mbits = (mag & 0x007fffffu) | ((uint32_t)(rem + 127) << 23);
    float mant;
    std::memcpy(&mant, &mbits, 4);       // in [0.125, 1)
    const double t = mant;
    const double p = ((((45.2548339756803022511987494 * t + 192.2798368355061050458134625) * t +
                        119.1654824285581628956914143) * t + 13.43250139086239872172837314) * t +
                      0.1636161226585754240958355063);
    const double q = ((((14.80884093219134573786480845 * t + 151.9714051044435648658557668) * t +
                        168.5254414101568283957668343) * t + 33.9905941350215598754191872) * t + 1.0);
    const float r = (float)(p / q);
    uint32_t rbits;
    std::memcpy(&rbits, &r, 4);
    rbits = (rbits + ((uint32_t)e3 << 23)) | sign;
    float out;
    std::memcpy(&out, &rbits, 4);
    return out;
}

void vp_host_tables(uint16_t* gamma, uint16_t* cbrt_tab, int32_t* sdiv, int32_t* hdiv180, int32_t* labC)
{
    if (gamma)
        for (int i = 0; i < 256; i++) {
            const float xf = (float)i / 255.0f;
            const double x = xf;
            const double lin = x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
            gamma[i] = (uint16_t)round_even(2040.0f * (float)lin);
        }
    if (cbrt_tab) {
        const float thr = 216.0f / 24389.0f, bias = 16.0f / 116.0f, slope = 841.0f / 108.0f;
        for (int i = 0; i < 3072; i++) {
            const float x = (float)i / 2040.0f;
            const float f = x < thr ? std::fmaf(x, slope, bias) : cube_root32(x);
            cbrt_tab[i] = (uint16_t)round_even(32768.0f * f);
        }
    }
    if (sdiv || hdiv180)
        for (int i = 0; i < 256; i++) {
            if (sdiv) sdiv[i] = i ? round_even((255 << 12) / (double)i) : 0;
            if (hdiv180) hdiv180[i] = i ? round_even((180 << 12) / (6.0 * i)) : 0;
        }
    if (labC) {
        const double m[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
        const double white[3] = {0.950456, 1.0, 1.088754};
        for (int k = 0; k < 9; k++) labC[k] = round_even(4096.0 * m[k] / white[k / 3]);
    }
}
