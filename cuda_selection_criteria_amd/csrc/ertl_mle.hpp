// ertl_mle.hpp -- HyperLogLog cardinality estimator used by the selection path (host + gfx950 device).
//
// Reference behaviour reproduced (paths relative to the reference repository):
//   sketch/include/sketch/hll.h:629-688   detail::ertl_ml_estimate(counts, p, q, relerr)   (Ertl MLE, secant)
//   sketch/include/sketch/hll.h:255-258   calculate_estimate(... ERTL_MLE ...) -> ertl_ml_estimate(c, p, 64-p, 1e-2)
// The secant iteration stops at a coarse relative step (1e-2/sqrt(m)), so the result depends on the
// exact sequence of IEEE-754 double operations.  Two sequences exist in the wild for the SAME reference
// source: g++ -O3 -march=<FMA-capable host> (the reference Makefile:32 build; GCC's default
// -ffp-contract=fast fuses every product whose only use is an add/sub of the same basic block) and
// the unfused one (-ffp-contract=off or a pre-FMA host).  Both are provided, selected by the template
// flag; this translation unit must be compiled with -ffp-contract=off so that nothing else is fused.
//
// Everything here is plain IEEE double arithmetic: +,-,*,/ (correctly rounded on gfx950: the f64
// division expands to v_div_scale/v_rcp/v_fma/v_div_fmas/v_div_fixup), fma (v_fma_f64), exact scaling
// by powers of two, and exponent extraction.  sqrt(m) is taken on the host and handed in as
// relerr_scaled so that no device sqrt is involved.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SELHIP_HD __host__ __device__ __forceinline__
#else
#define SELHIP_HD inline
#endif

namespace selhip {

SELHIP_HD double bits2d(uint64_t u) { union { uint64_t u; double d; } x; x.u = u; return x.d; }
SELHIP_HD uint64_t d2bits(double d) { union { uint64_t u; double d; } x; x.d = d; return x.u; }

// x * 2^e for finite normal x >= 0 results that stay normal (all uses below): exact, like ldexp().
// Falls back to repeated scaling at the extremes so that subnormal results are still rounded once.
SELHIP_HD double scale2(double x, int e) {
    if (x == 0.0) return x;
    // split the scaling so each factor is a normal power of two
    while (e > 1000) { x *= bits2d(0x7E70000000000000ull); e -= 1000; }            // 2^1000
    while (e < -1000) { x *= bits2d(0x0170000000000000ull); e += 1000; }           // 2^-1000
    return x * bits2d((uint64_t)(1023 + e) << 52);
}

// frexp() exponent of a positive finite double: x = f * 2^exp, f in [0.5, 1)
SELHIP_HD int frexp_exp(double x) {
    uint64_t b = d2bits(x);
    int be = (int)((b >> 52) & 0x7FF);
    if (be == 0) {                 // zero or subnormal
        if ((b << 1) == 0) return 0;
        uint64_t mant = b & 0x000FFFFFFFFFFFFFull;
        int lz = __builtin_clzll(mant) - 11;       // leading zeros inside the 52-bit field
        return -1022 - lz;
    }
    return be - 1022;
}

// log1p: restatement of the fdlibm algorithm (s_log1p.c) that glibc's generic dbl-64 __log1p follows; only reached in the
// start-point branch hll.h:659 (gprev > 1.5*a), i.e. for sketches whose registers are all near saturation.
// The algorithm, its constants and the structure of the code below come from fdlibm, whose notice is preserved here as its licence asks:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
//
SELHIP_HD double log1p_fdlibm(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 two54 = 1.80143985094819840000e+16,
                 Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01,
                 Lp3 = 2.857142874366239149e-01, Lp4 = 2.222219843214978396e-01,
                 Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    double hfsq, f = 0, c = 0, s, z, R, u;
    int32_t k, hx, hu = 0, ax;
    hx = (int32_t)(d2bits(x) >> 32);
    ax = hx & 0x7fffffff;
    k = 1;
    if (hx < 0x3FDA827A) {                         /* x < 0.41422 */
        if (ax >= 0x3ff00000) {                    /* x <= -1.0 */
            if (x == -1.0) return -two54 / 0.0;
            return (x - x) / (x - x);
        }
        if (ax < 0x3e200000) {                     /* |x| < 2**-29 */
            if (two54 + x > 0.0 && ax < 0x3c900000) return x;   /* |x| < 2**-54 */
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= ((int32_t)0xbfd2bec3)) { k = 0; f = x; hu = 1; }   /* -0.2929<x<0.41422 */
    } else if (hx >= 0x7ff00000) return x + x;
    if (k != 0) {
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = (int32_t)(d2bits(u) >> 32);
            k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);      /* correction term */
            c /= u;
        } else {
            u = x;
            hu = (int32_t)(d2bits(u) >> 32);
            k = (hu >> 20) - 1023;
            c = 0;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = bits2d((d2bits(u) & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32));   /* normalize u */
        } else {
            k += 1;
            u = bits2d((d2bits(u) & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32));   /* normalize u/2 */
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    hfsq = 0.5 * f * f;
    if (hu == 0) {                                 /* |f| < 2**-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    s = f / (2.0 + f);
    z = s * s;
    {
        double R1 = z * Lp1, z2 = z * z;
        double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
        double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
        double R4 = Lp6 + z * Lp7;
        R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    }
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

template <bool FMA>
SELHIP_HD double muladd(double a, double b, double c) {
    if (FMA) return __builtin_fma(a, b, c);
    return a * b + c;       // stays two roundings: the TU is built with -ffp-contract=off
}

// hll.h:629-688.  c[0..q+1] are the register-value counts (Σ = 2^p); relerr_scaled = relerr / sqrt(2^p)
// computed by the caller on the host (hll.h:662).  Returns +inf when every register is saturated (:642).
// `c` is anything indexable with operator[](int) that yields an unsigned count (pointer, LDS view, ...).
template <bool FMA, typename Counts>
SELHIP_HD double ertl_ml_estimate(const Counts& c, unsigned p, unsigned q, double relerr_scaled) {
    const uint64_t m = 1ull << p;
    if ((uint64_t)c[q + 1] == m) return bits2d(0x7FF0000000000000ull);        // :642 +inf

    int kMin, kMax;
    for (kMin = 0; c[kMin] == 0; ++kMin) {}                                    // :645
    const int kMinPrime = kMin > 1 ? kMin : 1;                                 // :646
    for (kMax = (int)q + 1; kMax && c[kMax] == 0; --kMax) {}                   // :647
    const int kMaxPrime = (int)q < kMax ? (int)q : kMax;                       // :648
    double z = 0.;
    for (int k = kMaxPrime; k >= kMinPrime; --k) z = 0.5 * z + (double)(uint32_t)c[k];   // :650
    z = scale2(z, -kMinPrime);                                                 // :651
    unsigned cPrime = (unsigned)c[q + 1];                                      // :652
    if (q) cPrime += (unsigned)c[kMaxPrime];                                   // :653
    const double a = z + (double)(uint32_t)c[0];                               // :656
    const int mPrime = (int)(m - (uint64_t)c[0]);                              // :657
    double gprev = z + scale2((double)(uint32_t)c[q + 1], -(int)q);            // :658
    double x = gprev <= 1.5 * a ? (double)mPrime / (0.5 * gprev + a)           // :659
                                : ((double)mPrime / gprev) * log1p_fdlibm(gprev / a);
    gprev = 0;
    double deltaX = x;
    while (deltaX > x * relerr_scaled) {                                       // :663
        const int kappaMinus1 = frexp_exp(x);                                  // :665
        const int sh = kMaxPrime + 1 > kappaMinus1 + 2 ? kMaxPrime + 1 : kappaMinus1 + 2;
        double xPrime = scale2(x, -sh);                                        // :666
        const double xPrime2 = xPrime * xPrime;
        // :668  h = xPrime - xPrime2/3 + (xPrime2*xPrime2)*(1./45. - xPrime2/472.5)
        double h = muladd<FMA>(xPrime2 * xPrime2, 1. / 45. - xPrime2 / 472.5, xPrime - xPrime2 / 3);
        for (int k = kappaMinus1; k >= kMaxPrime; --k) {                       // :669
            const double hPrime = 1. - h;
            h = muladd<FMA>(h, hPrime, xPrime) / (xPrime + hPrime);            // :671
            xPrime += xPrime;
        }
        double g = (double)cPrime * h;                                         // :674
        for (int k = kMaxPrime - 1; k >= kMinPrime; --k) {                     // :675
            const double hPrime = 1. - h;
            h = muladd<FMA>(h, hPrime, xPrime) / (xPrime + hPrime);            // :677
            xPrime += xPrime;
            g = muladd<FMA>((double)(uint32_t)c[k], h, g);                     // :679
        }
        g = muladd<FMA>(x, a, g);                                              // :681
        if (gprev < g && g <= (double)mPrime) deltaX *= (g - (double)mPrime) / (gprev - g);   // :682
        else                                  deltaX = 0;
        x += deltaX;
        gprev = g;
    }
    return x * (double)m;                                                      // :687
}

// (size_t)card as the reference's `size_t e1 = card_name[i].second` (src/selection.cpp:275,280).
// Defined here for finite 0 <= card < 2^63 (callers reject anything else at upload time).
SELHIP_HD uint64_t trunc_card(double card) { return (uint64_t)(long long)card; }

}  // namespace selhip
