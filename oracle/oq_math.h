/* oq_math.h -- the oracle's own float tanh(x/2) and log((1+c)/(1-c)).
 *
 * TEST INFRASTRUCTURE ONLY.  ldpc's product-sum update (bp.hpp) evaluates both in double through libm; the device computes in
 * float, where a libm result is not reproducible, so the HIP kernel (quits_amd/csrc/qd_math.h) builds them from IEEE basic
 * operations in a fixed order.  This file restates that arithmetic independently -- written from the description below, not
 * included from the product -- so that the checker and the product share no code; tests/test_oracle.py holds both to libm in
 * double (a few ulp), and the GPU parity tests hold the kernel to this file bit for bit.  Compile with -ffp-contract=off.
 *
 *   tanh(x/2), a = |x|:   a < 1/2  : h = a/2, odd Taylor polynomial  h (1 - h^2/3 + 2h^4/15 - 17h^6/315 + 62h^8/2835 - 1382h^10/155925), Horner in h^2
 *                         a >= 1/2 : e = exp(min(a, 40)) by Cody-Waite (k = floor(a log2(e) + 1/2), r = a - k ln2_hi - k ln2_lo,
 *                                    degree-6 Taylor of exp(r), scaled by 2^k), t = 1 - 2/(e + 1), capped at 1 - 2^-24; sign restored
 *   log((1+c)/(1-c)):     |c| <= 11/64 : 2 atanh(c) = c (2 + 2c^2/3 + 2c^4/5 + ... + 2c^12/13), Horner in c^2
 *                         otherwise    : q = (1+c)/(1-c) = 2^e m, m in (sqrt(1/2), sqrt 2], s = (m-1)/(m+1), e ln2 + 2 atanh(s) with the same series
 */
#ifndef OQ_MATH_H
#define OQ_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t oq_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float oq_float(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

static const float OQ_LN2_HI = 0.693145752f, OQ_LN2_LO = 1.42860677e-06f;

static inline float oq_tanh_half_f32(float x)
{
    const uint32_t xb = oq_bits(x);
    const float a = oq_float(xb & 0x7FFFFFFFu);
    float t;
    if (a >= 0.5f) {
        const float ac = a > 40.0f ? 40.0f : a;
        const float k = floorf(ac * 1.4426950f + 0.5f);
        float r = ac - k * OQ_LN2_HI;
        r = r - k * OQ_LN2_LO;
        static const float c[7] = {0.0013888889f, 0.008333334f, 0.041666668f, 0.16666667f, 0.5f, 1.0f, 1.0f};   /* 1/6!, 1/5!, ..., 1/1!, 1 */
        float p = c[0];
        for (int i = 1; i < 7; i++) p = p * r + c[i];
        const float e = p * oq_float((uint32_t)((int)k + 127) << 23);
        t = 1.0f - 2.0f / (e + 1.0f);
        if (t > 0.99999994f) t = 0.99999994f;
    } else {                                   /* also NaN, like the product's !(a >= 1/2) */
        const float h = a * 0.5f, h2 = h * h;
        static const float c[6] = {-0.0088632355f, 0.021869488f, -0.053968254f, 0.13333334f, -0.33333334f, 1.0f};
        float p = c[0];
        for (int i = 1; i < 6; i++) p = p * h2 + c[i];
        t = h * p;
    }
    return oq_float(oq_bits(t) | (xb & 0x80000000u));
}

static inline float oq_atanh2_series(float s)          /* 2 atanh(s) / s, |s| <= 0.1716 */
{
    static const float c[7] = {0.15384616f, 0.18181819f, 0.22222222f, 0.2857143f, 0.4f, 0.6666667f, 2.0f};
    const float s2 = s * s;
    float p = c[0];
    for (int i = 1; i < 7; i++) p = p * s2 + c[i];
    return p;
}

static inline float oq_log_ratio_f32(float c)
{
    const float ac = oq_float(oq_bits(c) & 0x7FFFFFFFu);
    if (!(ac > 0.171875f)) return c * oq_atanh2_series(c);
    const float q = (1.0f + c) / (1.0f - c);
    const uint32_t qb = oq_bits(q);
    int e = (int)((qb >> 23) & 255u) - 127;
    float m = oq_float((qb & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.4142135f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float ef = (float)e;
    return ef * OQ_LN2_HI + (s * oq_atanh2_series(s) + ef * OQ_LN2_LO);
}
#endif
