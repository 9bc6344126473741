/* oq_math.h -- the oracle's own float functions for the product-sum check update.
 *
 * TEST INFRASTRUCTURE ONLY.  ldpc's product-sum update (bp.hpp) evaluates tanh(b2c / 2), a product over a row and
 * log((1 + P) / (1 - P)) in double through libm.  The device computes in float, where tanh(x / 2) is 1 beyond |x| = 17.3, so it
 * keeps u = e^-|x| = (1 - t) / (1 + t) instead of t = tanh(|x| / 2) (quits_amd/csrc/qd_math.h explains why): a product of tanh
 * values t1 t2 is u12 = (u1 + u2) / (1 + u1 u2), and log((1 + P) / (1 - P)) = -log(u).  A libm result is not reproducible on
 * the GPU, so the kernel builds the three functions from IEEE basic operations in a fixed order.  This file restates that
 * arithmetic independently -- written from the description below, not included from the product -- so that the checker and
 * the product share no code; tests/test_oracle.py holds both to libm in double, and the GPU parity tests hold the kernel to
 * this file bit for bit.  Compile with -ffp-contract=off.
 *
 *   Every "x*y + z" below that is written fma(x, y, z) is ONE rounding (C99 fmaf); nothing else is fused.
 *
 *   E(x) = +-e^-a, a = min(|x|, 87), sign bit of x:   k = floor(fma(a, log2(e), 1/2)),  z = fma(k, ln2_lo, fma(k, ln2_hi, -a))  (= -r),
 *                         M = z * P(z), P = Horner by fma of 1 + z/2 + z^2/6 + ... + z^6/5040     (M = expm1(z)),   E = fma(M, 2^-k, 2^-k)
 *   C(a, b) = min(1, (a + b) / fma(a, b, 1))          for a, b in [0, 1]
 *   L(u) = -log(max(u, 2^-126)):  split u = 2^k m with m in [sqrt(1/2), sqrt 2) (adding 0x3F800000 - 0x3F3504F3 to the bit pattern
 *                         carries into the exponent field exactly when the mantissa reaches sqrt 2),  s = (m - 1) / (m + 1),
 *                         L = 0 - fma(k, ln2_hi, fma(s, Q(s^2), k * ln2_lo)),   Q = Horner by fma of 2 + 2y/3 + 2y^2/5 + ... + 2y^6/13
 */
#ifndef OQ_MATH_H
#define OQ_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t oq_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float oq_float(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

static const float OQ_LN2_HI = 0.693145752f, OQ_LN2_LO = 1.42860677e-06f;

static inline float oq_horner_fma(const float *c, int terms, float z)
{
    float p = c[0];
    for (int i = 1; i < terms; i++) p = fmaf(p, z, c[i]);
    return p;
}

static inline float oq_exp_neg_f32(float x)
{
    static const float expm1_over_z[7] = {1.9841270e-04f, 0.0013888889f, 0.008333334f, 0.041666668f, 0.16666667f, 0.5f, 1.0f};   /* 1/7!, ..., 1/2!, 1 */
    const uint32_t xb = oq_bits(x);
    float a = oq_float(xb & 0x7FFFFFFFu);
    if (a > 87.0f) a = 87.0f;
    const float k = floorf(fmaf(a, 1.4426950f, 0.5f));
    const float z = fmaf(k, OQ_LN2_LO, fmaf(k, OQ_LN2_HI, -a));
    const float M = z * oq_horner_fma(expm1_over_z, 7, z);
    const float scale = oq_float((uint32_t)(127 - (int)k) << 23);
    return oq_float(oq_bits(fmaf(M, scale, scale)) | (xb & 0x80000000u));
}

static inline float oq_ucomb_f32(float a, float b)
{
    const float q = (a + b) / fmaf(a, b, 1.0f);
    return q > 1.0f ? 1.0f : q;
}

static inline float oq_neg_log_f32(float u)
{
    static const float atanh2_over_s[7] = {0.15384616f, 0.18181819f, 0.22222222f, 0.2857143f, 0.4f, 0.6666667f, 2.0f};   /* 2/13, ..., 2/3, 2 */
    if (u < 1.17549435e-38f) u = 1.17549435e-38f;
    const uint32_t shifted = oq_bits(u) + (0x3F800000u - 0x3F3504F3u);
    const float k = (float)((int)(shifted >> 23) - 127);
    const float m = oq_float((shifted & 0x007FFFFFu) + 0x3F3504F3u);
    const float s = (m - 1.0f) / (m + 1.0f);
    return 0.0f - fmaf(k, OQ_LN2_HI, fmaf(s, oq_horner_fma(atanh2_over_s, 7, s * s), k * OQ_LN2_LO));
}
#endif
