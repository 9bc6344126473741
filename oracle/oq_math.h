/* oq_math.h -- the oracle's own float tanh(x/2) and log((1+c)/(1-c)).
 *
 * TEST INFRASTRUCTURE ONLY.  ldpc's product-sum update (bp.hpp) evaluates both in double through libm; the device computes in
 * float, where a libm result is not reproducible, so the HIP kernel (quits_amd/csrc/qd_math.h) builds them from IEEE basic
 * operations in a fixed order.  This file restates that arithmetic independently -- written from the description below, not
 * included from the product -- so that the checker and the product share no code; tests/test_oracle.py holds both to libm in
 * double (a few ulp), and the GPU parity tests hold the kernel to this file bit for bit.  Compile with -ffp-contract=off.
 *
 *   Every "x*y + z" below that is written fma(x, y, z) is ONE rounding (C99 fmaf); nothing else is fused (-ffp-contract=off).
 *
 *   tanh(x/2), a = min(|x|, 40):  k = floor(fma(a, log2(e), 1/2)),  r = fma(k, -ln2_lo, fma(k, -ln2_hi, a)),
 *                         E = r * P(r), P = Horner by fma of 1 + r/2 + r^2/6 + ... + r^6/5040        (E = expm1(r)),
 *                         D = fma(E, 2^k, 2^k - 1)   (= e^a - 1),   t = D / (D + 2), capped at 1 - 2^-24; sign restored
 *   log((1+c)/(1-c)):     split 1+c = 2^ku mu and 1-c = 2^kv mv with mu, mv in [sqrt(1/2), sqrt 2): adding 0x3F800000 - 0x3F3504F3 to
 *                         the bit pattern carries into the exponent field exactly when the mantissa reaches sqrt 2;
 *                         s = (mu - mv) / (mu + mv), replaced by c itself when ku = kv (it is c, exactly, in that case);
 *                         result = fma(e, ln2_hi, fma(s, Q(s^2), e * ln2_lo)),  e = ku - kv,
 *                         Q = Horner by fma of 2 + 2z/3 + 2z^2/5 + ... + 2z^7/15            (s Q(s^2) = 2 atanh(s), |s| <= 1/3)
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

static inline float oq_tanh_half_f32(float x)
{
    static const float expm1_over_r[7] = {1.9841270e-04f, 0.0013888889f, 0.008333334f, 0.041666668f, 0.16666667f, 0.5f, 1.0f};   /* 1/7!, ..., 1/2!, 1 */
    const uint32_t xb = oq_bits(x);
    float a = oq_float(xb & 0x7FFFFFFFu);
    if (a > 40.0f) a = 40.0f;
    const float k = floorf(fmaf(a, 1.4426950f, 0.5f));
    const float r = fmaf(k, -OQ_LN2_LO, fmaf(k, -OQ_LN2_HI, a));
    const float E = r * oq_horner_fma(expm1_over_r, 7, r);
    const float pow2k = oq_float((uint32_t)((int)k + 127) << 23);
    const float D = fmaf(E, pow2k, pow2k - 1.0f);
    float t = D / (D + 2.0f);
    if (t > 0.99999994f) t = 0.99999994f;
    return oq_float(oq_bits(t) | (xb & 0x80000000u));
}

static inline float oq_mantissa_sqrt2(float x, int *exponent)      /* x > 0 normal: x = 2^(exponent - bias) * result, result in [sqrt(1/2), sqrt 2) */
{
    const uint32_t shifted = oq_bits(x) + (0x3F800000u - 0x3F3504F3u);
    *exponent = (int)(shifted >> 23);
    return oq_float((shifted & 0x007FFFFFu) + 0x3F3504F3u);
}

static inline float oq_log_ratio_f32(float c)
{
    static const float atanh2_over_s[8] = {0.13333334f, 0.15384616f, 0.18181819f, 0.22222222f, 0.2857143f, 0.4f, 0.6666667f, 2.0f};   /* 2/15, 2/13, ..., 2/3, 2 */
    int ku, kv;
    const float mu = oq_mantissa_sqrt2(1.0f + c, &ku), mv = oq_mantissa_sqrt2(1.0f - c, &kv);
    const float s = (ku == kv) ? c : (mu - mv) / (mu + mv);
    const float ef = (float)(ku - kv);
    return fmaf(ef, OQ_LN2_HI, fmaf(s, oq_horner_fma(atanh2_over_s, 8, s * s), ef * OQ_LN2_LO));
}
#endif
