/* qd_math.h -- the two elementary functions of product-sum BP, in float, built from IEEE basic operations only.
 *
 * ldpc's product-sum check update (src_cpp/bp.hpp; call sites quits/decoder/sliding_window.py:149,171) evaluates
 *      tanh(b2c / 2)           and          log((1 + x) / (1 - x))
 * in double through libm.  The device path computes in float, and a libm result is not reproducible across a CPU and a
 * GPU math library, so both the HIP kernel (quits_amd/csrc/bp_general.hip) and its CPU mirror (oracle/oq_math.h, written
 * independently from the description there) evaluate the same expressions: additions, multiplications, FUSED multiply-adds
 * (fmaf, one rounding), correctly rounded divisions, floorf and bit casts, in a fixed order, compiled with
 * -ffp-contract=off on both sides so that nothing else is fused.  Same input bits -> same output bits.
 * tests/test_oracle.py checks them against libm in double so that mirroring them cannot hide an error.
 *
 * Both functions are ONE straight line of code (round 3): a wavefront whose lanes are shots or edges takes both sides of
 * every data-dependent branch, so the two-branch forms of rounds 1-2 (polynomial near zero, exp / log elsewhere) cost the
 * sum of their branches -- ~125 instructions per edge against ~70 here (one division each).
 *
 * Float-specific conventions (documented deviations from the double arithmetic of ldpc):
 *   - tanh saturates to 1 in float for |x/2| > 9; the result is clamped to +-(1 - 2^-24) so that (1+x)/(1-x) stays
 *     finite.  A check-to-bit message is therefore bounded by log(2^25) = 17.33 (a double reaches 37 before the same
 *     happens); posteriors are sums and are not clamped.
 */
#ifndef QD_MATH_H
#define QD_MATH_H
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define QD_MATH_FN __host__ __device__ static inline
#else
#define QD_MATH_FN static inline
#endif

QD_MATH_FN uint32_t qd_f2u(float x)
{
    union { float f; uint32_t u; } v;
    v.f = x;
    return v.u;
}
QD_MATH_FN float qd_u2f(uint32_t x)
{
    union { float f; uint32_t u; } v;
    v.u = x;
    return v.f;
}

#define QD_TANH_MAX 0.99999994f /* 1 - 2^-24, the largest float below 1 */

/* tanh(x / 2) = (e^a - 1) / (e^a + 1), a = |x|:  e^a - 1 = 2^k (1 + expm1(r)) - 1 with a = k ln2 + r (Cody-Waite), expm1(r) =
 * r (1 + r/2 + ... + r^6/5040) for |r| <= ln2 / 2.  For k = 0 this is expm1(r) itself: no cancellation near zero. */
QD_MATH_FN float qd_tanh_half(float x)
{
    const float ax = qd_u2f(qd_f2u(x) & 0x7FFFFFFFu);
    const uint32_t sign = qd_f2u(x) & 0x80000000u;
    const float a = ax > 40.0f ? 40.0f : ax;
    const float kf = floorf(fmaf(a, 1.4426950f, 0.5f));
    float r = fmaf(kf, -0.693145752f, a);  /* ln2 high part (exact product for |k| < 2^11) */
    r = fmaf(kf, -1.42860677e-06f, r);     /* ln2 low part  */
    float p = 1.9841270e-04f;              /* 1/5040 */
    p = fmaf(p, r, 0.0013888889f);         /* 1/720  */
    p = fmaf(p, r, 0.008333334f);          /* 1/120  */
    p = fmaf(p, r, 0.041666668f);          /* 1/24   */
    p = fmaf(p, r, 0.16666667f);           /* 1/6    */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    const float em1r = r * p;                                              /* expm1(r) */
    const float two_k = qd_u2f((uint32_t)((int)kf + 127) << 23);           /* 2^k, k in [0, 58] */
    const float em1 = fmaf(em1r, two_k, two_k - 1.0f);                     /* e^a - 1 */
    float t = em1 / (em1 + 2.0f);
    if (t > QD_TANH_MAX) t = QD_TANH_MAX;
    return qd_u2f(qd_f2u(t) | sign);
}

/* x = 2^k m with m in [sqrt(1/2), sqrt 2) for a positive normal x: adding (1 - sqrt(1/2)) in units of the last place to the
 * bit pattern carries into the exponent exactly when the mantissa is at least sqrt 2 */
#define QD_SQRT_HALF_BITS 0x3F3504F3u
QD_MATH_FN float qd_split_sqrt2(float x, int *k)
{
    const uint32_t b = qd_f2u(x) + (0x3F800000u - QD_SQRT_HALF_BITS);
    *k = (int)(b >> 23);                                                   /* biased; only differences are used */
    return qd_u2f((b & 0x007FFFFFu) + QD_SQRT_HALF_BITS);
}

/* log((1 + c) / (1 - c)) = 2 atanh(c) for |c| <= 1 - 2^-24:  1 + c = 2^ku mu, 1 - c = 2^kv mv with mu, mv in [sqrt(1/2), sqrt 2),
 * so the ratio is 2^(ku-kv) (1 + s) / (1 - s) with s = (mu - mv) / (mu + mv), |s| <= 1/3, and the result
 * (ku - kv) ln2 + 2 atanh(s).  When ku = kv (|c| < 0.17 at least) s is c itself, exactly: no division error, no cancellation. */
QD_MATH_FN float qd_log_ratio(float c)
{
    int ku, kv;
    const float mu = qd_split_sqrt2(1.0f + c, &ku);
    const float mv = qd_split_sqrt2(1.0f - c, &kv);
    float s = (mu - mv) / (mu + mv);
    if (ku == kv) s = c;
    const float s2 = s * s;
    float p = 0.13333334f;                 /* 2/15 */
    p = fmaf(p, s2, 0.15384616f);          /* 2/13 */
    p = fmaf(p, s2, 0.18181819f);          /* 2/11 */
    p = fmaf(p, s2, 0.22222222f);          /* 2/9  */
    p = fmaf(p, s2, 0.2857143f);           /* 2/7  */
    p = fmaf(p, s2, 0.4f);                 /* 2/5  */
    p = fmaf(p, s2, 0.6666667f);           /* 2/3  */
    p = fmaf(p, s2, 2.0f);
    const float ef = (float)(ku - kv);
    return fmaf(ef, 0.693145752f, fmaf(s, p, ef * 1.42860677e-06f));
}

#endif
