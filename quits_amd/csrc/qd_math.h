/* qd_math.h -- the two elementary functions of product-sum BP, in float, built from IEEE basic operations only.
 *
 * ldpc's product-sum check update (src_cpp/bp.hpp; call sites quits/decoder/sliding_window.py:149,171) evaluates
 *      tanh(b2c / 2)           and          log((1 + x) / (1 - x))
 * in double through libm.  The device path computes in float, and a libm result is not reproducible across a CPU and a
 * GPU math library, so both the HIP kernel (quits_amd/csrc/bp_general.hip) and its CPU mirror (oracle/bp_core.inc with
 * REAL = float) evaluate the same expressions below: additions, multiplications, correctly rounded divisions, floorf
 * and bit casts, in a fixed order, compiled with -ffp-contract=off on both sides.  Same input bits -> same output bits.
 * tests/test_oracle.py checks them against libm in double (a few ulp) so that sharing them cannot hide an error.
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

/* tanh(x / 2) */
QD_MATH_FN float qd_tanh_half(float x)
{
    const float ax = qd_u2f(qd_f2u(x) & 0x7FFFFFFFu);
    const uint32_t sign = qd_f2u(x) & 0x80000000u;
    float t;
    if (!(ax >= 0.5f)) {
        /* |x/2| < 1/4: odd Taylor polynomial of tanh, Horner in h^2 (next term 3e-10 relative) */
        const float h = ax * 0.5f;
        const float h2 = h * h;
        float p = -0.0088632355f;          /* -1382/155925 */
        p = p * h2 + 0.021869488f;         /*  62/2835     */
        p = p * h2 + -0.053968254f;        /* -17/315      */
        p = p * h2 + 0.13333334f;          /*  2/15        */
        p = p * h2 + -0.33333334f;         /* -1/3         */
        p = p * h2 + 1.0f;
        t = h * p;
    } else {
        /* tanh(x/2) = 1 - 2 / (e^x + 1), e^x by Cody-Waite reduction and a degree-6 polynomial */
        const float a = ax > 40.0f ? 40.0f : ax;
        const float kf = floorf(a * 1.4426950f + 0.5f);
        float r = a - kf * 0.693145752f;   /* ln2 high part (exact product for |k| < 2^11) */
        r = r - kf * 1.42860677e-06f;      /* ln2 low part  */
        float p = 0.0013888889f;           /* 1/720 */
        p = p * r + 0.008333334f;          /* 1/120 */
        p = p * r + 0.041666668f;          /* 1/24  */
        p = p * r + 0.16666667f;           /* 1/6   */
        p = p * r + 0.5f;
        p = p * r + 1.0f;
        p = p * r + 1.0f;
        const float e = p * qd_u2f((uint32_t)((int)kf + 127) << 23);
        t = 1.0f - 2.0f / (e + 1.0f);
        if (t > QD_TANH_MAX) t = QD_TANH_MAX;
    }
    return qd_u2f(qd_f2u(t) | sign);
}

/* log((1 + c) / (1 - c)) = 2 atanh(c) for |c| <= 1 - 2^-24 */
QD_MATH_FN float qd_log_ratio(float c)
{
    const float ac = qd_u2f(qd_f2u(c) & 0x7FFFFFFFu);
    if (!(ac > 0.171875f)) {
        /* (q - 1) / (q + 1) = c exactly, so the atanh series applies to c itself: no division, no cancellation */
        const float c2 = c * c;
        float p = 0.15384616f;             /* 2/13 */
        p = p * c2 + 0.18181819f;          /* 2/11 */
        p = p * c2 + 0.22222222f;          /* 2/9  */
        p = p * c2 + 0.2857143f;           /* 2/7  */
        p = p * c2 + 0.4f;                 /* 2/5  */
        p = p * c2 + 0.6666667f;           /* 2/3  */
        p = p * c2 + 2.0f;
        return c * p;
    }
    const float q = (1.0f + c) / (1.0f - c);      /* in [2^-25, 2^25]: positive and normal */
    const uint32_t qb = qd_f2u(q);
    int e = (int)((qb >> 23) & 255u) - 127;
    float m = qd_u2f((qb & 0x007FFFFFu) | 0x3F800000u);          /* [1, 2) */
    if (m > 1.4142135f) {
        m = m * 0.5f;
        e += 1;
    }
    const float s = (m - 1.0f) / (m + 1.0f);      /* |s| <= 0.1716 */
    const float s2 = s * s;
    float p = 0.15384616f;
    p = p * s2 + 0.18181819f;
    p = p * s2 + 0.22222222f;
    p = p * s2 + 0.2857143f;
    p = p * s2 + 0.4f;
    p = p * s2 + 0.6666667f;
    p = p * s2 + 2.0f;
    const float ef = (float)e;
    return ef * 0.693145752f + (s * p + ef * 1.42860677e-06f);
}

#endif
