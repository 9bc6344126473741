/* qd_math.h -- the elementary functions of product-sum BP, in float, built from IEEE basic operations only.
 *
 * ldpc's product-sum check update (src_cpp/bp.hpp; call sites quits/decoder/sliding_window.py:149,171) evaluates, in double
 * through libm,       c2b(i -> j) = +- log((1 + P) / (1 - P)),   P = product over the other edges of tanh(b2c / 2).
 * A float cannot hold tanh(x/2) beyond |x| = 17.3 (1 - 2 e^-x rounds to 1), so a float copy of that formula has to clamp the
 * tanh at 1 - 2^-24 and caps every check->bit message at 17.33 where the double reaches 37.4 -- and that cap is measurable:
 * it acts as a damping and LOWERS the logical error rate (rounds 1-3 of this repo: 127 against 139 failures of 2048 on the
 * reference's HGP phenomenological example, -0.5 sigma / McNemar z = -2.7 on 400 000 headline shots; a double-precision run
 * with the same clamp reproduces the float numbers, profiles/r03w_*).  So the device keeps, instead of t = tanh(|x| / 2),
 *        u = (1 - t) / (1 + t) = e^-|x|              (the full float range: |x| up to 87)
 * with the sign of x in the sign bit.  The product rule t12 = t1 t2 becomes
 *        u12 = (u1 + u2) / (1 + u1 u2)               (identity 0, all terms non-negative: no cancellation anywhere)
 * and log((1 + P) / (1 - P)) = -log(u).  A relative error in u is an ABSOLUTE error in the LLR, ~1e-7 per operation.
 * A libm result is not reproducible across a CPU and a GPU math library, so both the HIP kernels (bp_general.hip) and the CPU
 * mirror (oracle/oq_math.h, written independently from the description there) evaluate the same expressions: additions,
 * multiplications, FUSED multiply-adds (fmaf, one rounding), correctly rounded divisions, floorf and bit casts, in a fixed
 * order, compiled with -ffp-contract=off on both sides so that nothing else is fused.  Same input bits -> same output bits.
 * Each function is one straight line of code: a wavefront whose lanes are shots or edges takes both sides of every
 * data-dependent branch.  tests/test_oracle.py checks them against libm in double.
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

#define QD_U_MIN 1.17549435e-38f  /* 2^-126: the smallest u kept (a message of 87.3) */

/* +-e^-|x| with the sign bit of x:  |x| = k ln2 + r (Cody-Waite, |r| <= ln2 / 2),  e^-r = 1 + expm1(-r),
 * expm1(z) = z (1 + z/2 + ... + z^6/5040),  scaled by 2^-k. */
QD_MATH_FN float qd_exp_neg(float x)
{
    const float ax = qd_u2f(qd_f2u(x) & 0x7FFFFFFFu);
    const uint32_t sign = qd_f2u(x) & 0x80000000u;
    const float a = ax > 87.0f ? 87.0f : ax;
    const float kf = floorf(fmaf(a, 1.4426950f, 0.5f));
    float z = fmaf(kf, 0.693145752f, -a);  /* -(a - k ln2_hi): exact product for |k| < 2^11 */
    z = fmaf(kf, 1.42860677e-06f, z);      /* + k ln2_lo:  z = -r */
    float p = 1.9841270e-04f;              /* 1/5040 */
    p = fmaf(p, z, 0.0013888889f);         /* 1/720  */
    p = fmaf(p, z, 0.008333334f);          /* 1/120  */
    p = fmaf(p, z, 0.041666668f);          /* 1/24   */
    p = fmaf(p, z, 0.16666667f);           /* 1/6    */
    p = fmaf(p, z, 0.5f);
    p = fmaf(p, z, 1.0f);
    const float em1 = z * p;                                               /* expm1(-r) */
    const float two_mk = qd_u2f((uint32_t)(127 - (int)kf) << 23);          /* 2^-k, k in [0, 126] */
    const float u = fmaf(em1, two_mk, two_mk);
    return qd_u2f(qd_f2u(u) | sign);
}

/* (a + b) / (1 + a b) for a, b in [0, 1]: the u of a product of two tanh values; capped at 1 (the quotient of the two rounded
 * sums can exceed it by an ulp) */
QD_MATH_FN float qd_ucomb(float a, float b)
{
    const float q = (a + b) / fmaf(a, b, 1.0f);
    return q > 1.0f ? 1.0f : q;
}

/* x = 2^k m with m in [sqrt(1/2), sqrt 2) for a positive normal x: adding (1 - sqrt(1/2)) in units of the last place to the
 * bit pattern carries into the exponent exactly when the mantissa is at least sqrt 2 */
#define QD_SQRT_HALF_BITS 0x3F3504F3u

/* -log(u) for u in [0, 1] (u below 2^-126 counts as 2^-126):  u = 2^k m,  s = (m - 1) / (m + 1), |s| <= 0.1716,
 * -log(u) = -(k ln2 + 2 atanh(s)),  2 atanh(s) = s (2 + 2 s^2/3 + ... + 2 s^12/13) */
QD_MATH_FN float qd_neg_log(float u)
{
    const float uc = u < QD_U_MIN ? QD_U_MIN : u;
    const uint32_t b = qd_f2u(uc) + (0x3F800000u - QD_SQRT_HALF_BITS);
    const float kf = (float)((int)(b >> 23) - 127);
    const float m = qd_u2f((b & 0x007FFFFFu) + QD_SQRT_HALF_BITS);
    const float s = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p = 0.15384616f;                 /* 2/13 */
    p = fmaf(p, s2, 0.18181819f);          /* 2/11 */
    p = fmaf(p, s2, 0.22222222f);          /* 2/9  */
    p = fmaf(p, s2, 0.2857143f);           /* 2/7  */
    p = fmaf(p, s2, 0.4f);                 /* 2/5  */
    p = fmaf(p, s2, 0.6666667f);           /* 2/3  */
    p = fmaf(p, s2, 2.0f);
    const float lg = fmaf(kf, 0.693145752f, fmaf(s, p, kf * 1.42860677e-06f));   /* log(u) <= 0 */
    return 0.0f - lg;
}

#endif
