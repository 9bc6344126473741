// bp_scatter_edge.h -- the per-edge steps of the scatter form of flooding min-sum (gather pass: QS_EDGE, scatter pass: QS_SCAT), shared by
// qd_bp_scatter_kernel (bp_scatter.hip: one check per lane) and qd_bp_scatter_wide_kernel (bp_scatter_wide.hip: several checks per lane).
// The macros work on the enclosing scope's names: s1, s2, kold, sgnw (what the check sent last), a1, a2, ltw, neww, hp (what this pass
// collects), dc (the check's degree), pdif, pxq, own, xw (scatter pass).
#pragma once
#include "qd_internal.h"
#include <float.h>

__device__ __forceinline__ float qs_min_abs(float a, float b)       // min(a, |b|) as one instruction (see bp_kernels.hip)
{
    float r;
    asm("v_min_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
typedef uint32_t qs_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 qs_as_uint4(qs_u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); }
typedef __attribute__((address_space(3))) int32_t qs_lds_i32;
#define QS_LDS(off) ((qs_lds_i32 *)(uintptr_t)(uint32_t)(off))
#define QS_BIG 1.0e30f
#ifndef QS_PRIO
#define QS_PRIO 1         // wavefront priority during the scatter pass (short, bound by the LDS): 53.4 -> 52.8 ms at the headline
#endif

// Gather pass, one edge.  off = LDS byte offset of the fault's accumulator (L - 1) in the buffer being read; k_ = position of the
// edge in the check's walk (wave-uniform), compared with the argmin label of the last pass; sb = bit of `sgnw` that holds the sign of the message this check sent
// on the edge.  d_ = (L - 1) - prev = bm - 1 is an integer-valued float, so (bm <= 0) is its sign bit (-0.0 cannot occur: an
// integer converted to float is never -0, and x - y is -0 only for x = -0).
// HP: how the accumulator's bits enter `hp` (only bit 31, the hard decision, is used): QS_HP1 one XOR per edge; QS_HPA / QS_HPB on the
// two edges of a pair -- the first is remembered in `hpa`, the second folds both in with one v_bitop3_b32 (a ^ b ^ c): half an instruction per edge
// QS_SIGN31(sb): the sign this check sent on the edge (bit `sb` of sgnw) moved to bit 31.  (Keeping sgnw pre-shifted so that the shift amounts are
// immediates instead of one scalar each measured slower: 42.6 -> 43.1 ms, profiles/r05_k1sw_micro_ab.txt.)
#define QS_SIGN31(sb) (((sgnw >> (sb)) & 1u) << 31)
#define QS_HP1(A_) hp ^= (uint32_t)(A_);
#define QS_HPA(A_) hpa = (uint32_t)(A_);
#define QS_HPB(A_) hp = __builtin_amdgcn_bitop3_b32(hp, hpa, (uint32_t)(A_), 0x96);
#define QS_EDGE(off, k_, sb, TAILFIX) QS_EDGE_H(off, k_, sb, TAILFIX, QS_HP1)
#define QS_EDGE_H(off, k_, sb, TAILFIX, HP)                                                                  \
    {                                                                                                        \
        const int A_ = *QS_LDS(QS_ADDR(off));                                                                \
        const float mag_ = ((uint32_t)(k_) == kold) ? s2 : s1;                                               \
        const float prev_ = __uint_as_float(QS_SIGN31(sb) | __float_as_uint(mag_));                          \
        float d_ = (float)A_ - prev_;                                                                        \
        TAILFIX(d_, k_)                                                                                      \
        const float bm_ = d_ + 1.0f;                                                                         \
        HP(A_)                                                                                               \
        neww = __builtin_amdgcn_alignbit(neww, __float_as_uint(d_), 31);                                     \
        /* ltw = ltw << 1 | (|bm| < a1): the argmin is the edge of the LAST strict improvement */           \
        asm("v_cmp_lt_f32 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(ltw) : "v"(bm_), "v"(a1) : "vcc"); \
        a2 = __builtin_amdgcn_fmed3f(a1, a2, fabsf(bm_));                                                    \
        a1 = qs_min_abs(a1, bm_);                                                                            \
    }
#define QS_NOFIX(x_, k_)
// beyond this lane's degree the walk reads the trash slot (0 for ever: no hard decision); the difference must be neither negative
// nor a minimum
#define QS_TAILFIX(x_, k_) x_ = ((int)(k_) < dc) ? x_ : QS_BIG;

#if defined(QS_ABL_NOCONF)      /* timing experiment only (wrong results): every lane's accumulator address moved to bank (lane mod 32) of its 128-byte row */
#define QS_ADDR(off) ((((uint32_t)(off)) & ~0x7Cu) | ((threadIdx.x & 31u) << 2))
#elif defined(QS_ABL_ADDRCTL)   /* ... its control: the same extra instruction per access, addresses unchanged */
__device__ __forceinline__ uint32_t qs_opaque(uint32_t x) { asm volatile("" : "+s"(x)); return x; }
#define QS_ADDR(off) ((((uint32_t)(off)) & qs_opaque(0xFFFFFFFFu)) | (threadIdx.x & qs_opaque(0u)))
#else
#define QS_ADDR(off) (off)
#endif
#if defined(QS_ABL_STORE)       /* timing experiments only (wrong results): a plain store instead of the atomic add ... */
#define QS_ADD(off, v_) *QS_LDS(off) = v_;
#elif defined(QS_ABL_CONSTV)    /* ... the atomic add of a constant (the value's arithmetic is dead code) */
#define QS_ADD(off, v_) (void)__hip_atomic_fetch_add(QS_LDS(off), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif defined(QS_ABL_READ)      /* ... a plain LDS read per edge instead of the atomic add (what a gather-form bit pass would issue at least) */
#define QS_ADD(off, v_) { const int r_ = *QS_LDS(off) ^ (v_); asm volatile("" ::"v"(r_)); }
#elif defined(QS_ABL_NOADD)     /* ... no LDS operation at all in the scatter pass (its vector arithmetic stays) */
#define QS_ADD(off, v_) asm volatile("" ::"v"(v_));
#else
#define QS_ADD(off, v_) (void)__hip_atomic_fetch_add(QS_LDS(QS_ADDR(off)), v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
// Scatter pass, one edge: (new message) - (message sent in the last iteration) = sn a - so b with a, b = min1 of the two passes
// (the argmin edges are corrected after the loop) and sn, so = +-1 the outgoing signs: sn (a - b) where the signs agree, sn (a + b)
// where they differ.  `own` holds the new signs, `xw` = new ^ sent, edge i of the group at bit 31 - i; pdif = a - b, pxq = (a - b) ^ (a + b).
#define QS_SCAT(off, k_, bit_, TAILFIX)                                                                      \
    {                                                                                                        \
        const int dm_ = __builtin_amdgcn_sbfe((int)xw, bit_, 1), sm_ = __builtin_amdgcn_sbfe((int)own, bit_, 1); \
        const int mg_ = (int)__builtin_amdgcn_bitop3_b32((uint32_t)pdif, (uint32_t)pxq, (uint32_t)dm_, 0x78);   /* pdif ^ (pxq & dm) */ \
        int v_ = (mg_ ^ sm_) - sm_;                                                                          \
        TAILFIX(v_, k_)                                                                                      \
        QS_ADD(off, v_)                                                                                      \
    }
#define QS_TAILZERO(v_, k_) v_ = ((int)(k_) < dc) ? v_ : 0;

