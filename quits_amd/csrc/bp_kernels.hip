// bp_kernels.hip -- flooding min-sum belief propagation, one workgroup per shot, all message state in LDS.
//
// Replaces ldpc.BpOsdDecoder.decode -> BpDecoder::bp_decode_parallel (MINIMUM_SUM) as the reference calls it at
// quits/decoder/sliding_window.py:171,182.  CPU restatement with the same arithmetic: oracle/bp_core.inc,
// bp_minsum_compressed (REAL=float) -- the two agree bit for bit (tests/test_gpu_parity.py).
//
// Design (gfx950):
//   * A check keeps only (min1, min2, argmin, parity, per-edge sign bits) of its incoming messages: 16 bytes instead
//     of one float per edge; a fault keeps its posterior LLR and one sign bit per edge.  For the [[144,12,12]] R=12
//     window (1008 x 9504, 33192 edges) that is 68 KB of LDS per shot, so two shots are resident per CU and the
//     message traffic never leaves the CU.  HBM only sees the syndrome bytes in and the packed decision bits out.
//   * lane = node.  Check pass: lane owns a check, walks its faults (ELL-transposed adjacency -> coalesced index
//     loads, LDS gathers of the posteriors).  Bit pass: lane owns a fault, gathers the <= 16 checks' 16-byte states
//     with one ds_read_b128 each.  Nodes are degree-sorted so a wavefront's lanes share a trip count.
//   * The syndrome test of iteration t rides on check pass t+1 (it needs the same gathers), so an iteration costs
//     one check pass + one bit pass + two barriers.
//   * No atomics on floats, fixed summation order (ascending detector index) -> bit-reproducible.
#include "qd_internal.h"
#include <float.h>

#ifndef QD_ABLATE
#define QD_ABLATE 0        // timing experiments only (tools/ablate_bp.sh); any value but 0 breaks the results
#endif
#ifndef QD_BP_MINWAVES
#define QD_BP_MINWAVES 8   // waves per SIMD the register allocator must leave room for (8 = two 1024-thread workgroups per CU)
#endif

// One edge of the check pass.  L = posterior of the fault, k = edge position inside the check, kk = its bit inside the
// current 32-edge sign word `sgnw` (bit = sign of the previous check->bit message on that edge).
// Branch-free: the second minimum is the median of (min1, min2, |b|).
#define QD_CHECK_EDGE(L, k, kk)                                                                              \
    {                                                                                                        \
        us ^= ((L) <= 0.f);                                                                                  \
        const float mag_ = ((k) == idx_old) ? st.y : st.x;                                                   \
        const float prev_ = __uint_as_float(__builtin_amdgcn_ubfe(sgnw, (kk), 1) << 31 | __float_as_uint(mag_)); \
        const float bm_ = (L) - prev_;                 /* bit->check message, "total minus own" */            \
        const float ab_ = fabsf(bm_);                                                                        \
        neww |= ((bm_ <= 0.f) ? 1u : 0u) << (kk);      /* bp.hpp: a message <= 0 counts as negative */        \
        idx = (ab_ < a1) ? (k) : idx;                                                                        \
        a2 = __builtin_amdgcn_fmed3f(a1, a2, ab_);                                                           \
        a1 = fminf(a1, ab_);                                                                                 \
    }

// State of a check in LDS (16 bytes, one ds_read_b128):  x = min1 * alpha,  y = min2 * alpha,
//   z = argmin position | syndrome bit << 9,  w = SIGN bits of the outgoing messages on edges 0..31
//   (sign of check->bit message k = syndrome ^ parity of all incoming signs ^ incoming sign k).
// Edges 32.. of wide checks keep their sign words in `csgn_hi`.
template <int T, int MAXCD, bool WIDE>
__global__ void __launch_bounds__(T, QD_BP_MINWAVES) qd_bp_minsum_kernel(BpGraphDev g, DecodeArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float4 *chk = reinterpret_cast<float4 *>(smem + g.off_chk);
    uint32_t *csgn_hi = reinterpret_cast<uint32_t *>(smem + g.off_cneg);
    float *llr = reinterpret_cast<float *>(smem + g.off_llr);
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + g.off_out);
    volatile int *misc = reinterpret_cast<volatile int *>(smem + g.off_misc);   // [0..31] OR flags, [32] fail slot
    constexpr int NW = T / 64;

    const int tid = threadIdx.x;
    const int64_t shot = blockIdx.x;
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    const int m_pad = g.m_pad, n_pad = g.n_pad;

    // ---- load the window syndrome (sliding_window.py:168-169) and reset the state
    int any = 0;
    for (int c = tid; c < g.m; c += T) {
        const uint32_t o = g.chk_orig[c];
        uint32_t s = det[o] & 1u;
        if (upd && (int)o < a.upd_rows) s ^= upd[o] & 1u;
        any |= (int)s;
        chk[c] = make_float4(0.f, 0.f, __uint_as_float(0xFFu | (s << 9)), __uint_as_float(0u));   // no message yet
        if (WIDE)
            for (int w = 1; w < g.neg_words; ++w) csgn_hi[(w - 1) * m_pad + c] = 0u;
    }
    for (int b = tid; b < g.n; b += T) llr[b] = g.bit_llr0[b];
    for (int w = tid; w < g.out_words; w += T) outw[w] = 0u;
    if (tid == 0) {
        llr[g.dummy_bit] = __builtin_inff();                                            // padding edge of a short row: |b| = inf, never a minimum, never negative
        chk[g.dummy_chk] = make_float4(0.f, 0.f, __uint_as_float(0xFFu), __uint_as_float(0u));   // padding edge of a short column: message +0
    }
    any = qd_block_or(any, misc, NW, 0);
    if (!any) {   // bposd_decoder.pyx: an all-zero syndrome returns the zero vector without running BP
        for (int w = tid; w < g.out_words; w += T) a.err_bits[shot * g.out_words + w] = 0u;
        if (tid == 0) a.status[shot] = (1 << 16) | (1 << 19);
        return;
    }

    int t = 0, converged = 0, phase = 1;
    for (;;) {
        const float alpha = (a.ms_scale == 0.f) ? (1.0f - ldexpf(1.0f, -(t + 1))) : a.ms_scale;
        // ---- check pass t+1; its parity test is the convergence test of iteration t
        bool unsat = false;
        for (int c = tid; c < g.m; c += T) {
            const float4 st = chk[c];
            const uint32_t meta = __float_as_uint(st.z);
            const int idx_old = (int)(meta & 0xFFu);
            const uint32_t synd = (meta >> 9) & 1u;
            const int degp = __builtin_amdgcn_readfirstlane((int)g.chk_degp[c]);   // wave-uniform, multiple of 4
            bool us = (synd != 0u);
            int idx = 255;
            float a1 = FLT_MAX, a2 = FLT_MAX;
            uint32_t neg0 = 0u, npar = 0u;
            for (int k0 = 0; k0 < degp; k0 += 32) {
                const uint32_t sgnw = (!WIDE || k0 == 0) ? __float_as_uint(st.w) : csgn_hi[((k0 >> 5) - 1) * m_pad + c];
                uint32_t neww = 0u;
                const int kend = min(degp - k0, 32);                // multiple of 4
                const uint16_t *adj = g.chk_adj + (size_t)k0 * m_pad + c;
#if QD_ABLATE == 1 || QD_ABLATE == 12
                uint32_t j0 = (c * 7 + k0 * 13) % g.n, j1 = j0 + 1, j2 = j0 + 2, j3 = j0 + 3;
#elif QD_ABLATE == 2
                uint32_t j0 = c, j1 = c, j2 = c, j3 = c;
#else
                uint32_t j0 = adj[0], j1 = adj[m_pad], j2 = adj[2 * m_pad], j3 = adj[3 * m_pad];
#endif
#pragma unroll 1
                for (int kk = 0; kk < kend; kk += 4) {
                    const float L0 = llr[j0], L1 = llr[j1], L2 = llr[j2], L3 = llr[j3];
#if QD_ABLATE == 1 || QD_ABLATE == 12
                    j0 = (j0 + 977) % g.n; j1 = (j1 + 977) % g.n; j2 = (j2 + 977) % g.n; j3 = (j3 + 977) % g.n;
#elif QD_ABLATE == 2
#else
                    if (kk + 4 < kend) {                             // next four fault indices while these are processed
                        const uint16_t *nx = adj + (size_t)(kk + 4) * m_pad;
                        j0 = nx[0]; j1 = nx[m_pad]; j2 = nx[2 * m_pad]; j3 = nx[3 * m_pad];
                    }
#endif
                    const int kb = k0 + kk;
                    QD_CHECK_EDGE(L0, kb + 0, kk + 0)
                    QD_CHECK_EDGE(L1, kb + 1, kk + 1)
                    QD_CHECK_EDGE(L2, kb + 2, kk + 2)
                    QD_CHECK_EDGE(L3, kb + 3, kk + 3)
                }
                npar ^= neww;
                if (k0 == 0) neg0 = neww;
                else if (WIDE) csgn_hi[((k0 >> 5) - 1) * m_pad + c] = neww;   // fixed up below once the parity is known
            }
            unsat |= us;
            // outgoing sign on edge k = syndrome ^ (parity of all incoming signs) ^ incoming sign k
            const uint32_t flip = 0u - ((synd ^ (uint32_t)__popc(npar)) & 1u);
            if (WIDE)
                for (int k0 = 32; k0 < degp; k0 += 32) csgn_hi[((k0 >> 5) - 1) * m_pad + c] ^= flip;
            chk[c] = make_float4(a1 * alpha, a2 * alpha, __uint_as_float((uint32_t)idx | (synd << 9)), __uint_as_float(neg0 ^ flip));
        }
        const int anyun = qd_block_or(unsat ? 1 : 0, misc, NW, phase);
        phase ^= 1;
        if (t >= 1 && !anyun) { converged = 1; break; }
        if (t == a.max_iter) break;
        // ---- bit pass t+1: posterior = prior + sum of check->bit messages, in ascending detector order
        for (int b = tid; b < g.n; b += T) {
            const int d = __builtin_amdgcn_readfirstlane((int)g.bit_degp[b]);     // wave-uniform
            uint32_t aj[MAXCD];
#pragma unroll
#if QD_ABLATE == 3 || QD_ABLATE == 12
            for (int q = 0; q < MAXCD; ++q) aj[q] = (uint32_t)(((b * 5 + q * 131) % g.m) << 16) | (uint32_t)q;
#elif QD_ABLATE == 4
            for (int q = 0; q < MAXCD; ++q) aj[q] = (uint32_t)((b % g.m) << 16) | (uint32_t)q;
#else
            for (int q = 0; q < MAXCD; ++q) aj[q] = (q < d) ? g.bit_adj[(size_t)q * n_pad + b] : 0u;
#endif
            float acc = g.bit_llr0[b];
#pragma unroll
            for (int q = 0; q < MAXCD; ++q) {
                if (q < d) {
                    const uint32_t cs = aj[q] >> 16, pos = aj[q] & 0xFFu;       // check slot | edge position inside it
                    const float4 st = chk[cs];
                    const uint32_t meta = __float_as_uint(st.z);
                    const float mag = (pos == (meta & 0xFFu)) ? st.y : st.x;
                    uint32_t sw = __float_as_uint(st.w);
                    if (WIDE && pos >= 32u) sw = csgn_hi[((pos >> 5) - 1) * m_pad + cs];
                    acc += __uint_as_float(__builtin_amdgcn_ubfe(sw, pos & 31u, 1) << 31 | __float_as_uint(mag));
                }
            }
            llr[b] = acc;
        }
        __syncthreads();
        ++t;
    }

    // ---- hard decision, packed by fault index
    for (int b = tid; b < g.n; b += T)
        if (llr[b] <= 0.f) {
            const uint32_t j = g.bit_orig[b];
            atomicOr(&outw[j >> 5], 1u << (j & 31u));
        }
    if (!converged && a.want_llr && tid == 0) misc[32] = atomicAdd(a.fail_count, 1);
    __syncthreads();
    for (int w = tid; w < g.out_words; w += T) a.err_bits[shot * g.out_words + w] = outw[w];
    if (!converged && a.want_llr) {
        const int slot = misc[32];
        float *dst = a.llr_ws + (int64_t)slot * n_pad;
        for (int b = tid; b < g.n; b += T) dst[b] = llr[b];
        if (tid == 0) a.fail_list[slot] = (int32_t)shot;
    }
    if (tid == 0) a.status[shot] = t | (converged << 16);
}

// ---- launch wrappers -------------------------------------------------------------------------------------------------
template <int T, int MAXCD, bool WIDE>
static hipError_t launch_bp_k(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    auto k = qd_bp_minsum_kernel<T, MAXCD, WIDE>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(T), g.lds_bytes, s, g, a);
    return hipGetLastError();
}

template <int T>
static hipError_t launch_bp_t(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    const bool wide = g.max_rdeg_pad > 32;
    if (g.max_cdeg <= 8) return wide ? launch_bp_k<T, 8, true>(g, a, B, s) : launch_bp_k<T, 8, false>(g, a, B, s);
    return wide ? launch_bp_k<T, QD_MAX_COL_DEG, true>(g, a, B, s) : launch_bp_k<T, QD_MAX_COL_DEG, false>(g, a, B, s);
}

hipError_t qd_launch_bp(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    switch (g.threads) {
    case 256: return launch_bp_t<256>(g, a, B, s);
    case 512: return launch_bp_t<512>(g, a, B, s);
    default: return launch_bp_t<1024>(g, a, B, s);
    }
}
