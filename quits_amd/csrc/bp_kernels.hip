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

template <int T, int MAXCD>
__global__ void __launch_bounds__(T, 8) qd_bp_minsum_kernel(BpGraphDev g, DecodeArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float4 *chk = reinterpret_cast<float4 *>(smem + g.off_chk);        // {min1*alpha, min2*alpha, meta, sign bits 0..31}
    uint32_t *cneg_hi = reinterpret_cast<uint32_t *>(smem + g.off_cneg); // sign bits 32.. of wide checks
    float *llr = reinterpret_cast<float *>(smem + g.off_llr);
    uint16_t *bneg = reinterpret_cast<uint16_t *>(smem + g.off_bneg);
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + g.off_out);
    volatile int *misc = reinterpret_cast<volatile int *>(smem + g.off_misc);   // [0..31] OR flags, [32] fail slot
    constexpr int NW = T / 64;

    const int tid = threadIdx.x;
    const int64_t shot = blockIdx.x;
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;

    // ---- load the window syndrome (sliding_window.py:168-169) and reset the state
    int any = 0;
    for (int c = tid; c < g.m; c += T) {
        const uint32_t o = g.chk_orig[c];
        uint32_t s = det[o] & 1u;
        if (upd && (int)o < a.upd_rows) s ^= upd[o] & 1u;
        any |= (int)s;
        const uint32_t meta = 0xFFu | (s << 9) | ((uint32_t)g.chk_deg[c] << 16);     // argmin = none, parity = 0
        chk[c] = make_float4(0.f, 0.f, __uint_as_float(meta), __uint_as_float(0u));
        for (int w = 1; w < g.neg_words; ++w) cneg_hi[(w - 1) * g.m_pad + c] = 0u;
    }
    for (int b = tid; b < g.n; b += T) {
        const float l0 = g.bit_llr0[b];
        llr[b] = l0;
        bneg[b] = (l0 <= 0.f) ? (uint16_t)((1u << g.bit_deg[b]) - 1u) : (uint16_t)0;
    }
    for (int w = tid; w < g.out_words; w += T) outw[w] = 0u;
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
        int unsat = 0;
        for (int c = tid; c < g.m; c += T) {
            const float4 st = chk[c];
            const uint32_t meta = __float_as_uint(st.z);
            const int idx_old = (int)(meta & 0xFFu);
            const uint32_t par_old = (meta >> 8) & 1u, synd = (meta >> 9) & 1u;
            const int deg = (int)((meta >> 16) & 0xFFu);
            uint32_t us = synd, p = synd;
            int idx = 255;
            float a1 = FLT_MAX, a2 = FLT_MAX;
            uint32_t neg0 = 0u;
            for (int k0 = 0; k0 < deg; k0 += 32) {
                const uint32_t negw = (k0 == 0) ? __float_as_uint(st.w) : cneg_hi[((k0 >> 5) - 1) * g.m_pad + c];
                uint32_t neww = 0u;
                const int kend = min(deg - k0, 32);
                for (int kk = 0; kk < kend; ++kk) {
                    const int k = k0 + kk;
                    const int j = g.chk_adj[k * g.m_pad + c];
                    const float L = llr[j];
                    us ^= (L <= 0.f) ? 1u : 0u;
                    const float mag = (k == idx_old) ? st.y : st.x;
                    const float prev = ((par_old ^ (negw >> kk)) & 1u) ? -mag : mag;
                    const float bm = L - prev;                       // bit->check message, "total minus own"
                    const uint32_t ng = (bm <= 0.f) ? 1u : 0u;      // bp.hpp: a message <= 0 counts as negative
                    const float ab = fabsf(bm);
                    neww |= ng << kk;
                    p ^= ng;
                    if (ab < a1) { a2 = a1; a1 = ab; idx = k; }
                    else if (ab < a2) a2 = ab;
                }
                if (k0 == 0) neg0 = neww;
                else cneg_hi[((k0 >> 5) - 1) * g.m_pad + c] = neww;
            }
            unsat |= (int)us;
            const uint32_t nmeta = (uint32_t)idx | (p << 8) | (synd << 9) | ((uint32_t)deg << 16);
            chk[c] = make_float4(a1 * alpha, a2 * alpha, __uint_as_float(nmeta), __uint_as_float(neg0));
        }
        const int anyun = qd_block_or(unsat, misc, NW, phase);
        phase ^= 1;
        if (t >= 1 && !anyun) { converged = 1; break; }
        if (t == a.max_iter) break;
        // ---- bit pass t+1: posterior = prior + sum of check->bit messages, in ascending detector order
        for (int b = tid; b < g.n; b += T) {
            const int d = g.bit_deg[b];
            float acc = g.bit_llr0[b];
            const uint32_t bn = bneg[b];
            float vals[MAXCD];
#pragma unroll
            for (int q = 0; q < MAXCD; ++q) {
                vals[q] = 0.f;
                if (q < d) {
                    const uint32_t aj = g.bit_adj[q * g.n_pad + b];
                    const float4 st = chk[aj & 0xFFFFu];
                    const uint32_t meta = __float_as_uint(st.z);
                    const float mag = ((aj >> 16) == (meta & 0xFFu)) ? st.y : st.x;
                    const float v = (((meta >> 8) ^ (bn >> q)) & 1u) ? -mag : mag;
                    vals[q] = v;
                    acc += v;
                }
            }
            llr[b] = acc;
            uint32_t nb = 0u;
#pragma unroll
            for (int q = 0; q < MAXCD; ++q)
                if (q < d) nb |= (((acc - vals[q]) <= 0.f) ? 1u : 0u) << q;
            bneg[b] = (uint16_t)nb;
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
        float *dst = a.llr_ws + (int64_t)slot * g.n_pad;
        for (int b = tid; b < g.n; b += T) dst[b] = llr[b];
        if (tid == 0) a.fail_list[slot] = (int32_t)shot;
    }
    if (tid == 0) a.status[shot] = t | (converged << 16);
}

// ---- launch wrappers -------------------------------------------------------------------------------------------------
template <int T>
static hipError_t launch_bp_t(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    if (g.max_cdeg <= 8) {
        auto k = qd_bp_minsum_kernel<T, 8>;
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(T), g.lds_bytes, s, g, a);
    } else {
        auto k = qd_bp_minsum_kernel<T, QD_MAX_COL_DEG>;
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(T), g.lds_bytes, s, g, a);
    }
    return hipGetLastError();
}

hipError_t qd_launch_bp(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    switch (g.threads) {
    case 256: return launch_bp_t<256>(g, a, B, s);
    case 512: return launch_bp_t<512>(g, a, B, s);
    default: return launch_bp_t<1024>(g, a, B, s);
    }
}
