// bp_scatter.hip -- flooding min-sum belief propagation on an LLR grid, scatter form: one workgroup per shot, one lane per check,
// the check's message state in that lane's REGISTERS, the faults' posteriors as 32-bit integers (grid units) in LDS.
//
// Replaces ldpc.BpOsdDecoder.decode -> BpDecoder::bp_decode_parallel (MINIMUM_SUM, ms_scaling_factor 1) as the reference calls it at
// quits/decoder/sliding_window.py:171,182, for the windows ScatGraphDev::ok admits; returns exactly what qd_bp_minsum_kernel
// (bp_kernels.hip) returns -- hard decisions, iteration counts, status words, posteriors of the shots that go to OSD.  Oracle:
// oracle/bp_core.inc bp_parallel_edge in double on the same grid.
//
// Why a second kernel.  bp_kernels.hip spends 8.3 vector instructions and one 16-byte LDS gather per edge in its bit pass (a fault
// rebuilding every check's message from the check's packed state).  On the grid every message is an integer multiple of 2^-k, so a
// posterior is an INTEGER sum -- and integer LDS atomics run at the rate of a 4-byte gather (ds_add_u32: 2.2 ns per wavefront
// instruction per CU, scattered addresses; ds_add_f32: 80 ns, profiles/r03z_lds_atomic_rates.txt).  So the check that has just
// computed (min1, min2, argmin, signs) adds +-min1 to each of its faults' accumulators itself (one sign mask, one xor, one subtract,
// one ds_add_u32 per edge), corrects the argmin fault by +-(min2 - min1), and the bit pass is gone.  The check state never leaves the
// lane, so the LDS holds two posterior buffers (read one, accumulate into the other) and still two shots per CU at the headline
// window (2 x 38 KB + 2 KB each).
//
// One iteration = [gather pass over L(t): new minima and signs, and the syndrome of L(t)'s hard decision] barrier [converged? |
// scatter pass into the other buffer | the buffer just read goes back to the priors] barrier.
// An accumulator holds L - 1, so that (L <= 0) is its sign bit: a check gets the parity of its faults' hard decisions with one
// XOR per edge, and (bit->check message <= 0) is the sign bit of the difference it computes anyway.
//
// Exactness.  Integer accumulation cannot round; what must not round are the float operations of the gather
// pass, on integer-valued floats: exact below 2^24.  Every |posterior partial sum| and |bit->check message| of fault j is at most
// S_j = |prior_j| + sum_i |c2b_ij| <= max|prior| + max_cdeg * max_i min2_i, so max over checks and iterations of min2 <
// (2^23 - max|prior|) / max_cdeg - 1 (ScatArgs::m2_limit; one v_max per CHECK per iteration) certifies the run.  That bound is
// looser than qd_bp_minsum_kernel's per-fault one, which the oracle restates: a shot it cannot certify is parked and decoded again
// by qd_bp_minsum_kernel (recheck pass), which then decides about the coarse grid as before.  A certified shot is exact in both
// kernels, hence identical.
#include "qd_internal.h"
#include "../../include/quits_amd.h"
#include <float.h>

#include "bp_scatter_edge.h"

template <int T, int MW>
__global__ void __launch_bounds__(T, MW) qd_bp_scatter_kernel(BpGraphDev g, ScatGraphDev sg, DecodeArgs a, ScatArgs x)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    if (lds_base != 0u) __builtin_trap();   // no static LDS in this kernel: byte offsets into smem are LDS addresses
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + sg.off_out);
    int *misc = reinterpret_cast<int *>(smem + sg.off_misc);             // [0..31] qd_block_or, [32..47] convergence flags, [48] fail slot
    constexpr int NW = T / 64;

    const int tid = threadIdx.x;
    const int c = tid;                                   // my check slot
    const bool active = c < g.m;
    const int64_t shot = blockIdx.x;
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    const int m_pad = g.m_pad, n_pad = g.n_pad;

    // ---- the window syndrome (sliding_window.py:168-169), the two buffers at the priors (minus one)
    int any = 0;
    uint32_t synd = 0u;
    int dc = 0;
    if (tid < 64) misc[tid] = 0;
    if (active) {
        const uint32_t o = g.chk_orig[c];
        synd = det[o] & 1u;
        if (upd && (int)o < a.upd_rows) synd ^= upd[o] & 1u;
        any = (int)synd;
        dc = sg.chk_deg[c];
    }
    {
        int32_t *bA = reinterpret_cast<int32_t *>(smem + sg.offA);
        for (int b = tid; b < sg.nslots; b += T) bA[b] = x.prior_g[b];                 // (unused slots and the trash slots of the short rows hold 0 and stay 0)
    }
    for (int w = tid; w < g.out_words; w += T) outw[w] = 0u;
    __syncthreads();
    any = qd_block_or(any, misc, NW, 0);
    if (!any) {   // bposd_decoder.pyx: an all-zero syndrome returns the zero vector without running BP
        for (int w = tid; w < g.out_words; w += T) a.err_bits[shot * g.out_words + w] = 0u;
        if (tid == 0) a.status[shot] = (1 << 16) | (1 << 19) | a.status_or;
        return;
    }

    const int dw = (int)sg.deg_w[__builtin_amdgcn_readfirstlane(c) >> 6];     // scalar
    const int trip = dw & 0xFF, wmax = (dw >> 8) & 0xFF, wmin4 = ((dw >> 16) & 0xFF) & ~3;
    // check state, in registers: what this check SENT in the last scatter pass -- the two minima, the position of the argmin edge,
    // the outgoing signs (edge k of a word of kend edges at bit kend - 1 - k, as in bp_kernels.hip) -- and the largest second
    // minimum so far
    float s1 = 0.f, s2 = 0.f, mx2 = 0.f;
    uint32_t kold = 0xFFFFFFFFu, o0 = 0u, o1 = 0u;
    const uint32_t cur = (uint32_t)sg.offA;
    // offsets through buffer loads: the per-lane offset c * 16 never changes, the group row is a scalar offset -- no vector address
    // arithmetic per group
    const __amdgpu_buffer_rsrc_t adj_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sg.adjA, 0, (g.max_rdeg_pad / 4 + 2) * m_pad * 16, 0x00020000);
    const int adj_voff = c * 16, adj_row = m_pad * 16;
#define QS_ADJ(row_) qs_as_uint4(__builtin_amdgcn_raw_buffer_load_b128(adj_rsrc, adj_voff, (row_) * adj_row, 0))
    const uint4 g0 = QS_ADJ(0);                                                 // the first four offsets of my walk: never reloaded
    int t = 0, converged = 0;
    for (;;) {
        // ---- gather pass t+1 over L(t); the parity of the hard decisions it meets is the convergence test of iteration t
        bool us = false;
        float a1 = FLT_MAX, a2 = FLT_MAX;
        uint32_t kst = 0u, q0 = 0u, q1 = 0u;
#ifndef QS_ABL_NOGATHER
        if (active) {
            uint32_t neg0 = 0u, neg1 = 0u, hp = 0u, hpa = 0u;
            for (int k0 = 0; k0 < trip; k0 += 32) {
                const uint32_t sgnw = k0 ? o1 : o0;
                uint32_t neww = 0u, ltw = 0u;
                const int kend = min(trip - k0, 32);                  // multiple of 4
                const int kplain = min(max(wmin4 - k0, 0), kend);     // groups every lane of the wavefront has in full
                const int row0 = k0 >> 2;
                uint4 nx = k0 ? QS_ADJ(row0) : g0;
                int kk = 0;
                {
                    // two groups per trip, two register sets: the next group's offsets are requested while this one is worked on,
                    // without copying them
                    uint4 eb;
#pragma unroll 1
                    for (; kk + 8 <= kplain; kk += 8) {
                        eb = QS_ADJ(row0 + (kk >> 2) + 1);            // (the table has spare group rows)
                        {
                            const int sb = kend - 1 - kk, k = k0 + kk;
                            QS_EDGE_H(nx.x, k, sb, QS_NOFIX, QS_HPA) QS_EDGE_H(nx.y, k + 1, sb - 1, QS_NOFIX, QS_HPB)
                            QS_EDGE_H(nx.z, k + 2, sb - 2, QS_NOFIX, QS_HPA) QS_EDGE_H(nx.w, k + 3, sb - 3, QS_NOFIX, QS_HPB)
                        }
                        nx = QS_ADJ(row0 + (kk >> 2) + 2);
                        {
                            const int sb = kend - 5 - kk, k = k0 + kk + 4;
                            QS_EDGE_H(eb.x, k, sb, QS_NOFIX, QS_HPA) QS_EDGE_H(eb.y, k + 1, sb - 1, QS_NOFIX, QS_HPB)
                            QS_EDGE_H(eb.z, k + 2, sb - 2, QS_NOFIX, QS_HPA) QS_EDGE_H(eb.w, k + 3, sb - 3, QS_NOFIX, QS_HPB)
                        }
                    }
                }
#pragma unroll 1
                for (; kk < kplain; kk += 4) {
                    const uint4 e4 = nx;
                    nx = QS_ADJ(row0 + (kk >> 2) + 1);
                    const int sb = kend - 1 - kk, k = k0 + kk;
                    QS_EDGE_H(e4.x, k, sb, QS_NOFIX, QS_HPA)
                    QS_EDGE_H(e4.y, k + 1, sb - 1, QS_NOFIX, QS_HPB)
                    QS_EDGE_H(e4.z, k + 2, sb - 2, QS_NOFIX, QS_HPA)
                    QS_EDGE_H(e4.w, k + 3, sb - 3, QS_NOFIX, QS_HPB)
                }
#pragma unroll 1
                for (; kk < kend; kk += 4) {
                    const uint4 e4 = nx;
                    nx = QS_ADJ(row0 + (kk >> 2) + 1);
                    const int sb = kend - 1 - kk, k = k0 + kk;
                    QS_EDGE(e4.x, k, sb, QS_TAILFIX)                  // (k < wmax: a group starts below the largest degree)
                    if (k + 1 < wmax) QS_EDGE(e4.y, k + 1, sb - 1, QS_TAILFIX) else { neww <<= 1; ltw <<= 1; }
                    if (k + 2 < wmax) QS_EDGE(e4.z, k + 2, sb - 2, QS_TAILFIX) else { neww <<= 1; ltw <<= 1; }
                    if (k + 3 < wmax) QS_EDGE(e4.w, k + 3, sb - 3, QS_TAILFIX) else { neww <<= 1; ltw <<= 1; }
                }
                if (k0 == 0) neg0 = neww; else neg1 = neww;
                if (ltw) kst = (uint32_t)(k0 + kend - 1 - (int)__builtin_ctz(ltw));   // a later word's improvement overrides an earlier one's
            }
            // outgoing sign on edge k = syndrome ^ (parity of all incoming signs) ^ incoming sign k
            const uint32_t flip = 0u - ((synd ^ (uint32_t)__popc(neg0 ^ neg1)) & 1u);
            q0 = neg0 ^ flip; q1 = neg1 ^ flip;
            mx2 = fmaxf(mx2, a2);
            us = ((synd ^ (hp >> 31)) & 1u) != 0u;
        }
#endif
        {
            const unsigned long long bal = __ballot(us);
            if ((tid & 63) == 0) misc[32 + (tid >> 6)] = (bal != 0ull);
        }
        __syncthreads();
        int anyun = 0;
        {
            const int4 *f4 = reinterpret_cast<const int4 *>(misc + 32);
            for (int w = 0; w < (NW + 3) / 4; ++w) {
                const int4 v = f4[w];
                anyun |= v.x | v.y | v.z | v.w;
            }
        }
        if (t >= 1 && !anyun) { converged = 1; break; }
        if (t == a.max_iter) break;
        // ---- scatter pass, in place (every gather of this iteration is done): each edge's accumulator moves by (new message) -
        // (message sent last time), so L(t+1) = prior + sum of the new messages without a second buffer and without a reset
        __builtin_amdgcn_s_setprio(QS_PRIO);
#ifndef QS_ABL_NOSCAT
        if (active) {
            const int n1i = (int)a1, s1i = (int)s1;
            const int pdif = n1i - s1i, pxq = pdif ^ (n1i + s1i);
            const uint32_t ko = kold == 0xFFFFFFFFu ? 0u : kold;     // (no message sent yet: s1 = s2 = 0, any edge will do)
            const uint32_t fixn_off = __builtin_amdgcn_raw_buffer_load_b32(adj_rsrc, (int)((kst >> 2) * (uint32_t)adj_row + (kst & 3u) * 4u) + adj_voff, 0, 0);
            const uint32_t fixo_off = __builtin_amdgcn_raw_buffer_load_b32(adj_rsrc, (int)((ko >> 2) * (uint32_t)adj_row + (ko & 3u) * 4u) + adj_voff, 0, 0);
            // groups of four edges, two in flight: a group's offsets are requested while the previous group is added (a group is
            // a few dozen cycles of work: a load per group on the critical path made this pass latency-bound); the first group's
            // offsets stay in registers for the whole shot
            const int ng = trip >> 2, kend0 = min(trip, 32);
            uint32_t own = q0 << (32 - kend0), xw = (q0 ^ o0) << (32 - kend0);
            uint4 ea = g0, eb;
#define QS_GROUP(e4, gi_)                                                                                                            \
            {                                                                                                                        \
                const int k = (gi_) * 4;                                                                                             \
                if (k + 4 <= wmin4) {                                                                                                \
                    QS_SCAT(e4.x, 0, 31, QS_NOFIX) QS_SCAT(e4.y, 0, 30, QS_NOFIX) QS_SCAT(e4.z, 0, 29, QS_NOFIX) QS_SCAT(e4.w, 0, 28, QS_NOFIX) \
                } else {                                                                                                             \
                    QS_SCAT(e4.x, k, 31, QS_TAILZERO)                                                                                \
                    if (k + 1 < wmax) QS_SCAT(e4.y, k + 1, 30, QS_TAILZERO)                                                          \
                    if (k + 2 < wmax) QS_SCAT(e4.z, k + 2, 29, QS_TAILZERO)                                                          \
                    if (k + 3 < wmax) QS_SCAT(e4.w, k + 3, 28, QS_TAILZERO)                                                          \
                }                                                                                                                    \
                own <<= 4; xw <<= 4;                                                                                                 \
            }
            const int ng0 = min(ng, 8);                               // groups of the first sign word
#pragma unroll 1
            for (int gi = 0; gi < ng0; gi += 2) {
                eb = QS_ADJ(gi + 1);                                  // (the table has two spare group rows)
                QS_GROUP(ea, gi)
                ea = QS_ADJ(gi + 2);
                if (gi + 1 < ng0) QS_GROUP(eb, gi + 1)
            }
            if (ng > 8) {                                             // ... and of the second: `ea` already holds group 8
                own = q1 << (64 - trip); xw = (q1 ^ o1) << (64 - trip);
#pragma unroll 1
                for (int gi = 8; gi < ng; gi += 2) {
                    eb = QS_ADJ(gi + 1);
                    QS_GROUP(ea, gi)
                    ea = QS_ADJ(gi + 2);
                    if (gi + 1 < ng) QS_GROUP(eb, gi + 1)
                }
            }
#undef QS_GROUP
            // the argmin edges carry min2, not min1: the new one gains +-(min2 - min1), the old one gives its own back
            {
                const int kw = (int)(kst >> 5), kendw = min(trip - 32 * kw, 32);
                const uint32_t sg_ = ((kw ? q1 : q0) >> (kendw - 1 - (int)(kst & 31u))) & 1u;
                const int dlt = (int)a2 - n1i;
                (void)__hip_atomic_fetch_add(QS_LDS(fixn_off), sg_ ? -dlt : dlt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            {
                const int kw = (int)(ko >> 5), kendw = min(trip - 32 * kw, 32);
                const uint32_t sg_ = ((kw ? o1 : o0) >> (kendw - 1 - (int)(ko & 31u))) & 1u;
                const int dlt = (int)s2 - s1i;
                (void)__hip_atomic_fetch_add(QS_LDS(fixo_off), sg_ ? dlt : -dlt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#endif
        s1 = a1; s2 = a2; kold = kst; o0 = q0; o1 = q1;
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        ++t;
    }
    // L(t) - 1 is in the buffer: the scatter pass of the last iteration did not run

    // ---- did the bound hold?
    {
        const int tripped = qd_block_or((active && !(mx2 < x.m2_limit)) ? 1 : 0, misc, NW, 0);
        if (tripped) {
            if (tid == 0) {
                const int at = atomicAdd(x.recheck_count, 1);
                if (at < x.recheck_cap) x.recheck_list[at] = (int32_t)shot;
            }
            return;
        }
    }
    // ---- hard decision, packed by fault index
    for (int b = tid; b < sg.nslots; b += T)
        if (*QS_LDS(cur + 4u * (uint32_t)b) < 0) {
            const uint32_t j = sg.slot_fault[b];
            if (j != 0xFFFFFFFFu) atomicOr(&outw[j >> 5], 1u << (j & 31u));      // (unused and trash slots stay at 0 today; never index LDS with their marker)
        }
    if (!converged && a.want_llr && tid == 0) misc[48] = atomicAdd(a.fail_count, 1);
    __syncthreads();
    for (int w = tid; w < g.out_words; w += T) a.err_bits[shot * g.out_words + w] = outw[w];
    if (!converged && a.want_llr) {
        const int slot = misc[48];
        float *dst = a.llr_ws + (int64_t)slot * n_pad;
        // (rows of the OSD workspace are in the gather kernel's bit-slot order: walk THAT order, so that the stores are whole lines)
        for (int k1 = tid; k1 < g.n; k1 += T) dst[k1] = (float)(*QS_LDS(cur + 4u * sg.k1_slot[k1]) + 1) * x.grid_inv;
        if (tid == 0) a.fail_list[slot] = (int32_t)shot;
    }
    if (tid == 0) a.status[shot] = t | (converged << 16) | a.status_or;
}

template <int T, int MW>
static hipError_t launch_scatter_t(const BpGraphDev &g, const ScatGraphDev &sg, const DecodeArgs &a, const ScatArgs &x, int64_t B, hipStream_t s)
{
    auto k = qd_bp_scatter_kernel<T, MW>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, sg.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(T), sg.lds_bytes, s, g, sg, a, x);
    return hipGetLastError();
}

hipError_t qd_launch_bp_scatter(const BpGraphDev &g, const ScatGraphDev &sg, const DecodeArgs &a, const ScatArgs &x, int64_t B, hipStream_t s)
{
    switch (g.threads) {
    case 256: return launch_scatter_t<256, 8>(g, sg, a, x, B, s);
    case 512: return launch_scatter_t<512, 8>(g, sg, a, x, B, s);
    default: return launch_scatter_t<1024, 8>(g, sg, a, x, B, s);
    }
}
