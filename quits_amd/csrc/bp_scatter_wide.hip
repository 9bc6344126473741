// bp_scatter_wide.hip -- the scatter form of flooding min-sum BP (bp_scatter.hip) with CPL checks per lane and NSW 32-bit sign words
// per check.  The default shape of flooding min-sum on the LLR grid:
//   * windows bp_scatter.hip could take with one check per lane run here on HALF the lanes with two checks each -- the headline window
//     (1008 checks) on 512 lanes: the same 32 wavefronts per CU in four workgroups instead of two, i.e. half as many wavefronts per
//     barrier (BP 50.3 -> 44.2 ms per 65 536 shots); 512-thread windows on 256 lanes, windows of <= 128 checks on 128;
//   * windows it cannot take -- more checks than a workgroup has lanes, or rows of 65..96 faults (three sign words) -- on 512 lanes x 3
//     checks (or 704 / 1024 x 2): the windows of BASELINE configs[4] (QLP [[1020,136]], W = 3: 1350 checks of up to 78 faults, 18 900 faults).
//
// Replaces ldpc.BpOsdDecoder.decode -> BpDecoder::bp_decode_parallel (MINIMUM_SUM, ms_scaling_factor 1) as the reference calls it at
// quits/decoder/sliding_window.py:171,182.  Same arithmetic, same certificate and the same outputs as qd_bp_scatter_kernel and
// qd_bp_minsum_kernel, bit for bit; oracle: oracle/bp_core.inc bp_parallel_edge in double on the same LLR grid.
//
// Why for the QLP windows.  qd_bp_minsum_kernel keeps 16 bytes of packed state per check plus sign words plus a float posterior per
// fault in LDS: 112 KB per QLP shot, ONE workgroup of 1024 threads per CU, four wavefronts per SIMD.  In the scatter form the LDS holds
// one int32 accumulator per fault (76 KB), so TWO shots share a CU (BP 36.8 -> 19.0 ms per 8192-shot window launch).
//
// A lane's checks keep their state in registers (two minima, argmin position, the sign words, twice: what was sent and what the gather
// pass just found).  Check slots are sorted by degree; a wavefront's 64 checks of one round are 64 consecutive slots (sg.wave_map), so
// the trip count stays wave-uniform, and the rounds are dealt out so that the wavefronts of a workgroup walk equally many edges between
// two barriers.
//
// One iteration = [gather pass of check 0 .. CPL-1] barrier [converged? | scatter pass of check 0 .. CPL-1] barrier, in place as in
// bp_scatter.hip (every gather of the iteration precedes every add).
#include "qd_internal.h"
#include "../../include/quits_amd.h"
#include "bp_scatter_edge.h"

// Wavefront priorities.  A wavefront raises its priority for its last gather round (0 -> 2; the scatter pass runs at QS_PRIO = 1 as
// in bp_scatter.hip): the wavefronts closest to the barrier go first, so a workgroup's stragglers are fewer.  Headline BP stage 44.1 ->
// 43.2 ms per 65 536 shots; the other way round (first round high) 43.7, scatter pass at 0 / 2 / 3 no change, alternate wavefronts
// high no change (profiles/r03x_wavefront_priority_ab.txt).
// Scatter pass: the sign / difference bit of an edge is picked with a scalar bit position (v_bfe_i32 with an SGPR offset) instead of shifting
// the two words by 4 per group: half a vector instruction per edge moves to the scalar unit.  BP stage 42.35 -> 42.24 ms per 65 536 headline
// shots (profiles/r05_k1sw_micro_ab.txt); 0 = the shifting form.
#ifndef QSW_SCAT_SGPR_POS
#define QSW_SCAT_SGPR_POS 1
#endif
#ifndef QSW_GPRIO
#define QSW_GPRIO 1
#endif
// QSW_PRIO_BASE: added to every priority this kernel sets (gather rounds 0 / 1 / 2, scatter pass QS_PRIO, 0 in between): above 0 the wavefronts of a
// kernel running beside this one at the default priority (the pipelined driver's post-processing) only issue when these do not
#ifndef QSW_PRIO_BASE
#define QSW_PRIO_BASE 0
#endif

__device__ __forceinline__ uint4 qs_reuse4(uint4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); return v; }   // (QS_ABL_NOADJ: opaque, so that the reads it feeds stay in the loops)

template <int T, int MW, int CPL, int NSW>
__global__ void __launch_bounds__(T, MW) qd_bp_scatter_wide_kernel(BpGraphDev g, ScatGraphDev sg, DecodeArgs a, ScatArgs x)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    if (lds_base != 0u) __builtin_trap();   // no static LDS in this kernel: byte offsets into smem are LDS addresses
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + sg.off_out);
    int *misc = reinterpret_cast<int *>(smem + sg.off_misc);             // [0..31] qd_block_or, [32..47] convergence flags, [48] fail slot
    constexpr int NW = T / 64;
    static_assert(T % 64 == 0 && NW <= 16, "workgroup shape");

    const int tid = threadIdx.x;
    const int64_t shot = blockIdx.x;
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    const int m_pad = g.m_pad, n_pad = g.n_pad;

    // ---- the window syndrome (sliding_window.py:168-169), the accumulators at the priors (minus one)
    int any = 0;
    uint32_t synd[CPL];
    int dcs[CPL];
    bool act[CPL];
    if (tid < 64) misc[tid] = 0;
    int cs[CPL];                                       // my check slot of round j (sg.wave_map: 64 consecutive slots per wavefront and round)
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int sw = __builtin_amdgcn_readfirstlane(sg.wave_map[j * NW + (tid >> 6)]);
        const int c = sw * 64 + (tid & 63);
        cs[j] = c;
        act[j] = sw >= 0 && c < g.m;
        synd[j] = 0u; dcs[j] = 0;
        if (act[j]) {
            const uint32_t o = g.chk_orig[c];
            synd[j] = det[o] & 1u;
            if (upd && (int)o < a.upd_rows) synd[j] ^= upd[o] & 1u;
            any |= (int)synd[j];
            dcs[j] = sg.chk_deg[c];
        }
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

    // per wavefront and round: trip count | largest degree << 8 | smallest << 16 (scalar); a round whose 64 slots lie beyond the window has 0
    int dws[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int wslot = __builtin_amdgcn_readfirstlane(cs[j] >> 6);
        dws[j] = wslot >= 0 ? (int)sg.deg_w[wslot] : 0;
    }
    // check state, in registers: what each check SENT in the last scatter pass (the two minima, the position of the argmin edge, the
    // outgoing signs: edge k of a word of kend edges at bit kend - 1 - k) and what the gather pass of this iteration found
    float S1[CPL], S2[CPL], A1[CPL], A2[CPL], mx2 = 0.f;
    uint32_t KOLD[CPL], KST[CPL], O[CPL][NSW], Q[CPL][NSW];
    uint32_t FOFF[CPL];                                 // where and what the last scatter pass added for its argmin edge (min2 - min1, signed): given back by the next
    int FVAL[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        FOFF[j] = (uint32_t)sg.offA + (uint32_t)(sg.nslots - 32 + (tid & 31)) * 4u; FVAL[j] = 0;      // (a trash slot)
        S1[j] = 0.f; S2[j] = 0.f; A1[j] = FLT_MAX; A2[j] = FLT_MAX; KOLD[j] = 0xFFFFFFFFu; KST[j] = 0u;
#pragma unroll
        for (int w = 0; w < NSW; ++w) { O[j][w] = 0u; Q[j][w] = 0u; }
    }
    const uint32_t cur = (uint32_t)sg.offA;
    const __amdgpu_buffer_rsrc_t adj_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sg.adjA, 0, (g.max_rdeg_pad / 4 + 2) * m_pad * 16, 0x00020000);
    const int adj_row = m_pad * 16;
#define QS_ADJ_LOAD(row_) qs_as_uint4(__builtin_amdgcn_raw_buffer_load_b128(adj_rsrc, adj_voff, (row_) * adj_row, 0))
#ifdef QS_ABL_NOADJ      /* timing experiment only (wrong results): one group of offsets per check and pass, reused for every step -- no adjacency traffic in the loops */
#define QS_ADJ(row_) qs_reuse4(adjc)
#define QS_ABL_ADJC uint4 adjc = QS_ADJ_LOAD(0);
#else
#define QS_ADJ(row_) QS_ADJ_LOAD(row_)
#define QS_ABL_ADJC
#endif
    // The first group of offsets of a pass (round 0, word 0) is requested BEFORE the barrier the pass starts behind: every wavefront of the workgroup
    // would otherwise begin the pass by waiting for the same L2 round trip at the same time (QSW_PREFETCH=0: not).
#ifndef QSW_PREFETCH
#define QSW_PREFETCH 1
#endif
#define QS_ADJ_FIRST qs_as_uint4(__builtin_amdgcn_raw_buffer_load_b128(adj_rsrc, cs[0] * 16, 0, 0))
    uint4 pf = QS_ADJ_FIRST;
    int t = 0, converged = 0;
    for (;;) {
        // ---- gather pass t+1 over L(t); the parity of the hard decisions it meets is the convergence test of iteration t
        bool us = false;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            float a1 = FLT_MAX, a2 = FLT_MAX;
            uint32_t kst = 0u;
#if QSW_GPRIO       /* the later gather rounds of a wavefront run at a higher priority (see QSW_GPRIO above) */
            if (j == 0) __builtin_amdgcn_s_setprio(QSW_PRIO_BASE); else if (j == CPL - 1) __builtin_amdgcn_s_setprio((QSW_PRIO_BASE + 2) & 3); else __builtin_amdgcn_s_setprio((QSW_PRIO_BASE + 1) & 3);
#endif
            if (act[j]) {
                // (the round's loop bounds are re-derived from one scalar every pass: hoisted out of the iteration loop they, and everything computed
                //  from them for CPL rounds x NSW words, outgrow the scalar registers and come back through v_readlane)
                int dwj = dws[j];
                asm volatile("" : "+s"(dwj));
                const int trip = dwj & 0xFF, wmax = (dwj >> 8) & 0xFF, wmin = (dwj >> 16) & 0xFF, wmin4 = wmin & ~3;
                const int adj_voff = cs[j] * 16;
                QS_ABL_ADJC
                const float s1 = S1[j], s2 = S2[j];
                const uint32_t kold = KOLD[j];
                const int dc = dcs[j];
                uint32_t hp = 0u, hpa = 0u, par = 0u;
                uint32_t neg[NSW];
#pragma unroll
                for (int w = 0; w < NSW; ++w) {
                    neg[w] = 0u;
                    const int k0 = 32 * w;
                    if (k0 < trip) {
                        const uint32_t sgnw = O[j][w];
                        uint32_t neww = 0u, ltw = 0u;
                        const int kend = min(trip - k0, 32);                  // multiple of 4
                        const int kplain = min(max(wmin4 - k0, 0), kend);     // groups every lane of the wavefront has in full
                        const int row0 = k0 >> 2;
                        uint4 nx = (QSW_PREFETCH && j == 0 && w == 0) ? pf : QS_ADJ(row0);
                        int kk = 0;
                        {
                            uint4 eb;                                         // two groups per trip on two register sets (bp_scatter.hip)
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
                        // an edge below the smallest degree of the wavefront is real on every lane (no fix), one at or beyond the largest is
                        // nobody's; only in between does a lane have to ask (wave-uniform tests; k < wmax: a group starts below the largest degree)
#define QS_TAIL_EDGE(off, q_)                                                                                     \
                            if (k + (q_) < wmin) QS_EDGE(off, k + (q_), sb - (q_), QS_NOFIX)                      \
                            else if (k + (q_) < wmax) QS_EDGE(off, k + (q_), sb - (q_), QS_TAILFIX)               \
                            else { neww <<= 1; ltw <<= 1; }
#pragma unroll 1
                        for (; kk + 4 < kend; kk += 4) {
                            const uint4 e4 = nx;
                            nx = QS_ADJ(row0 + (kk >> 2) + 1);
                            const int sb = kend - 1 - kk, k = k0 + kk;
                            QS_TAIL_EDGE(e4.x, 0) QS_TAIL_EDGE(e4.y, 1) QS_TAIL_EDGE(e4.z, 2) QS_TAIL_EDGE(e4.w, 3)
                        }
                        if (kk < kend) {              // the word's last group (for rows of 33..36 faults the second word's only one): nothing to request behind it, no copy
                            const int sb = kend - 1 - kk, k = k0 + kk;
                            QS_TAIL_EDGE(nx.x, 0) QS_TAIL_EDGE(nx.y, 1) QS_TAIL_EDGE(nx.z, 2) QS_TAIL_EDGE(nx.w, 3)
                        }
#undef QS_TAIL_EDGE
                        neg[w] = neww;
                        par ^= neww;
                        if (ltw) kst = (uint32_t)(k0 + kend - 1 - (int)__builtin_ctz(ltw));   // a later word's improvement overrides an earlier one's
                    }
                }
                // outgoing sign on edge k = syndrome ^ (parity of all incoming signs) ^ incoming sign k
                const uint32_t flip = 0u - ((synd[j] ^ (uint32_t)__popc(par)) & 1u);
#pragma unroll
                for (int w = 0; w < NSW; ++w) Q[j][w] = neg[w] ^ flip;
                mx2 = fmaxf(mx2, a2);
                us = us || (((synd[j] ^ (hp >> 31)) & 1u) != 0u);
            }
            A1[j] = a1; A2[j] = a2; KST[j] = kst;
        }
#if QSW_GPRIO
        __builtin_amdgcn_s_setprio(QSW_PRIO_BASE);
#endif
        if (QSW_PREFETCH) pf = QS_ADJ_FIRST;              // for the scatter pass behind the barrier
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
#ifndef QS_ABL_FORCE_ITERS       /* timing experiments only: every shot runs max_iter iterations, whatever the (wrong) arithmetic of an ablation build does */
        if (t >= 1 && !anyun) { converged = 1; break; }
#endif
        if (t == a.max_iter) break;
        // ---- scatter pass, in place: each edge's accumulator moves by (new message) - (message sent last time)
        __builtin_amdgcn_s_setprio((QSW_PRIO_BASE + QS_PRIO) & 3);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
#ifdef QSW_STATS
            {
                bool same = A1[j] == S1[j] && A2[j] == S2[j] && (KST[j] == KOLD[j] || A1[j] == A2[j]);
#pragma unroll
                for (int w = 0; w < NSW; ++w) same = same && Q[j][w] == O[j][w];
                const unsigned long long ba = __ballot(act[j]), bs = __ballot(act[j] && same);
                if ((tid & 63) == 0 && ba) {
                    const int bk = min(t / 10, 4);
                    atomicAdd(&a.dbg[0], 1ull); atomicAdd(&a.dbg[2 + bk], 1ull);
                    if (bs == ba) { atomicAdd(&a.dbg[1], 1ull); atomicAdd(&a.dbg[7 + bk], 1ull); }
                    atomicAdd(&a.dbg[12], (unsigned long long)__popcll(bs)); atomicAdd(&a.dbg[13], (unsigned long long)__popcll(ba));
                }
            }
#endif
            if (act[j]) {
                int dwj = dws[j];
                asm volatile("" : "+s"(dwj));
                const int trip = dwj & 0xFF, wmax = (dwj >> 8) & 0xFF, wmin = (dwj >> 16) & 0xFF, wmin4 = wmin & ~3;
                const int adj_voff = cs[j] * 16;
                QS_ABL_ADJC
                const int dc = dcs[j];
                const int n1i = (int)A1[j], s1i = (int)S1[j];
                const int pdif = n1i - s1i, pxq = pdif ^ (n1i + s1i);
                const uint32_t kst = KST[j];
                const uint32_t fixn_off = __builtin_amdgcn_raw_buffer_load_b32(adj_rsrc, (int)((kst >> 2) * (uint32_t)adj_row + (kst & 3u) * 4u) + adj_voff, 0, 0);
                const int ng = trip >> 2;
#if QSW_SCAT_SGPR_POS
#define QS_GROUP_POS(gi_) const int b0_ = kendw - 1 - 4 * ((gi_) - 8 * w);      /* (the words are used as they are: edge k of the word at bit kendw - 1 - k) */
#define QS_GROUP_STEP
#else
#define QS_GROUP_POS(gi_) constexpr int b0_ = 31;
#define QS_GROUP_STEP own <<= 4; xw <<= 4;
#endif
                // a group every lane of the wavefront has in full ...
#define QS_GROUP_PLAIN(e4, gi_)                                                                                                      \
                {                                                                                                                    \
                    QS_GROUP_POS(gi_)                                                                                                \
                    QS_SCAT(e4.x, 0, b0_, QS_NOFIX) QS_SCAT(e4.y, 0, b0_ - 1, QS_NOFIX) QS_SCAT(e4.z, 0, b0_ - 2, QS_NOFIX) QS_SCAT(e4.w, 0, b0_ - 3, QS_NOFIX) \
                    QS_GROUP_STEP                                                                                                    \
                }
                // ... and one that reaches beyond the smallest degree: lanes past their degree add 0 to a trash slot
#define QS_SCAT_TAIL(off, q_)                                                                                                        \
                    if (k + (q_) < wmin) QS_SCAT(off, 0, b0_ - (q_), QS_NOFIX)                                                       \
                    else if (k + (q_) < wmax) QS_SCAT(off, k + (q_), b0_ - (q_), QS_TAILZERO)
#define QS_GROUP_TAIL(e4, gi_)                                                                                                       \
                {                                                                                                                    \
                    const int k = (gi_) * 4;                                                                                         \
                    QS_GROUP_POS(gi_)                                                                                                \
                    QS_SCAT_TAIL(e4.x, 0) QS_SCAT_TAIL(e4.y, 1) QS_SCAT_TAIL(e4.z, 2) QS_SCAT_TAIL(e4.w, 3)                          \
                    QS_GROUP_STEP                                                                                                    \
                }
#pragma unroll
                for (int w = 0; w < NSW; ++w) {
                    if (32 * w < trip) {
                        const int kendw = min(trip - 32 * w, 32);
#if QSW_SCAT_SGPR_POS
                        const uint32_t own = Q[j][w], xw = Q[j][w] ^ O[j][w];
#else
                        uint32_t own = Q[j][w] << (32 - kendw), xw = (Q[j][w] ^ O[j][w]) << (32 - kendw);
#endif
                        const int g1 = min(ng, 8 * w + 8);                    // groups of this word: [8 w, g1)
                        const int gp = min(g1, max(wmin4 >> 2, 8 * w));       // ... of which [8 w, gp) are plain (as in the gather pass: three loops, no test per group)
                        uint4 ea = (QSW_PREFETCH && j == 0 && w == 0) ? pf : QS_ADJ(8 * w), eb;
                        int gi = 8 * w;
#pragma unroll 1
                        for (; gi + 2 <= gp; gi += 2) {                       // two groups per trip on two register sets
                            eb = QS_ADJ(gi + 1);                              // (the table has two spare group rows)
                            QS_GROUP_PLAIN(ea, gi)
                            ea = QS_ADJ(gi + 2);
                            QS_GROUP_PLAIN(eb, gi + 1)
                        }
#pragma unroll 1
                        for (; gi < gp; ++gi) {
                            const uint4 e4 = ea;
                            ea = QS_ADJ(gi + 1);
                            QS_GROUP_PLAIN(e4, gi)
                        }
#pragma unroll 1
                        for (; gi + 1 < g1; ++gi) {
                            const uint4 e4 = ea;
                            ea = QS_ADJ(gi + 1);
                            QS_GROUP_TAIL(e4, gi)
                        }
                        if (gi < g1) QS_GROUP_TAIL(ea, gi)          // the word's last group: nothing to request behind it, no copy
                    }
                }
#undef QS_GROUP_PLAIN
#undef QS_GROUP_TAIL
#undef QS_SCAT_TAIL
#undef QS_GROUP_POS
#undef QS_GROUP_STEP
                // the argmin edges carry min2, not min1: the new one gains +-(min2 - min1), the old one gives its own back
                {
                    const int kw = (int)(kst >> 5), kendw = min(trip - 32 * kw, 32);
                    uint32_t wd = Q[j][0];
#pragma unroll
                    for (int w = 1; w < NSW; ++w) wd = (kw == w) ? Q[j][w] : wd;
                    const uint32_t sg_ = (wd >> (kendw - 1 - (int)(kst & 31u))) & 1u;
                    const int dlt = (int)A2[j] - n1i;
                    const int vn = sg_ ? -dlt : dlt;
                    (void)__hip_atomic_fetch_add(QS_LDS(fixn_off), vn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // ... which is exactly what the last pass added for ITS argmin edge, kept from then (0 at the trash slot before any message was sent)
                    (void)__hip_atomic_fetch_add(QS_LDS(FOFF[j]), -FVAL[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    FOFF[j] = fixn_off; FVAL[j] = vn;
                }
            }
            S1[j] = A1[j]; S2[j] = A2[j]; KOLD[j] = KST[j];
#pragma unroll
            for (int w = 0; w < NSW; ++w) O[j][w] = Q[j][w];
        }
        __builtin_amdgcn_s_setprio(QSW_PRIO_BASE);
        if (QSW_PREFETCH) pf = QS_ADJ_FIRST;              // for the next gather pass
        __syncthreads();
        ++t;
    }
#undef QS_ADJ
#undef QS_ADJ_FIRST
#undef QS_ADJ_LOAD
#undef QS_ABL_ADJC
    // L(t) - 1 is in the buffer: the scatter pass of the last iteration did not run

    // ---- did the bound hold?
    {
        const int tripped = qd_block_or(!(mx2 < x.m2_limit) ? 1 : 0, misc, NW, 0);
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
            const uint32_t jf = sg.slot_fault[b];
            if (jf != 0xFFFFFFFFu) atomicOr(&outw[jf >> 5], 1u << (jf & 31u));   // (unused and trash slots stay at 0 today; never index LDS with their marker)
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

template <int T, int MW, int CPL, int NSW>
static hipError_t launch_scatter_wide_t(const BpGraphDev &g, const ScatGraphDev &sg, const DecodeArgs &a, const ScatArgs &x, int64_t B, hipStream_t s)
{
    auto k = qd_bp_scatter_wide_kernel<T, MW, CPL, NSW>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, sg.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(T), sg.lds_bytes, s, g, sg, a, x);
    return hipGetLastError();
}

// (sg.wide_threads, sg.wide_cpl) is one of the shapes instantiated here (qd_graph_create chooses it); rows of up to 64 faults take two
// sign words, up to 96 three
hipError_t qd_launch_bp_scatter_wide(const BpGraphDev &g, const ScatGraphDev &sg, const DecodeArgs &a, const ScatArgs &x, int64_t B, hipStream_t s)
{
    const bool two_words = g.max_rdeg_pad <= 64;
    switch (sg.wide_threads * 8 + sg.wide_cpl) {
    case 512 * 8 + 2:           // headline-size windows (513..1024 checks) on half the lanes: four workgroups per CU
        return two_words ? launch_scatter_wide_t<512, 8, 2, 2>(g, sg, a, x, B, s) : hipErrorInvalidValue;
    case 256 * 8 + 2:           // 257..512 checks: eight workgroups per CU
        return two_words ? launch_scatter_wide_t<256, 8, 2, 2>(g, sg, a, x, B, s) : hipErrorInvalidValue;
    case 128 * 8 + 2:           // <= 256 checks: sixteen workgroups of two wavefronts per CU (ONE wavefront with four checks per lane, no barrier partner: 7.73 -> 8.99 ms on 216-check windows, profiles/r05_scatter_small_ab.txt)
        return two_words ? launch_scatter_wide_t<128, 8, 2, 2>(g, sg, a, x, B, s) : hipErrorInvalidValue;
    case 704 * 8 + 2:           // 11 wavefronts; two workgroups per CU: <= 6 per SIMD, 80 registers
        return launch_scatter_wide_t<704, 6, 2, 3>(g, sg, a, x, B, s);
    case 1024 * 8 + 2: return launch_scatter_wide_t<1024, 4, 2, 3>(g, sg, a, x, B, s);      // (a 64-register budget -- two workgroups of 16 wavefronts per CU -- measured no faster on the QLP windows and spills with the prefetch registers)
    case 512 * 8 + 3: return launch_scatter_wide_t<512, 4, 3, 3>(g, sg, a, x, B, s);      // QLP windows: 8 wavefronts x 3 rounds, 89 registers, two workgroups per CU
    default: return hipErrorInvalidValue;
    }
}
