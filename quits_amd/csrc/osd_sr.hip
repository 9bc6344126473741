// osd_sr.hip -- OSD-0 for the shots BP could not finish, with SIMULTANEOUS pivots (round 4).
//
// Replaces ldpc.BpOsdDecoder.decode -> OsdDecoder::decode (osd.hpp) with osd_method = OSD_0 / osd_order = 0, as the reference
// reaches it through quits/decoder/sliding_window.py:171,182.  CPU restatement: oracle/qd_oracle.c (oq_osd_column_order +
// elim_run + oq_osd0); identical corrections, pivot counts and flags bit for bit (tests/test_gpu_parity.py).
//
// What OSD-0 returns depends on the column order alone: the pivot columns are the lexicographically first independent set S of
// that order, and e_S = H_S^-1 s is unique.  WHICH row a pivot column is eliminated on, and in which order, is free.  qd_osd0_reg_kernel
// (osd_kernels.hip) takes one pivot per barrier round -- lowest column, lowest row -- and a shot is a chain of ~100 (p = 3e-3) to
// ~470 (p = 6e-3) such rounds.  Here a round takes MANY pivots:
//
//   * the 64 columns of a batch are transformed by the pivots found so far into one 64-bit word per row (T-form, as before);
//   * the LOWEST column c of the word of an unpivoted row p is certainly a pivot column: the reduced row p is a functional that
//     is 1 on c and 0 on every column before it (earlier columns of the batch: bits below the lowest are 0; pivot columns: cleared
//     from unpivoted rows; dependent columns: combinations of pivot columns), so c is not in the span of its predecessors.  Every
//     live row nominates that column (one LDS atomicMin: per column the sparsest, then lowest, nominating row is picked);
//   * pivots (c_i, p_i) whose rows hold no other nominated column commute -- no p_i is touched by another pivot of the round, and
//     bit c_j of any row is the same before and after pivot i -- so they are all applied in the same round: the picked rows publish
//     their batch word, Q words and syndrome bit, and every row adds the published rows of the picked columns its word holds.
//     The highest nominated column always qualifies, so every round makes progress; a column that no live row nominates any more
//     and that was not pivoted is zero on the unpivoted rows, i.e. dependent.  Hence the same S as the sequential elimination,
//     in ~4 rounds per batch instead of 40-64;
//   * early stop as before (syndrome zero on every unpivoted row).  A round may pivot columns BEYOND the one that resolved the
//     syndrome; their coefficients are 0.  The reported pivot count is the sequential one: the pivots up to the last column
//     with coefficient 1 -- the columns of that batch below it are drained (classified) before the shot ends.
//
// A syndrome outside the column space has no unique answer: the oracle defines it by the lowest-row rule, so such shots (never
// produced by a detector error model) are handed to qd_osd0_reg_kernel through the hard list.
//
// State: a thread owns RPT rows and keeps their batch word, syndrome bit, pivot flag and the first KWR Q planes in REGISTERS; there
// is no LDS mirror (pivot rows are published per round: 64 x (KWR + 1) words), so a shot needs ~26 KB of LDS at the headline
// window.  Q planes beyond KWR * 64 pivots live in an L2-resident spill, addressed by the owner of the row.
// Registers: 512 threads x 2 rows at a 128-register budget (two workgroups per CU) -- NO scratch.  At 80 registers (three per CU)
// the kernel spilled ~50 vector registers next to ~100 scalar registers the compiler parks in vector-register lanes; spill-heavy
// builds of this and of the column kernel faulted on the GPU until scalar spills were sent to memory instead, which cost 20 % and
// 0.8 GB of scratch writes per launch.  Same box, 65 536 headline shots / p = 6e-3: 80 registers + scalar spills to memory 5.05 /
// 66.4 ms, 128 registers 4.40 / 72 ms (profiles/r04_osd_sr_register_budget_ab.txt); tests/test_api.py holds every instantiation
// to ScratchSize 0.
#include "osd_shared.h"
#include <cstdlib>
#include <algorithm>

#ifndef QD_SR_TIER_GUESS
#define QD_SR_TIER_GUESS 1       // 0: every first tier through the radix selection (A/B)
#endif
struct OsdSrArgs {
    int m, n, m_pad, n_pad, mw, out_words, upd_rows, ell_log2;
    int o_tb, o_rowpiv, o_prow, o_pcol, o_ppos, o_nz, o_cand, o_stq, o_stsp, o_bcols, o_red, o_out, o_order;
    int tier_first;
    const uint16_t *csc_ell;    // [n][1 << ell_log2] rows of a fault, ascending, 0xFFFF beyond its weight
    const uint32_t *bit_orig;   // [n_pad] bit slot -> fault
    const uint8_t *det, *upd;
    int64_t det_stride, det_offset, upd_stride;
    const float *llr_ws;
    const int32_t *fail_list, *fail_count;
    uint64_t *q_spill;          // [blocks][mw - KWR][m_pad] Q planes beyond the registers
    int32_t *hard_list, *hard_count;
    uint32_t *err_bits;
    int32_t *status;
    unsigned long long *dbg;
};

template <int T, int RPT, int KWR>
__global__ void __launch_bounds__(T, QD_SR_WPS_OF(RPT)) qd_osd0_sr_kernel(OsdSrArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
#if QD_SR_PRIO
    __builtin_amdgcn_s_setprio(QD_SR_PRIO);              // (see QD_SR_PRIO in qd_internal.h)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NW = T / 64;
    static_assert(NW <= 16 && T >= 256, "key / flag records hold 16 wavefronts; the tier drawing wants >= 256 threads");
    uint64_t *tb = reinterpret_cast<uint64_t *>(smem + a.o_tb);          // [m_pad] batch words by row; the tier buffer between batches
    uint64_t *sortbuf = tb;
    int16_t *rowpiv = reinterpret_cast<int16_t *>(smem + a.o_rowpiv);    // [m_pad] row -> pivot order or -1
    uint16_t *prow = reinterpret_cast<uint16_t *>(smem + a.o_prow);      // [m_pad] pivot order -> row
    uint32_t *pcol = reinterpret_cast<uint32_t *>(smem + a.o_pcol);      // [m_pad] pivot order -> fault
    uint16_t *ppos = reinterpret_cast<uint16_t *>(smem + a.o_ppos);      // [m_pad] pivot order -> batch number << 6 | column in batch
    uint64_t *nz = reinterpret_cast<uint64_t *>(smem + a.o_nz);          // [mw] pivots whose row meets a column of the batch
    uint32_t *cand = reinterpret_cast<uint32_t *>(smem + a.o_cand);      // [3][80] per round (rotating): [0..63] the row picked for a column (bits << 16 | row), [64] residual flag
    uint64_t *stq = reinterpret_cast<uint64_t *>(smem + a.o_stq);        // [KWR + 1][64] published rows, by column: Q words, [KWR]: batch word
    uint32_t *stsp = reinterpret_cast<uint32_t *>(smem + a.o_stsp);      // [64] ... and syndrome bit
    uint32_t *bcols = reinterpret_cast<uint32_t *>(smem + a.o_bcols);    // [64] faults of the batch
    uint32_t *red = reinterpret_cast<uint32_t *>(smem + a.o_red);        // [72] drain maximum; [80] gather counter; [96..223] block sums
    uint32_t *sumbuf = red + 96;
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + a.o_out);
    uint16_t *order = reinterpret_cast<uint16_t *>(smem + a.o_order);    // [QD_OSD_TIER]
    const int m = a.m, m_pad = a.m_pad, mw = a.mw, dlog = a.ell_log2;
    uint64_t *qglb = a.q_spill + (size_t)blockIdx.x * (size_t)(mw > KWR ? mw - KWR : 0) * m_pad;
    const int nfail = *a.fail_count;
    if (tid == 0) red[81] = 0u;                             // red[81]: carried from shot to shot -- where the last first tier was cut, nudged towards 70 % fill
                                                            // (in LDS, not in a register: the kernel sits at its register budget; the barriers of a shot's set-up order the accesses)
    for (int item = blockIdx.x; item < nfail; item += gridDim.x) {
        const int slot = item;
        const int64_t shot = a.fail_list[slot];
        const float *llr = a.llr_ws + (int64_t)slot * a.n_pad;
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
#ifdef QD_OSD_TIMING
        unsigned long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long tick_ = wall_clock64();
#endif
        uint64_t my_tb[RPT], my_q[RPT][KWR];
        uint32_t my_sp = 0, my_piv = 0;                  // bit i: row tid + i * T
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = tid + i * T;
            uint32_t sbit = 0;
            if (r < m) {
                sbit = det[r] & 1u;
                if (upd && r < a.upd_rows) sbit ^= upd[r] & 1u;
            }
            my_sp |= sbit << i;
            my_tb[i] = 0ull;
#pragma unroll
            for (int w = 0; w < KWR; ++w) my_q[i][w] = 0ull;
            if (r < m_pad) rowpiv[r] = -1;
        }
        for (int w = tid; w < a.out_words; w += T) outw[w] = 0u;
        if (tid < 240) cand[tid] = (tid % 80) < 64 ? QD_NOKEY : 0u;
        if (tid == 0) red[72] = 0u;
        __syncthreads();
        QD_TICK(10)

        int npiv = 0, done = 0, draining = 0, cmaxp1 = 64, sphase = 0, rd = 0, bseq = 0, ntier = 0;
        uint32_t my_bp = 0;                              // rows pivoted in the current batch
        uint32_t lo_key = 0, lo_idx = 0;                 // every column with (key, fault index) < (lo_key, lo_idx) has been consumed
        while (!done) {
            TierState ts{lo_key, lo_idx, sphase, 0, ntier == 0 ? a.tier_first : QD_OSD_TIER};
            const bool first_tier = ntier == 0;
            if (first_tier) ts.guess = red[81];             // (TierState::guess)
            ++ntier;
            const int cnt = qd_osd_draw_tier<T, QD_OSD_KPT, OsdSrArgs, 6>(a, llr, sortbuf, order, red, sumbuf, ts);
            lo_key = ts.lo_key; lo_idx = ts.lo_idx; sphase = ts.sphase;
            if (first_tier && QD_SR_TIER_GUESS) {
                // aim the next shot's first tier at 55..85 % of its size: fuller, and a shot with a few more unreliable columns overflows it (one pass
                // wasted, then the radix selection); emptier, and more shots need a second tier
                const uint32_t lim1 = (uint32_t)a.tier_first, step = 1u << 20;      // an eighth of a binade of the monotone key
                uint32_t g = ts.guess;
                if (ts.cut) g = ts.count * 20u > lim1 * 17u ? ts.cut - step : (ts.count * 20u < lim1 * 11u ? ts.cut + step : ts.cut);
                else if (g) g = ts.count > lim1 ? g - step : g;                     // (overflowed and the selection then drew ties by index: rare)
                if (g > 0xFFFFFFFFu - step) g = 0u;
                if (tid == 0) red[81] = g;                  // (read again by the next shot, behind its set-up barriers)
            }
            QD_TICK(0)
            if (ts.exhausted) break;
            for (int base = 0; base < cnt && !done; base += 64, ++bseq) {
                const int nb = min(64, cnt - base);
                // ---- raw batch words: one LDS atomic per entry of the 64 sparse columns; nz = pivots whose row they meet
#pragma unroll
                for (int i = 0; i < RPT; ++i) { const int r = tid + i * T; if (r < m_pad) tb[r] = 0ull; }
                if (tid < mw) nz[tid] = 0ull;
                if (tid < 64) bcols[tid] = tid < nb ? (uint32_t)order[base + tid] : 0xFFFFFFFFu;
                __syncthreads();
                for (int x = tid; x < (64 << dlog); x += T) {
                    const int c = x >> dlog, q = x & ((1 << dlog) - 1);
                    const uint32_t col = bcols[c];
                    if (col != 0xFFFFFFFFu) {
                        const uint32_t r = a.csc_ell[((size_t)col << dlog) + q];
                        if (r != 0xFFFFu) {
                            atomicXor(reinterpret_cast<unsigned long long *>(&tb[r]), 1ull << c);
                            const int k = rowpiv[r];
                            if (k >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&nz[k >> 6]), 1ull << (k & 63));
                        }
                    }
                }
                __syncthreads();
                // ---- transform by the pivots so far, row-wise: x[r] = raw[r] ^ XOR_{k in Q[r], raw[prow[k]] != 0} raw[prow[k]]
                {
                    const int nplanes = (npiv + 63) >> 6;
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * T;
                        uint64_t x = 0ull;
                        if (r < m) {
                            x = tb[r];
#pragma unroll
                            for (int w = 0; w < KWR; ++w)
                                if (w < nplanes) {
                                    uint64_t q = my_q[i][w] & nz[w];
                                    while (q) {
                                        const int k = w * 64 + __builtin_ctzll(q);
                                        q &= q - 1ull;
                                        x ^= tb[prow[k]];
                                    }
                                }
                            for (int w = KWR; w < nplanes; ++w) {
                                uint64_t q = qglb[(size_t)(w - KWR) * m_pad + r] & nz[w];
                                while (q) {
                                    const int k = w * 64 + __builtin_ctzll(q);
                                    q &= q - 1ull;
                                    x ^= tb[prow[k]];
                                }
                            }
                        }
                        my_tb[i] = x;
                    }
                }
                QD_TICK(1)
                // ---- rounds
                const int bpiv0 = npiv;
                my_bp = 0u;
                for (;;) {
                    // A: every live row nominates the lowest column of its word (see the header); per column the sparsest, then lowest, row
                    uint32_t resid = 0u;
                    uint32_t *cnd = cand + rd * 80;                              // [0..63] per column, [64] residual flag
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * T;
                        if (r < m && !((my_piv >> i) & 1u)) {
                            resid |= (my_sp >> i) & 1u;
                            const uint64_t x = my_tb[i];
                            if (x) atomicMin(&cnd[__builtin_ctzll(x)], ((uint32_t)__popcll(x) << 16) | (uint32_t)r);
                        }
                    }
                    if (__ballot(resid != 0u) != 0ull && lane == 0) cnd[64] = 1u;
                    QD_TICK(4)
                    __syncthreads();
                    QD_TICK(5)
                    // B: the nominated rows publish themselves (every lane of every wavefront sees the same candidates)
                    const uint32_t cv = cnd[lane];
                    unsigned long long C = __ballot(cv != QD_NOKEY);
                    const uint32_t anyres = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnd[64]);
                    if (tid < 80) cand[((rd + 2) % 3) * 80 + tid] = tid < 64 ? QD_NOKEY : 0u;      // clean for the round after next
                    rd = (rd + 1) % 3;
                    if (!anyres && !draining) {
                        // the syndrome is in the span of the pivots found.  The sequential elimination would have stopped at the last
                        // column with coefficient 1: classify the columns of this batch below it, nothing beyond
                        if (npiv == bpiv0) { done = 1; break; }
                        uint32_t cm = 0u;
#pragma unroll
                        for (int i = 0; i < RPT; ++i)
                            if (((my_bp >> i) & 1u) && ((my_sp >> i) & 1u)) cm = max(cm, (uint32_t)(ppos[rowpiv[tid + i * T]] & 63u) + 1u);
                        if (cm) atomicMax(&red[72], cm);
                        __syncthreads();
                        cmaxp1 = (int)red[72];
                        draining = 1;
                        const uint64_t keep = cmaxp1 >= 64 ? ~0ull : ((1ull << cmaxp1) - 1ull);
#pragma unroll
                        for (int i = 0; i < RPT; ++i) my_tb[i] &= keep;
                        C &= keep;                                               // (a nominated column below the cut stays the lowest of its row)
                    }
                    if (C == 0ull) {                                             // no live column left in the batch
                        if (draining) done = 1;
                        break;
                    }
                    uint32_t nom = 0u;                                           // my rows that were picked for their column
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * T;
                        const uint64_t x = my_tb[i];
                        if (r < m && !((my_piv >> i) & 1u) && x) {
                            const int c = (int)__builtin_ctzll(x);
                            if ((cnd[c] & 0xFFFFu) == (uint32_t)r) {                 // (a row nominates one column, so the row alone identifies the pick)
#pragma unroll
                                for (int w = 0; w < KWR; ++w) stq[w * 64 + c] = my_q[i][w];
                                stq[KWR * 64 + c] = x;
                                stsp[c] = (my_sp >> i) & 1u;
                                nom |= 1u << i;
                            }
                        }
                    }
                    QD_TICK(6)
                    __syncthreads();
                    // the picked rows that hold no other nominated column are eliminated at once: none of them is touched by another
                    // pivot of the round (the highest nominated column always qualifies, so every round makes progress); one Q plane per round
                    const int K_base = npiv, kw = K_base >> 6;
                    const bool fresh = (K_base & 63) == 0 && kw >= KWR;          // first pivot of a spilled plane: still uninitialised
                    unsigned long long M;
                    {
                        const uint64_t wl = stq[KWR * 64 + lane];                // (stale for columns outside C: masked by the test)
                        M = __ballot(((C >> lane) & 1ull) && (wl & C) == (1ull << lane));
                        const int room = 64 - (K_base & 63);
                        while ((int)__popcll(M) > room) M &= ~(1ull << (63 - __builtin_clzll(M)));
                    }
                    uint32_t newp = 0u;
#pragma unroll
                    for (int i = 0; i < RPT; ++i)
                        if ((nom >> i) & 1u) {
                            const int r = tid + i * T;
                            const int c = (int)__builtin_ctzll(my_tb[i]);
                            if ((M >> c) & 1ull) {
                                const int K = K_base + (int)__popcll(M & ((1ull << c) - 1ull));
                                rowpiv[r] = (int16_t)K; prow[K] = (uint16_t)r; pcol[K] = bcols[c]; ppos[K] = (uint16_t)((bseq << 6) | c);
                                newp |= 1u << i;
                            }
                        }
                    my_piv |= newp; my_bp |= newp;
                    uint64_t hs[RPT];
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * T;
                        hs[i] = (r < m && !((newp >> i) & 1u)) ? (my_tb[i] & M) : 0ull;
                        uint64_t h = hs[i];
                        while (h) {
                            const int c = (int)__builtin_ctzll(h);
                            h &= h - 1ull;
                            const uint64_t kb = 1ull << ((K_base + (int)__popcll(M & ((1ull << c) - 1ull))) & 63);   // the new pivot's own bit (plane kw)
                            my_tb[i] ^= stq[KWR * 64 + c];
#pragma unroll
                            for (int w = 0; w < KWR; ++w)
                                if (w <= kw) my_q[i][w] ^= stq[w * 64 + c] ^ ((w == kw) ? kb : 0ull);
                            my_sp ^= stsp[c] << i;
                        }
                    }
                    if (kw >= KWR) {
                        for (int w = KWR; w <= kw; ++w) {
                            const bool fr = fresh && w == kw;
#pragma unroll
                            for (int i = 0; i < RPT; ++i) {
                                const int r = tid + i * T;
                                if (r >= m) continue;
                                uint64_t h = hs[i], acc = 0ull;
                                while (h) {
                                    const int c = (int)__builtin_ctzll(h);
                                    h &= h - 1ull;
                                    if (!fr) acc ^= qglb[(size_t)(w - KWR) * m_pad + ((cnd[c] & 0xFFFFu))];
                                    if (w == kw) acc ^= 1ull << ((K_base + (int)__popcll(M & ((1ull << c) - 1ull))) & 63);
                                }
                                uint64_t *mine = &qglb[(size_t)(w - KWR) * m_pad + r];
                                if (fr) *mine = acc;
                                else if (hs[i]) *mine ^= acc;
                            }
                        }
                    }
                    npiv = K_base + (int)__popcll(M);
                    QD_TICK(7)
#ifdef QD_OSD_TIMING
                    ++acc_[11];
#endif
                }
                QD_TICK(2)
                __syncthreads();          // the batch words, bcols and nz are recycled by the next batch / the next tier's sort buffer
            }
        }
        // ---- residual left on a non-pivot row <=> syndrome outside the column space
        uint32_t resid = 0u, over = 0u;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = tid + i * T;
            if (r < m && !((my_piv >> i) & 1u)) resid |= (my_sp >> i) & 1u;
            if (draining && ((my_bp >> i) & 1u) && (int)(ppos[rowpiv[r]] & 63u) >= cmaxp1) ++over;    // pivots beyond the sequential stop
        }
        // `over` totals at most 64 (one batch), the residual flags at most T <= 1024: the flags go in the high half so that neither
        // field can carry into -- or wrap inside -- the other (256 residual rows used to read as "consistent")
        const uint32_t tot = qd_block_sum<T>(over | (resid << 16), sumbuf, sphase);
        const int inconsistent = (tot >> 16) != 0u;
        const int npiv_rep = npiv - (int)(tot & 0xFFFFu);
        if (inconsistent) {
            // defined by the lowest-row rule (oracle): the mirrored kernel decodes the shot again
            if (tid == 0) a.hard_list[atomicAdd(a.hard_count, 1)] = slot;
        } else {
            // ---- OSD-0 solution: e[pivot column k] = transformed syndrome at pivot row k
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                if (((my_piv >> i) & 1u) && ((my_sp >> i) & 1u)) {
                    const uint32_t j = pcol[rowpiv[tid + i * T]];
                    atomicOr(&outw[j >> 5], 1u << (j & 31u));
                }
            __syncthreads();
            for (int w = tid; w < a.out_words; w += T) a.err_bits[shot * a.out_words + w] = outw[w];
            if (tid == 0) a.status[shot] = (a.status[shot] & 0xFFFF) | (1 << 17) | (min(npiv_rep, 4095) << 20);
        }
        QD_TICK(3)
#ifdef QD_OSD_TIMING
        if (tid == 0) {
            for (int i = 0; i < 8; ++i) atomicAdd(&a.dbg[i], acc_[i]);
            atomicAdd(&a.dbg[8], 1ull); atomicAdd(&a.dbg[9], (unsigned long long)npiv); atomicAdd(&a.dbg[10], acc_[10]); atomicAdd(&a.dbg[11], acc_[11]); atomicAdd(&a.dbg[12], (unsigned long long)ntier); atomicAdd(&a.dbg[13], (unsigned long long)bseq);
        }
#endif
        __syncthreads();   // LDS is recycled by the next shot
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// LDS layout of the kernel above for a window of m detectors (m_pad rows), out_words words of output; returns the bytes, 0 when
// the kernel does not take the window.  threads / rpt name the instantiation.
int qd_osd_sr_layout(int m, int m_pad, int n, int out_words, int *off13, int *threads, int *rpt)
{
    if (m > 2048 || n > 49152) return 0;              // rows per thread <= 4 at 512 threads; ppos holds 10 bits of batch number
    const int T = m <= 256 ? 256 : (m <= 4 * QD_SR_TSMALL ? QD_SR_TSMALL : 512);     // windows of <= 256 checks: one row per thread on four wavefronts
    *threads = T; *rpt = (m + T - 1) / T;
    auto al = [](int x) { return (x + 15) & ~15; };
    int o = 0;
    off13[0] = o; o += al(std::max(m_pad * 8, QD_OSD_TIER * 8));   // tb / tier buffer
    off13[1] = o; o += al(m_pad * 2);                 // rowpiv
    off13[2] = o; o += al(m_pad * 2);                 // prow
    off13[3] = o; o += al(m_pad * 4);                 // pcol
    off13[4] = o; o += al(m_pad * 2);                 // ppos
    off13[5] = o; o += al(((m + 63) / 64) * 8);       // nz
    off13[6] = o; o += 3 * 80 * 4;                    // cand
    off13[7] = o; o += (QD_SR_KWR_MAX + 1) * 64 * 8;  // stq
    off13[8] = o; o += 64 * 4;                        // stsp
    off13[9] = o; o += 64 * 4;                        // bcols
    off13[10] = o; o += 1024;                         // red
    off13[11] = o; o += al(out_words * 4);            // out
    off13[12] = o; o += al(QD_OSD_TIER * 2);          // order
    return o;
}

// uint64 words of workspace per workgroup: the Q planes beyond the registers
size_t qd_osd_sr_ws_words(int m_pad, int mw, int threads, int rpt)
{
    (void)threads;
    const int kwr = QD_SR_KWR_OF(rpt);
    return (size_t)(mw > kwr ? mw - kwr : 0) * m_pad + 64;
}

hipError_t qd_launch_osd0_sr(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks, hipStream_t s)
{
    OsdSrArgs r{};
    r.m = g.m; r.n = g.n; r.m_pad = g.m_pad; r.n_pad = bg.n_pad; r.mw = g.mw; r.out_words = bg.out_words; r.upd_rows = a.upd_rows;
    r.ell_log2 = g.ell_log2;
    const int *o = g.s_off;
    r.o_tb = o[0]; r.o_rowpiv = o[1]; r.o_prow = o[2]; r.o_pcol = o[3]; r.o_ppos = o[4]; r.o_nz = o[5]; r.o_cand = o[6]; r.o_stq = o[7];
    r.o_stsp = o[8]; r.o_bcols = o[9]; r.o_red = o[10]; r.o_out = o[11]; r.o_order = o[12];
    r.tier_first = QD_OSD_TIER_FIRST;
    if (const char *ev = std::getenv("QD_SR_TIER_FIRST")) { const int v = std::atoi(ev); if (v >= 64 && v <= QD_OSD_TIER) r.tier_first = v; }
    r.csc_ell = g.csc_ell; r.bit_orig = bg.bit_orig;
    r.det = a.det; r.upd = a.upd; r.det_stride = a.det_stride; r.det_offset = a.det_offset; r.upd_stride = a.upd_stride;
    r.llr_ws = a.llr_ws; r.fail_list = a.fail_list; r.fail_count = a.fail_count; r.q_spill = a.q_spill_sr;
    r.hard_list = a.hard_list; r.hard_count = a.hard_count;
    r.err_bits = a.err_bits; r.status = a.status; r.dbg = a.dbg;
    const int lds = g.s_lds_bytes;
#define QD_SR_CASE(TT, RR)                                                                                                        \
    {                                                                                                                             \
        auto k = qd_osd0_sr_kernel<TT, RR, QD_SR_KWR_OF(RR)>;                                                                            \
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                     \
        if (e != hipSuccess) return e;                                                                                            \
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(TT), lds, s, r);                                                       \
        return hipGetLastError();                                                                                                 \
    }
    if (g.s_threads == 256) {
        switch (g.s_rpt) {
        case 1: QD_SR_CASE(256, 1)
        case 2: QD_SR_CASE(256, 2)
        case 3: QD_SR_CASE(256, 3)
        default: QD_SR_CASE(256, 4)
        }
    }
    switch (g.s_rpt) {
    case 1: QD_SR_CASE(512, 1)
    case 2: QD_SR_CASE(512, 2)
    case 3: QD_SR_CASE(512, 3)
    default: QD_SR_CASE(512, 4)
    }
#undef QD_SR_CASE
}
