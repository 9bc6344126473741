// osd_kernels.hip -- OSD-0 post-processing for the shots whose BP did not converge: one workgroup per shot.
//
// Replaces ldpc.BpOsdDecoder.decode -> OsdDecoder::decode (osd.hpp) with osd_method = OSD_0 / osd_order = 0, as the
// reference reaches it through quits/decoder/sliding_window.py:171,182.  CPU restatement: oracle/qd_oracle.c
// (oq_osd_column_order + elim_run + oq_osd0); identical results bit for bit (tests/test_gpu_parity.py).
//
//   1. column order: ascending posterior LLR, ties by ascending fault index  (ldpc soft_decision_col_sort; std::sort
//      leaves ties open, this build fixes them).  Keys are 64-bit (monotone-float << 32 | index) words, bitonic-sorted
//      in LDS.
//   2. Gaussian elimination over GF(2) in that column order, bit-packed:
//        - rows are only ever modified by adding a pivot row, so the accumulated row transformation is the identity
//          plus columns that belong to pivot rows; row r keeps those as a bit vector Q[r] indexed by pivot ORDER
//          (uint64 word-planes Q[w][r], lane = row -> conflict-free LDS access, the pivot row is a broadcast read);
//        - the image of a (sparse, weight <= 16) column under the transformation is
//              t[r] = [r in column] xor parity(Q[r] & {pivot order of the column's pivoted rows});
//          64 columns are transformed at once into one uint64 per row, then pivots are taken from that word in
//          order: one workgroup min-reduction of (first set column, lowest row) per pivot; dependent columns are
//          skipped for free;
//        - the syndrome rides along as one more bit per row; the loop stops as soon as it is zero on every
//          non-pivot row: the syndrome is then in the span of the pivot columns found so far and, the pivot set being
//          independent, their coefficients are final (the remaining pivots of ldpc's full elimination get 0).
//          At p = 0.003 on the [[144,12,12]] window this is ~100 pivots instead of rank 1002.
//   3. e[pivot column k] = transformed syndrome at pivot row k; everything else 0  (OSD-0).
//
// Two kernels share the elimination:
//   qd_osd0_reg_kernel   draws the order lazily in tiers of 1024 columns (bisection on the key value), keeps each thread's
//                        rows in registers with a write-through LDS mirror, ~80 KB of LDS -> two workgroups per CU.
//   qd_osd0_full_kernel  sorts every column up front, all state in LDS, one workgroup per CU; used when a window has
//                        more than 2048 detectors.  Same results by construction: both consume the same column order.
#include "osd_shared.h"
#include <cstdlib>
#include <algorithm>

struct OsdLds {
    uint64_t *q;          // Q planes resident in LDS
    uint64_t *tb;         // [m_pad] image of the 64 batch columns
    uint8_t *sp;          // [m_pad] transformed syndrome
    int16_t *rowpiv;      // [m_pad] row -> pivot order or -1
    uint16_t *prow;       // [m_pad] pivot order -> row
    uint32_t *pcol;       // [m_pad] pivot order -> fault
    uint32_t *pairs;      // [64 * max_cdeg] column-in-batch | pivot order << 8
    uint32_t *bcols;      // [64]
    uint32_t *red;        // [0..15] keys A, [16..31] keys B, [32..47] flags A, [48..63] flags B, [64..] counters
    uint32_t *outw;       // packed solution
};

template <bool SPILL>
__device__ __forceinline__ uint64_t qd_q_load(const OsdLds &S, uint64_t *qglb, int kw_lds, int m_pad, int w, int r)
{
    if (SPILL && w >= kw_lds) return qglb[(size_t)(w - kw_lds) * m_pad + r];
    return S.q[(size_t)w * m_pad + r];
}
template <bool SPILL>
__device__ __forceinline__ void qd_q_store(const OsdLds &S, uint64_t *qglb, int kw_lds, int m_pad, int w, int r, uint64_t v)
{
    if (SPILL && w >= kw_lds) qglb[(size_t)(w - kw_lds) * m_pad + r] = v;
    else S.q[(size_t)w * m_pad + r] = v;
}

// Elimination over order[0..ncols).  kcap = number of pivots the Q storage can hold.
// Returns 0 when finished (early stop, rank exhausted or every column consumed with ncols == n), 1 when it ran out of
// sorted columns (ncols < n) or of Q capacity before finishing -- the caller must then redo the shot with more.
//
// One barrier per pivot: a round is  [own rows -> candidate key] -> wave min -> LDS -> BARRIER -> every thread reads the
// 16 partials -> every thread updates its own rows from row p (which its owner leaves alone this round).  The barrier
// of round i+1 also separates the updates of round i from those of round i+1, and the two reduction buffers alternate.
template <int T, bool SPILL>
__device__ int qd_osd_eliminate(const OsdGraphDev &g, const OsdLds &S, uint64_t *qglb, int kw_lds, int kcap,
                                const uint16_t *order, int ncols, const uint8_t *det, const uint8_t *upd,
                                int upd_rows, int out_words, int *npiv_out, int *inconsistent_out,
                                unsigned long long *a_dbg = nullptr)
{
    const int tid = threadIdx.x;
    constexpr int NW = T / 64;
    uint32_t *red = S.red;
    for (int r = tid; r < g.m_pad; r += T) {
        uint8_t s = 0;
        if (r < g.m) {
            s = det[r] & 1u;
            if (upd && r < upd_rows) s ^= upd[r] & 1u;
        }
        S.sp[r] = s;
        S.rowpiv[r] = -1;
    }
    for (int i = tid; i < kw_lds * g.m_pad; i += T) S.q[i] = 0ull;
    if (SPILL && qglb)
        for (int i = tid; i < (g.mw - kw_lds) * g.m_pad; i += T) qglb[i] = 0ull;
    for (int w = tid; w < out_words; w += T) S.outw[w] = 0u;
    if (tid < 32) red[tid] = QD_NOKEY;
    else if (tid < 64) red[tid] = 0u;
    __syncthreads();

    int npiv = 0, done = 0, hard = 0, phase = 0;
    for (int base = 0; base < ncols && !done && !hard; base += 64) {
        // ---- transform the next 64 columns: tb[r] bit c = (T * column_c)[r]
        for (int r = tid; r < g.m_pad; r += T) S.tb[r] = 0ull;
        if (tid == 0) red[64] = 0u;
        if (tid < 64) S.bcols[tid] = (base + tid < ncols) ? (uint32_t)order[base + tid] : 0xFFFFFFFFu;
        __syncthreads();
        for (int x = tid; x < 64 * g.max_cdeg; x += T) {
            const int c = x / g.max_cdeg, q = x - c * g.max_cdeg;
            const uint32_t col = S.bcols[c];
            if (col != 0xFFFFFFFFu) {
                const uint32_t e0 = g.csc_ptr[col], e1 = g.csc_ptr[col + 1];
                if (e0 + q < e1) {
                    const int r = g.csc_row[e0 + q];
                    atomicXor(reinterpret_cast<unsigned long long *>(&S.tb[r]), 1ull << c);
                    const int k = S.rowpiv[r];
                    if (k >= 0) S.pairs[atomicAdd(&red[64], 1u)] = (uint32_t)c | ((uint32_t)k << 8);
                }
            }
        }
        __syncthreads();
        const int np = (int)red[64];
        if (np)
            for (int r = tid; r < g.m; r += T) {
                uint64_t x = S.tb[r];
                for (int i = 0; i < np; ++i) {
                    const uint32_t pr = S.pairs[i];
                    const int k = (int)(pr >> 8);
                    const uint64_t qw = qd_q_load<SPILL>(S, qglb, kw_lds, g.m_pad, k >> 6, r);
                    x ^= ((qw >> (k & 63)) & 1ull) << (pr & 63u);
                }
                S.tb[r] = x;
            }
        // ---- take pivots out of the batch, in column order
        for (;;) {
            uint32_t key = QD_NOKEY;
            int resid = 0;
            for (int r = tid; r < g.m; r += T)
                if (S.rowpiv[r] < 0) {
                    const uint64_t x = S.tb[r];
                    if (x) key = min(key, ((uint32_t)__builtin_ctzll(x) << 16) | (uint32_t)r);
                    resid |= S.sp[r];
                }
            key = qd_wave_umin(key);
            const unsigned long long bal = __ballot(resid);
            if ((tid & 63) == 0) { red[phase * 16 + (tid >> 6)] = key; red[32 + phase * 16 + (tid >> 6)] = (bal != 0ull); }
            __syncthreads();
            key = QD_NOKEY;
            int anyres = 0;
            {
                const uint4 *kv = reinterpret_cast<const uint4 *>(red + phase * 16);
                const uint4 *fv = reinterpret_cast<const uint4 *>(red + 32 + phase * 16);
#pragma unroll
                for (int w = 0; w < (NW + 3) / 4; ++w) {       // entries beyond NW hold NOKEY / 0 (set once per call)
                    const uint4 k4 = kv[w], f4 = fv[w];
                    key = min(key, min(min(k4.x, k4.y), min(k4.z, k4.w)));
                    anyres |= (int)(f4.x | f4.y | f4.z | f4.w);
                }
            }
            phase ^= 1;
            if (!anyres) { done = 1; break; }            // syndrome already in the span of the pivots found
            if (key == QD_NOKEY) break;                   // rest of the batch depends on earlier pivots
            if (npiv >= kcap) { hard = 1; break; }        // no room for another pivot in this kernel's Q storage
            const int c = (int)(key >> 16), p = (int)(key & 0xFFFFu);
            const int K = npiv, kw = K >> 6;
            const uint64_t kb = 1ull << (K & 63);
            const uint64_t tp = S.tb[p];
            const uint8_t spp = S.sp[p];
            for (int r = tid; r < g.m; r += T) {
                if (r == p) { S.rowpiv[r] = (int16_t)K; S.prow[K] = (uint16_t)p; S.pcol[K] = S.bcols[c]; }   // the owner records the pivot
                else if ((S.tb[r] >> c) & 1ull) {
                    S.tb[r] ^= tp;
                    S.sp[r] ^= spp;
                    for (int w = 0; w <= kw; ++w) {
                        uint64_t v = qd_q_load<SPILL>(S, qglb, kw_lds, g.m_pad, w, r) ^ qd_q_load<SPILL>(S, qglb, kw_lds, g.m_pad, w, p);
                        if (w == kw) v ^= kb;
                        qd_q_store<SPILL>(S, qglb, kw_lds, g.m_pad, w, r, v);
                    }
                }
            }
            npiv = K + 1;
        }
        __syncthreads();                                  // last round's updates, before the next batch re-uses tb / reads rowpiv
    }
    if (!done && !hard && ncols < g.n) {
        // out of sorted columns: finished only if the syndrome happens to be resolved already or no row is left
        int resid = 0;
        for (int r = tid; r < g.m; r += T)
            if (S.rowpiv[r] < 0) resid |= S.sp[r];
        const unsigned long long bal = __ballot(resid);
        if ((tid & 63) == 0) red[32 + phase * 16 + (tid >> 6)] = (bal != 0ull);
        __syncthreads();
        int anyres = 0;
        for (int w = 0; w < NW; ++w) anyres |= (int)red[32 + phase * 16 + w];
        if (anyres && npiv < g.m) hard = 1;
        phase ^= 1;
        __syncthreads();
    }
    if (hard) return 1;
    // residual left on a non-pivot row <=> syndrome outside the column space
    int inconsistent = 0;
    {
        int resid = 0;
        for (int r = tid; r < g.m; r += T)
            if (S.rowpiv[r] < 0) resid |= S.sp[r];
        const unsigned long long bal = __ballot(resid);
        if ((tid & 63) == 0) red[32 + phase * 16 + (tid >> 6)] = (bal != 0ull);
        __syncthreads();
        for (int w = 0; w < NW; ++w) inconsistent |= (int)red[32 + phase * 16 + w];
    }
    // OSD-0 solution
    for (int k = tid; k < npiv; k += T)
        if (S.sp[S.prow[k]]) {
            const uint32_t j = S.pcol[k];
            atomicOr(&S.outw[j >> 5], 1u << (j & 31u));
        }
    __syncthreads();
    *npiv_out = npiv;
    *inconsistent_out = inconsistent;
    return 0;
}

__device__ __forceinline__ void qd_osd_carve(unsigned char *smem, const int *off, OsdLds &S)
{
    S.q = reinterpret_cast<uint64_t *>(smem + off[0]);
    S.tb = reinterpret_cast<uint64_t *>(smem + off[1]);
    S.sp = smem + off[2];
    S.rowpiv = reinterpret_cast<int16_t *>(smem + off[3]);
    S.prow = reinterpret_cast<uint16_t *>(smem + off[4]);
    S.pcol = reinterpret_cast<uint32_t *>(smem + off[5]);
    S.pairs = reinterpret_cast<uint32_t *>(smem + off[6]);
    S.bcols = reinterpret_cast<uint32_t *>(smem + off[7]);
    S.red = reinterpret_cast<uint32_t *>(smem + off[8]);
    S.outw = reinterpret_cast<uint32_t *>(smem + off[9]);
}

// ---- register-resident path ------------------------------------------------------------------------------------------------

// ---- higher-order OSD: candidate sweep after a FULL-rank elimination (ldpc osd.hpp, OsdDecoder::decode) ----------------
// Flipping a non-pivot column c on changes the pivot coefficients by t_c[k] = (T H_c)[p_k] = [p_k in c] xor parity(Q[p_k] &
// {pivot order of c's pivoted rows}); cost change = w_c + sum_k t_c[k] * (+-w_{pcol k}).  Thread k-owner accumulates 32
// candidates at a time in registers; one workgroup reduction per 32 candidates.  Costs are integers (wfix), so the result
// does not depend on summation order; ties go to the earliest candidate in ldpc's enumeration order (OSD-0, then single
// columns in sorted order, then pairs / patterns), expressed as the lexicographic minimum of (cost change, class, order key).
struct SweepBest { long long delta; uint32_t cls; unsigned long long tie, what; };

__device__ __forceinline__ bool qd_sweep_less(long long d1, uint32_t c1, unsigned long long t1, long long d2, uint32_t c2,
                                              unsigned long long t2)
{
    if (d1 != d2) return d1 < d2;
    if (c1 != c2) return c1 < c2;
    return t1 < t2;
}

template <int T>
__device__ __noinline__ void qd_osd_sweep(const OsdRegArgs &a, unsigned char *smem, const float *llr, uint64_t *qglb,
                                          int npiv, int nnp)
{
    const int tid = threadIdx.x;
    constexpr int NW = T / 64;
    constexpr int KP = 4;                                  // pivots per thread: npiv <= m <= 4 * T for this kernel
    OsdLds S;
    qd_osd_carve(smem, a.off, S);
    const uint32_t *pivmask = reinterpret_cast<const uint32_t *>(smem + a.off_pivmask);
    const uint32_t *npl = reinterpret_cast<const uint32_t *>(smem + a.off_npl);
    unsigned char *scr = smem + a.off_sort;                // 8 KB, dead after the elimination
    uint16_t *J = reinterpret_cast<uint16_t *>(scr);                        // [32][16] pivot orders touched by candidate q
    int32_t *wsum = reinterpret_cast<int32_t *>(scr + 1024);                // [NW][32]
    SweepBest *bests = reinterpret_cast<SweepBest *>(scr + 1024 + NW * 128);// [32]
    uint32_t *maskw = reinterpret_cast<uint32_t *>(scr + 4096);             // [2][npiv <= 4T ... ] not used: masks stay in registers
    (void)maskw;
    const int m_pad = a.m_pad, kw_lds = a.f_kw, n = a.n;

    int32_t sw[KP];
    uint32_t rowk[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int k = tid + i * T;
        sw[i] = 0; rowk[i] = 0;
        if (k < npiv) {
            const uint32_t row = S.prow[k];
            const int32_t w = (int32_t)a.wfix[S.pcol[k]];
            rowk[i] = row;
            sw[i] = S.sp[row] ? -w : w;                    // a pivot that is on in the OSD-0 solution gets cheaper when flipped
        }
    }
    // bits[i] bit q = t_{candidate q}[k_i] for the 32 candidates whose pivot lists sit in J.  The Q words of one candidate
    // are fetched with independent loads (planes beyond the LDS budget, pivot order >= 64 * f_kw, come from HBM)
    auto eval_bits = [&](uint32_t bits[KP]) {
        const uint4 *J4 = reinterpret_cast<const uint4 *>(J);
#pragma unroll
        for (int i = 0; i < KP; ++i) bits[i] = 0u;
        for (int q = 0; q < 32; ++q) {
            const uint4 ja = J4[2 * q], jb = J4[2 * q + 1];
            const uint32_t jw[8] = {ja.x, ja.y, ja.z, ja.w, jb.x, jb.y, jb.z, jb.w};
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int k = tid + i * T;
                if (k >= npiv) continue;
                uint32_t b = 0u;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (e < a.max_cdeg) {
                        const uint32_t kk = (e & 1) ? (jw[e >> 1] >> 16) : (jw[e >> 1] & 0xFFFFu);
                        const bool valid = kk != 0xFFFFu;
                        const uint32_t w = valid ? (kk >> 6) : 0u;
                        uint64_t word = S.q[(size_t)min(w, (uint32_t)kw_lds - 1u) * m_pad + rowk[i]];
                        if (w >= (uint32_t)kw_lds) word = qglb[(size_t)(w - kw_lds) * m_pad + rowk[i]];
                        const uint32_t t = ((kk == (uint32_t)k) ? 1u : 0u) ^ ((uint32_t)(word >> (kk & 63u)) & 1u);
                        b ^= valid ? t : 0u;
                    }
                }
                bits[i] |= b << q;
            }
        }
    };
    auto fill_J = [&](auto col_of) {            // col_of(q) -> fault index or 0xFFFFFFFF
        for (int x = tid; x < 32 * 16; x += T) {
            const int q = x >> 4, e = x & 15;
            uint16_t v = 0xFFFFu;
            const uint32_t col = col_of(q);
            if (col != 0xFFFFFFFFu && e < a.max_cdeg) {
                const uint32_t e0 = a.csc_ptr[col], e1 = a.csc_ptr[col + 1];
                if (e0 + e < e1) { const int kk = S.rowpiv[a.csc_row[e0 + e]]; if (kk >= 0) v = (uint16_t)kk; }
            }
            J[x] = v;
        }
    };
    // workgroup sums of 32 per-thread accumulators -> every thread q < 32 returns the total of accumulator q
    auto reduce32 = [&](const int32_t acc[32]) -> long long {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const uint32_t v = qd_wave_add((uint32_t)acc[q]);
            if ((tid & 63) == 0) wsum[(tid >> 6) * 32 + q] = (int32_t)v;
        }
        __syncthreads();
        long long tot = 0;
        if (tid < 32)
            for (int w = 0; w < NW; ++w) tot += (long long)wsum[w * 32 + tid];
        __syncthreads();
        return tot;
    };

    SweepBest best{0x7FFFFFFFFFFFFFFFll, 3u, ~0ull, 0ull};     // lane q < 32 tracks the best candidate it has seen
    if (a.osd_w == 1) {
        // ---- singles: every non-pivot column, 32 consecutive fault indices at a time
        for (int w0 = 0; w0 < a.out_words; ++w0) {
            const uint32_t pm = pivmask[w0];
            auto col_of = [&](int q) -> uint32_t {
                const uint32_t j = (uint32_t)(w0 * 32 + q);
                return (j < (uint32_t)n && !((pm >> q) & 1u)) ? j : 0xFFFFFFFFu;
            };
            if (__builtin_amdgcn_readfirstlane(pm) == 0xFFFFFFFFu) continue;
            fill_J(col_of);
            __syncthreads();
            uint32_t bits[KP];
            eval_bits(bits);
            int32_t acc[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                int32_t s = 0;
#pragma unroll
                for (int i = 0; i < KP; ++i) s += ((bits[i] >> q) & 1u) ? sw[i] : 0;
                acc[q] = s;
            }
            const long long tot = reduce32(acc);
            if (tid < 32) {
                const uint32_t col = col_of(tid);
                if (col != 0xFFFFFFFFu) {
                    const long long d = tot + (long long)a.wfix[col];
                    const unsigned long long tie = ((unsigned long long)qd_mono_key(llr[a.bit_slot_of[col]]) << 32) | col;
                    if (qd_sweep_less(d, 1u, tie, best.delta, best.cls, best.tie)) best = SweepBest{d, 1u, tie, (unsigned long long)col};
                }
            }
        }
    }
    // ---- patterns over the first lam non-pivot columns of the order: pairs (combination sweep) or all subsets (exhaustive)
    const int lam = min(min(a.osd_order, nnp), 64);
    if (lam >= (a.osd_w == 1 ? 2 : 1)) {
        unsigned long long mask[KP];
#pragma unroll
        for (int i = 0; i < KP; ++i) mask[i] = 0ull;
        for (int h = 0; h * 32 < lam; ++h) {
            auto col_of = [&](int q) -> uint32_t { return (h * 32 + q < lam) ? npl[h * 32 + q] : 0xFFFFFFFFu; };
            fill_J(col_of);
            __syncthreads();
            uint32_t bits[KP];
            eval_bits(bits);
#pragma unroll
            for (int i = 0; i < KP; ++i) mask[i] |= (unsigned long long)bits[i] << (32 * h);
            __syncthreads();
        }
        const unsigned long long npat = (a.osd_w == 1) ? (unsigned long long)lam * (lam - 1) / 2 : ((1ull << lam) - 1ull);
        for (unsigned long long p0 = 0; p0 < npat; p0 += 32) {
            // pattern of candidate q = tid & 31 (all threads need all 32 patterns: recompute per q)
            auto pat_of = [&](unsigned long long ic) -> unsigned long long {
                if (ic >= npat) return 0ull;
                if (a.osd_w == 2) return ic + 1ull;
                unsigned long long qq = ic; int x = 0;                 // pairs (x, y), x < y < lam, lexicographic
                while (qq >= (unsigned long long)(lam - 1 - x)) { qq -= (unsigned long long)(lam - 1 - x); ++x; }
                return (1ull << x) | (1ull << (x + 1 + (int)qq));
            };
            int32_t acc[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const unsigned long long pat = pat_of(p0 + q);
                int32_t s = 0;
#pragma unroll
                for (int i = 0; i < KP; ++i) s += (__popcll(mask[i] & pat) & 1) ? sw[i] : 0;
                acc[q] = s;
            }
            const long long tot = reduce32(acc);
            if (tid < 32 && p0 + tid < npat) {
                const unsigned long long pat = pat_of(p0 + tid);
                long long d = tot;
                for (int b = 0; b < lam; ++b) if ((pat >> b) & 1ull) d += (long long)a.wfix[npl[b]];
                const unsigned long long tie = p0 + tid;
                if (qd_sweep_less(d, 2u, tie, best.delta, best.cls, best.tie)) best = SweepBest{d, 2u, tie, pat};
            }
        }
    }
    // ---- winner over the 32 lanes; keep OSD-0 unless a candidate is strictly cheaper
    if (tid < 32) bests[tid] = best;
    __syncthreads();
    SweepBest win = bests[0];
    for (int q = 1; q < 32; ++q) { const SweepBest c = bests[q]; if (qd_sweep_less(c.delta, c.cls, c.tie, win.delta, win.cls, win.tie)) win = c; }
    __syncthreads();
    unsigned long long wpat = 0ull;                         // winner as a pattern over (<= 64) columns listed in J rows
    int wcols = 0;
    uint32_t wcol[64];
    if (win.delta < 0) {
        if (win.cls == 1u) { wcol[0] = (uint32_t)win.what; wcols = 1; }
        else for (int b = 0; b < lam; ++b) if ((win.what >> b) & 1ull) wcol[wcols++] = npl[b];
    }
    (void)wpat;
    // ---- solution: pivots = OSD-0 coefficients xor t_winner; winner columns on
    for (int w = tid; w < a.out_words; w += T) S.outw[w] = 0u;
    uint32_t flip[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) flip[i] = 0u;
    for (int h = 0; h * 32 < wcols; ++h) {
        auto col_of = [&](int q) -> uint32_t { return (h * 32 + q < wcols) ? wcol[h * 32 + q] : 0xFFFFFFFFu; };
        __syncthreads();
        fill_J(col_of);
        __syncthreads();
        uint32_t bits[KP];
        eval_bits(bits);
#pragma unroll
        for (int i = 0; i < KP; ++i) flip[i] ^= (uint32_t)__popc(bits[i]) & 1u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int k = tid + i * T;
        if (k < npiv && ((uint32_t)S.sp[rowk[i]] ^ flip[i])) { const uint32_t j = S.pcol[k]; atomicOr(&S.outw[j >> 5], 1u << (j & 31u)); }
    }
    if (tid == 0)
        for (int c = 0; c < wcols; ++c) atomicOr(&S.outw[wcol[c] >> 5], 1u << (wcol[c] & 31u));
    __syncthreads();
}

// ---- the same sweep with Q transposed --------------------------------------------------------------------------------------
// t_c = XOR over the pivoted rows r of column c of (e_j + column j of Q), j = pivot order of r, as a bit vector over pivot
// order k.  With Q stored by row that is <= max_cdeg word reads per (candidate, pivot); transposed (MT[w][j] = bits k in
// word w of column j) it is <= max_cdeg * W reads per CANDIDATE, and a whole candidate fits one thread: no workgroup
// reduction per 32 candidates, only one for the winner.  The transpose goes through a global scratch (64 x 64 bit blocks,
// one per wavefront at a time) and replaces Q in LDS.  Same costs, same tie rule, same result as qd_osd_sweep.
// SW = pivot-order words a candidate vector may span: 16 (rank <= 1024) or 32 (rank <= 2048)
template <int T, int SW>
__device__ __noinline__ void qd_osd_sweep_t(const OsdRegArgs &a, unsigned char *smem, const float *llr, uint64_t *qglb, uint64_t *mt,
                                            int npiv, int nnp)
{
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NW = T / 64;
    OsdLds S;
    qd_osd_carve(smem, a.off, S);
    const uint32_t *pivmask = reinterpret_cast<const uint32_t *>(smem + a.off_pivmask);
    const uint32_t *npl = reinterpret_cast<const uint32_t *>(smem + a.off_npl);
    unsigned char *scr = smem + a.off_sort;                                  // 8 KB, dead after the elimination
    int32_t *swl = reinterpret_cast<int32_t *>(scr);                         // [<= 2048] signed pivot weights
    unsigned char *scr2 = smem + a.off_order;                                // 2 KB, the tier order: dead as well
    SweepBest *bests = reinterpret_cast<SweepBest *>(scr2);                  // [NW]
    uint64_t *twin = reinterpret_cast<uint64_t *>(scr2 + 1024);              // [SW] winner's vector
    const int m_pad = a.m_pad, kw_lds = a.f_kw, n = a.n;
    const int Wp = (npiv + 63) >> 6;
    uint64_t *tv = mt + (size_t)a.mw * m_pad;                                // [64][SW] vectors of the first non-pivot columns

    // ---- MT = transpose of (Q rows in pivot order), block by block
    for (int bi = tid >> 6; bi < Wp * Wp; bi += NW) {
        const int kb = bi / Wp, jb = bi - kb * Wp;
        const int k = kb * 64 + lane;
        uint64_t x = 0ull;
        if (k < npiv) x = qd_q_load<true>(S, qglb, kw_lds, m_pad, jb, S.prow[k]);
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            const uint64_t msk = sh == 32 ? 0x00000000FFFFFFFFull : sh == 16 ? 0x0000FFFF0000FFFFull : sh == 8 ? 0x00FF00FF00FF00FFull
                               : sh == 4 ? 0x0F0F0F0F0F0F0F0Full : sh == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
            const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(x >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)x, sh);
            x = (lane & sh) ? (((other >> sh) & msk) | (x & ~msk)) : ((x & msk) | ((other & msk) << sh));
        }
        mt[(size_t)kb * m_pad + jb * 64 + lane] = x;                          // word kb of column j = jb * 64 + lane
    }
    for (int k = tid; k < Wp * 64; k += T) {
        int32_t v = 0;
        if (k < npiv) {
            const int32_t w = (int32_t)a.wfix[S.pcol[k]];
            v = S.sp[S.prow[k]] ? -w : w;                  // a pivot that is on in the OSD-0 solution gets cheaper when flipped
        }
        swl[k] = v;
    }
    __syncthreads();                                       // every Q word has been read, every MT word written
    // (word kw_lds, when the rank needs one more plane than Q had in LDS, goes where the batch words lived: S.tb is idle now)
    // words beyond that stay in the scratch (L2)
    {
        for (int i = tid; i < min(Wp, kw_lds + 1) * m_pad; i += T) {
            if (i < kw_lds * m_pad) S.q[i] = mt[i];
            else S.tb[i - kw_lds * m_pad] = mt[i];
        }
        __syncthreads();
    }

    auto add_col = [&](uint32_t col, uint64_t t[SW]) {               // t ^= vector of fault `col`
        const uint32_t e0 = a.csc_ptr[col], e1 = a.csc_ptr[col + 1];
        for (uint32_t e = e0; e < e1; ++e) {
            const int rr = a.csc_row[e];
            const int j = S.rowpiv[rr];
            if (j < 0) continue;
            const int eb = j;                                                // where the identity part of the column sits
#pragma unroll
            for (int w = 0; w < SW; ++w)
                if (w < Wp) {                                               // (uniform branches: no speculative loads)
                    uint64_t v;
                    if (w < kw_lds) v = S.q[(size_t)w * m_pad + j];
                    else if (w == kw_lds) v = S.tb[j];
                    else v = mt[(size_t)w * m_pad + j];
                    t[w] ^= v ^ ((w == (eb >> 6)) ? (1ull << (eb & 63)) : 0ull);
                }
        }
    };
    auto wsum = [&](const uint64_t t[SW]) -> long long {
        long long s = 0;
#pragma unroll
        for (int w = 0; w < SW; ++w)
            if (w < Wp) {
                uint64_t x = t[w];
                while (x) { s += (long long)swl[w * 64 + __builtin_ctzll(x)]; x &= x - 1ull; }
            }
        return s;
    };

    SweepBest best{0x7FFFFFFFFFFFFFFFll, 3u, ~0ull, 0ull};
    if (a.osd_w == 1) {
        // ---- singles: every non-pivot column
        for (int col = tid; col < n; col += T) {
            if ((pivmask[col >> 5] >> (col & 31)) & 1u) continue;
            uint64_t t[SW];
#pragma unroll
            for (int w = 0; w < SW; ++w) t[w] = 0ull;
            add_col((uint32_t)col, t);
            const long long d = wsum(t) + (long long)a.wfix[col];
            if (d <= best.delta) {
                const unsigned long long tie = ((unsigned long long)qd_mono_key(llr[a.bit_slot_of[col]]) << 32) | (uint32_t)col;
                if (qd_sweep_less(d, 1u, tie, best.delta, best.cls, best.tie)) best = SweepBest{d, 1u, tie, (unsigned long long)col};
            }
        }
    }
    // ---- patterns over the first lam non-pivot columns of the order: pairs (combination sweep) or all subsets (exhaustive)
    const int lam = min(min(a.osd_order, nnp), 64);
    if (lam >= (a.osd_w == 1 ? 2 : 1)) {
        if (tid < lam) {
            uint64_t t[SW];
#pragma unroll
            for (int w = 0; w < SW; ++w) t[w] = 0ull;
            add_col(npl[tid], t);
#pragma unroll
            for (int w = 0; w < SW; ++w) tv[tid * SW + w] = t[w];
        }
        __syncthreads();
        const unsigned long long npat = (a.osd_w == 1) ? (unsigned long long)lam * (lam - 1) / 2 : ((1ull << lam) - 1ull);
        for (unsigned long long ic = tid; ic < npat; ic += T) {
            unsigned long long pat;
            if (a.osd_w == 2) pat = ic + 1ull;
            else {
                unsigned long long qq = ic; int x = 0;                 // pairs (x, y), x < y < lam, lexicographic
                while (qq >= (unsigned long long)(lam - 1 - x)) { qq -= (unsigned long long)(lam - 1 - x); ++x; }
                pat = (1ull << x) | (1ull << (x + 1 + (int)qq));
            }
            uint64_t t[SW];
#pragma unroll
            for (int w = 0; w < SW; ++w) t[w] = 0ull;
            long long d = 0;
            for (int b = 0; b < lam; ++b)
                if ((pat >> b) & 1ull) {
                    d += (long long)a.wfix[npl[b]];
#pragma unroll
                    for (int w = 0; w < SW; ++w)
                        if (w < Wp) t[w] ^= tv[b * SW + w];
                }
            d += wsum(t);
            if (qd_sweep_less(d, 2u, ic, best.delta, best.cls, best.tie)) best = SweepBest{d, 2u, ic, pat};
        }
    }
    // ---- winner: wavefront minimum by shuffles, then the NW partials
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        SweepBest o;
        o.delta = ((long long)__shfl_xor((int)(best.delta >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)best.delta, sh);
        o.cls = (uint32_t)__shfl_xor((int)best.cls, sh);
        o.tie = ((unsigned long long)(uint32_t)__shfl_xor((int)(best.tie >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)best.tie, sh);
        o.what = ((unsigned long long)(uint32_t)__shfl_xor((int)(best.what >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)best.what, sh);
        if (qd_sweep_less(o.delta, o.cls, o.tie, best.delta, best.cls, best.tie)) best = o;
    }
    if (lane == 0) bests[tid >> 6] = best;
    __syncthreads();
    SweepBest win = bests[0];
    for (int q = 1; q < NW; ++q) { const SweepBest c = bests[q]; if (qd_sweep_less(c.delta, c.cls, c.tie, win.delta, win.cls, win.tie)) win = c; }
    // ---- solution: pivots = OSD-0 coefficients xor the winner's vector; winner columns on (OSD-0 unless strictly cheaper)
    for (int w = tid; w < a.out_words; w += T) S.outw[w] = 0u;
    const bool take = win.delta < 0;
    if (tid == 0) {
        uint64_t t[SW];
#pragma unroll
        for (int w = 0; w < SW; ++w) t[w] = 0ull;
        if (take) {
            if (win.cls == 1u) add_col((uint32_t)win.what, t);
            else for (int b = 0; b < lam; ++b) if ((win.what >> b) & 1ull) add_col(npl[b], t);
        }
#pragma unroll
        for (int w = 0; w < SW; ++w) twin[w] = t[w];
    }
    __syncthreads();
    for (int k = tid; k < npiv; k += T) {
        const int tbit = k;
        if ((uint32_t)S.sp[S.prow[k]] ^ (uint32_t)((twin[tbit >> 6] >> (tbit & 63)) & 1ull)) {
            const uint32_t j = S.pcol[k];
            atomicOr(&S.outw[j >> 5], 1u << (j & 31u));
        }
    }
    if (tid == 0 && take) {
        if (win.cls == 1u) atomicOr(&S.outw[(uint32_t)win.what >> 5], 1u << ((uint32_t)win.what & 31u));
        else for (int b = 0; b < lam; ++b) if ((win.what >> b) & 1ull) atomicOr(&S.outw[npl[b] >> 5], 1u << (npl[b] & 31u));
    }
    __syncthreads();
}

// one call site in the kernel (two would change its register allocation: the pivot rounds measured 40 % slower)
template <int T>
__device__ __noinline__ void qd_osd_sweep_pick(const OsdRegArgs &a, unsigned char *smem, const float *llr, uint64_t *qglb, uint64_t *mt,
                                               int npiv, int nnp)
{
    if (npiv <= 1024) qd_osd_sweep_t<T, 16>(a, smem, llr, qglb, mt, npiv, nnp);
    else qd_osd_sweep_t<T, 32>(a, smem, llr, qglb, mt, npiv, nnp);
}

// ---- one batch of the OSD-0 elimination by a single wavefront ---------------------------------------------------------
// A pivot only ever touches rows whose 64-column panel word is non-zero, and a batch of sparse columns leaves most rows
// zero (64 x ~3.5 entries on ~1000 rows).  Those rows are compacted (in row order) into `list`; wavefront 0 keeps them in
// registers -- QD_PANEL_SLOTS per lane: panel word, syndrome bit, pivot flag, the first Q planes -- and runs the pivot loop on
// its own: no workgroup barrier, no LDS round trip per pivot (the pivot row is broadcast with v_readlane).  Same pivot
// choice as the workgroup-wide loop (lowest column, then lowest row: list positions ascend with the row index), same
// updates, so the results are identical.  Preconditions checked by the caller: nL <= 64 * QD_PANEL_SLOTS, every pivot of
// the batch stays in Q planes 0..QD_PANEL_PLANES-1.  Returns the new pivot count; *done_out = syndrome explained.
#define QD_PANEL_SLOTS 4
#define QD_PANEL_PLANES 2   // (3 measured slower: late batches hold few pivots, the fixed cost of compaction + call exceeds a handful of barrier rounds)
#ifndef QD_OSD_PANEL_INLINE
#define QD_OSD_PANEL_INLINE __forceinline__
#endif
__device__ QD_OSD_PANEL_INLINE int qd_osd_panel_wave0(const OsdLds &S, const uint16_t *list, int nL, int npiv0, uint32_t outside_resid,
                                               int m_pad, int *done_out)
{
    const int lane = threadIdx.x & 63;
    int row[QD_PANEL_SLOTS];
    uint64_t tb[QD_PANEL_SLOTS], q[QD_PANEL_SLOTS][QD_PANEL_PLANES];
    uint32_t spb = 0u, pivb = 0u, valid = 0u;
    const int planes0 = (npiv0 + 63) >> 6;                            // planes that hold anything yet
#pragma unroll
    for (int s = 0; s < QD_PANEL_SLOTS; ++s) {
        const int pos = s * 64 + lane;
        row[s] = 0; tb[s] = 0ull;
#pragma unroll
        for (int w = 0; w < QD_PANEL_PLANES; ++w) q[s][w] = 0ull;
        if (pos < nL) {
            const int r = list[pos];
            row[s] = r; valid |= 1u << s;
            tb[s] = S.tb[r];
            spb |= (uint32_t)(S.sp[r] & 1u) << s;
            pivb |= (S.rowpiv[r] >= 0 ? 1u : 0u) << s;
#pragma unroll
            for (int w = 0; w < QD_PANEL_PLANES; ++w)
                if (w < planes0) q[s][w] = S.q[(size_t)w * m_pad + r];
        }
    }
    int npiv = npiv0, done = 0;
    // (a column-driven variant -- one ballot per slot and column instead of the key minimum -- measured 7 % slower)
    for (;;) {
        uint32_t key = QD_NOKEY, resid = 0u;
#pragma unroll
        for (int s = 0; s < QD_PANEL_SLOTS; ++s)
            if (((valid & ~pivb) >> s) & 1u) {
                if (tb[s]) key = min(key, ((uint32_t)__builtin_ctzll(tb[s]) << 16) | (uint32_t)(s * 64 + lane));
                resid |= (spb >> s) & 1u;
            }
        key = qd_wave_umin(key);
        const unsigned long long bal = __ballot(resid != 0u);
        if (bal == 0ull && !outside_resid) { done = 1; break; }      // syndrome already in the span of the pivots found
        if (key == QD_NOKEY) break;                                  // rest of the batch depends on earlier pivots
        const int c = (int)(key >> 16), pos = (int)(key & 0xFFFFu);
        const int ol = pos & 63, os = pos >> 6;                      // owner lane / slot: uniform
        const int K = npiv, kw = K >> 6;
        uint64_t tp = 0ull, qp[QD_PANEL_PLANES];
        int prow = 0;
#pragma unroll
        for (int w = 0; w < QD_PANEL_PLANES; ++w) qp[w] = 0ull;
#pragma unroll
        for (int s = 0; s < QD_PANEL_SLOTS; ++s)
            if (s == os) {
                tp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(tb[s] >> 32), ol) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)tb[s], ol);
#pragma unroll
                for (int w = 0; w < QD_PANEL_PLANES; ++w)
                    if (w <= kw)
                        qp[w] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(q[s][w] >> 32), ol) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)q[s][w], ol);
                prow = __builtin_amdgcn_readlane(row[s], ol);
            }
        const uint32_t spp = ((uint32_t)__builtin_amdgcn_readlane((int)spb, ol) >> os) & 1u;
#pragma unroll
        for (int w = 0; w < QD_PANEL_PLANES; ++w)
            if (w == kw) qp[w] ^= 1ull << (K & 63);                  // the new pivot's own bit rides along
        if (lane == 0) { S.rowpiv[prow] = (int16_t)K; S.prow[K] = (uint16_t)prow; S.pcol[K] = S.bcols[c]; }
#pragma unroll
        for (int s = 0; s < QD_PANEL_SLOTS; ++s) {
            if (!((valid >> s) & 1u)) continue;
            if (s * 64 + lane == pos) pivb |= 1u << s;
            else if ((tb[s] >> c) & 1ull) {
                tb[s] ^= tp;
                spb ^= spp << s;
#pragma unroll
                for (int w = 0; w < QD_PANEL_PLANES; ++w)
                    if (w <= kw) q[s][w] ^= qp[w];
            }
        }
        npiv = K + 1;
    }
    // publish what the other wavefronts keep in registers or read later: syndrome bits and Q planes of the touched rows
    const int planes = (npiv + 63) >> 6;
#pragma unroll
    for (int s = 0; s < QD_PANEL_SLOTS; ++s)
        if ((valid >> s) & 1u) {
            const int r = row[s];
            S.sp[r] = (uint8_t)((spb >> s) & 1u);
#pragma unroll
            for (int w = 0; w < QD_PANEL_PLANES; ++w)
                if (w < planes) S.q[(size_t)w * m_pad + r] = q[s][w];
        }
    *done_out = done;
    return npiv;
}

template <int T, int RPT, bool WFULL>
#ifndef QD_OSD0_WPS
#define QD_OSD0_WPS (T == 512 ? 4 : T / 128)   // 512 threads: two workgroups per CU, 128 registers.  (Rounds 2-5: three per CU at 85 registers and 88-144 bytes of
                                               // scratch per lane -- the faster trade while this was THE OSD-0 kernel; since round 4 it only takes the shots qd_osd0_sr_kernel
                                               // hands over and the windows that kernel does not fit, and a fall-back does not get to spill: ScratchSize 0, tests/test_api.py)
#endif
__global__ void __launch_bounds__(T, (WFULL ? T / 256 : QD_OSD0_WPS)) qd_osd0_reg_kernel(OsdRegArgs a)   // full-rank instantiation: one workgroup per CU, so twice the registers
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    constexpr int NW = T / 64;
    constexpr int KWR = WFULL ? QD_OSD_KWR : QD_OSD_KWR0;      // Q planes kept in registers (~100 pivots at the headline, ~470 at p = 6e-3; 2, 4, 6 planes measured 10.5, 10.0, 10.6 ms at the headline and 178, 165, 161 ms at p = 6e-3)
    const int nfail = a.slot_list ? *a.slot_count : *a.fail_count;
    OsdLds S;
    qd_osd_carve(smem, a.off, S);
    uint64_t *sortbuf = reinterpret_cast<uint64_t *>(smem + a.off_sort);       // [QD_OSD_TIER]
    uint16_t *order = reinterpret_cast<uint16_t *>(smem + a.off_order);        // [QD_OSD_TIER]
    uint32_t *red = S.red;            // [0..31] pivot keys A/B, [32..63] flags A/B, [64] pair counter, [96..127] block sums A/B, [80] gather counter
    uint32_t *sumbuf = red + 96;
    uint32_t *pivmask = reinterpret_cast<uint32_t *>(smem + a.off_pivmask);    // [out_words] bit j: fault j is a pivot column (higher-order OSD)
    uint32_t *npl = reinterpret_cast<uint32_t *>(smem + a.off_npl);            // [64] first non-pivot columns of the order
    constexpr bool want_full = WFULL;                                           // OSD-CS / OSD-E need the complete factorisation (separate instantiation:
                                                                                // the OSD-0 kernel must not carry the sweep's registers)
    const int lam_max = want_full ? min(a.osd_order, 64) : 0;
    const int m = a.m, m_pad = a.m_pad, kw_lds = a.f_kw;
    uint64_t *qglb = a.q_spill_fast ? a.q_spill_fast + (int64_t)blockIdx.x * (int64_t)(a.mw - kw_lds) * m_pad : nullptr;
    for (int item = blockIdx.x; item < nfail; item += gridDim.x) {
        const int slot = a.slot_list ? a.slot_list[item] : item;
        const int64_t shot = a.fail_list[slot];
        const float *llr = a.llr_ws + (int64_t)slot * a.n_pad;
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
#ifdef QD_OSD_TIMING
        unsigned long long *a_dbg = a.dbg;
        unsigned long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long tick_ = wall_clock64();
#endif
        // ---- elimination state: registers + LDS mirror
        uint64_t my_tb[RPT], my_q[RPT][KWR];
        uint32_t my_sp = 0, my_piv = 0;              // bit i: row tid + i*T
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
            if (r < m_pad) {
                S.sp[r] = (uint8_t)sbit; S.rowpiv[r] = -1;
#pragma unroll
                for (int w = 0; w < KWR; ++w)
                    if (w < kw_lds) S.q[(size_t)w * m_pad + r] = 0ull;      // the register planes' mirror; later planes are cleared when first used
            }
        }
        for (int w = tid; w < a.out_words; w += T) { S.outw[w] = 0u; if (want_full) pivmask[w] = 0u; }
        if (tid < 64) red[tid] = ((tid & 16) == 0) ? QD_NOKEY : 0u;      // per phase: 16 keys then 16 flags
        __syncthreads();

        int npiv = 0, done = 0, phase = 0, sphase = 0, nnp = 0, ntier = 0;
        QD_TICK(10)
        uint32_t lo_key = 0, lo_idx = 0;             // every column with (key, fault index) < (lo_key, lo_idx) has been consumed
        while (!done) {
            // =============== draw the next tier of the column order (kept out of line: its 20 key registers and unrolled
            // compares must not push the elimination state of this loop into scratch)
            // OSD-0 stops after ~100 columns at the usual operating points: a first tier of 256 costs a third of a full one
            TierState ts{lo_key, lo_idx, sphase, 0, (!want_full && ntier == 0) ? QD_OSD_TIER_FIRST : QD_OSD_TIER};
            ++ntier;
            // The keys are re-read from the posteriors (L2) at every radix level instead of being held 20 per thread: those
            // registers were what pushed the OSD-0 instantiation (three workgroups per CU, 85 registers) into scratch, 500 -> 116 B
            // per lane.  Same-box A/B: headline OSD-0 10.2-11.5 -> 8.3-8.4 ms, p = 6e-3 152-154 -> 145-148 ms; OSD-CS(1) through
            // qd_osdw_col_kernel 106.8 -> 98.8 ms per 32768-shot launch.  Same keys, same order.
            const int cnt = qd_osd_draw_tier<T>(a, llr, sortbuf, order, red, sumbuf, ts);
            lo_key = ts.lo_key; lo_idx = ts.lo_idx; sphase = ts.sphase;
            if (ts.exhausted) break;
            // The tier code is inlined and wants the registers (20 keys per thread): the elimination state is not carried
            // across it but read back from its write-through LDS mirror, so nothing of it has to be spilled.
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = tid + i * T;
                my_tb[i] = 0ull;
#pragma unroll
                for (int w = 0; w < KWR; ++w) my_q[i][w] = 0ull;
                if (r < m) {
                    my_tb[i] = S.tb[r];
#pragma unroll
                    for (int w = 0; w < KWR; ++w)
                        if (w < kw_lds && w < ((npiv + 63) >> 6)) my_q[i][w] = S.q[(size_t)w * m_pad + r];
                }
            }
            QD_TICK(0)
            if (want_full && npiv >= a.rank) {
                // factorisation complete: every further column of the order is a non-pivot column
                if (tid == 0)
                    for (int i = 0; i < cnt && nnp + i < lam_max; ++i) npl[nnp + i] = order[i];
                nnp = min(lam_max, nnp + cnt);
                __syncthreads();
                if (nnp >= lam_max) break;
                continue;
            }

            // =============== eliminate over this tier, 64 columns at a time
            for (int base = 0; base < cnt && !done; base += 64) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) { const int r = tid + i * T; if (r < m_pad) S.tb[r] = 0ull; }
                if (tid == 0) red[64] = 0u;
                if (tid < 64) S.bcols[tid] = (base + tid < cnt) ? (uint32_t)order[base + tid] : 0xFFFFFFFFu;
                __syncthreads();
                if constexpr (want_full && QD_OSD_FULL_PAIRS) {
                for (int x = tid; x < 64 * a.max_cdeg; x += T) {
                    const int c = x / a.max_cdeg, q = x - c * a.max_cdeg;
                    const uint32_t col = S.bcols[c];
                    if (col != 0xFFFFFFFFu) {
                        const uint32_t e0 = a.csc_ptr[col], e1 = a.csc_ptr[col + 1];
                        if (e0 + q < e1) {
                            const int r = a.csc_row[e0 + q];
                            atomicXor(reinterpret_cast<unsigned long long *>(&S.tb[r]), 1ull << c);
                            const int k = S.rowpiv[r];
                            if (k >= 0) S.pairs[atomicAdd(&red[64], 1u)] = (uint32_t)c | ((uint32_t)k << 8);
                        }
                    }
                }
                __syncthreads();
                const int np = (int)red[64];
                {
                    // incidences eight at a time: the pair words first, then the 8 x RPT independent Q loads, then the bits
                    // (one pair -> one dependent Q load per step left the loop waiting on LDS latency)
                    uint64_t xr[RPT];
#pragma unroll
                    for (int i = 0; i < RPT; ++i) { const int r = tid + i * T; xr[i] = r < m ? S.tb[r] : 0ull; }
                    for (int p0 = 0; p0 < np; p0 += 8) {
                        uint32_t pr[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) pr[q] = (p0 + q < np) ? S.pairs[p0 + q] : 0xFFFFFFFFu;
                        uint64_t qw[RPT][8];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int i = 0; i < RPT; ++i) {
                                const int r = tid + i * T;
                                qw[i][q] = 0ull;
                                if (pr[q] != 0xFFFFFFFFu && r < m) qw[i][q] = qd_q_load<true>(S, qglb, kw_lds, m_pad, (int)(pr[q] >> 14), r);
                            }
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int i = 0; i < RPT; ++i)
                                if (pr[q] != 0xFFFFFFFFu) xr[i] ^= ((qw[i][q] >> ((pr[q] >> 8) & 63u)) & 1ull) << (pr[q] & 63u);
                    }
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * T;
                        if (np && r < m) S.tb[r] = xr[i];
                        my_tb[i] = xr[i];
                    }
                }
                } else {
                // OSD-0 (early stop, ~100 pivots): Q rows are sparse, so the transform is applied row-wise -- row r of T is
                // e_r plus e_{pivot row k} for every bit k of Q[r]:  x[r] = raw[r] ^ XOR_{k in Q[r]} raw[prow[k]].
                for (int x = tid; x < 64 * a.max_cdeg; x += T) {
                    const int c = x / a.max_cdeg, q = x - c * a.max_cdeg;
                    const uint32_t col = S.bcols[c];
                    if (col != 0xFFFFFFFFu) {
                        const uint32_t e0 = a.csc_ptr[col], e1 = a.csc_ptr[col + 1];
                        if (e0 + q < e1)
                            atomicXor(reinterpret_cast<unsigned long long *>(&S.tb[a.csc_row[e0 + q]]), 1ull << c);
                    }
                }
                __syncthreads();
                const int nplanes = (npiv + 63) >> 6;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int r = tid + i * T;
                    uint64_t x = 0ull;
                    if (r < m) {
                        x = S.tb[r];
#pragma unroll
                        for (int w = 0; w < KWR; ++w)
                            if (w < nplanes) {
                                uint64_t q = my_q[i][w];
                                while (q) {
                                    const int k = w * 64 + __builtin_ctzll(q);
                                    q &= q - 1ull;
                                    x ^= S.tb[S.prow[k]];
                                }
                            }
                        for (int w = KWR; w < nplanes; ++w) {
                            uint64_t q = qd_q_load<true>(S, qglb, kw_lds, m_pad, w, r);
                            while (q) {
                                const int k = w * 64 + __builtin_ctzll(q);
                                q &= q - 1ull;
                                x ^= S.tb[S.prow[k]];
                            }
                        }
                    }
                    my_tb[i] = x;
                }
                if (npiv > 0) {
                    __syncthreads();                       // every raw word has been read
#pragma unroll
                    for (int i = 0; i < RPT; ++i) { const int r = tid + i * T; if (r < m) S.tb[r] = my_tb[i]; }
                }
                }
                QD_TICK(1)
                bool did_panel = false;
                if constexpr (!want_full) {
                    if (npiv + 64 <= 64 * min(min(QD_PANEL_PLANES, KWR), kw_lds) && NW <= 8 && RPT <= 4) {
                        // ---- compact the rows with a non-zero panel word (row order), then one wavefront does the batch
                        uint16_t *list = reinterpret_cast<uint16_t *>(sortbuf);          // the tier buffer is idle between draws
                        uint32_t *buf = sumbuf + sphase * 64;                            // [i * 8 + wave] counts, [32 + wave] flags, [48..49] results
                        uint32_t pre[RPT], outres = 0u;
                        bool nz[RPT];
#pragma unroll
                        for (int i = 0; i < RPT; ++i) {
                            const int r = tid + i * T;
                            nz[i] = r < m && my_tb[i] != 0ull;
                            if (r < m && !nz[i] && !((my_piv >> i) & 1u)) outres |= (my_sp >> i) & 1u;
                            const unsigned long long bal = __ballot(nz[i]);
                            pre[i] = (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull));
                            if ((tid & 63) == 0) buf[i * 8 + (tid >> 6)] = (uint32_t)__popcll(bal);
                        }
                        {
                            const unsigned long long bo = __ballot(outres != 0u);
                            if ((tid & 63) == 0) buf[32 + (tid >> 6)] = (bo != 0ull);
                        }
                        __syncthreads();
                        uint32_t at = 0u, nL = 0u, oflag = 0u;
#pragma unroll
                        for (int i = 0; i < RPT; ++i) {
                            uint32_t before = 0u, tot = 0u;
                            for (int w = 0; w < NW; ++w) { const uint32_t cw = buf[i * 8 + w]; tot += cw; if (w < (tid >> 6)) before += cw; }
                            const uint32_t posn = nL + before + pre[i];
                            if (nz[i] && posn < 64u * QD_PANEL_SLOTS) list[posn] = (uint16_t)(tid + i * T);
                            nL += tot;
                        }
                        (void)at;
                        for (int w = 0; w < NW; ++w) oflag |= buf[32 + w];
                        sphase ^= 1;
                        nL = (uint32_t)__builtin_amdgcn_readfirstlane((int)nL);
                        oflag = (uint32_t)__builtin_amdgcn_readfirstlane((int)oflag);
                        if (nL <= 64u * QD_PANEL_SLOTS) {
                            did_panel = true;
                            __syncthreads();                                             // list, S.tb, S.sp, S.q are in place
                            if (tid < 64) {
                                int dn = 0;
                                const int np2 = qd_osd_panel_wave0(S, list, (int)nL, npiv, oflag, m_pad, &dn);
                                if (tid == 0) { buf[48] = (uint32_t)np2; buf[49] = (uint32_t)dn; }
                            }
                            __syncthreads();
                            npiv = (int)buf[48];
                            done = (int)buf[49];
                            // refresh the register copies of what wavefront 0 changed
#pragma unroll
                            for (int i = 0; i < RPT; ++i) {
                                const int r = tid + i * T;
                                if (r < m) {
                                    my_sp = (my_sp & ~(1u << i)) | ((uint32_t)(S.sp[r] & 1u) << i);
                                    if (S.rowpiv[r] >= 0) my_piv |= 1u << i;
#pragma unroll
                                    for (int w = 0; w < QD_PANEL_PLANES; ++w)
                                        if (w < ((npiv + 63) >> 6)) my_q[i][w] = S.q[(size_t)w * m_pad + r];
                                }
                            }
                        }
                    }
                }
                int bpos = 0;                               // batch columns before bpos have been classified
                const int nb = min(64, cnt - base);
                // ---- pivots of this batch, one barrier per round
                if (!did_panel)
                for (;;) {
                    uint32_t key = QD_NOKEY;
                    uint32_t resid = 0;
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * T;
                        if (r < m && !((my_piv >> i) & 1u)) {
                            if (my_tb[i]) key = min(key, ((uint32_t)__builtin_ctzll(my_tb[i]) << 16) | (uint32_t)r);
                            resid |= (my_sp >> i) & 1u;
                        }
                    }
                    key = qd_wave_umin(key);
                    const unsigned long long bal = __ballot(resid != 0u);
                    if ((tid & 63) == 0) { red[phase * 32 + (tid >> 6)] = key; red[phase * 32 + 16 + (tid >> 6)] = (bal != 0ull); }
                    QD_TICK(4)
                    __syncthreads();
                    QD_TICK(5)
                    key = QD_NOKEY;
                    uint32_t anyres = 0;
                    {
                        // partial keys and flags sit in one 128-byte record per phase: [0..15] keys, [16..31] flags
                        const uint4 *kv = reinterpret_cast<const uint4 *>(red + phase * 32);
                        uint4 k4[(NW + 3) / 4], f4[(NW + 3) / 4];
#pragma unroll
                        for (int w = 0; w < (NW + 3) / 4; ++w) { k4[w] = kv[w]; f4[w] = kv[4 + w]; }
#pragma unroll
                        for (int w = 0; w < (NW + 3) / 4; ++w) {
                            key = min(key, min(min(k4[w].x, k4[w].y), min(k4[w].z, k4[w].w)));
                            anyres |= f4[w].x | f4[w].y | f4[w].z | f4[w].w;
                        }
                    }
                    phase ^= 1;
                    // every lane holds the same values; tell the compiler, so that the round runs on scalar control flow
                    key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
                    anyres = (uint32_t)__builtin_amdgcn_readfirstlane((int)anyres);
                    if (!anyres && !want_full) { done = 1; break; }   // syndrome already in the span of the pivots found
                    if (want_full && npiv >= a.rank) key = QD_NOKEY;  // rank reached: the rest of the batch is non-pivot
                    {
                        // batch columns [bpos, c) depend on earlier pivots: record the first lam_max of them in order
                        const int cend = (key == QD_NOKEY) ? nb : (int)(key >> 16);
                        if (want_full && nnp < lam_max) {
                            if (tid == 0)
                                for (int x = bpos; x < cend && nnp + (x - bpos) < lam_max; ++x) npl[nnp + (x - bpos)] = S.bcols[x];
                            nnp = min(lam_max, nnp + max(0, cend - bpos));
                        }
                        bpos = cend + 1;
                    }
                    if (key == QD_NOKEY) break;               // rest of the batch depends on earlier pivots
                    const int c = (int)(key >> 16), p = (int)(key & 0xFFFFu);
                    const int K = npiv, kw = K >> 6;
                    const uint64_t kb = 1ull << (K & 63);
                    // pivot row, from the LDS mirror (its owner does not touch it this round)
                    const uint64_t tp = S.tb[p];
                    const uint32_t spp = S.sp[p];
                    uint64_t qp[KWR];
#pragma unroll
                    for (int w = 0; w < KWR; ++w) qp[w] = (w <= kw) ? S.q[(size_t)w * m_pad + p] : 0ull;   // wave-uniform guards: most shots stay in plane 0..1
                    // The host guarantees f_kw >= min(KWR, mw) LDS planes; planes >= mw (tiny windows) read past the Q
                    // region: the values are never stored nor used because a pivot index never reaches such a plane.
                    QD_TICK(6)
                    const bool fresh = (K & 63) == 0 && kw >= KWR;     // first pivot of a memory-only plane: still uninitialised
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * T;
                        if (r >= m) continue;
                        if (fresh) qd_q_store<true>(S, qglb, kw_lds, m_pad, kw, r, 0ull);
                        if (r == p) {
                            my_piv |= 1u << i;
                            const uint32_t pc = S.bcols[c];
                            S.rowpiv[r] = (int16_t)K; S.prow[K] = (uint16_t)p; S.pcol[K] = pc;      // the owner records the pivot
                            if (want_full) atomicOr(&pivmask[pc >> 5], 1u << (pc & 31u));
                        } else if ((my_tb[i] >> c) & 1ull) {
                            my_tb[i] ^= tp;
                            S.tb[r] = my_tb[i];
                            if (spp) { my_sp ^= 1u << i; S.sp[r] = (uint8_t)((my_sp >> i) & 1u); }
#pragma unroll
                            for (int w = 0; w < KWR; ++w)
                                if (w <= kw) {
                                    uint64_t v = my_q[i][w] ^ qp[w];
                                    if (w == kw) v ^= kb;
                                    my_q[i][w] = v;
                                    S.q[(size_t)w * m_pad + r] = v;
                                }
                            for (int w = KWR; w <= kw; ++w) {       // beyond the register planes: memory only
                                uint64_t v = qd_q_load<true>(S, qglb, kw_lds, m_pad, w, r);
                                if (!(fresh && w == kw)) v ^= qd_q_load<true>(S, qglb, kw_lds, m_pad, w, p);
                                if (w == kw) v ^= kb;
                                qd_q_store<true>(S, qglb, kw_lds, m_pad, w, r, v);
                            }
                        }
                    }
                    npiv = K + 1;
                    QD_TICK(7)
                }
                __syncthreads();          // last round's mirror updates, before the next batch re-uses tb / reads rowpiv
                QD_TICK(2)
                if (want_full && npiv >= a.rank) {
                    // the rest of this tier is non-pivot as well
                    const int rest0 = base + 64;
                    if (tid == 0)
                        for (int i = rest0; i < cnt && nnp + (i - rest0) < lam_max; ++i) npl[nnp + (i - rest0)] = order[i];
                    nnp = min(lam_max, nnp + max(0, cnt - rest0));
                    __syncthreads();
                    break;
                }
            }
            if (want_full && npiv >= a.rank && nnp >= lam_max) break;
        }
        // ---- residual left on a non-pivot row <=> syndrome outside the column space
        uint32_t resid = 0;
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            if (tid + i * T < m && !((my_piv >> i) & 1u)) resid |= (my_sp >> i) & 1u;
        const int inconsistent = qd_block_sum<T>(resid, sumbuf, sphase) != 0u;
        if constexpr (want_full) {
            // writes the winning candidate (or OSD-0) into outw.  The first-generation sweep stays as the fallback for ranks above
            // 2048 (unreachable here: m <= 2048) -- and because the code generated for the pivot loop of THIS kernel depends on the
            // call sites around it: with this call removed the full-rank elimination measured 1.8x slower (418 -> 748 ms).
            if (a.mt_ws && ((npiv + 63) >> 6) <= 32)
            {
#ifdef QD_OSD_TIMING
                if (tid == 0) atomicAdd(&a.dbg[11], 1ull);
#endif
                qd_osd_sweep_pick<T>(a, smem, llr, qglb, a.mt_ws + (size_t)blockIdx.x * ((size_t)a.mw * m_pad + 64 * 32), npiv, nnp);
            }
            else
                qd_osd_sweep<T>(a, smem, llr, qglb, npiv, nnp);
        } else {
            // ---- OSD-0 solution: e[pivot column k] = transformed syndrome at pivot row k
            for (int k = tid; k < npiv; k += T)
                if (S.sp[S.prow[k]]) {
                    const uint32_t j = S.pcol[k];
                    atomicOr(&S.outw[j >> 5], 1u << (j & 31u));
                }
            __syncthreads();
        }
        for (int w = tid; w < a.out_words; w += T) a.err_bits[shot * a.out_words + w] = S.outw[w];
        if (tid == 0) a.status[shot] = (a.status[shot] & 0xFFFF) | (1 << 17) | (inconsistent ? (1 << 18) : 0) | (min(npiv, 4095) << 20);
        QD_TICK(3)
#ifdef QD_OSD_TIMING
        if (tid == 0) {
            for (int i = 0; i < 8; ++i) atomicAdd(&a_dbg[i], acc_[i]);
            atomicAdd(&a_dbg[8], 1ull); atomicAdd(&a_dbg[9], (unsigned long long)npiv); atomicAdd(&a_dbg[10], acc_[10]);
        }
#endif
        __syncthreads();   // LDS is recycled by the next shot
    }
}

// ---- full path: every column sorted, every Q plane available ------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(T) qd_osd0_full_kernel(OsdGraphDev g, BpGraphDev bg, DecodeArgs a,
                                                         const int32_t *in_list, const int32_t *in_count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int nlist = *in_count;
    OsdLds S;
    qd_osd_carve(smem, g.off, S);
    uint64_t *sortbuf = S.q;                                                    // phase 1; phase 2 reuses it as Q planes
    // persistent: a fixed grid walks the list; order/spill workspace is per workgroup
    for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int slot = in_list ? in_list[li] : li;
        const int64_t shot = a.fail_list[slot];
        uint16_t *order = a.order_ws + (int64_t)blockIdx.x * g.n;
        uint64_t *qglb = a.q_spill ? a.q_spill + (int64_t)blockIdx.x * (int64_t)(g.mw - g.kw_lds) * g.m_pad : nullptr;
        const float *llr = a.llr_ws + (int64_t)slot * bg.n_pad;
        for (int i = tid; i < g.npow2; i += T) sortbuf[i] = ~0ull;
        __syncthreads();
        for (int b = tid; b < g.n; b += T) {
            const uint32_t j = bg.bit_orig[b];
            sortbuf[j] = ((uint64_t)qd_mono_key(llr[b]) << 32) | j;
        }
        __syncthreads();
        qd_bitonic_u64<T>(sortbuf, g.npow2, tid);
        for (int i = tid; i < g.n; i += T) order[i] = (uint16_t)(sortbuf[i] & 0xFFFFu);
        __syncthreads();   // order[] is re-read by this workgroup only (same CU; the barrier drains the stores)
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
        int npiv = 0, inconsistent = 0;
        qd_osd_eliminate<T, true>(g, S, qglb, g.kw_lds, g.m, order, g.n, det, upd, a.upd_rows, bg.out_words, &npiv, &inconsistent);
        for (int w = tid; w < bg.out_words; w += T) a.err_bits[shot * bg.out_words + w] = S.outw[w];
        if (tid == 0) a.status[shot] = (a.status[shot] & 0xFFFF) | (1 << 17) | (inconsistent ? (1 << 18) : 0) | (min(npiv, 4095) << 20);
        __syncthreads();   // LDS is recycled by the next shot
    }
}


template <int TF, int RPT>
static hipError_t launch_reg(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks, hipStream_t s, bool handed_over)
{
    const bool wl = a.osd_w != 0;                              // higher-order OSD uses the one-workgroup-per-CU layout
    auto k = wl ? qd_osd0_reg_kernel<TF, RPT, true> : qd_osd0_reg_kernel<TF, RPT, false>;
    // higher-order OSD here = the full-rank elimination by ROW (one workgroup per CU): since round 5 only the windows the panel kernel
    // (osd_cs.hip) does not take -- more than 1408 detectors, or more faults than its sort holds -- and QD_OSDCS_OLD=1 (A/B)
    const int lds = wl ? g.w_lds_bytes : g.f_lds_bytes;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    OsdRegArgs r{};
    r.m = g.m; r.n = g.n; r.m_pad = g.m_pad; r.n_pad = bg.n_pad; r.max_cdeg = g.max_cdeg; r.mw = g.mw; r.f_kw = wl ? g.w_kw : g.f_kw;
    r.out_words = bg.out_words; r.upd_rows = a.upd_rows;
    for (int i = 0; i < 10; ++i) r.off[i] = wl ? g.w_off[i] : g.f_off[i];
    r.off_sort = wl ? g.w_off_sort : g.f_off_sort; r.off_order = wl ? g.w_off_order : g.f_off_order;
    r.csc_ptr = g.csc_ptr; r.csc_row = g.csc_row; r.bit_orig = bg.bit_orig;
    r.det = a.det; r.upd = a.upd; r.det_stride = a.det_stride; r.det_offset = a.det_offset; r.upd_stride = a.upd_stride;
    r.llr_ws = a.llr_ws; r.fail_list = a.fail_list; r.fail_count = a.fail_count; r.q_spill_fast = a.q_spill_fast; r.mt_ws = a.mt_ws;
    if (handed_over) { r.slot_list = a.hard_list; r.slot_count = a.hard_count; }
    r.err_bits = a.err_bits; r.status = a.status; r.dbg = a.dbg;
    r.off_pivmask = wl ? g.w_off_pivmask : g.f_off_pivmask; r.off_npl = wl ? g.w_off_npl : g.f_off_npl;
    r.osd_w = a.osd_w; r.osd_order = a.osd_order; r.rank = a.rank; r.wfix = g.wfix; r.bit_slot_of = bg.bit_slot_of;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(TF), lds, s, r);
    return hipGetLastError();
}

template <int T>
static hipError_t launch_full(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks, hipStream_t s)
{
    auto kf = qd_osd0_full_kernel<T>;
    hipError_t e = hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kf, dim3((unsigned)blocks), dim3(T), g.lds_bytes, s, g, bg, a, (const int32_t *)nullptr,
                       (const int32_t *)a.fail_count);
    return hipGetLastError();
}

hipError_t qd_launch_osd0(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks_fast,
                          int blocks_full, hipStream_t s, bool handed_over)
{
    if (g.f_lds_bytes > 0) {
        const bool ho = handed_over;                       // only the fail-list slots on the hard list (osd_sr.hip hands them over)
        const int rpt = (g.m + g.f_threads - 1) / g.f_threads;
        if (g.f_threads == 256) {
            switch (rpt) {
            case 1: return launch_reg<256, 1>(g, bg, a, blocks_fast, s, ho);
            case 2: return launch_reg<256, 2>(g, bg, a, blocks_fast, s, ho);
            case 3: return launch_reg<256, 3>(g, bg, a, blocks_fast, s, ho);
            default: return launch_reg<256, 4>(g, bg, a, blocks_fast, s, ho);
            }
        }
        switch (rpt) {
        case 1: return launch_reg<512, 1>(g, bg, a, blocks_fast, s, ho);
        case 2: return launch_reg<512, 2>(g, bg, a, blocks_fast, s, ho);
        case 3: return launch_reg<512, 3>(g, bg, a, blocks_fast, s, ho);
        default: return launch_reg<512, 4>(g, bg, a, blocks_fast, s, ho);
        }
    }
    switch (g.threads) {
    case 256: return launch_full<256>(g, bg, a, blocks_full, s);
    case 512: return launch_full<512>(g, bg, a, blocks_full, s);
    default: return launch_full<1024>(g, bg, a, blocks_full, s);
    }
}
