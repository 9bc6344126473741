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
//   qd_osd0_fast_kernel  sorts only the head of the order -- the <= 2048 columns with the smallest LLRs, picked by
//                        bisecting on the key's top 12 bits -- and keeps <= 6..8 Q planes; ~75 KB of LDS, two
//                        workgroups per CU.  Because elimination stops early this is enough for almost every shot; a
//                        shot that runs out of sorted columns or of Q planes is appended to the "hard" list untouched.
//   qd_osd0_full_kernel  sorts every column and keeps (or spills) all Q planes; one workgroup per CU; runs over the
//                        hard list.  Same results by construction: both consume the same column order.
#include "qd_internal.h"

#define QD_NOKEY 0xFFFFFFFFu

#ifdef QD_OSD_TIMING
#define QD_TICK(slot)                                                                   \
    {                                                                                   \
        const unsigned long long now_ = wall_clock64();                                 \
        if (threadIdx.x == 0 && a_dbg) atomicAdd(&a_dbg[slot], now_ - tick_);           \
        tick_ = now_;                                                                   \
    }
#else
#define QD_TICK(slot)
#endif

__device__ __forceinline__ uint64_t &qd_qword(uint64_t *q_lds, uint64_t *q_glb, int kw_lds, int m_pad, int w, int r)
{
    return (w < kw_lds) ? q_lds[(size_t)w * m_pad + r] : q_glb[(size_t)(w - kw_lds) * m_pad + r];
}

__device__ __forceinline__ uint32_t qd_mono_key(float llr)
{
    const float f = llr + 0.0f;                    // -0 -> +0, so that +-0 tie on the index like the oracle's '<'
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone float -> unsigned
}

template <int T>
__device__ __forceinline__ void qd_bitonic_u64(uint64_t *buf, int P, int tid)
{
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int pi = tid; pi < (P >> 1); pi += T) {
                const int i = ((pi & ~(j - 1)) << 1) | (pi & (j - 1));
                const int l = i | j;
                const uint64_t x = buf[i], y = buf[l];
                const bool up = ((i & k) == 0);
                if ((x > y) == up) { buf[i] = y; buf[l] = x; }
            }
            __syncthreads();
        }
}

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
#ifdef QD_OSD_TIMING
    unsigned long long tick_ = wall_clock64();
#endif
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

    QD_TICK(4)
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
        QD_TICK(5)
#ifdef QD_OSD_TIMING
        if (tid == 0 && a_dbg) atomicAdd(&a_dbg[12], 1ull);
#endif
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
        QD_TICK(6)
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
    QD_TICK(7)
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

// ---- fast path: head of the order only --------------------------------------------------------------------------------
// `cap` = number of columns to sort (<= QD_OSD_FAST_CAP).  Reads the shots to do from in_list (or the BP fail list
// itself when in_list is null) and appends the ones it cannot finish to out_list.
#define QD_OSD_KPT 24     // monotone keys a thread keeps in registers (covers n <= 24 * T; larger windows re-read the LLRs)
template <int T>
__global__ void __launch_bounds__(T, T / 256) qd_osd0_fast_kernel(OsdGraphDev g, BpGraphDev bg, DecodeArgs a, int cap,
                                                                  const int32_t *in_list, const int32_t *in_count,
                                                                  int32_t *out_list, int32_t *out_count)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int nlist = *in_count;
    OsdLds S;
    qd_osd_carve(smem, g.f_off, S);
    uint64_t *sortbuf = reinterpret_cast<uint64_t *>(smem + g.f_off_sort);     // [QD_OSD_FAST_CAP], overlaps the Q planes
    uint16_t *order = reinterpret_cast<uint16_t *>(smem + g.f_off_order);      // [QD_OSD_FAST_CAP]
    uint32_t *red = S.red;
    const bool in_regs = g.n <= QD_OSD_KPT * T;
    for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
        const int slot = in_list ? in_list[li] : li;
        const int64_t shot = a.fail_list[slot];
        const float *llr = a.llr_ws + (int64_t)slot * bg.n_pad;
#ifdef QD_OSD_TIMING
        unsigned long long *a_dbg = a.dbg;
        unsigned long long tick_ = wall_clock64();
#endif
        uint32_t kreg[QD_OSD_KPT];
        if (in_regs) {
#pragma unroll
            for (int i = 0; i < QD_OSD_KPT; ++i) {
                const int b = tid + i * T;
                kreg[i] = (b < g.n) ? qd_mono_key(llr[b]) : 0xFFFFFFFFu;
            }
        }
        // ---- 1a. hi = largest number of leading key bins (top 12 bits) whose population fits `cap`.
        //      Bisection on counts: 12 rounds of (compare, wave sum, one LDS add per wave, one barrier).
        uint32_t lo_b = 0, hi_b = 4096;                      // invariant: count(bin < lo_b) <= cap
        if (tid < 16) red[64 + tid] = 0u;
        __syncthreads();
        for (int it = 0; it < 12; ++it) {
            const uint32_t mid = (lo_b + hi_b + 1) >> 1;
            uint32_t c = 0;
            if (in_regs) {
#pragma unroll
                for (int i = 0; i < QD_OSD_KPT; ++i) c += ((kreg[i] >> 20) < mid) ? 1u : 0u;
            } else {
                for (int b = tid; b < g.n; b += T) c += ((qd_mono_key(llr[b]) >> 20) < mid) ? 1u : 0u;
            }
            c = qd_wave_add(c);
            if ((tid & 63) == 0) atomicAdd(&red[64 + it], c);
            __syncthreads();
            if (red[64 + it] <= (uint32_t)cap) lo_b = mid; else hi_b = mid - 1;
        }
        const uint32_t hi = lo_b;
        QD_TICK(0)
        // ---- 1b. gather and sort the head (one LDS counter bump per wavefront)
        if (tid == 0) red[80] = 0u;
        __syncthreads();
        for (int i = 0; i * T < g.n; ++i) {
            const int b = tid + i * T;
            uint32_t u = 0xFFFFFFFFu;
            if (in_regs) {
                // kreg is indexed with a compile-time constant only when the loop is unrolled; re-derive instead
                u = (b < g.n) ? qd_mono_key(llr[b]) : 0xFFFFFFFFu;
            } else if (b < g.n) u = qd_mono_key(llr[b]);
            const bool take = (b < g.n) && (u >> 20) < hi;
            const unsigned long long bal = __ballot(take);
            uint32_t wbase = 0;
            if ((tid & 63) == 0 && bal) wbase = atomicAdd(&red[80], (uint32_t)__popcll(bal));
            wbase = (uint32_t)__shfl((int)wbase, 0);
            if (take) sortbuf[wbase + (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull))] = ((uint64_t)u << 32) | bg.bit_orig[b];
        }
        __syncthreads();
        const int cnt = (int)red[80];
        int P = 64;
        while (P < cnt) P <<= 1;
        for (int i = cnt + tid; i < P; i += T) sortbuf[i] = ~0ull;
        __syncthreads();
        QD_TICK(1)
        qd_bitonic_u64<T>(sortbuf, P, tid);
        for (int i = tid; i < cnt; i += T) order[i] = (uint16_t)(sortbuf[i] & 0xFFFFu);
        __syncthreads();
        QD_TICK(2)
        // ---- 2./3. elimination on the head
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
        int npiv = 0, inconsistent = 0;
#ifdef QD_OSD_TIMING
        const int rc = qd_osd_eliminate<T, false>(g, S, nullptr, g.f_kw, g.f_kw * 64, order, cnt, det, upd, a.upd_rows,
                                                  bg.out_words, &npiv, &inconsistent, a_dbg);
        tick_ = wall_clock64();
#else
        const int rc = qd_osd_eliminate<T, false>(g, S, nullptr, g.f_kw, g.f_kw * 64, order, cnt, det, upd, a.upd_rows,
                                                  bg.out_words, &npiv, &inconsistent);
#endif
#ifdef QD_OSD_TIMING
        if (tid == 0) { atomicAdd(&a_dbg[8], 1ull); atomicAdd(&a_dbg[9], (unsigned long long)npiv); atomicAdd(&a_dbg[10], (unsigned long long)cnt); atomicAdd(&a_dbg[11], (unsigned long long)rc); }
#endif
        if (rc) {
            if (tid == 0) out_list[atomicAdd(out_count, 1)] = slot;
        } else {
            for (int w = tid; w < bg.out_words; w += T) a.err_bits[shot * bg.out_words + w] = S.outw[w];
            if (tid == 0) a.status[shot] = (a.status[shot] & 0xFFFF) | (1 << 17) | (inconsistent ? (1 << 18) : 0) | (min(npiv, 4095) << 20);
        }
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

// Three passes over ever shorter lists: head of 512 columns, head of 2048 columns, everything.
template <int TF, int T>
static hipError_t launch_osd_t(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks_fast,
                               int blocks_full, hipStream_t s)
{
    hipError_t e;
    auto kf = qd_osd0_full_kernel<T>;
    e = hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    if (g.f_lds_bytes > 0) {
        auto kq = qd_osd0_fast_kernel<TF>;
        e = hipFuncSetAttribute((const void *)kq, hipFuncAttributeMaxDynamicSharedMemorySize, g.f_lds_bytes);
        if (e != hipSuccess) return e;
        int32_t *cnt = a.hard_count;            // [0] after pass 1, [1] after pass 2
        hipLaunchKernelGGL(kq, dim3((unsigned)blocks_fast), dim3(TF), g.f_lds_bytes, s, g, bg, a, 512,
                           (const int32_t *)nullptr, (const int32_t *)a.fail_count, a.hard_list, cnt);
        hipLaunchKernelGGL(kq, dim3((unsigned)blocks_fast), dim3(TF), g.f_lds_bytes, s, g, bg, a, QD_OSD_FAST_CAP,
                           (const int32_t *)a.hard_list, (const int32_t *)cnt, a.hard_list2, cnt + 1);
        hipLaunchKernelGGL(kf, dim3((unsigned)blocks_full), dim3(T), g.lds_bytes, s, g, bg, a, (const int32_t *)a.hard_list2,
                           (const int32_t *)(cnt + 1));
    } else {
        hipLaunchKernelGGL(kf, dim3((unsigned)blocks_full), dim3(T), g.lds_bytes, s, g, bg, a, (const int32_t *)nullptr,
                           (const int32_t *)a.fail_count);
    }
    return hipGetLastError();
}

hipError_t qd_launch_osd0(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks_fast,
                          int blocks_full, hipStream_t s)
{
    switch (g.threads) {
    case 256: return launch_osd_t<256, 256>(g, bg, a, blocks_fast, blocks_full, s);
    case 512: return launch_osd_t<512, 512>(g, bg, a, blocks_fast, blocks_full, s);
    default: return launch_osd_t<512, 1024>(g, bg, a, blocks_fast, blocks_full, s);
    }
}
