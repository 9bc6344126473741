// osd_shared.h -- pieces shared by the OSD kernels (osd_kernels.hip: row-form / column-form kernels; osd_sr.hip: the
// OSD-0 kernel with simultaneous singleton pivots): key order, workgroup sums, the lazily drawn column order.
#pragma once
#include "qd_internal.h"

#define QD_NOKEY 0xFFFFFFFFu

#ifdef QD_OSD_TIMING
// phase timers accumulate in registers and are flushed once per shot (a global atomic per tick would stall every
// following barrier on vmcnt(0) and measure itself)
#define QD_TICK(slot)                                                                   \
    {                                                                                   \
        const unsigned long long now_ = wall_clock64();                                 \
        acc_[slot] += now_ - tick_;                                                     \
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

// One kernel, no restarts: the column order is produced lazily in TIERS of <= QD_OSD_TIER columns (the next-smallest keys,
// found by bisecting on the key value), each tier is bitonic-sorted and eliminated before the next one is drawn; the
// elimination state survives between tiers.  A thread owns RPT rows and keeps their batch image, syndrome bit, pivot flag
// and the first QD_OSD_KWR Q planes in registers; LDS holds a write-through mirror that other threads read when one of
// those rows becomes the pivot row.  Q planes beyond the LDS budget spill to HBM (rare: > 64 * f_kw pivots).
#define QD_OSD_TIER 1024
#ifndef QD_OSD_FULL_PAIRS
#define QD_OSD_FULL_PAIRS 1   // full-rank elimination: loop over the (column, pivot) incidences (the row-wise form, Q rows being dense by then, measured 30 % slower)
#endif
#ifndef QD_OSD_TIER_FIRST
#define QD_OSD_TIER_FIRST 256
#endif
#define QD_OSD_KWR 6
#ifndef QD_OSD_KWR0
#define QD_OSD_KWR0 2
#endif
#ifndef QD_OSD_KPT
#define QD_OSD_KPT 1      // (20, i.e. every key of the headline window held in registers, was the round-1 choice: see qd_osd_draw_tier)
#endif
//      QD_OSD_KPT        // monotone keys a thread keeps in registers while a tier is drawn (windows with n <= KPT * T; else re-read)

// Sum of v over the workgroup; one barrier; `buf` = 2 x 64 words alternating with `phase` (entries beyond the wave count must be zero).
template <int T>
__device__ __forceinline__ uint32_t qd_block_sum(uint32_t v, uint32_t *buf, int &phase)
{
    constexpr int NW = T / 64;
    v = qd_wave_add(v);
    if ((threadIdx.x & 63) == 0) buf[phase * 64 + (threadIdx.x >> 6)] = v;     // 64-word stride: the 3-way counter of the tier search shares these buffers
    __syncthreads();
    const uint4 *p4 = reinterpret_cast<const uint4 *>(buf + phase * 64);
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < (NW + 3) / 4; ++w) { const uint4 x = p4[w]; tot += x.x + x.y + x.z + x.w; }
    phase ^= 1;
    return tot;
}

// Only what this kernel needs, so the scalar registers are not flooded with three full descriptor structs.
struct OsdRegArgs {
    int m, n, m_pad, n_pad, max_cdeg, mw, f_kw, out_words, upd_rows;
    int off[10], off_sort, off_order, off_pivmask, off_npl;
    int osd_w;                  // 0 = OSD-0 (early stop), 1 = combination sweep, 2 = exhaustive
    int osd_order, rank;
    const uint32_t *wfix;       // [n] round(log(1/p_j) * 2^18): candidate cost per fault
    const uint32_t *bit_slot_of;// [n] fault -> bit slot (row of llr_ws)
    const uint32_t *csc_ptr;
    const uint16_t *csc_row;
    const uint32_t *bit_orig;
    const uint8_t *det, *upd;
    int64_t det_stride, det_offset, upd_stride;
    const float *llr_ws;
    const int32_t *fail_list, *fail_count;
    const int32_t *slot_list, *slot_count;      // when set: only these fail-list slots (a subset handed over by an earlier pass)
    uint64_t *q_spill_fast;
    uint64_t *mt_ws;
    uint32_t *err_bits;
    int32_t *status;
    unsigned long long *dbg;
};

struct TierState {
    uint32_t lo_key, lo_idx; int sphase, exhausted, limit;    // limit: tier size wanted (<= QD_OSD_TIER)
    // guess (in, 0: none): an upper key bound to try first -- every key in [lo_key, guess) is gathered in ONE pass over the posteriors; if that is between
    // 1 and `limit` columns it IS the tier (any cut gives a valid tier: the elimination depends on the column order, not on where tiers end), otherwise the
    // radix selection below runs as before.  A tier's cut moves little from shot to shot, so the caller carries it (cut / count, out).
    uint32_t guess = 0u, cut = 0u, count = 0u;
};

// Draws the next tier: the <= QD_OSD_TIER not yet consumed columns with the smallest (key, fault index), sorted, as fault
// indices in order[0..cnt).  State: every column with (key, index) < (lo_key, lo_idx) has been consumed.
#ifndef QD_OSD_TIER_INLINE
#define QD_OSD_TIER_INLINE __forceinline__
#endif
// KPT = monotone keys a thread keeps in registers while the tier is drawn (used when n <= KPT * T; otherwise, and always with
// KPT = 1 on windows of more than T faults, every radix level and the gather re-read the posteriors, which sit in L2).
template <int T, int KPT = QD_OSD_KPT, class ARGS = OsdRegArgs, int SCAN = 0>     // ARGS: anything with n and bit_orig; SCAN: 16-byte loads in flight per pass over the posteriors (0: one float at a time)
__device__ QD_OSD_TIER_INLINE int qd_osd_draw_tier(const ARGS &a, const float *llr, uint64_t *sortbuf, uint16_t *order,
                                             uint32_t *red, uint32_t *sumbuf, TierState &ts)
{
    const int tid = threadIdx.x;
    const int n = a.n;
    const bool in_regs = n <= KPT * T;
    uint32_t lo_key = ts.lo_key, lo_idx = ts.lo_idx;
    int sphase = ts.sphase;
    const uint32_t lim = (uint32_t)ts.limit;
    bool exhausted = false;
    int cnt = 0;
    {
            uint32_t kreg[KPT];
            if (in_regs) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const int b = tid + i * T;
                    kreg[i] = (b < n) ? qd_mono_key(llr[b]) : 0xFFFFFFFFu;
                }
            }
            // One pass over the posteriors (L2).  SCAN > 0: 16-byte loads, SCAN of them in flight per thread before the first is used -- a
            // plain loop waits for every load in turn (~19 L2 round trips per pass at the headline window: most of what a tier used to
            // cost).  The register kernels of osd_kernels.hip keep the plain loop: they are at their register budget already.
            auto scan_keys = [&](auto fn) {                                    // fn(monotone key, bit slot)
                if constexpr (SCAN == 0) {
                    for (int b = tid; b < n; b += T) fn(qd_mono_key(llr[b]), b);
                } else {
                    const float4 *l4 = reinterpret_cast<const float4 *>(llr);  // rows of llr_ws start on 256-byte boundaries (n_pad = 64 k)
                    const int n4 = n >> 2;
                    for (int b0 = 0; b0 < n4; b0 += SCAN * T) {
                        float4 v[SCAN > 0 ? SCAN : 1];
#pragma unroll
                        for (int u = 0; u < SCAN; ++u) {
                            const int b4 = b0 + u * T + tid;
                            v[u] = l4[b4 < n4 ? b4 : 0];
                        }
#pragma unroll
                        for (int u = 0; u < SCAN; ++u) {
                            const int b4 = b0 + u * T + tid;
                            if (b4 < n4) {
                                fn(qd_mono_key(v[u].x), 4 * b4); fn(qd_mono_key(v[u].y), 4 * b4 + 1);
                                fn(qd_mono_key(v[u].z), 4 * b4 + 2); fn(qd_mono_key(v[u].w), 4 * b4 + 3);
                            }
                        }
                    }
                    for (int b = (n4 << 2) + tid; b < n; b += T) fn(qd_mono_key(llr[b]), b);
                }
            };
            // tie group at key == lo_key that is only partly consumed (or too big for one tier): ordered by fault index;
            // rare, so it simply re-reads the LLRs and fault indices
            auto count_ties = [&](uint32_t key, uint32_t ilo, uint32_t ihi) -> uint32_t {
                uint32_t c = 0;
                for (int b = tid; b < n; b += T)
                    if (qd_mono_key(llr[b]) == key) { const uint32_t j = a.bit_orig[b]; c += (j >= ilo && j < ihi) ? 1u : 0u; }
                return qd_block_sum<T>(c, sumbuf, sphase);
            };
            bool by_index = false;
            uint32_t t_lo = lo_key, t_hi = lo_key, i_lo = 0, i_hi = 0;
            if (lo_idx > 0) {
                if (count_ties(lo_key, lo_idx, 0xFFFFFFFFu) > 0) by_index = true;
                else { lo_key += 1; lo_idx = 0; }
            }
            // the guess first (TierState::guess): the gather below runs with [lo_key, guess) as the cut; if that is not 1..lim columns the selection runs and
            // the gather is repeated (one instance of the pass over the posteriors serves both)
            bool spec = !by_index && !in_regs && ts.guess > lo_key;
            for (;;) {
            if (spec) { t_lo = lo_key; t_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)ts.guess); }
            else {
            if (!by_index) {
                // largest bin boundary t_hi with  #{lo_key <= key < t_hi} <= lim   (key 0xFFFFFFFF is reserved for "no
                // column").  Radix selection: a histogram of the keys over 2048 bins of 2^21 (one LDS atomic per key), a scan
                // for the bin where the running count passes `lim`; when the cut in front of that bin would leave a thin tier,
                // the bin itself is resolved with 2048 bins of 2^10, then 1024 bins of 1.
                uint32_t *hist = reinterpret_cast<uint32_t *>(sortbuf);            // 8 KB, free until the gather
                uint32_t base = 0u, cur_lo = lo_key, cum = 0u;
                uint64_t thi64 = 0;
                for (int level = 0; level < 3; ++level) {
                    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
                    const int nb = level == 2 ? 1024 : 2048;
                    for (int i = tid; i < nb; i += T) hist[i] = 0u;
                    __syncthreads();
                    auto tally = [&](uint32_t u) {
                        if (u != 0xFFFFFFFFu && u >= cur_lo) {
                            const uint32_t bin = (u - base) >> shift;              // cur_lo >= base
                            if (bin < (uint32_t)nb) atomicAdd(&hist[bin], 1u);
                        }
                    };
                    if (in_regs) {
#pragma unroll
                        for (int i = 0; i < KPT; ++i) tally(kreg[i]);
                    } else {
                        scan_keys([&](uint32_t u, int) { tally(u); });
                    }
                    __syncthreads();
                    // every thread owns nb / T consecutive bins; exclusive prefix over the workgroup
                    const int per = nb / T;                                        // 2..8
                    uint32_t h[8], mysum = 0u;
#pragma unroll
                    for (int q = 0; q < 8; ++q) { h[q] = (q < per) ? hist[tid * per + q] : 0u; mysum += h[q]; }
                    uint32_t incl = mysum;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if ((tid & 63) >= d) incl += o; }
                    uint32_t *buf = sumbuf + sphase * 64;
                    if ((tid & 63) == 63) buf[tid >> 6] = incl;
                    if (tid == 0) { buf[32] = 0xFFFFFFFFu; }                       // first overflowing bin (minimum over threads)
                    __syncthreads();
                    uint32_t before = cum;
                    for (int w = 0; w < (tid >> 6); ++w) before += buf[w];
                    uint32_t run = before + incl - mysum;                          // keys in [lo_key, first key of my first bin)
                    int kk = -1;
                    uint32_t run_at = 0u;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < per && kk < 0) { if (run + h[q] > lim) { kk = tid * per + q; run_at = run; } else run += h[q]; }
                    if (kk >= 0) atomicMin(&buf[32], (uint32_t)kk);
                    __syncthreads();
                    const uint32_t kmin = buf[32];
                    if (kk >= 0 && (uint32_t)kk == kmin) buf[33] = run_at;         // exactly one thread owns that bin
                    if (kmin == 0xFFFFFFFFu && tid == T - 1) buf[33] = run;        // nothing overflows: everything fits
                    __syncthreads();
                    const uint32_t k = (kmin == 0xFFFFFFFFu) ? (uint32_t)nb : kmin;
                    const uint32_t cnt_here = buf[33];
                    sphase ^= 1;
                    thi64 = (uint64_t)base + ((uint64_t)k << shift);
                    cnt = (int)cnt_here;
                    if (k == (uint32_t)nb || cnt_here >= lim / 4u || level == 2) break;
                    // descend into the overflowing bin
                    base = (uint32_t)thi64; cum = cnt_here; cur_lo = max(cur_lo, base);
                }
                __syncthreads();                                                   // hist (= sortbuf) is reused by the gather
                t_lo = lo_key; t_hi = thi64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thi64;
                if (cnt > 0) { ts.cut = t_hi; ts.count = (uint32_t)cnt; }
                if (cnt == 0) {
                    if (t_hi == 0xFFFFFFFFu) exhausted = true;                 // nothing at or above lo_key
                    else { lo_key = t_hi; lo_idx = 0; by_index = true; }       // more than `lim` columns share the next key value
                } else lo_key = t_hi;
            }
            if (by_index) {
                uint64_t L = lo_idx, H = (uint64_t)n;
                uint32_t cntL = 0;
                while (L < H) {
                    const uint64_t mid = (L + H + 1) >> 1;
                    const uint32_t c = count_ties(lo_key, lo_idx, (uint32_t)mid);
                    if (c <= lim) { L = mid; cntL = c; } else H = mid - 1;
                }
                i_lo = lo_idx; i_hi = (uint32_t)L;
                cnt = (int)cntL;
                t_lo = lo_key;
                if (i_hi >= (uint32_t)n) { lo_key += 1; lo_idx = 0; } else lo_idx = i_hi;
            }
            if (exhausted) { ts.lo_key = lo_key; ts.lo_idx = lo_idx; ts.sphase = sphase; ts.exhausted = 1; return 0; }
            }
            // gather and sort the tier
            if (tid == 0) red[80] = 0u;
            __syncthreads();
            if (in_regs && !by_index) {
                // keys are in registers: count the takes, one workgroup scan for the write offsets (order inside the tier
                // buffer is irrelevant, it is sorted next)
                uint32_t mask = 0u;
#pragma unroll
                for (int i = 0; i < KPT; ++i) mask |= (kreg[i] >= t_lo && kreg[i] < t_hi) ? (1u << i) : 0u;
                const uint32_t mine = (uint32_t)__popc(mask);
                uint32_t incl = mine;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if ((tid & 63) >= d) incl += o; }
                uint32_t *buf = sumbuf + sphase * 64;
                if ((tid & 63) == 63) buf[tid >> 6] = incl;
                __syncthreads();
                uint32_t at = incl - mine;
                for (int w = 0; w < (tid >> 6); ++w) at += buf[w];
                sphase ^= 1;
#pragma unroll
                for (int i = 0; i < KPT; ++i)
                    if ((mask >> i) & 1u) sortbuf[at++] = ((uint64_t)kreg[i] << 32) | a.bit_orig[tid + i * T];
            } else
            scan_keys([&](uint32_t u, int b) {
                // (one LDS counter bump per taken column: a tier takes a few hundred of the n columns)
                bool take = false;
                uint32_t j = 0;
                if (by_index) { if (u == t_lo) { j = a.bit_orig[b]; take = (j >= i_lo && j < i_hi); } }
                else if (u >= t_lo && u < t_hi) { j = a.bit_orig[b]; take = true; }
                if (take) {
                    const uint32_t at = atomicAdd(&red[80], 1u);
                    if (at < (uint32_t)QD_OSD_TIER) sortbuf[at] = ((uint64_t)u << 32) | j;      // (a guess may overflow the buffer; a selected cut never does)
                }
            });
            __syncthreads();
            if (!spec) break;
            {
                const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)red[80]);
                ts.count = c;
                if (c > 0u && c <= lim) { cnt = (int)c; lo_key = t_hi; ts.cut = t_hi; break; }
                spec = false;                                                      // no column or too many: select, gather again
                __syncthreads();
            }
            }
            int P = 64;
            while (P < cnt) P <<= 1;
            for (int i = cnt + tid; i < P; i += T) sortbuf[i] = ~0ull;
            __syncthreads();
            if (cnt <= 256 && cnt <= T) {
                // small tier: rank by counting (the keys are distinct), one pass of broadcast reads instead of 36 barrier stages
                if ((tid & ~63) < cnt) {
                    const uint64_t mine = tid < cnt ? sortbuf[tid] : ~0ull;
                    const uint4 *sb4 = reinterpret_cast<const uint4 *>(sortbuf);
                    int rank = 0;
                    for (int i = 0; i < (cnt + 1) / 2; ++i) {                 // sortbuf[cnt] is padding (~0) when cnt is odd
                        const uint4 v = sb4[i];
                        const uint64_t k0 = (uint64_t)v.x | ((uint64_t)v.y << 32), k1 = (uint64_t)v.z | ((uint64_t)v.w << 32);
                        rank += (k0 < mine ? 1 : 0) + (k1 < mine ? 1 : 0);
                    }
                    if (tid < cnt) order[rank] = (uint16_t)(mine & 0xFFFFu);
                }
            } else {
                qd_bitonic_u64<T>(sortbuf, P, tid);
                for (int i = tid; i < cnt; i += T) order[i] = (uint16_t)(sortbuf[i] & 0xFFFFu);
            }
            __syncthreads();
    }
    ts.lo_key = lo_key; ts.lo_idx = lo_idx; ts.sphase = sphase; ts.exhausted = 0;
    return cnt;
}
