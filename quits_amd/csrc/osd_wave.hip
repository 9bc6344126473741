// osd_wave.hip -- OSD-0 for the common case, one WAVEFRONT per shot.
//
// Replaces ldpc.BpOsdDecoder.decode -> OsdDecoder::decode (osd.hpp) with osd_method = OSD_0, like osd_kernels.hip, and
// produces the same bits (same column order, same pivots, same early stop; oracle: oq_osd0 in oracle/qd_oracle.c).  At the
// operating points of interest OSD-0 stops after ~100 pivots found among the ~130 most suspicious columns.  That is a chain
// of a hundred small dependent steps: the workgroup-per-shot kernel (qd_osd0_reg_kernel) keeps two shots per CU in flight and
// most of its 512 threads wait at a barrier per pivot.  Here a shot gets ONE wavefront and ~33 KB of LDS, so five shots run
// per CU and nobody waits for anybody:
//   1. the PREFIX of the column order: the largest key threshold T with at most `cap` (256) columns below it, found by radix
//      selection (LDS histograms of 2^21-, 2^10-, 1-wide bins); every column with key < T is gathered and sorted by
//      (key, fault index) -- so the prefix is exactly the head of ldpc's order, ties included;
//   2. elimination column by column in T-form (Q[r] = pivot rows added to row r, two 64-bit planes in LDS: up to 128 pivots).
//      A column none of whose rows is a pivot row yet is its own image: its first row is the pivot, its other <= 5 rows are
//      updated -- a few dozen instructions.  Otherwise the image needs Q: every lane scans its 16 rows once;
//   3. the syndrome bit of every row lives in a lane register; after each pivot one ballot tells whether it has vanished on
//      all unpivoted rows (exact early stop).
// A shot that needs more than the prefix or more than 128 pivots is appended to `hard_list` and redone from scratch by
// qd_osd0_reg_kernel, which handles everything.
#include "qd_internal.h"
#include "../../include/quits_amd.h"
#include <algorithm>

#ifndef QW_DBG_MODE
#define QW_DBG_MODE 1
#endif
#ifdef QD_OSD_TIMING
#define QW_TICK(i) { const unsigned long long now_ = wall_clock64(); wacc_[i] += now_ - wtick_; wtick_ = now_; }
#else
#define QW_TICK(i)
#endif
#define QW_NOKEY 0xFFFFFFFFu
#define QW_PLANES 2
#define QW_MAXPIV (64 * QW_PLANES)
#define QW_SLOTS 32                 // rows per lane: windows up to 2048 detectors
#define QW_MLP 16                   // posteriors a lane requests before it uses the first (the passes over the n keys are latency-bound otherwise)

struct OsdWaveArgs {
    int m, n, m_pad, n_pad, max_cdeg, out_words, upd_rows, cap;
    const uint32_t *csc_ptr;
    const uint16_t *csc_row;
    const uint32_t *bit_orig;      // posterior column (bit slot) -> fault
    const uint8_t *det, *upd;
    int64_t det_stride, det_offset, upd_stride;
    const float *llr_ws;
    const int32_t *fail_list, *fail_count;
    int32_t *hard_list, *hard_count;
    uint32_t *err_bits;
    int32_t *status;
    unsigned long long *dbg;       // phase counters of -DQD_OSD_TIMING builds (tools/osd_timing.py)
    int off_q, off_sort, off_cols, off_rowpiv, off_prow, off_pcol, off_raw, off_sp, off_out, off_misc;
};

__device__ __forceinline__ uint32_t qw_mono_key(float llr)
{
    const float f = llr + 0.0f;                    // -0 -> +0, so that +-0 tie on the index like the oracle's '<'
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint32_t qw_wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v += o; }
    return v;
}

template <int SLOTS>
__global__ void __launch_bounds__(64) qd_osd0_wave_kernel(OsdWaveArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *Q = reinterpret_cast<uint64_t *>(smem + a.off_q);              // [m_pad][2]
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + a.off_q);           // 2048 bins, before the elimination starts
    uint64_t *sortbuf = reinterpret_cast<uint64_t *>(smem + a.off_sort);     // [cap]  key << 32 | fault
    uint16_t *colrows = reinterpret_cast<uint16_t *>(smem + a.off_cols);     // [cap][max_cdeg]
    int16_t *rowpiv = reinterpret_cast<int16_t *>(smem + a.off_rowpiv);
    uint16_t *prow = reinterpret_cast<uint16_t *>(smem + a.off_prow);        // [QW_MAXPIV]
    uint32_t *pcol = reinterpret_cast<uint32_t *>(smem + a.off_pcol);        // [QW_MAXPIV]
    uint8_t *rawf = smem + a.off_raw;
    uint8_t *spl = smem + a.off_sp;
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + a.off_out);
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + a.off_misc);        // [0] counter, [1] first overflowing bin, [2] count before it
    const int lane = threadIdx.x;
    const int m = a.m, n = a.n, cdeg = a.max_cdeg;
    const int nfail = *a.fail_count;
    const uint32_t lim = (uint32_t)a.cap;

    for (int slot = blockIdx.x; slot < nfail; slot += gridDim.x) {
        const int64_t shot = a.fail_list[slot];
        const float *llr = a.llr_ws + (int64_t)slot * a.n_pad;
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;

#ifdef QD_OSD_TIMING
        unsigned long long wacc_[4] = {0, 0, 0, 0}, wtick_ = wall_clock64();
        int ncols_ = 0, ngen_ = 0;
#endif
        // ---- 1. threshold: the largest T with #{key < T} <= lim
        uint32_t base = 0u, cur_lo = 0u, cum = 0u, cnt = 0u;
        uint64_t thi = 0ull;
        for (int level = 0; level < 3; ++level) {
            const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
            const int nb = level == 2 ? 1024 : 2048;
            for (int i = lane; i < nb; i += 64) hist[i] = 0u;
            if (lane == 0) { misc[1] = QW_NOKEY; misc[2] = 0u; }
            __syncthreads();
            for (int b0 = 0; b0 < n; b0 += 64 * QW_MLP) {                     // QW_MLP independent loads in flight per lane
                float v[QW_MLP];
#pragma unroll
                for (int x = 0; x < QW_MLP; ++x) { const int b = b0 + 64 * x + lane; v[x] = b < n ? llr[b] : 0.f; }
#pragma unroll
                for (int x = 0; x < QW_MLP; ++x) {
                    const uint32_t u = qw_mono_key(v[x]);
                    if (b0 + 64 * x + lane < n && u != QW_NOKEY && u >= cur_lo) {
                        const uint32_t bin = (u - base) >> shift;
                        if (bin < (uint32_t)nb) atomicAdd(&hist[bin], 1u);
                    }
                }
            }
            __syncthreads();
            const int per = nb / 64;                                           // 32 or 16 consecutive bins per lane
            uint32_t mysum = 0u;
            for (int q = 0; q < per; ++q) mysum += hist[lane * per + q];
            const uint32_t incl = qw_wave_incl_scan(mysum, lane);
            uint32_t run = cum + incl - mysum;                                 // keys in [lo, first key of my first bin)
            int kk = -1;
            uint32_t run_at = 0u;
            for (int q = 0; q < per && kk < 0; ++q) {
                const uint32_t h = hist[lane * per + q];
                if (run + h > lim) { kk = lane * per + q; run_at = run; } else run += h;
            }
            if (kk >= 0) atomicMin(&misc[1], (uint32_t)kk);
            __syncthreads();
            const uint32_t kmin = misc[1];
            if (kk >= 0 && (uint32_t)kk == kmin) misc[2] = run_at;             // exactly one lane owns that bin
            if (kmin == QW_NOKEY && lane == 63) misc[2] = run;                 // nothing overflows: everything fits
            __syncthreads();
            const uint32_t k = (kmin == QW_NOKEY) ? (uint32_t)nb : kmin;
            cnt = misc[2];
            thi = (uint64_t)base + ((uint64_t)k << shift);
            __syncthreads();
            if (k == (uint32_t)nb || cnt >= lim - lim / 4u || level == 2) break;
            base = (uint32_t)thi; cum = cnt; cur_lo = base;                    // descend into the overflowing bin
        }
        QW_TICK(0)
        const uint32_t T = thi > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thi;
        bool give_up = (cnt == 0u);                                            // more than `lim` columns share the lowest key value

        // ---- 2. gather + sort the prefix, fetch its columns
        int P = 64;
        if (!give_up) {
            uint32_t wbase = 0u;                                               // uniform running offset into the prefix buffer
            for (int b0 = 0; b0 < n; b0 += 64 * QW_MLP) {
                float v[QW_MLP];
#pragma unroll
                for (int x = 0; x < QW_MLP; ++x) { const int b = b0 + 64 * x + lane; v[x] = b < n ? llr[b] : 0.f; }
#pragma unroll
                for (int x = 0; x < QW_MLP; ++x) {
                    const int b = b0 + 64 * x + lane;
                    const uint32_t u = qw_mono_key(v[x]);
                    const bool take = b < n && u < T;
                    const unsigned long long bal = __ballot(take);
                    if (take) sortbuf[wbase + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = ((uint64_t)u << 32) | a.bit_orig[b];
                    wbase += (uint32_t)__popcll(bal);
                }
            }
            __syncthreads();
            while (P < (int)cnt) P <<= 1;
            for (int i = (int)cnt + lane; i < P; i += 64) sortbuf[i] = ~0ull;
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int pi = lane; pi < (P >> 1); pi += 64) {
                        const int i = ((pi & ~(j - 1)) << 1) | (pi & (j - 1));
                        const int l = i | j;
                        const uint64_t x = sortbuf[i], y = sortbuf[l];
                        const bool up = ((i & k) == 0);
                        if ((x > y) == up) { sortbuf[i] = y; sortbuf[l] = x; }
                    }
                    __syncthreads();
                }
            for (int x0 = 0; x0 < (int)cnt * cdeg; x0 += 64 * 8) {             // eight independent (pointer, row) chains per lane
                uint32_t e0[8], e1[8];
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    const int x = x0 + 64 * y + lane;
                    e0[y] = 0u; e1[y] = 0u;
                    if (x < (int)cnt * cdeg) {
                        const uint32_t col = (uint32_t)(sortbuf[x / cdeg] & 0xFFFFFFFFull);
                        e0[y] = a.csc_ptr[col]; e1[y] = a.csc_ptr[col + 1];
                    }
                }
                uint16_t rw[8];
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    const int x = x0 + 64 * y + lane, q = x % cdeg;
                    rw[y] = (x < (int)cnt * cdeg && e0[y] + q < e1[y]) ? a.csc_row[e0[y] + q] : (uint16_t)0xFFFFu;
                }
#pragma unroll
                for (int y = 0; y < 8; ++y) { const int x = x0 + 64 * y + lane; if (x < (int)cnt * cdeg) colrows[x] = rw[y]; }
            }
        }
        __syncthreads();

        QW_TICK(1)
        // ---- 3. elimination state
        uint32_t spb = 0u, pivb = 0u;                                          // bit s: row lane + 64 s
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int r = lane + 64 * s;
            if (r < m) {
                uint32_t sb = det[r] & 1u;
                if (upd && r < a.upd_rows) sb ^= upd[r] & 1u;
                spb |= sb << s;
                rowpiv[r] = -1; rawf[r] = 0;
                Q[2 * r] = 0ull; Q[2 * r + 1] = 0ull;
            } else
                pivb |= 1u << s;                                               // rows beyond m never qualify
        }
        for (int w = lane; w < a.out_words; w += 64) outw[w] = 0u;
        __syncthreads();

        int npiv = 0;
        bool done = false;
        for (int c = 0; c < (int)cnt && !done && !give_up; ++c) {
            const uint32_t myrow = lane < cdeg ? (uint32_t)colrows[c * cdeg + lane] : 0xFFFFu;
            const bool valid = myrow != 0xFFFFu;
            const int mypk = valid ? (int)rowpiv[myrow] : -1;
            const unsigned long long bv = __ballot(valid), bp = __ballot(valid && mypk >= 0);
            const int deg = (int)__popcll(bv);                                 // rows are packed into the first lanes, ascending
#ifdef QD_OSD_TIMING
            ++ncols_; ngen_ += bp != 0ull;
#endif
            int p = -1;
            uint32_t tmask = 0u;                                               // general path: my rows that hold the column's image
            if (bp == 0ull) {
                p = __builtin_amdgcn_readlane((int)myrow, 0);                  // the column is its own image: lowest row = pivot
            } else {
                uint64_t mk0 = 0ull, mk1 = 0ull;
                for (unsigned long long bb = bp; bb; bb &= bb - 1ull) {
                    const int pk = __builtin_amdgcn_readlane(mypk, (int)__builtin_ctzll(bb));
                    if (pk < 64) mk0 |= 1ull << pk; else mk1 |= 1ull << (pk - 64);
                }
                if (valid) rawf[myrow] = 1;
                __syncthreads();
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int r = lane + 64 * s;
                    if (r < m) {
                        const uint32_t t = (uint32_t)rawf[r] ^ ((uint32_t)(__popcll(Q[2 * r] & mk0) + __popcll(Q[2 * r + 1] & mk1)) & 1u);
                        tmask |= t << s;
                    }
                }
                const uint32_t cand = tmask & ~pivb;
                uint32_t key = cand ? (((uint32_t)__builtin_ctz(cand) << 6) | (uint32_t)lane) : QW_NOKEY;
                key = qd_wave_umin(key);
                __syncthreads();
                if (valid) rawf[myrow] = 0;
                if (key != QW_NOKEY) p = (int)(((key & 63u)) + 64u * (key >> 6));       // row = lane + 64 * slot
            }
            if (p < 0) { __syncthreads(); continue; }                          // depends on earlier pivots
            if (npiv == QW_MAXPIV) { give_up = true; break; }
            const int K = npiv, ol = p & 63, os = p >> 6;
            const uint64_t qp0 = Q[2 * p] ^ (K < 64 ? (1ull << K) : 0ull), qp1 = Q[2 * p + 1] ^ (K < 64 ? 0ull : (1ull << (K - 64)));
            const uint32_t spp = ((uint32_t)__builtin_amdgcn_readlane((int)spb, ol) >> os) & 1u;
            __syncthreads();                                                   // Q[p] read by everybody before anybody writes
            if (bp == 0ull) {
                for (int x = 1; x < deg; ++x) {
                    const int r = __builtin_amdgcn_readlane((int)myrow, x);
                    if (lane == (r & 63)) {
                        Q[2 * r] ^= qp0; Q[2 * r + 1] ^= qp1;
                        if (spp) spb ^= 1u << (r >> 6);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int r = lane + 64 * s;
                    if (((tmask >> s) & 1u) && r != p) {
                        Q[2 * r] ^= qp0; Q[2 * r + 1] ^= qp1;
                        if (spp) spb ^= 1u << s;
                    }
                }
            }
            if (lane == ol) { pivb |= 1u << os; rowpiv[p] = (int16_t)K; prow[K] = (uint16_t)p; pcol[K] = (uint32_t)(sortbuf[c] & 0xFFFFFFFFull); }
            npiv = K + 1;
            done = __ballot((spb & ~pivb) != 0u) == 0ull;                      // syndrome in the span of the pivots found: finished
            __syncthreads();
        }
        QW_TICK(2)
#ifdef QD_OSD_TIMING
        if (lane == 0) {            // slots 12..15 (0..11 belong to qd_osd0_reg_kernel); QW_DBG_MODE picks what they hold
#if QW_DBG_MODE == 2
            atomicAdd(&a.dbg[12], wacc_[0]); atomicAdd(&a.dbg[13], wacc_[1]); atomicAdd(&a.dbg[14], wacc_[2]); atomicAdd(&a.dbg[15], 1ull);
#elif QW_DBG_MODE == 3
            atomicAdd(&a.dbg[12], (unsigned long long)cnt); atomicAdd(&a.dbg[13], (unsigned long long)npiv); atomicAdd(&a.dbg[14], (unsigned long long)ngen_); atomicAdd(&a.dbg[15], 1ull);
#else
            atomicAdd(&a.dbg[12], 1ull); atomicAdd(&a.dbg[13], done ? 0ull : 1ull); atomicAdd(&a.dbg[14], (unsigned long long)ncols_); atomicAdd(&a.dbg[15], give_up ? 1ull : 0ull);
#endif
        }
#endif
        if (!done) {
            // the prefix or the two planes did not suffice: the workgroup-per-shot kernel redoes this shot
            if (lane == 0) a.hard_list[atomicAdd(a.hard_count, 1)] = slot;
            __syncthreads();
            continue;
        }
        // ---- OSD-0 solution: e[pivot column k] = transformed syndrome at pivot row k
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) { const int r = lane + 64 * s; if (r < m) spl[r] = (uint8_t)((spb >> s) & 1u); }
        __syncthreads();
        for (int k = lane; k < npiv; k += 64)
            if (spl[prow[k]]) { const uint32_t j = pcol[k]; atomicOr(&outw[j >> 5], 1u << (j & 31u)); }
        __syncthreads();
        for (int w = lane; w < a.out_words; w += 64) a.err_bits[shot * a.out_words + w] = outw[w];
        if (lane == 0) a.status[shot] = (a.status[shot] & 0xFFFF) | QD_STATUS_OSD | (min(npiv, 4095) << 20);
        __syncthreads();
    }
}

static int wave_layout(OsdWaveArgs &a)
{
    auto al = [](int x) { return (x + 15) & ~15; };
    int o = 0;
    a.off_q = o; o += al(std::max(a.m_pad * 16, 2048 * 4));
    a.off_sort = o; o += al(std::max(a.cap, 64) * 8);
    a.off_cols = o; o += al(a.cap * a.max_cdeg * 2);
    a.off_pcol = o; o += al(QW_MAXPIV * 4);
    a.off_out = o; o += al(a.out_words * 4);
    a.off_misc = o; o += 64;
    a.off_rowpiv = o; o += al(a.m_pad * 2);
    a.off_prow = o; o += al(QW_MAXPIV * 2);
    a.off_raw = o; o += al(a.m_pad);
    a.off_sp = o; o += al(a.m_pad);
    return o;
}

static int wave_cap(int max_cdeg) { return std::max(32, std::min(384, (4096 / std::max(1, max_cdeg)) & ~31)); }

// LDS bytes per shot, or 0 when the window does not fit this kernel (more than 2048 detectors)
int qd_osd_wave_lds_bytes(int m, int m_pad, int max_cdeg, int out_words)
{
    if (m > 64 * QW_SLOTS) return 0;
    OsdWaveArgs a{};
    a.m_pad = m_pad; a.max_cdeg = max_cdeg; a.out_words = out_words; a.cap = wave_cap(max_cdeg);
    return wave_layout(a);
}

hipError_t qd_launch_osd0_wave(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &d, int blocks, hipStream_t s)
{
    OsdWaveArgs a{};
    a.m = g.m; a.n = g.n; a.m_pad = g.m_pad; a.n_pad = bg.n_pad; a.max_cdeg = g.max_cdeg; a.out_words = bg.out_words;
    a.upd_rows = d.upd_rows; a.cap = wave_cap(g.max_cdeg);
    a.csc_ptr = g.csc_ptr; a.csc_row = g.csc_row; a.bit_orig = bg.bit_orig;
    a.det = d.det; a.upd = d.upd; a.det_stride = d.det_stride; a.det_offset = d.det_offset; a.upd_stride = d.upd_stride;
    a.llr_ws = d.llr_ws; a.fail_list = d.fail_list; a.fail_count = d.fail_count;
    a.hard_list = d.hard_list; a.hard_count = d.hard_count;
    a.err_bits = d.err_bits; a.status = d.status; a.dbg = d.dbg;
    const int lds = wave_layout(a);
    auto k = g.m <= 1024 ? qd_osd0_wave_kernel<16> : qd_osd0_wave_kernel<QW_SLOTS>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64), lds, s, a);
    return hipGetLastError();
}
