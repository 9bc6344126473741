// osd_cs.hip -- OSD-CS / OSD-E (ldpc osd.hpp, OsdDecoder::decode with osd_order > 0): the full-rank elimination by column, rebuilt
// in round 5.  Replaces qd_osdw_col_kernel (osd_kernels.hip) wherever the window fits; same oracle
// (oracle/qd_oracle.c: osd_w_impl / elim_run), same column order, same pivot rule (first independent column of the order, lowest
// unpivoted row), hence the same pivots, transformed syndrome, candidate costs and corrections, bit for bit.
//
// What changed against qd_osdw_col_kernel, and why:
//   * The column order is produced ONCE, completely: a sample sort of the (posterior key, fault index) pairs in LDS -- 8 samples per
//     bucket ranked by counting, one pass that counts, one that scatters into <= 32 buckets, then every wavefront sorts whole buckets
//     in place with a direction-free bitonic network (no workgroup barrier inside).  The old kernel drew the order in tiers of 1024
//     by radix selection: three passes over all n posteriors + a 55-stage barrier sort per tier, 13 % of a shot.
//   * A batch of 64 sorted columns is one PANEL.  The image of a column under the pivots found before the batch is pushed into LDS
//     by the owners of the Q columns (as before); then ONE wavefront finds the batch's pivots on the panel alone -- liveness of all
//     64 columns in one sweep, a pivot step = three LDS round trips and no barrier -- and records (pivot row, image) per pivot.
//     The Q columns in the other wavefronts' registers are brought up to date AFTERWARDS, all pivots of the batch in one go, with the
//     image of each pivot broadcast through SCALAR registers (one 16-lane LDS read + v_readlane per word instead of a 16-word LDS
//     read per thread): the old kernel paid two workgroup barriers and 64 KB of LDS reads per pivot.
//     Three barriers per BATCH (~130 batches per headline shot) instead of two per PIVOT (~1000) plus five per batch.
//     Round 6: the pivot step of phase [B] uses the ballot as the execution mask of the XOR (499 -> 409 ticks per batch); the transformed syndrome is
//     carried by wavefront 1, not by the searching wavefront; wavefront 0 keeps its priority to the batch's last barrier; the next batch's rows are
//     requested two barriers ahead; publish / poll are release / acquire at workgroup scope.  Restructurings of the batch sequence (pipelined
//     batches, a register-built panel pulled from an L2 mirror, a Q update carried into the next batch, other ownership maps of the Q columns) were
//     measured and NOT kept: profiles/r06_osdcs_steps.txt.
//   * The candidate sweep sums the signed pivot weights four rows at a time (a 16-entry table per nibble of rows, built once per
//     shot) instead of one set bit at a time, and breaks ties on the sorted position instead of re-deriving the key.
//   * One kernel body, every loop that touches the Q columns unrolled over compile-time bounds: ScratchSize 0 in every instantiation
//     (tests/test_api.py compiles this file and asserts it).
#include "osd_shared.h"

#include <cstdlib>

#ifndef QD_CS_NSAMP_PER_BUCKET
#define QD_CS_NSAMP_PER_BUCKET 8
#endif
#define QD_CS_MAX_BUCKETS 64

// sub-phase timers of a -DQD_OSD_TIMING build: -DQD_CS_SUB=1 (default) panel phase, 2 the sort, 3 the sweep (tools/osdcs_timing.py)
#ifndef QD_CS_SUB
#define QD_CS_SUB 1
#endif
// priority of wavefront 0 from the head of phase [B] to the batch's last barrier (1 and 2 measured no different: profiles/r06_osdcs_steps.txt)
#ifndef QD_CS_B_PRIO
#define QD_CS_B_PRIO 3
#endif
#ifdef QD_OSD_TIMING
#define QD_SUBT(mode, slot) if constexpr (QD_CS_SUB == mode) { const unsigned long long n2_ = wall_clock64(); acc_[slot] += n2_ - sub_; sub_ = n2_; }
#define QD_SUBT0(mode) if constexpr (QD_CS_SUB == mode) { sub_ = wall_clock64(); }
#else
#define QD_SUBT(mode, slot)
#define QD_SUBT0(mode)
#endif

// Lanes of ONE wavefront exchanging data through LDS: the hardware runs them in lockstep and completes a wavefront's LDS operations in
// order, but the compiler reasons per thread -- it may forward a lane's own store to its later load and hoist the other lanes' load
// above the (to them absent) store.  A wavefront-scope fence costs no instruction and forbids that.
#define QD_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

struct OsdCsArgs {
    int m, n, n_pad, out_words, upd_rows, ell_log2, rank;
    int osd_w, osd_order;                 // 1 = combination sweep, 2 = exhaustive
    // LDS: everything whose size the instantiation fixes sits at compile-time offsets (CsLds below: one scalar register per runtime offset
    // is what pushed the first build of this kernel into scratch); only the three arrays sized by the window follow at runtime offsets
    int o_out, o_order, o_sort_aux;       // pivmask at CsLds::o_var, out words, sorted order, splitters / counters of the sort
    const uint16_t *csc_ell;              // [n][1 << ell_log2] detector indices of a fault, ascending, 0xFFFF beyond its weight
    const uint32_t *wfix;                 // [n] round(log(1/p_j) * 2^18)
    const uint32_t *bit_orig;             // [n_pad] fault index of a bit slot (rows of llr_ws are in slot order)
    const uint8_t *det, *upd;
    int64_t det_stride, det_offset, upd_stride;
    const float *llr_ws;
    const int32_t *fail_list, *fail_count;
    uint64_t *ws;                         // [blocks][ws_words]: the Q columns where the sweep reads them
    size_t ws_words;
    uint32_t *err_bits;
    int32_t *status;
    unsigned long long *dbg;
};

// LDS layout shared by the kernel and the host (bytes).  The sweep's tables overlay the panel and `need` buffers, dead by then.
template <int T, int CPT, int NWD>
struct CsLds {
    static constexpr int PSTR = NWD + 1, NPIV = T * CPT;
    static constexpr int o_p = 0;                                   // [2][64][PSTR] u64 panel
    static constexpr int o_need = o_p + 2 * 64 * PSTR * 8;          // [2][NPIV] u64
    static constexpr int o_tp = o_need + 2 * NPIV * 8;              // [64][NWD] u64
    static constexpr int o_sv = o_tp + 64 * NWD * 8;                // [NWD] u64
    static constexpr int o_unp = o_sv + NWD * 8;                    // [NWD] u64
    static constexpr int o_rowpiv = o_unp + NWD * 8;                // [NWD * 64] i16
    static constexpr int o_prow = o_rowpiv + NWD * 64 * 2;          // [NPIV] u16
    static constexpr int o_pcol = o_prow + NPIV * 2;                // [NPIV] u16
    static constexpr int o_misc = o_pcol + NPIV * 2;                // 2048 bytes
    static constexpr int o_var = o_misc + 2048;                     // pivmask [out_words], then out, then order (runtime sizes)
    static constexpr int o_swl = o_p;                               // [NWD * 64] i32
    static constexpr int o_nib = o_swl + NWD * 64 * 4;              // [NWD * 16][16] i32
    // the nibble tables of one 64-row word take 256 entries; each word's block is shifted by NIBS entries so that the four lanes of a
    // candidate (words sub * WPL + u) read four different quarters of the 64 banks: WPL * NIBS = 16 (mod 64)
    static constexpr int WPL = (NWD + 3) / 4, NIBS = WPL == 6 ? 24 : 16 / WPL, NIBW = 256 + NIBS;
    static constexpr int o_tv = o_nib + ((NWD * NIBW * 4 + 15) & ~15);   // [64][NWD] u64
    static_assert(o_tv + 64 * NWD * 8 <= o_tp, "the sweep's tables must fit over the panel and need buffers");
    static_assert((NWD * 8) % 16 == 0 && (o_need % 16) == 0 && (o_tp % 16) == 0 && (o_var % 16) == 0, "16-byte alignment");
};

struct CsBest { long long delta; uint32_t cls; unsigned long long tie, what; };

__device__ __forceinline__ bool qd_cs_less(long long d1, uint32_t c1, unsigned long long t1, long long d2, uint32_t c2, unsigned long long t2)
{
    if (d1 != d2) return d1 < d2;
    if (c1 != c2) return c1 < c2;
    return t1 < t2;
}

// Sorts buf[0 .. len) ascending, in place, by ONE wavefront (all 64 lanes call it with the same arguments).  Direction-free bitonic
// network: the first step of every merge compares i with its mirror inside the block, the rest are half-cleaners, so every
// comparator puts the smaller key at the lower index and the elements beyond `len` (virtual +infinity) never move.  A wavefront's
// LDS operations complete in order, so the stages need no barrier.
__device__ __forceinline__ void qd_cs_wave_sort(uint64_t *buf, int len, int lane)
{
    if (len < 2) return;
    int P = 2;
    while (P < len) P <<= 1;
    const int half = P >> 1;
    // one pass of the network: comparator i of `half`; four comparators per lane are loaded before the first is decided (the pairs of
    // a pass are disjoint), so a pass costs one LDS round trip per 256 comparators instead of one per 64
    auto pass = [&](int k, int j) {                                    // j == 0: the flip step of merge size k; else half-cleaner of distance j
        const int kh = k >> 1;
        for (int i0 = lane; i0 < half; i0 += 256) {
            int lo[4], hi[4];
            uint64_t x[4], y[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u;
                if (j == 0) { const int blk = i / kh, off = i - blk * kh; lo[u] = blk * k + off; hi[u] = blk * k + (k - 1 - off); }
                else { lo[u] = ((i & ~(j - 1)) << 1) | (i & (j - 1)); hi[u] = lo[u] + j; }
                ok[u] = i < half && hi[u] < len;
                x[u] = ok[u] ? buf[lo[u]] : 0ull;
                y[u] = ok[u] ? buf[hi[u]] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u] && x[u] > y[u]) { buf[lo[u]] = y[u]; buf[hi[u]] = x[u]; }
        }
        QD_WAVE_SYNC();
    };
    for (int k = 2; k <= P; k <<= 1) {
        pass(k, 0);
        for (int j = k >> 2; j > 0; j >>= 1) pass(k, j);
    }
}

// The same for a short range (len <= 64 * QD_CS_RANK_KM), by counting: a lane keeps its <= KM keys in registers and counts, for each, the keys
// of the range below it -- broadcast reads that depend on nothing, where the network above is ~40 dependent LDS round trips --, then every
// key goes to the position of its rank (the keys are distinct: the fault index is part of them).  All reads precede all writes.
#define QD_CS_RANK_KM 6
__device__ __forceinline__ void qd_cs_wave_ranksort(uint64_t *buf, int len, int lane)
{
    uint64_t own[QD_CS_RANK_KM];
    int rk[QD_CS_RANK_KM];
#pragma unroll
    for (int u = 0; u < QD_CS_RANK_KM; ++u) {
        const int idx = lane + 64 * u;
        own[u] = idx < len ? buf[idx] : ~0ull;
        rk[u] = 0;
    }
#pragma unroll 4
    for (int j = 0; j < len; ++j) {
        const uint64_t x = buf[j];
#pragma unroll
        for (int u = 0; u < QD_CS_RANK_KM; ++u) rk[u] += x < own[u] ? 1 : 0;
    }
    QD_WAVE_SYNC();
#pragma unroll
    for (int u = 0; u < QD_CS_RANK_KM; ++u)
        if (lane + 64 * u < len) buf[rk[u]] = own[u];
    QD_WAVE_SYNC();
}

// T threads hold CPT columns of Q each (rank <= T * CPT), NWD words per column (m <= 64 * NWD); WPS = wavefronts per SIMD the
// register budget is cut for; IPT = sorted items per thread (n <= T * IPT).
template <int T, int CPT, int NWD, int WPS, int IPT>
__global__ void __launch_bounds__(T, WPS) qd_osdcs_kernel(OsdCsArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NW = T / 64;
    constexpr int LPS = NWD <= 8 ? 8 : (NWD <= 16 ? 16 : 32);     // lanes per panel vector in the single-wavefront phase
    constexpr int NSLOT = 64 / LPS;
    constexpr unsigned long long LPSMASK = (LPS == 32) ? 0xFFFFFFFFull : ((1ull << LPS) - 1ull);
    constexpr int PSTR = NWD + 1;                                  // padded column stride (words): a lane-per-column read of one word spreads over the banks
    constexpr int NPIV = T * CPT;
    constexpr int SPT = (64 * 16 + (T - 64) - 1) / (T - 64);      // (column, incidence) pairs of a batch per thread of wavefronts 1.. (at most 16 incidences per fault)

    using L = CsLds<T, CPT, NWD>;
    uint64_t *sb = reinterpret_cast<uint64_t *>(smem);                                   // [n] sort buffer (the sort phase owns all of LDS)
    uint16_t *order = reinterpret_cast<uint16_t *>(smem + a.o_order);                    // [n] faults in sorted order
    uint64_t *Pbuf = reinterpret_cast<uint64_t *>(smem + L::o_p);                        // [2][64][PSTR] panel: images of the batch's columns, by row
    uint64_t *needb = reinterpret_cast<uint64_t *>(smem + L::o_need);                    // [2][NPIV] pivot order -> batch columns that contain its row
    uint64_t *Tp = reinterpret_cast<uint64_t *>(smem + L::o_tp);                         // [64][NWD] images of the batch's pivot columns (without the pivot bit)
    uint64_t *sv = reinterpret_cast<uint64_t *>(smem + L::o_sv);                         // [NWD] transformed syndrome
    uint64_t *unpm = reinterpret_cast<uint64_t *>(smem + L::o_unp);                      // [NWD] rows that are not pivot rows yet
    int16_t *rowpiv = reinterpret_cast<int16_t *>(smem + L::o_rowpiv);                   // [NWD * 64] row -> pivot order or -1
    uint16_t *prow = reinterpret_cast<uint16_t *>(smem + L::o_prow);                     // [NPIV] pivot order -> row
    uint16_t *pcol = reinterpret_cast<uint16_t *>(smem + L::o_pcol);                     // [NPIV] pivot order -> fault
    uint32_t *pivmask = reinterpret_cast<uint32_t *>(smem + L::o_var);                   // [out_words] faults that are pivot columns
    uint32_t *misc = reinterpret_cast<uint32_t *>(smem + L::o_misc);                     // [0] pivots of the batch, [64..127] pivp, [128..] bests, [256..] npl
    uint32_t *pivp = misc + 64;                                                          // [64] pivot row | batch column << 16
    int32_t *swl = reinterpret_cast<int32_t *>(smem + L::o_swl);                         // [NWD * 64] signed pivot weight of a row (sweep)
    int32_t *nib = reinterpret_cast<int32_t *>(smem + L::o_nib);                         // [NWD * 16][16] sums of swl over the rows of a nibble (sweep)
    uint64_t *tvl = reinterpret_cast<uint64_t *>(smem + L::o_tv);                        // [64][NWD] images of the first non-pivot columns (patterns)

    const int nfail = *a.fail_count;
    int tid = threadIdx.x, n = a.n;
    for (int item = blockIdx.x; item < nfail; item += gridDim.x) {
        // The thread index is made opaque once per shot: everything derived from it (the IPT item indices of the sort, their bounds
        // tests and global addresses, ...) is invariant across shots, and the compiler otherwise hoists all of it out of this loop
        // and keeps it live -- in scratch -- for the whole kernel (the first build: 628 bytes per lane)
        asm volatile("" : "+v"(tid));                                  // (loop-carried and opaque: ONE register holds the thread index across shots)
        asm volatile("" : "+s"(n));                                    // (likewise: a copy of n in a vector register, made before the loop, was kept in scratch)
        const int m = a.m, dlog = a.ell_log2, ellw = 1 << dlog;
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int slot = item;
        const int64_t shot = a.fail_list[slot];
        const float *llr = a.llr_ws + (int64_t)slot * a.n_pad;
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
#ifdef QD_OSD_TIMING
        unsigned long long acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long sub_ = 0;
        unsigned long long tick_ = wall_clock64();
#endif
        // ================================================================== the column order: sample sort of (key, fault) in LDS
        {
            QD_SUBT0(2)
            uint64_t *spl = reinterpret_cast<uint64_t *>(smem + a.o_sort_aux);                   // [64] splitters
            uint32_t *bcnt = reinterpret_cast<uint32_t *>(smem + a.o_sort_aux + 512);            // [65] bucket starts
            uint32_t *bcur = bcnt + 72;                                                          // [64] bucket cursors
            const int nbk = min(min(QD_CS_MAX_BUCKETS, T / QD_CS_NSAMP_PER_BUCKET), max(1, n >> 7));   // ~150 keys per bucket
            const int ns = nbk * QD_CS_NSAMP_PER_BUCKET;                     // <= T
            auto key_of = [&](int b) -> uint64_t { return ((uint64_t)qd_mono_key(llr[b]) << 32) | (uint64_t)a.bit_orig[b]; };
            if (tid < ns) sb[tid] = key_of((int)(((long long)tid * n) / ns));
            {
                uint64_t ones = ~0ull;
                asm volatile("" : "+v"(ones));                         // (made here, every shot: hoisted out of the shot loop this constant was kept in scratch)
                if (tid < 64) spl[tid] = ones;
            }
            if (tid < 72) { bcnt[tid] = 0u; }
            __syncthreads();
            if (nbk > 1 && tid < ns) {
                // rank by counting (the keys are distinct: the fault index is part of them); every 8th sample is a splitter
                const uint64_t mine = sb[tid];
                int rank = 0;
                for (int u = 0; u < ns; ++u) rank += sb[u] < mine ? 1 : 0;
                if (rank > 0 && (rank % QD_CS_NSAMP_PER_BUCKET) == 0) spl[rank / QD_CS_NSAMP_PER_BUCKET - 1] = mine;
            }
            __syncthreads();
            // bucket of a key = number of splitters <= key (branch-free binary search over 31 + 1 entries)
            auto bucket_of = [&](uint64_t x) -> uint32_t {
                uint32_t lo = 0;
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) lo += (spl[lo + step - 1] <= x) ? step : 0;
                return lo;
            };
            QD_SUBT(2, 11)
            uint32_t bkreg[(IPT + 3) / 4];
#pragma unroll
            for (int i = 0; i < (IPT + 3) / 4; ++i) bkreg[i] = 0u;
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int b = tid + i * T;
                if (b < n) {
                    const uint32_t bk = nbk > 1 ? bucket_of(key_of(b)) : 0u;
                    bkreg[i >> 2] |= bk << (8 * (i & 3));
                    atomicAdd(&bcnt[bk], 1u);
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four items' searches in flight at a time, not IPT: the sort must not be the kernel's register peak)
            }
            __syncthreads();
            QD_SUBT(2, 12)
            if (tid < 64) {
                // exclusive scan of the <= 64 counts; bcnt[k] becomes the start of bucket k, bcnt[nbk .. 64] = n
                const uint32_t c = bcnt[tid];
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += o; }
                bcnt[tid] = incl - c; bcur[tid] = incl - c;
                if (tid == 0) bcnt[64] = (uint32_t)n;
            }
            __syncthreads();                                                  // (the samples in sb[0 .. ns) are dead: everybody has its bucket numbers)
            {
                uint32_t pos[IPT];                                    // all the cursor bumps first, then the stores: one round trip, not IPT
#pragma unroll
                for (int i = 0; i < IPT; ++i) {
                    const int b = tid + i * T;
                    pos[i] = b < n ? atomicAdd(&bcur[(bkreg[i >> 2] >> (8 * (i & 3))) & 0xFFu], 1u) : 0u;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < IPT; ++i) {
                    const int b = tid + i * T;
                    if (b < n) sb[pos[i]] = key_of(b);
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            QD_SUBT(2, 13)
            for (int bk = wave; bk < nbk; bk += NW) {
                const int lo = (int)bcnt[bk], hi = (int)bcnt[bk + 1];
                if (hi - lo <= 64 * QD_CS_RANK_KM) qd_cs_wave_ranksort(sb + lo, hi - lo, lane);
                else qd_cs_wave_sort(sb + lo, hi - lo, lane);            // (a bucket several times its expected size: the sorting network takes any length)
            }
            __syncthreads();
            QD_SUBT(2, 14)
            uint16_t myord[IPT];
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int b = tid + i * T;
                myord[i] = b < n ? (uint16_t)(sb[b] & 0xFFFFull) : (uint16_t)0;
            }
            __syncthreads();                                                  // the sort buffer is dead; everything below lives in the same LDS
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int b = tid + i * T;
                if (b < n) order[b] = myord[i];
            }
        }
        QD_TICK(0)
        // ================================================================== elimination state
        uint64_t mycol[CPT][NWD];
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int w = 0; w < NWD; ++w) mycol[i][w] = 0ull;
        for (int x = tid; x < 2 * 64 * PSTR; x += T) Pbuf[x] = 0ull;
        for (int x = tid; x < 2 * NPIV; x += T) needb[x] = 0ull;
        if (tid < NWD) {
            sv[tid] = 0ull;
            const int lo = tid * 64;
            unpm[tid] = (m - lo >= 64) ? ~0ull : (m > lo ? ((1ull << (m - lo)) - 1ull) : 0ull);
        }
        for (int r = tid; r < NWD * 64; r += T) rowpiv[r] = -1;
        for (int w = tid; w < a.out_words; w += T) { reinterpret_cast<uint32_t *>(smem + a.o_out)[w] = 0u; pivmask[w] = 0u; }
        if (tid < 64) misc[tid] = 0u;
        __syncthreads();
        for (int r = tid; r < m; r += T) {
            uint32_t sbit = det[r] & 1u;
            if (upd && r < a.upd_rows) sbit ^= upd[r] & 1u;
            if (sbit) atomicOr(reinterpret_cast<unsigned long long *>(&sv[r >> 6]), 1ull << (r & 63));
        }
        // scatter of batch `b0`: raw columns by row into the panel, and for every pivot the batch columns that contain its row
        auto scatter = [&](int base, uint64_t *Pc, uint64_t *nd) {
            const int nb = min(64, n - base);
            for (int x = tid; x < (64 << dlog); x += T) {
                const int c = x >> dlog, q = x & (ellw - 1);
                if (c < nb) {
                    const uint32_t col = order[base + c];
                    const uint32_t r = a.csc_ell[((size_t)col << dlog) + q];
                    if (r != 0xFFFFu) {
                        atomicXor(reinterpret_cast<unsigned long long *>(&Pc[c * PSTR + (r >> 6)]), 1ull << (r & 63));
                        const int k = rowpiv[r];
                        if (k >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&nd[k]), 1ull << c);
                    }
                }
            }
        };
        scatter(0, Pbuf, needb);
        __syncthreads();

        // Which Q columns a thread owns.  Wavefront 0 -- the one that finds the pivots -- owns the HIGHEST pivot orders, which only exist in the last
        // batches of a shot; the other wavefronts share the rest.  So while wavefront 0 works on a panel the others apply each pivot to their columns
        // as soon as it is published (phase [B] below), and wavefront 0 itself has nothing to update until late in the shot.
        auto kown = [&](int c) -> int { return wave == 0 ? NPIV - 64 * CPT + c * 64 + lane : (tid - 64) + c * (T - 64); };
        const int kmin_wave = wave == 0 ? NPIV - 64 * CPT : (wave - 1) * 64;       // the lowest pivot order any lane of this wavefront owns
        // Round 6: the transformed syndrome lives in a register of wavefront 1 during the elimination (lane w holds word w) and takes the pivots
        // with the Q columns; wavefront 0 used to carry it through phase [B], one more vector to update per pivot on the chain everything waits for
        uint64_t svx = (wave == 1 && lane < NWD) ? sv[lane] : 0ull;
        int npiv = 0;
        for (int base = 0, bi = 0; base < n; base += 64, ++bi) {
            const int nb = min(64, n - base);
            uint64_t *Pc = Pbuf + (bi & 1) * 64 * PSTR, *Pn = Pbuf + ((bi & 1) ^ 1) * 64 * PSTR;
            uint64_t *ndc = needb + (bi & 1) * NPIV, *ndn = needb + ((bi & 1) ^ 1) * NPIV;
            // The rows of the NEXT batch's columns (two or three (column, incidence) pairs per thread of wavefronts 1..) are requested here, two barriers
            // before they are scattered, and stay in registers: round 5 scattered by loading them behind the Q update, an L2 round trip on the
            // other wavefronts' way to the barrier.
            uint32_t rpre[SPT];
            int tid_y = tid;
            asm volatile("" : "+v"(tid_y));                           // (opaque per batch: the addresses below are invariant in the batch loop, and hoisted out of it they were kept in scratch)
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                rpre[j] = 0xFFFFu;
                const int x = (tid_y - 64) + j * (T - 64);
                if (wave != 0 && x < (64 << dlog) && base + 64 + (x >> dlog) < n)
                    rpre[j] = (uint32_t)a.csc_ell[((size_t)order[base + 64 + (x >> dlog)] << dlog) + (x & (ellw - 1))];
            }
            // ---- [A] images: the owner of Q column k adds it to the batch columns that contain pivot row k; the other buffers are cleared
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int k = kown(i);
                if (k < npiv)
                    for (uint64_t bits = ndc[k]; bits; bits &= bits - 1ull) {
                        const int c = (int)__builtin_ctzll(bits);
#pragma unroll
                        for (int w = 0; w < NWD; ++w)
                            atomicXor(reinterpret_cast<unsigned long long *>(&Pc[c * PSTR + w]), (unsigned long long)mycol[i][w]);
                    }
            }
            for (int x = tid; x < 64 * PSTR; x += T) Pn[x] = 0ull;
            for (int x = tid; x < NPIV; x += T) ndn[x] = 0ull;
            if (tid == 0) { misc[6] = 0u; misc[7] = 0u; }              // pivots published / panel finished (phase [B])
#ifdef QD_OSD_TIMING
            ++acc_[10];
#endif
            QD_TICK(1)
            __syncthreads();
            QD_TICK(5)
            // ---- [B] one wavefront: the pivots of the batch, on the panel alone.  Lane w holds word w of whatever vector is being looked
            // at (the lanes beyond NWD idle; the LPS-lane slots only matter to the liveness sweep).  Live columns are taken CH at a time
            // into registers -- one LDS round trip per chunk, none per pivot: bit p of a register-held vector is read with v_readlane
            // from the lane of its word --, the syndrome and the unpivoted-row mask stay in registers for the whole batch, and a chunk
            // loaded later first takes the batch's earlier pivots (only the columns that hold one of their rows at all).
            if (wave == 0) {
                __builtin_amdgcn_s_setprio(QD_CS_B_PRIO);              // the one wavefront everybody waits for
                const int w = lane & (LPS - 1), s = lane / LPS;
                const bool wv = w < NWD;
                uint64_t unp = wv ? unpm[w] : 0ull;
                uint64_t bm = 0ull;                                    // rows that became pivot rows in this batch
                QD_SUBT0(1)
                // liveness, one lane per column: OR over the words of (column & unpivoted rows); the padded column stride keeps the 64
                // lanes' reads of one word on different banks
                uint64_t live;                                         // batch columns with a one on a row that is not a pivot row
                {
                    uint64_t acc = 0ull;
#pragma unroll
                    for (int ww = 0; ww < NWD; ++ww) acc |= Pc[lane * PSTR + ww] & unpm[ww];
                    live = __ballot(acc != 0ull && lane < nb);
                }
                QD_SUBT(1, 11)
                // rank -> column table of the live columns (chunks are consecutive rank ranges; a pivot's record keeps the rank)
                uint32_t *tab = misc + 192;                            // [64]
                const int nlive = (int)__popcll(live);
                if ((live >> lane) & 1ull) tab[__builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u))] = (uint32_t)lane;   // (rank = set bits below my lane)
                QD_WAVE_SYNC();
                // A chunk = NR registers x NSLOT slots: the column of rank c0 + r * NSLOT + q has word w in lane (q * LPS + w) of x[r].
                constexpr int NR = NWD <= 8 ? 4 : (NWD <= 16 ? 7 : 8), CHC = NR * NSLOT;   // (7, not 8, at 16 words: the eighth row costs the 128-register instantiation its zero scratch)
                // Round 6: "add the pivot's image to the columns that hold the pivot row" is: the lanes of word w0 test their bit (`pm` has the
                // pivot bit in those lanes only), one compare puts the four (eight, two) slots' answers into a scalar pair, three scalar
                // instructions spread each slot's bit over the slot's lanes, and that mask IS the execution mask of the XOR
                // (__builtin_amdgcn_inverse_ballot_w64).  Round 5's form went back through per-lane shifts and compares of the ballot: ~14 vector
                // instructions per register where this takes 5, in chains half as long.
                constexpr unsigned long long SLOTBITS = LPS == 8 ? 0x0101010101010101ull : (LPS == 16 ? 0x0001000100010001ull : 0x0000000100000001ull);
                auto slot_mask = [&](unsigned long long bal, int w0) -> unsigned long long {
                    const unsigned long long mb = (bal >> w0) & SLOTBITS;
                    const uint32_t lo = (uint32_t)mb, hi = (uint32_t)(mb >> 32);
                    if constexpr (LPS == 32) return (unsigned long long)(0u - lo) | ((unsigned long long)(0u - hi) << 32);
                    else {
                        constexpr uint32_t MUL = LPS == 8 ? 0xFFu : 0xFFFFu;
                        return (unsigned long long)(lo * MUL) | ((unsigned long long)(hi * MUL) << 32);
                    }
                };
                auto apply_row = [&](uint64_t &v, uint64_t tq, uint64_t pm, int w0) {
                    const unsigned long long bal = __ballot((v & pm) != 0ull);
                    if (__builtin_amdgcn_inverse_ballot_w64(slot_mask(bal, w0))) v ^= tq;
                };
                int g = 0, gpub = 0;
                const int room = a.rank - npiv;
                for (int c0 = 0; c0 < nlive && g < room; c0 += CHC) {
                    uint64_t x[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int rank = c0 + r * NSLOT + s;
                        const bool ok = wv && rank < nlive;
                        const uint32_t c = ok ? tab[rank] : 0u;
                        x[r] = ok ? Pc[c * PSTR + w] : 0ull;
                    }
                    if (g > 0) {
                        QD_WAVE_SYNC();
                        uint32_t flag = 0u;
#pragma unroll
                        for (int r = 0; r < NR; ++r)
                            if (__ballot((x[r] & bm) != 0ull) != 0ull) flag |= 1u << r;
                        if (flag != 0u)
                            for (int i = 0; i < g; ++i) {
                                const uint32_t pj = (uint32_t)__builtin_amdgcn_readfirstlane((int)pivp[i]);
                                const int p = (int)(pj & 0xFFFFu), w0 = p >> 6;
                                const uint64_t tq = wv ? Tp[i * NWD + w] : 0ull;
                                const uint64_t pm = (w == w0) ? (1ull << (p & 63)) : 0ull;
#pragma unroll
                                for (int r = 0; r < NR; ++r)
                                    if ((flag >> r) & 1u) apply_row(x[r], tq, pm, w0);
                            }
                    }
                    QD_SUBT(1, 12)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        // (the slot loop stays unrolled -- 28 copies of the pivot step: rolled up it was slower, 419 -> 513 ticks per batch, profiles/r06_osdcs_steps.txt)
                        for (int q = 0; q < NSLOT; ++q) {
                            if (c0 + r * NSLOT + q >= nlive || g >= room) break;
                            const uint64_t y = x[r] & unp;
                            const unsigned long long nz = (__ballot(y != 0ull) >> (q * LPS)) & LPSMASK;
                            if (nz == 0ull) continue;                  // the pivots of this batch made it dependent
                            const int w0 = (int)__builtin_ctzll(nz), src = q * LPS + w0;
                            const uint32_t ylo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)y, src);
                            const uint32_t yhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(y >> 32), src);
                            const int pbit = ylo ? (int)__builtin_ctz(ylo) : 32 + (int)__builtin_ctz(yhi);
                            const int p = w0 * 64 + pbit;
                            const uint64_t pm = (w == w0) ? (1ull << pbit) : 0ull;        // the pivot bit, in the lanes of its word only
                            // the image without bit p: stored for phase [C], and copied from slot q into every slot -- by two lane-swap
                            // instructions per half (rows of 16 lanes, halves of 32), no LDS round trip; 8-lane slots go through LDS
                            uint64_t tq = x[r] ^ pm;
                            // the count published here is the PREVIOUS pivot's: its image and record were stored a whole step ago, so the release
                            // (workgroup scope, ADVICE r5: one s_waitcnt) finds nothing outstanding and costs the chain nothing
                            if (g > gpub) { if (lane == 0) __hip_atomic_store(&misc[6], (uint32_t)g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); gpub = g; }
                            if (s == q && wv) Tp[g * NWD + w] = tq;
                            if constexpr (LPS == 8) { QD_WAVE_SYNC(); tq = wv ? Tp[g * NWD + w] : 0ull; }
                            else {
                                uint32_t lo = (uint32_t)tq, hi = (uint32_t)(tq >> 32);
                                if constexpr (LPS == 16) {
                                    const auto a1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
                                    const auto a2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
                                    lo = (q & 1) ? a1[1] : a1[0];
                                    hi = (q & 1) ? a2[1] : a2[0];
                                }
                                const int qh = LPS == 16 ? (q >> 1) : q;
                                const auto b1 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
                                const auto b2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
                                lo = qh ? b1[1] : b1[0];
                                hi = qh ? b2[1] : b2[0];
                                tq = ((uint64_t)hi << 32) | (uint64_t)lo;
                            }
                            if (lane == 0) pivp[g] = (uint32_t)p | ((uint32_t)(c0 + r * NSLOT + q) << 16);
                            unp &= ~pm; bm |= pm;
#pragma unroll
                            for (int r2 = r; r2 < NR; ++r2)
                                if (c0 + r2 * NSLOT < nlive) apply_row(x[r2], tq, pm, w0);  // (columns already passed are dead: harmless)
                            ++g;
                        }
                    }
                    QD_SUBT(1, 13)
                }
                if (s == 0 && wv) unpm[w] = unp;                        // (the transformed syndrome is wavefront 1's: it takes the pivots with the Q columns)
                if (lane == 0) misc[0] = (uint32_t)g;
                QD_WAVE_SYNC();
                if (lane < g) {                                        // the pivots' records, one lane each (pivp: same wavefront, LDS in order)
                    const uint32_t pj = pivp[lane];
                    const int p = (int)(pj & 0xFFFFu), K = npiv + lane;
                    const uint32_t cb = tab[pj >> 16];                 // rank among the live columns -> column of the batch
                    const uint32_t pc = order[base + (int)cb];
                    rowpiv[p] = (int16_t)K; prow[K] = (uint16_t)p; pcol[K] = (uint16_t)pc;
                    atomicOr(&pivmask[pc >> 5], 1u << (pc & 31u));
                }
                QD_SUBT(1, 14)
                if (lane == 0) {                                       // the panel is finished: the last image(s), the count, the records -- then the flag
                    __hip_atomic_store(&misc[6], (uint32_t)g, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&misc[7], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef QD_OSD_TIMING
                    if constexpr (QD_CS_SUB == 5) { const unsigned long long tf_ = wall_clock64(); misc[20] = (uint32_t)tf_; misc[21] = (uint32_t)(tf_ >> 32); }
#endif
                }
            }
            QD_TICK(2)
            // ---- [C] every Q column takes the batch's pivots, in order: a column that has bit p set gets the pivot's image added (bit p stays: the
            // image is stored without it); column K, all zero until now, becomes that image.  Wavefronts 1.. do this WHILE wavefront 0 is still
            // at work on the panel: they poll the published count (two LDS reads, then a short sleep) and apply what has arrived; wavefront 0
            // applies the batch to its own columns afterwards, which exist in the last batches of a shot only.
            auto apply_pivot = [&](int i) {
                const int K = npiv + i;
                if (kmin_wave > K && wave != 1) return;               // none of this wavefront's columns exists yet (uniform)
                const uint32_t pj = (uint32_t)__builtin_amdgcn_readfirstlane((int)pivp[i]);
                const int p = (int)(pj & 0xFFFFu), pw = p >> 6;
                const uint64_t pb = 1ull << (p & 63);
                if (wave == 1) {
                    // the syndrome takes the pivot like any column: the lane of the pivot row's word tests, every lane adds its word of the image
                    if (__ballot(lane == pw && (svx & pb) != 0ull) != 0ull && lane < NWD) svx ^= Tp[i * NWD + lane];
                    if (kmin_wave > K) return;
                }
                uint64_t sel[CPT];
#pragma unroll
                for (int c = 0; c < CPT; ++c) sel[c] = 0ull;
                switch (pw) {
#define QD_X(W) case W: if constexpr (W < NWD) { _Pragma("unroll") for (int c = 0; c < CPT; ++c) sel[c] = mycol[c][W < NWD ? W : 0]; } break;
                    QD_X(0) QD_X(1) QD_X(2) QD_X(3) QD_X(4) QD_X(5) QD_X(6) QD_X(7) QD_X(8) QD_X(9) QD_X(10) QD_X(11)
                    QD_X(12) QD_X(13) QD_X(14) QD_X(15) QD_X(16) QD_X(17) QD_X(18) QD_X(19) QD_X(20) QD_X(21) QD_X(22) QD_X(23)
#undef QD_X
                    default: break;
                }
                bool hit[CPT];
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    const int k = kown(c);
                    hit[c] = (k < K && (sel[c] & pb) != 0ull) || k == K;
                }
                // the image comes as broadcast LDS reads (every lane the same address): the LDS pipe is idle in this phase, the vector ALU
                // is what bounds it (the first form moved the words through v_readlane: a quarter of this loop's instructions)
                const uint64_t *tq = Tp + i * NWD;
#pragma unroll
                for (int wb = 0; wb < NWD; wb += 8) {
                    uint64_t tv_[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (wb + u < NWD) tv_[u] = tq[wb + u];
#pragma unroll
                    for (int c = 0; c < CPT; ++c)
                        if (hit[c]) {
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (wb + u < NWD) mycol[c][wb + u] ^= tv_[u];
                        }
                }
            };
            int g = 0;                                                 // (one call site of apply_pivot: wavefront 0 arrives here with its panel finished)
            for (;;) {
                // acquire at workgroup scope: what is read after these loads -- images, records -- is at least as new as the count.
                // `finished` is read before the count: finished => the count is final
                const uint32_t fin = __hip_atomic_load(&misc[7], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int avail = (int)__hip_atomic_load(&misc[6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (; g < avail; ++g) apply_pivot(g);
                if (fin) break;
                __builtin_amdgcn_s_sleep(2);
            }
            npiv += g;
            const bool done = npiv >= a.rank || base + 64 >= n;
            if (!done && wave != 0) {
                // the next batch's raw columns and need words, from the rows requested before this batch's panel phase (rowpiv includes this batch's records)
#pragma unroll
                for (int j = 0; j < SPT; ++j) {
                    const uint32_t r = rpre[j];
                    if (r != 0xFFFFu) {
                        const int c = ((tid_y - 64) + j * (T - 64)) >> dlog;
                        atomicXor(reinterpret_cast<unsigned long long *>(&Pn[c * PSTR + (r >> 6)]), 1ull << (r & 63));
                        const int k = rowpiv[r];
                        if (k >= 0) atomicOr(reinterpret_cast<unsigned long long *>(&ndn[k]), 1ull << c);
                    }
                }
            }
#ifdef QD_OSD_TIMING
            if constexpr (QD_CS_SUB == 5) {                            // how long after the panel was finished does each wavefront reach the batch's last barrier?
                const unsigned long long tf_ = ((unsigned long long)misc[21] << 32) | misc[20];
                const int sl_ = wave == 1 ? 11 : (wave == 4 ? 12 : (wave == 2 ? 13 : (wave == 7 ? 14 : -1)));
                if (sl_ >= 0) acc_[sl_] += wall_clock64() - tf_;
            }
#endif
            if (wave == 0) __builtin_amdgcn_s_setprio(0);             // (held since the head of [B]: at priority 0 its few instructions between [B] and this barrier took ~100 ticks beside the others' Q update)
            QD_TICK(3)
            __syncthreads();
            QD_TICK(7)
            if (done) break;
        }
        if (wave == 1 && lane < NWD) sv[lane] = svx;                   // the syndrome back where the sweep reads it
        __syncthreads();
        asm volatile("" : "+v"(tid));                                  // (what the loops below derive from the thread index is not shared with -- kept live since -- the head of the shot)
        // ================================================================== OSD-0 solution, then the candidate sweep
        // residual on a non-pivot row <=> syndrome outside the column space (the answer is still the oracle's: same pivot rule)
        QD_SUBT0(3)
        uint64_t *mt = a.ws + (size_t)blockIdx.x * a.ws_words;                                   // [NPIV][NWD] Q columns, for the sweep (L2-resident)
        if (tid < NWD && (sv[tid] & unpm[tid]) != 0ull) atomicOr(&misc[2], 1u);      // (misc[0..63] was cleared at the head of the shot)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int k = kown(c);
            if (k < npiv) {
#pragma unroll
                for (int w = 0; w < NWD; ++w) mt[(size_t)k * NWD + w] = mycol[c][w];
            }
        }
        // signed pivot weights by row (a pivot that is on in the OSD-0 solution gets cheaper when flipped), and their sums per nibble
        for (int r = tid; r < NWD * 64; r += T) {
            int32_t v = 0;
            const int k = r < m ? (int)rowpiv[r] : -1;
            if (k >= 0) { const int32_t wgt = (int32_t)a.wfix[pcol[k]]; v = ((sv[r >> 6] >> (r & 63)) & 1ull) ? -wgt : wgt; }
            swl[r] = v;
        }
        __syncthreads();
        for (int x = tid; x < NWD * 16 * 16; x += T) {
            const int nb4 = x >> 4, pat = x & 15;
            int32_t s = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) s += ((pat >> b) & 1) ? swl[nb4 * 4 + b] : 0;
            nib[(x >> 8) * L::NIBW + (x & 255)] = s;                  // word (x >> 8), nibble position, pattern
        }
        __syncthreads();
        QD_SUBT(3, 11)
        const int nnp_all = n - npiv;
        // image of fault `col` under the complete transform, restricted to nothing: t ^= XOR over its pivoted rows r of (Q column of r + e_r)
        // image of fault `col` under the complete transform: t ^= XOR over its pivoted rows r of (Q column of r + e_r).  One thread, whole
        // vector: the patterns' base vectors and the winner only (the single-column candidates use the four-lane form below)
        auto add_col = [&](uint32_t col, uint64_t t[NWD]) {
            for (int q = 0; q < ellw; ++q) {
                const uint32_t r = a.csc_ell[((size_t)col << dlog) + q];
                if (r == 0xFFFFu) break;
                const int k = rowpiv[r];
                if (k < 0) continue;                                   // a non-pivot row weighs nothing
                const uint64_t *src = mt + (size_t)k * NWD;
                const int rw = (int)(r >> 6);
                const uint64_t rb = 1ull << (r & 63u);
#pragma unroll
                for (int ww = 0; ww < NWD; ++ww) t[ww] ^= src[ww] ^ ((ww == rw) ? rb : 0ull);
            }
        };
        auto wsum = [&](const uint64_t t[NWD]) -> long long {
            long long tot = 0;
#pragma unroll
            for (int w = 0; w < NWD; ++w) {
                const uint32_t lo = (uint32_t)t[w], hi = (uint32_t)(t[w] >> 32);
                int32_t s = 0;                                         // 16 nibbles x 4 weights below 2^25 each: no overflow (checked by the host)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    s += nib[w * L::NIBW + (q << 4) + ((lo >> (4 * q)) & 15u)];
                    s += nib[w * L::NIBW + ((8 + q) << 4) + ((hi >> (4 * q)) & 15u)];
                }
                tot += (long long)s;
            }
            return tot;
        };
        int dmax_hi_ = 0x7FFFFFFF;
        asm volatile("" : "+s"(dmax_hi_));                             // (a scalar made opaque here, every shot: as a vector constant hoisted out of the shot loop it was kept in scratch)
        CsBest best{(long long)(((unsigned long long)(uint32_t)dmax_hi_ << 32) | 0xFFFFFFFFull), 3u, ~0ull, 0ull};
        if (a.osd_w == 1) {
            // ---- singles: every non-pivot column; ties go to the earlier position of the order (ldpc's enumeration order).  FOUR lanes per
            // candidate, WPL words of the image each: a lane reads 16-byte pieces of the Q columns' lines, so one load instruction touches
            // 16 lines instead of 64 (the L1 looks up one line per clock: with a lane per candidate that lookup rate, not bandwidth, was
            // 13 % of a shot); the four partial sums meet by two lane exchanges.
            constexpr int WPL = (NWD + 3) / 4;
            const int sub = lane & 3, wbase = sub * WPL;
            for (int i0 = 0; i0 < n; i0 += T / 4) {
                const int i = i0 + (tid >> 2);
                uint32_t col = 0u;
                bool cand = false;
                if (i < n) { col = order[i]; cand = !((pivmask[col >> 5] >> (col & 31u)) & 1u); }
                if (__ballot(cand) == 0ull) continue;
                uint64_t t[WPL];
#pragma unroll
                for (int u = 0; u < WPL; ++u) t[u] = 0ull;
                if (cand) {
                    uint32_t wds[8];
                    if (dlog >= 3) {
                        const uint4 *e4 = reinterpret_cast<const uint4 *>(a.csc_ell + ((size_t)col << dlog));
                        const uint4 v0 = e4[0];
                        uint4 v1 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                        if (dlog == 4) v1 = e4[1];
                        wds[0] = v0.x; wds[1] = v0.y; wds[2] = v0.z; wds[3] = v0.w; wds[4] = v1.x; wds[5] = v1.y; wds[6] = v1.z; wds[7] = v1.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const uint32_t e0 = 2 * q < ellw ? (uint32_t)a.csc_ell[((size_t)col << dlog) + 2 * q] : 0xFFFFu;
                            const uint32_t e1 = 2 * q + 1 < ellw ? (uint32_t)a.csc_ell[((size_t)col << dlog) + 2 * q + 1] : 0xFFFFu;
                            wds[q] = e0 | (e1 << 16);
                        }
                    }
                    // four incidences at a time: their pivot orders, then their pieces of the Q columns, all in flight before the first is
                    // added (one candidate used to be a chain of up to eight dependent L2 round trips)
#pragma unroll
                    for (int q0 = 0; q0 < 16; q0 += 4) {
                        uint32_t r4[4];
#pragma unroll
                        for (int u4 = 0; u4 < 4; ++u4) { const int q = q0 + u4; r4[u4] = (q & 1) ? (wds[q >> 1] >> 16) : (wds[q >> 1] & 0xFFFFu); }
                        if (q0 >= ellw || __ballot(r4[0] != 0xFFFFu) == 0ull) break;      // (rows ascend: nothing behind the first padding entry)
                        int k4[4];
#pragma unroll
                        for (int u4 = 0; u4 < 4; ++u4) k4[u4] = r4[u4] != 0xFFFFu ? (int)rowpiv[r4[u4]] : -1;   // (a non-pivot row weighs nothing)
                        uint64_t v4[4][WPL];
#pragma unroll
                        for (int u4 = 0; u4 < 4; ++u4) {
                            const uint64_t *src = mt + (size_t)(k4[u4] >= 0 ? k4[u4] : 0) * NWD + wbase;
#pragma unroll
                            for (int u = 0; u < WPL; ++u) v4[u4][u] = (k4[u4] >= 0 && (NWD % 4 == 0 || wbase + u < NWD)) ? src[u] : 0ull;
                        }
#pragma unroll
                        for (int u4 = 0; u4 < 4; ++u4) {
                            const int rw = k4[u4] >= 0 ? (int)(r4[u4] >> 6) - wbase : -1;
                            const uint64_t rb = 1ull << (r4[u4] & 63u);
#pragma unroll
                            for (int u = 0; u < WPL; ++u) t[u] ^= v4[u4][u] ^ ((u == rw) ? rb : 0ull);
                        }
                    }
                }
                long long part = 0;
#pragma unroll
                for (int u = 0; u < WPL; ++u) {
                    const int ww = wbase + u;
                    if (NWD % 4 == 0 || ww < NWD) {
                        const uint32_t lo = (uint32_t)t[u], hi = (uint32_t)(t[u] >> 32);
                        int32_t sacc = 0;                              // 16 nibbles x 4 weights below 2^25 each: no overflow (checked by the host)
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            sacc += nib[ww * L::NIBW + (q << 4) + ((lo >> (4 * q)) & 15u)];
                            sacc += nib[ww * L::NIBW + ((8 + q) << 4) + ((hi >> (4 * q)) & 15u)];
                        }
                        part += (long long)sacc;
                    }
                }
#pragma unroll
                for (int sh = 1; sh <= 2; sh <<= 1)
                    part += ((long long)__shfl_xor((int)(part >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)part, sh);
                if (cand && sub == 0) {
                    const long long d = part + (long long)a.wfix[col];
                    if (qd_cs_less(d, 1u, (unsigned long long)i, best.delta, best.cls, best.tie)) best = CsBest{d, 1u, (unsigned long long)i, (unsigned long long)col};
                }
            }
        }
        QD_SUBT(3, 12)
        // ---- patterns over the first lam non-pivot columns of the order: pairs (combination sweep) or all subsets (exhaustive)
        const int lam = min(min(a.osd_order, nnp_all), 64);
        uint32_t *npl = misc + 256;                                    // [64] the first lam non-pivot faults of the order
        if (lam >= (a.osd_w == 1 ? 2 : 1)) {
            if (wave == 0) {
                int cnt = 0;
                for (int b0 = 0; b0 < n && cnt < lam; b0 += 64) {
                    const int i = b0 + lane;
                    uint32_t col = 0u;
                    bool np_ = false;
                    if (i < n) { col = order[i]; np_ = !((pivmask[col >> 5] >> (col & 31u)) & 1u); }
                    const unsigned long long bal = __ballot(np_);
                    const int at = cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    if (np_ && at < lam) npl[at] = col;
                    cnt += (int)__popcll(bal);
                }
            }
            __syncthreads();
            if (tid < lam) {
                uint64_t t[NWD];
#pragma unroll
                for (int w = 0; w < NWD; ++w) t[w] = 0ull;
                add_col(npl[tid], t);
#pragma unroll
                for (int w = 0; w < NWD; ++w) tvl[tid * NWD + w] = t[w];
            }
            __syncthreads();
            const unsigned long long npat = (a.osd_w == 1) ? (unsigned long long)lam * (lam - 1) / 2 : ((1ull << lam) - 1ull);
            for (unsigned long long ic = tid; ic < npat; ic += T) {
                unsigned long long pat;
                if (a.osd_w == 2) pat = ic + 1ull;
                else {
                    unsigned long long qq = ic; int x = 0;             // pairs (x, y), x < y < lam, lexicographic
                    while (qq >= (unsigned long long)(lam - 1 - x)) { qq -= (unsigned long long)(lam - 1 - x); ++x; }
                    pat = (1ull << x) | (1ull << (x + 1 + (int)qq));
                }
                uint64_t t[NWD];
#pragma unroll
                for (int w = 0; w < NWD; ++w) t[w] = 0ull;
                long long d = 0;
                for (int b = 0; b < lam; ++b)
                    if ((pat >> b) & 1ull) {
                        d += (long long)a.wfix[npl[b]];
#pragma unroll
                        for (int w = 0; w < NWD; ++w) t[w] ^= tvl[b * NWD + w];
                    }
                d += wsum(t);
                if (qd_cs_less(d, 2u, ic, best.delta, best.cls, best.tie)) best = CsBest{d, 2u, ic, pat};
            }
        }
        QD_SUBT(3, 13)
        // ---- winner: wavefront minimum by shuffles, then the NW partials
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            CsBest o;
            o.delta = ((long long)__shfl_xor((int)(best.delta >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)best.delta, sh);
            o.cls = (uint32_t)__shfl_xor((int)best.cls, sh);
            o.tie = ((unsigned long long)(uint32_t)__shfl_xor((int)(best.tie >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)best.tie, sh);
            o.what = ((unsigned long long)(uint32_t)__shfl_xor((int)(best.what >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)best.what, sh);
            if (qd_cs_less(o.delta, o.cls, o.tie, best.delta, best.cls, best.tie)) best = o;
        }
        CsBest *bests = reinterpret_cast<CsBest *>(misc + 128);       // [NW <= 16] x 32 bytes
        uint64_t *twin = Tp;                                           // [NWD] the winner's image (the panel scratch is idle)
        if (lane == 0) bests[wave] = best;
        __syncthreads();
        CsBest win = bests[0];
        for (int q = 1; q < NW; ++q) { const CsBest c = bests[q]; if (qd_cs_less(c.delta, c.cls, c.tie, win.delta, win.cls, win.tie)) win = c; }
        const bool take = win.delta < 0;                               // OSD-0 unless a candidate is strictly cheaper
        if (tid == 0) {
            uint64_t t[NWD];
#pragma unroll
            for (int w = 0; w < NWD; ++w) t[w] = 0ull;
            if (take) {
                if (win.cls == 1u) add_col((uint32_t)win.what, t);
                else for (int b = 0; b < lam; ++b) if ((win.what >> b) & 1ull) add_col(npl[b], t);
            }
#pragma unroll
            for (int w = 0; w < NWD; ++w) twin[w] = t[w];
        }
        __syncthreads();
        for (int k = tid; k < npiv; k += T) {
            const int r = (int)prow[k];
            if ((((sv[r >> 6] ^ twin[r >> 6]) >> (r & 63)) & 1ull) != 0ull) {
                const uint32_t j = pcol[k];
                atomicOr(&reinterpret_cast<uint32_t *>(smem + a.o_out)[j >> 5], 1u << (j & 31u));
            }
        }
        if (tid == 0 && take) {
            if (win.cls == 1u) atomicOr(&reinterpret_cast<uint32_t *>(smem + a.o_out)[(uint32_t)win.what >> 5], 1u << ((uint32_t)win.what & 31u));
            else for (int b = 0; b < lam; ++b) if ((win.what >> b) & 1ull) atomicOr(&reinterpret_cast<uint32_t *>(smem + a.o_out)[npl[b] >> 5], 1u << (npl[b] & 31u));
        }
        __syncthreads();
        for (int w = tid; w < a.out_words; w += T) a.err_bits[shot * a.out_words + w] = reinterpret_cast<uint32_t *>(smem + a.o_out)[w];
        if (tid == 0) a.status[shot] = (a.status[shot] & 0xFFFF) | (1 << 17) | (misc[2] ? (1 << 18) : 0) | (min(npiv, 4095) << 20);
        QD_SUBT(3, 14)
        QD_TICK(4)
#ifdef QD_OSD_TIMING
        if (tid == 0) {
            for (int i = 0; i < 8; ++i) atomicAdd(&a.dbg[i], acc_[i]);
            atomicAdd(&a.dbg[8], 1ull); atomicAdd(&a.dbg[9], (unsigned long long)npiv); atomicAdd(&a.dbg[10], acc_[10]);
            if constexpr (QD_CS_SUB != 5) for (int i = 11; i < 15; ++i) atomicAdd(&a.dbg[i], acc_[i]);
        }
        if constexpr (QD_CS_SUB == 5) {
            if (lane == 0 && (wave == 1 || wave == 4 || wave == 2 || wave == 7)) { const int sl_ = wave == 1 ? 11 : (wave == 4 ? 12 : (wave == 2 ? 13 : 14)); atomicAdd(&a.dbg[sl_], acc_[sl_]); }
        }
#endif
        __syncthreads();   // LDS is recycled by the next shot
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
// Instantiations: variant 1: m <= 512 (256 threads x 2 columns x 8 words), 2: m <= 1024 (512 x 2 x 16), 3: m <= 1408 (512 x 3 x 22).
struct CsShape { int T, CPT, NWD, WPS, IPT; };
static CsShape cs_shape(int variant)
{
    switch (variant) {
    case 1: return CsShape{256, 2, 8, 5, 16};
    case 2: return CsShape{512, 2, 16, 4, 20};
    default: return CsShape{512, 3, 22, 2, 40};
    }
}

// LDS layout for a window of m detectors, n faults; returns the bytes (0: this kernel does not take the window), the instantiation
// and how many workgroups share a CU.  off[0..2] = the runtime offsets of OsdCsArgs (out, order, sort_aux).
int qd_osdcs_layout(int m, int n, int out_words, uint32_t max_wfix, int *off, int *variant, int *per_cu)
{
    const int var = m <= 512 ? 1 : (m <= 1024 ? 2 : (m <= 1408 ? 3 : 0));
    if (var == 0 || n >= 65535) return 0;
    const CsShape sh = cs_shape(var);
    if (n > sh.T * sh.IPT) return 0;
    if ((uint64_t)max_wfix * 64ull >= 0x7FFFFFFFull) return 0;        // the sweep adds 64 weights in 32 bits
    auto al = [](int x) { return (x + 15) & ~15; };
    const int o_var = var == 1 ? CsLds<256, 2, 8>::o_var : (var == 2 ? CsLds<512, 2, 16>::o_var : CsLds<512, 3, 22>::o_var);
    int o = o_var + al(out_words * 4);                // pivmask
    off[0] = o; o += al(out_words * 4);               // out
    off[1] = o; o += al(n * 2);                       // order
    // the sort phase owns everything: n keys of 8 bytes, then splitters / counters
    off[2] = al(n * 8);
    const int sort_end = off[2] + 2048;
    const int total = std::max(o, sort_end);
    const int by_regs = sh.WPS * 256 / sh.T;          // workgroups per CU the register budget allows
    const int by_lds = QD_LDS_BYTES / total;
    if (by_lds < 1) return 0;
    *variant = var;
    *per_cu = std::max(1, std::min(by_regs, by_lds));
    return total;
}

size_t qd_osdcs_ws_words(int variant)
{
    const CsShape sh = cs_shape(variant);
    return (size_t)sh.T * sh.CPT * sh.NWD + 64;
}

hipError_t qd_launch_osdcs(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, const int *off, int variant, int lds,
                           uint64_t *ws, int blocks, hipStream_t s)
{
    OsdCsArgs r{};
    r.m = g.m; r.n = g.n; r.n_pad = bg.n_pad; r.out_words = bg.out_words; r.upd_rows = a.upd_rows; r.ell_log2 = g.ell_log2; r.rank = a.rank;
    r.osd_w = a.osd_w; r.osd_order = a.osd_order;
    r.o_out = off[0]; r.o_order = off[1]; r.o_sort_aux = off[2];
    r.csc_ell = g.csc_ell; r.wfix = g.wfix; r.bit_orig = bg.bit_orig;
    r.det = a.det; r.upd = a.upd; r.det_stride = a.det_stride; r.det_offset = a.det_offset; r.upd_stride = a.upd_stride;
    r.llr_ws = a.llr_ws; r.fail_list = a.fail_list; r.fail_count = a.fail_count;
    r.ws = ws; r.ws_words = qd_osdcs_ws_words(variant);
    r.err_bits = a.err_bits; r.status = a.status; r.dbg = a.dbg;
#define QD_CS_CASE(TT, CC, WW, SS, II)                                                                                            \
    {                                                                                                                             \
        auto k = qd_osdcs_kernel<TT, CC, WW, SS, II>;                                                                             \
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                     \
        if (e != hipSuccess) return e;                                                                                            \
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(TT), lds, s, r);                                                       \
        return hipGetLastError();                                                                                                 \
    }
    switch (variant) {
    case 1: QD_CS_CASE(256, 2, 8, 5, 16)
    case 2: QD_CS_CASE(512, 2, 16, 4, 20)
    default: QD_CS_CASE(512, 3, 22, 2, 40)
    }
#undef QD_CS_CASE
}
