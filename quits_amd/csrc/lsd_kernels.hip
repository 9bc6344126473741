// lsd_kernels.hip -- BP-LSD post-processing (localized statistics decoding, LSD-0) for the shots BP could not finish:
// one WAVEFRONT per shot.
//
// Replaces ldpc.bplsd_decoder.BpLsdDecoder.decode -> LsdDecoder::lsd_decode (ldpc 2.x src_cpp/lsd.hpp) with ldpc's defaults
// bits_per_step = 1, lsd_order = 0, as the reference reaches it through quits/decoder/bplsd.py:51,86.  CPU restatement with
// the same rules, bit for bit: oracle/qd_oracle.c, oq_lsd0 (which also states what is fixed where ldpc leaves it open).
//
//   - every unsatisfied check seeds a cluster; while invalid clusters exist, the clusters that are invalid at the start of a
//     round grow by one fault each in (size, id) order: the not yet used fault with the lowest posterior LLR among those
//     touching the cluster's checks joins, its checks join, clusters owning one of them are absorbed;
//   - validity = the cluster's part of the syndrome lies in the span of its faults' columns, by on-the-fly GF(2) elimination.
//     Clusters own disjoint checks, so ONE elimination over all rows serves them all: the T-form bookkeeping of the OSD
//     kernels (Q[r] = pivot rows added to row r, indexed by pivot order; transformed syndrome bit per row);
//   - correction: err[pivot column k] = transformed syndrome at pivot row k.
//
// Mapping: the work of one shot is a chain of ~100 dependent growth steps on clusters of a few dozen checks -- far too little
// for a workgroup, so a shot gets one wavefront and a CU runs eight shots side by side (19.7 KB of LDS per shot at the headline
// window; shots are handed out through a work counter, they differ by 10x in steps).  A lane owns a BLOCK of NR consecutive
// checks, so every scan over the rows (best candidate of a cluster, absorbing a cluster, image of the new column, pivot update,
// validity) is a few 128-bit LDS loads plus register work.  Each check caches its best candidate (lowest (LLR, index) among its
// unused faults); rescans -- checks that joined, or whose cached fault was just used -- read the row in ELL form {fault,
// posterior column}, QL_NB rows in flight at a time, two memory latencies per batch, overlapped with the step's elimination.
// The Q planes live in a per-slot HBM workspace (L2-resident; a lane reads and writes its own block of rows with 128-bit
// accesses): keeping two of them in LDS (39 KB per shot, four shots per CU) measured 38.4 ms per 65536 headline shots, none
// 32.6 ms -- a growth step is a chain of dependent instructions, and what hides it is more shots per SIMD.
#include "qd_internal.h"
#include "../../include/quits_amd.h"

#define QL_NONE 0xFFFFu
#define QL_NOKEY64 0xFFFFFFFFFFFFFFFFull

struct LsdArgs {
    int m, n, n_pad, mw, out_words, upd_rows;
    const int32_t *cp, *ri;        // CSC of the window matrix
    const int2 *ell;               // [m][ell_w] rows in ELL form: {fault, its posterior column}, {-1, 0} padding
    int ell_w;                     // multiple of 64
    const uint8_t *det, *upd;
    int64_t det_stride, det_offset, upd_stride;
    const float *llr_ws;           // [fail slot][n_pad]
    const int32_t *fail_list, *fail_count;
    uint64_t *q_ws;                // [blocks][mw][64 * NR] Q planes (planes 0 and 1 unused: they live in LDS)
    int32_t *next_slot;            // work counter, zero at launch
    uint16_t *pcol_ws;             // [blocks][64 * NR] pivot row -> its fault
    // higher-order LSD (lsd_method 'lsd_cs' / 'lsd_e' with lsd_order > 0): 0 = LSD-0, 1 = combination sweep, 2 = exhaustive
    int lsd_w, order;
    const uint32_t *wfix;          // [n] integer candidate costs round(log(1/p) * 2^18) (as for OSD-CS / OSD-E)
    const uint32_t *slot_of;       // [n] fault -> its posterior column in llr_ws
    uint16_t *npl_ws;              // [blocks][npl_cap] the added faults that did not become pivot columns, in order of addition
    uint16_t *nown_ws;             // [blocks][npl_cap] ... the cluster each belongs to at sweep time
    uint32_t *nkey_ws;             // [blocks][npl_cap] ... and its sort key (monotone image of the posterior LLR)
    int npl_cap;
    uint32_t *tv_ws;               // [blocks][QL_TV_WORDS] images of the first 64 sorted non-pivot faults of the cluster at hand
                                   //                       ([i][lane] = bit k: row base + k), then their fault indices
    uint32_t *err_bits;
    int32_t *status;
};

// LDS carve-up for NR checks per lane: compile-time offsets, so that a lane's block of NR consecutive rows is read with
// 128-bit LDS loads.  The two bitmaps (sizes depend on n) come last.
#ifndef QL_LDS_PLANES
#define QL_LDS_PLANES 0    // Q planes kept in LDS (pivot orders 0 .. 64 * QL_LDS_PLANES - 1); the rest lives in the HBM slot
#endif
template <int NR> struct QlLay {
    static constexpr int MP = 64 * NR;
    static constexpr int o_ql = 0;                       // u64 [2][MP]  Q planes 0 and 1 (pivot orders 0..127)
    static constexpr int o_bkey = o_ql + 8 * QL_LDS_PLANES * MP;        // u32 [MP]     check -> LLR key of its best unused fault
    static constexpr int o_rl = o_bkey + 4 * MP;         // u32 [MP]     the round: size at its start << 16 | cluster id
    static constexpr int o_owner = o_rl + 4 * MP;        // u16 [MP]     check -> cluster id (= seed check), QL_NONE = free
    static constexpr int o_bj = o_owner + 2 * MP;        // u16 [MP]     check -> its best unused fault
    static constexpr int o_rowpiv = o_bj + 2 * MP;       // i16 [MP]     check -> pivot order, -1 = not a pivot row
    static constexpr int o_cnbits = o_rowpiv + 2 * MP;   // u16 [MP]     per cluster id: faults added
    static constexpr int o_flags = o_cnbits + 2 * MP;    // u8  [MP]     bit 0 transformed syndrome, 1 image of the column at hand, 2 pivot row
    static constexpr int o_cstate = o_flags + MP;        // u8  [MP]     per cluster id: 0 none, 1 invalid, 2 valid, 3 gone
    static constexpr int o_rs = o_cstate + MP;           // u32 [32]     checks whose candidate has to be (re)computed
    static constexpr int o_added = o_rs + 128;           // u32 [ceil(n / 32)] fault bitmap
};
#define QL_F_SP 1u
#define QL_F_T 2u
#define QL_F_PIV 4u

__device__ __forceinline__ uint32_t ql_mono_key(float llr)
{
    const float f = llr + 0.0f;                    // -0 -> +0: they tie on the index like the oracle's '<' on doubles
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

#ifdef QD_LSD_TIMING       // debug build: cycles per phase, summed over all shots, printed after the launch
#define QL_T0() long long t0_ = clock64()
#define QL_T(k) do { const long long t1_ = clock64(); tacc[k] += t1_ - t0_; t0_ = t1_; } while (0)
#define QL_CNT(k, v) do { tacc[k] += (v); } while (0)
#else
#define QL_T0() do {} while (0)
#define QL_T(k) do {} while (0)
#define QL_CNT(k, v) do {} while (0)
#endif
#define QL_TV_MAX 64       // candidate positions whose images are kept (pairs of 'lsd_cs' among the first min(order, 64); 'lsd_e': 15)
#define QL_TV_WORDS (QL_TV_MAX * 64 + QL_TV_MAX)
#define QL_NB 8            // rows per batch of the candidate scan: their loads are in flight together (16 measured 18 % slower: registers)

__device__ __forceinline__ long long ql_wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NR>
__global__ void __launch_bounds__(64) qd_lsd0_kernel(LsdArgs a)
{
    using Lay = QlLay<NR>;
    constexpr int MP = Lay::MP;
    struct alignas(16) BlkU64 { uint64_t v[NR]; };
    struct alignas(16) BlkU32 { uint32_t v[NR]; };
    struct alignas(16) BlkU16 { uint16_t v[NR]; };
    struct alignas(8) BlkU8 { uint8_t v[NR]; };
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *ql = reinterpret_cast<uint64_t *>(smem + Lay::o_ql);
    uint32_t *bkey = reinterpret_cast<uint32_t *>(smem + Lay::o_bkey);
    uint32_t *rl = reinterpret_cast<uint32_t *>(smem + Lay::o_rl);
    uint16_t *owner = reinterpret_cast<uint16_t *>(smem + Lay::o_owner);
    uint16_t *bj = reinterpret_cast<uint16_t *>(smem + Lay::o_bj);
    int16_t *rowpiv = reinterpret_cast<int16_t *>(smem + Lay::o_rowpiv);
    uint16_t *pcol = a.pcol_ws + (size_t)blockIdx.x * MP;                     // pivot row -> its fault: written once per pivot, read at the end (HBM)
    uint16_t *cnbits = reinterpret_cast<uint16_t *>(smem + Lay::o_cnbits);
    uint8_t *flags = smem + Lay::o_flags;
    uint8_t *cstate = smem + Lay::o_cstate;
    uint32_t *rs = reinterpret_cast<uint32_t *>(smem + Lay::o_rs);
    uint32_t *added = reinterpret_cast<uint32_t *>(smem + Lay::o_added);
    const int added_words = ((a.n + 31) / 32 + 3) & ~3;
    const int lane = threadIdx.x, base = lane * NR;                           // the lane's block of rows
    const int m = a.m;
    const int nfail = *a.fail_count;
    uint64_t *Q = a.q_ws + (size_t)blockIdx.x * (size_t)a.mw * MP;            // planes >= 2 (rare): HBM, per resident slot
    uint16_t *npl = a.npl_ws + (size_t)blockIdx.x * a.npl_cap;
    uint16_t *nown = a.nown_ws + (size_t)blockIdx.x * a.npl_cap;
    uint32_t *nkey = a.nkey_ws + (size_t)blockIdx.x * a.npl_cap;
    uint32_t *tv = a.tv_ws + (size_t)blockIdx.x * QL_TV_WORDS;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    BlkU16 &own_blk = *reinterpret_cast<BlkU16 *>(owner + base);
    BlkU8 &fl_blk = *reinterpret_cast<BlkU8 *>(flags + base);
#ifdef QD_LSD_TIMING
    long long tacc[16] = {0};
#endif
    QL_T0();

    for (;;) {
        // shots differ by an order of magnitude in growth steps: slots are handed out one at a time
        int slot = 0;
        if (lane == 0) slot = atomicAdd(a.next_slot, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= nfail) break;
        const int64_t shot = a.fail_list[slot];
        const float *llr = a.llr_ws + (int64_t)slot * a.n_pad;
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;

        // best unused fault of the checks list[0..cnt): lane = entry of the row, QL_NB rows per trip so that a trip costs two
        // memory latencies (row entries, then their LLRs) whatever the number of rows
        auto scan_list = [&](const uint32_t *list, int cnt) {
            for (int b0 = 0; b0 < cnt; b0 += QL_NB) {
                int row[QL_NB];
                uint64_t key[QL_NB];
#pragma unroll
                for (int b = 0; b < QL_NB; ++b) {
                    row[b] = b0 + b < cnt ? __builtin_amdgcn_readfirstlane((int)(list[b0 + b] & 0xFFFFu)) : -1;
                    key[b] = QL_NOKEY64;
                }
                for (int wc = 0; wc < a.ell_w; wc += 64) {
                    int2 ent[QL_NB];
                    float lv[QL_NB];
#pragma unroll
                    for (int b = 0; b < QL_NB; ++b)
                        ent[b] = row[b] >= 0 ? a.ell[(size_t)row[b] * a.ell_w + wc + lane] : make_int2(-1, 0);
#pragma unroll
                    for (int b = 0; b < QL_NB; ++b) lv[b] = ent[b].x >= 0 ? llr[ent[b].y] : 0.0f;
#pragma unroll
                    for (int b = 0; b < QL_NB; ++b) {
                        const uint32_t j = (uint32_t)ent[b].x;
                        if (ent[b].x >= 0 && !((added[j >> 5] >> (j & 31u)) & 1u)) {
                            const uint64_t k = ((uint64_t)ql_mono_key(lv[b]) << 16) | j;
                            key[b] = k < key[b] ? k : key[b];
                        }
                    }
                }
#pragma unroll
                for (int b = 0; b < QL_NB; ++b) {
                    if (row[b] < 0) break;                                     // uniform
                    const uint32_t hi = (uint32_t)(key[b] >> 16), mh = qd_wave_umin(hi);
                    const uint32_t lo = (hi == mh) ? (uint32_t)(key[b] & 0xFFFFu) : 0xFFFFu;
                    const uint32_t ml = qd_wave_umin(lo);
                    if (lane == 0) { bkey[row[b]] = mh; bj[row[b]] = (uint16_t)ml; }     // nothing left: 0xFFFFFFFF, 0xFFFF
                }
            }
            __syncthreads();
        };

        // ---- per shot: every unsatisfied check seeds a cluster
        int nseed = 0;
        {
            BlkU16 own, allff;
            BlkU8 fl, cs;
            BlkU32 kff;
            BlkU64 zero;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int r = base + k;
                uint32_t s = 0;
                if (r < m) {
                    s = det[r] & 1u;
                    if (upd && r < a.upd_rows) s ^= upd[r] & 1u;
                }
                own.v[k] = s ? (uint16_t)r : (uint16_t)QL_NONE; fl.v[k] = (uint8_t)s; cs.v[k] = s ? 1 : 0;
                allff.v[k] = 0xFFFFu; kff.v[k] = 0xFFFFFFFFu; zero.v[k] = 0ull;
                const unsigned long long bs = __ballot(s != 0u);
                if (s) rl[nseed + __popcll(bs & lt_mask)] = (uint32_t)r;       // size 0 << 16 | id
                nseed += __popcll(bs);
            }
            own_blk = own; fl_blk = fl;
            *reinterpret_cast<BlkU8 *>(cstate + base) = cs;
            *reinterpret_cast<BlkU16 *>(bj + base) = allff;
            *reinterpret_cast<BlkU16 *>(rowpiv + base) = allff;                // -1
            *reinterpret_cast<BlkU32 *>(bkey + base) = kff;
#pragma unroll
            for (int w = 0; w < QL_LDS_PLANES; ++w) *reinterpret_cast<BlkU64 *>(ql + w * MP + base) = zero;   // later planes are cleared when first used
            BlkU16 z16;
#pragma unroll
            for (int k = 0; k < NR; ++k) z16.v[k] = 0;
            *reinterpret_cast<BlkU16 *>(cnbits + base) = z16;
        }
        for (int w = lane; w < added_words; w += 64) added[w] = 0u;
        for (int w = lane; w < a.out_words; w += 64) a.err_bits[shot * a.out_words + w] = 0u;   // the correction is ORed in at the end
        __syncthreads();
        QL_T(0);
        scan_list(rl, nseed);                                                  // seeds: the candidate of every unsatisfied check
        QL_T(1); QL_CNT(12, 1);

        int npiv = 0, inconsistent = 0, nrl = nseed, nnp = 0;
        int mode = 0, pc = -1, pcnt = 0;                                       // mode 1: growth stage of the higher orders (below)
        for (;;) {
            if (mode == 0) {
                // ---- this round: the clusters still invalid, by (size now, id); they all were in the previous round's list
                int out = 0;
                for (int b0 = 0; b0 < nrl; b0 += 64) {
                    const int idx = b0 + lane;
                    const int c = idx < nrl ? (int)(rl[idx] & 0xFFFFu) : 0;
                    const bool inv = idx < nrl && cstate[c] == 1;
                    const uint32_t v = ((uint32_t)cnbits[c] << 16) | (uint32_t)c;
                    const unsigned long long bi = __ballot(inv);
                    if (inv) rl[out + __popcll(bi & lt_mask)] = v;             // out <= b0: never ahead of the reads
                    out += __popcll(bi);
                }
                nrl = out;
                __syncthreads();
                QL_T(2); QL_CNT(13, 1);
                if (nrl == 0) {
                    if (a.lsd_w == 0) break;
                    mode = 1;
                }
            }
            if (mode == 1) {
                // ---- higher-order LSD, growth stage (oracle oq_lsd, step 1): every cluster is valid now; in ascending id, a
                // cluster with fewer than `order` non-pivot faults takes up to `order` further faults, by the same growth step
                // (merges included).  One step per trip of this loop: the "round" is that one cluster.
                int c1 = -1;
                for (;;) {
                    if (pc >= 0 && pcnt < a.order && __builtin_amdgcn_readfirstlane((int)cstate[pc]) == 2) {
                        const BlkU16 own = own_blk;
                        const BlkU8 fl = fl_blk;
                        int pv = 0;
#pragma unroll
                        for (int k = 0; k < NR; ++k) pv += (own.v[k] == (uint16_t)pc && (fl.v[k] & QL_F_PIV)) ? 1 : 0;
                        const int dim = __builtin_amdgcn_readfirstlane((int)cnbits[pc]) - (int)qd_wave_add((uint32_t)pv);
                        if (dim < a.order) { c1 = pc; break; }
                    }
                    const BlkU8 cs = *reinterpret_cast<const BlkU8 *>(cstate + base);
                    uint32_t cm = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = 0; k < NR; ++k) if (cs.v[k] == 2 && base + k > pc) cm = min(cm, (uint32_t)(base + k));
                    cm = qd_wave_umin(cm);
                    if (cm == 0xFFFFFFFFu) break;
                    pc = (int)cm; pcnt = 0;
                }
                if (c1 < 0) break;
                pcnt++;
                if (lane == 0) rl[0] = (uint32_t)c1;
                nrl = 1;
                __syncthreads();
            }
            long long last = -1;
            for (;;) {
                uint32_t kk = 0xFFFFFFFFu;
                for (int idx = lane; idx < nrl; idx += 64) { const uint32_t v = rl[idx]; if ((long long)v > last) kk = min(kk, v); }
                kk = qd_wave_umin(kk);
                if (kk == 0xFFFFFFFFu) break;
                last = (long long)kk;
                const int c = (int)(kk & 0xFFFFu);
                QL_T(3);
                if (mode == 0 && __builtin_amdgcn_readfirstlane((int)cstate[c]) != 1) continue;   // became valid or was absorbed earlier in the round
                QL_CNT(14, 1);
                // ---- the fault that joins: lowest (LLR, index) among the cached candidates of the cluster's checks
                int j;
                {
                    const BlkU16 own = own_blk;
                    const BlkU32 bk = *reinterpret_cast<const BlkU32 *>(bkey + base);
                    uint32_t kmin = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = 0; k < NR; ++k) if (own.v[k] == (uint16_t)c) kmin = min(kmin, bk.v[k]);
                    const uint32_t K = qd_wave_umin(kmin);
                    if (K == 0xFFFFFFFFu) {                                    // nothing left to add: the syndrome is outside the column space
                        if (mode == 1) { pcnt = a.order; continue; }           // (growth stage: the cluster is valid and simply complete)
                        if (lane == 0) cstate[c] = 3;
                        inconsistent = 1;
                        __syncthreads();
                        continue;
                    }
                    const BlkU16 bjb = *reinterpret_cast<const BlkU16 *>(bj + base);
                    uint32_t jm = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = 0; k < NR; ++k) if (own.v[k] == (uint16_t)c && bk.v[k] == K) jm = min(jm, (uint32_t)bjb.v[k]);
                    j = (int)qd_wave_umin(jm);
                }
                QL_T(4);
                if (lane == 0) { added[j >> 5] |= 1u << (j & 31); cnbits[c] = (uint16_t)(cnbits[c] + 1); }
                const int c0 = a.cp[j], c1 = a.cp[j + 1], deg = c1 - c0;
                // the column's rows, once, into the first lanes (one vector load instead of a dependent scalar load per row)
                const int myrow = lane < deg ? a.ri[c0 + lane] : -1;
                __syncthreads();
                // ---- its checks join; clusters owning one of them are absorbed; cached candidates that were this fault are redone.
                // Lane x < deg holds row x of the column: owners and cached candidates are read once, free checks join in one
                // store, and only the absorptions (about one per step) walk the owner blocks.
                const int dv = lane < deg ? (int)owner[myrow] : (int)c;
                const int bjv = lane < deg ? (int)bj[myrow] : -1;
                if (lane < deg && dv == QL_NONE) owner[myrow] = (uint16_t)c;
                const unsigned long long bres = __ballot(lane < deg && (dv == QL_NONE || bjv == j));
                if ((bres >> lane) & 1ull) rs[__popcll(bres & lt_mask)] = (uint32_t)myrow;
                const int nrs = __popcll(bres);
                for (unsigned long long bm = __ballot(dv != QL_NONE && dv != c); bm; ) {
                    const int d = __builtin_amdgcn_readlane(dv, (int)__builtin_ctzll(bm));
                    bm &= ~__ballot(dv == d);                                  // every row of the column owned by d
                    BlkU16 o2 = own_blk;
#pragma unroll
                    for (int k = 0; k < NR; ++k) o2.v[k] = o2.v[k] == (uint16_t)d ? (uint16_t)c : o2.v[k];
                    own_blk = o2;
                    if (lane == 0) { cnbits[c] = (uint16_t)(cnbits[c] + cnbits[d]); cstate[d] = 3; }
                }
                __syncthreads();
                QL_T(5);
                // the rescans: their loads are issued here and consumed after the elimination (nothing in between reads
                // the cached candidates), so the two memory latencies overlap with it
                const bool overlap = nrs <= QL_NB && a.ell_w == 64;
                if (!overlap) scan_list(rs, nrs);
                int srow[QL_NB];
                int2 sent[QL_NB];
#pragma unroll
                for (int b = 0; b < QL_NB; ++b) {
                    srow[b] = overlap && b < nrs ? __builtin_amdgcn_readfirstlane((int)rs[b]) : -1;
                    sent[b] = srow[b] >= 0 ? a.ell[(size_t)srow[b] * 64 + lane] : make_int2(-1, 0);
                }
                QL_T(6);
                // ---- the column through the elimination: t = column + the pivot columns of its pivoted rows, on the cluster's rows
                const int mypk = lane < deg ? (int)rowpiv[myrow] : -1;
                if (lane < deg) flags[myrow] |= (uint8_t)QL_F_T;
                const unsigned long long bpk = __ballot(mypk >= 0);
                __syncthreads();
                const BlkU16 own = own_blk;
                BlkU8 fl = fl_blk;
                uint32_t memb = 0, tbits = 0, pivb = 0, spb = 0;              // bit k: row base + k
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const uint32_t f = fl.v[k];
                    if (own.v[k] == (uint16_t)c) memb |= 1u << k;
                    tbits |= ((f >> 1) & 1u) << k; pivb |= ((f >> 2) & 1u) << k; spb |= (f & 1u) << k;
                }
                tbits &= memb;
                const int kwmax = (npiv + 63) / 64 - 1;
                for (int w = 0; w <= kwmax; ++w) {
                    uint64_t mkw = 0ull;                                       // orders, in plane w, of the column's pivoted rows
                    for (unsigned long long bb = bpk; bb; bb &= bb - 1ull) {
                        const int pk = __builtin_amdgcn_readlane(mypk, (int)__builtin_ctzll(bb));
                        if ((pk >> 6) == w) mkw |= 1ull << (pk & 63);
                    }
                    if (mkw == 0ull || memb == 0u) continue;
                    BlkU64 qb;
                    if (w < QL_LDS_PLANES) qb = *reinterpret_cast<const BlkU64 *>(ql + w * MP + base);
                    else qb = *reinterpret_cast<const BlkU64 *>(Q + (size_t)w * MP + base);
#pragma unroll
                    for (int k = 0; k < NR; ++k) tbits ^= ((uint32_t)__popcll(qb.v[k] & mkw) & 1u) << k;
                    tbits &= memb;
                }
                const uint32_t cand = tbits & ~pivb;                           // rows that may become the pivot: lowest index wins
                const uint32_t pkey = qd_wave_umin(cand ? (uint32_t)(base + __builtin_ctz(cand)) : 0xFFFFFFFFu);
                float slv[QL_NB];
#pragma unroll
                for (int b = 0; b < QL_NB; ++b) slv[b] = sent[b].x >= 0 ? llr[sent[b].y] : 0.0f;
                QL_T(7);
                if (pkey != 0xFFFFFFFFu) {
                    const int p = (int)pkey, K = npiv, kw = K >> 6;
                    const uint64_t kbit = 1ull << (K & 63);
                    if ((K & 63) == 0 && kw >= QL_LDS_PLANES) {                // a new HBM plane comes into use: clear it
                        BlkU64 zero;
#pragma unroll
                        for (int k = 0; k < NR; ++k) zero.v[k] = 0ull;
                        *reinterpret_cast<BlkU64 *>(Q + (size_t)kw * MP + base) = zero;
                        __syncthreads();
                    }
                    const uint32_t mine = (p >= base && p < base + NR) ? 1u << (p - base) : 0u;
                    const uint32_t spp = (uint32_t)__builtin_amdgcn_readfirstlane((int)flags[p]) & QL_F_SP;
                    const uint32_t updm = tbits & ~mine;                       // rows the pivot row is added to
                    for (int w = 0; w <= kw; ++w) {
                        if (w < QL_LDS_PLANES) {
                            uint64_t *qw = ql + w * MP;
                            uint64_t x = qw[p];
                            if (w == kw) x ^= kbit;
                            if (updm) {
                                BlkU64 qb = *reinterpret_cast<const BlkU64 *>(qw + base);
#pragma unroll
                                for (int k = 0; k < NR; ++k) qb.v[k] ^= ((updm >> k) & 1u) ? x : 0ull;
                                *reinterpret_cast<BlkU64 *>(qw + base) = qb;
                            }
                        } else {
                            uint64_t *qw = Q + (size_t)w * MP;
                            uint64_t x = qw[p];
                            if (w == kw) x ^= kbit;
                            if (updm) {
                                BlkU64 qb = *reinterpret_cast<const BlkU64 *>(qw + base);
#pragma unroll
                                for (int k = 0; k < NR; ++k) qb.v[k] ^= ((updm >> k) & 1u) ? x : 0ull;
                                *reinterpret_cast<BlkU64 *>(qw + base) = qb;
                            }
                        }
                    }
                    if (spp) spb ^= updm;
                    pivb |= mine;
                    if (lane == 0) { rowpiv[p] = (int16_t)K; pcol[p] = (uint16_t)j; }
                    npiv = K + 1;
                } else if (a.lsd_w) {                                          // dependent on the cluster's pivot columns: a candidate position of the sweep
                    if (lane == 0) npl[nnp] = (uint16_t)j;
                    nnp++;
                }
                // ---- flags back (image cleared); valid once no unpivoted check of the cluster carries syndrome
#pragma unroll
                for (int k = 0; k < NR; ++k) fl.v[k] = (uint8_t)(((spb >> k) & 1u) | (((pivb >> k) & 1u) << 2));
                fl_blk = fl;
                const bool anybad = __ballot((memb & spb & ~pivb) != 0u) != 0ull;
                if (lane == 0) cstate[c] = anybad ? 1 : 2;
                QL_T(9);
#pragma unroll
                for (int b = 0; b < QL_NB; ++b) {
                    if (srow[b] < 0) break;                                    // uniform
                    const uint32_t jj = (uint32_t)sent[b].x;
                    const bool ok = sent[b].x >= 0 && !((added[jj >> 5] >> (jj & 31u)) & 1u);
                    const uint32_t hi = ok ? ql_mono_key(slv[b]) : 0xFFFFFFFFu, mh = qd_wave_umin(hi);
                    const uint32_t ml = qd_wave_umin((ok && hi == mh) ? jj : 0xFFFFu);
                    if (lane == 0) { bkey[srow[b]] = mh; bj[srow[b]] = (uint16_t)ml; }
                }
                __syncthreads();
                QL_T(8);
            }
            QL_T(3);
            __syncthreads();
        }
        if (a.lsd_w && nnp > 0) {
            // ---- higher-order LSD, sweep (oracle oq_lsd, step 2): per valid cluster, ascending id, OSD-CS / OSD-E on the cluster's
            // own factorisation.  Candidate positions: its non-pivot faults by (posterior LLR, index); flipping a set of them
            // flips the pivot coefficients on the rows of t = XOR of their images; cost change = sum of their weights + the
            // weights of the pivot columns switched on - those switched off, in integers (exact); strict '<', earliest wins.
            __syncthreads();
            const int kwmax = (npiv + 63) / 64 - 1;
            auto image = [&](int j) -> uint32_t {                              // t = T * column j on this lane's rows
                const int c0 = a.cp[j], deg = a.cp[j + 1] - c0;
                const int myrow = lane < deg ? a.ri[c0 + lane] : -1;
                const int mypk = lane < deg ? (int)rowpiv[myrow] : -1;
                const unsigned long long bpk = __ballot(mypk >= 0);
                uint32_t tb = 0;
                for (int x = 0; x < deg; ++x) {
                    const int r = __builtin_amdgcn_readlane(myrow, x) - base;
                    if ((unsigned)r < (unsigned)NR) tb |= 1u << r;
                }
                for (int w = 0; w <= kwmax; ++w) {
                    uint64_t mkw = 0ull;
                    for (unsigned long long bb = bpk; bb; bb &= bb - 1ull) {
                        const int pk = __builtin_amdgcn_readlane(mypk, (int)__builtin_ctzll(bb));
                        if ((pk >> 6) == w) mkw |= 1ull << (pk & 63);
                    }
                    if (mkw == 0ull) continue;
                    BlkU64 qb;
                    if (w < QL_LDS_PLANES) qb = *reinterpret_cast<const BlkU64 *>(ql + w * MP + base);
                    else qb = *reinterpret_cast<const BlkU64 *>(Q + (size_t)w * MP + base);
#pragma unroll
                    for (int k = 0; k < NR; ++k) tb ^= ((uint32_t)__popcll(qb.v[k] & mkw) & 1u) << k;
                }
                return tb;
            };
            // owner and sort key of every candidate position, once per shot (ownership is final here)
            for (int x = lane; x < nnp; x += 64) {
                const uint32_t jj = npl[x];
                nown[x] = owner[a.ri[a.cp[jj]]];                               // all checks of an added fault are in its cluster
                nkey[x] = ql_mono_key(llr[a.slot_of[jj]]);
            }
            __syncthreads();
            int sc = -1;
            for (;;) {
                const BlkU8 cs = *reinterpret_cast<const BlkU8 *>(cstate + base);
                uint32_t cm = 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < NR; ++k) if (cs.v[k] == 2 && base + k > sc) cm = min(cm, (uint32_t)(base + k));
                cm = qd_wave_umin(cm);
                if (cm == 0xFFFFFFFFu) break;
                sc = (int)cm;
                int kk = 0;
                for (int i0 = 0; i0 < nnp; i0 += 64) kk += __popcll(__ballot(i0 + lane < nnp && nown[i0 + lane] == (uint16_t)sc));
                if (kk == 0) continue;
                const BlkU16 own = own_blk;
                BlkU8 fl = fl_blk;
                uint32_t memb = 0, pivb = 0, spb = 0;
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const uint32_t f = fl.v[k];
                    if (own.v[k] == (uint16_t)sc) memb |= 1u << k;
                    pivb |= ((f >> 2) & 1u) << k; spb |= (f & 1u) << k;
                }
                const uint32_t mp = memb & pivb;
                uint32_t wv[NR];                                               // cost of the pivot column of each of the lane's pivot rows
#pragma unroll
                for (int k = 0; k < NR; ++k) wv[k] = ((mp >> k) & 1u) ? a.wfix[pcol[base + k]] : 0u;
                auto delta_of = [&](uint32_t tb) -> long long {
                    long long d = 0;
                    const uint32_t on = tb & mp & ~spb, off = tb & mp & spb;
#pragma unroll
                    for (int k = 0; k < NR; ++k) d += ((on >> k) & 1u) ? (long long)wv[k] : (((off >> k) & 1u) ? -(long long)wv[k] : 0ll);
                    return ql_wave_sum_i64(d);
                };
                const int wmax = a.lsd_w == 1 ? QL_TV_MAX : 15;
                const int w = min(min(a.order, kk), wmax);
                // best so far: (cost change, class, key) -- class -1 = the LSD-0 solution (cost change 0), 0 = one position, 1 = a pair /
                // pattern.  The oracle walks OSD-0, the single positions in sorted order, then the pairs, keeping the first strict
                // minimum: among single positions that is the lexicographic minimum of (cost change, sort key), so they can be
                // taken in list order; only the first w positions -- the ones pairs and patterns are built from -- have to be found in order.
                long long best = 0;
                int bcls = -1;
                unsigned long long bkey = 0ull;
                uint32_t btb = 0;
                int bja = -1, bjb = -1;
                unsigned bpat = 0;
                const int nsorted = (a.lsd_w == 2 || w >= 2) ? w : 0;
                unsigned long long lastkey = 0ull;
                for (int i = 0; i < nsorted; ++i) {
                    unsigned long long kmin = QL_NOKEY64;
                    for (int i0 = 0; i0 < nnp; i0 += 64) {
                        const int x = i0 + lane;
                        if (x < nnp && nown[x] == (uint16_t)sc) {
                            const unsigned long long key = ((unsigned long long)nkey[x] << 16) | npl[x];
                            if ((i == 0 || key > lastkey) && key < kmin) kmin = key;
                        }
                    }
                    const uint32_t hi = (uint32_t)(kmin >> 16), mh = qd_wave_umin(hi);
                    const uint32_t ml = qd_wave_umin((hi == mh && kmin != QL_NOKEY64) ? (uint32_t)(kmin & 0xFFFFu) : 0xFFFFu);
                    lastkey = ((unsigned long long)mh << 16) | ml;
                    const uint32_t tb = image((int)ml) & memb;
                    tv[i * 64 + lane] = tb;
                    if (lane == 0) tv[QL_TV_MAX * 64 + i] = ml;
                }
                __syncthreads();
                if (a.lsd_w == 1) {
                    for (int i0 = 0; i0 < nnp; i0 += 64) {
                        const int x = i0 + lane;
                        const bool mem = x < nnp && nown[x] == (uint16_t)sc;
                        const uint32_t myj = mem ? (uint32_t)npl[x] : 0u, myk = mem ? nkey[x] : 0u;
                        for (unsigned long long bm = __ballot(mem); bm; bm &= bm - 1ull) {
                            const int src = (int)__builtin_ctzll(bm);
                            const int j = __builtin_amdgcn_readlane((int)myj, src);
                            const unsigned long long key = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)myk, src) << 16) | (uint32_t)j;
                            const uint32_t tb = image(j) & memb;
                            const long long d = delta_of(tb) + (long long)a.wfix[j];
                            if (d < best || (d == best && bcls == 0 && key < bkey)) { best = d; bcls = 0; bkey = key; btb = tb; bja = j; bjb = -1; }
                        }
                    }
                    for (int x = 0; x < nsorted; ++x)
                        for (int y = x + 1; y < nsorted; ++y) {
                            const uint32_t tb = tv[x * 64 + lane] ^ tv[y * 64 + lane];
                            const int ja = (int)tv[QL_TV_MAX * 64 + x], jb = (int)tv[QL_TV_MAX * 64 + y];
                            const long long d = delta_of(tb) + (long long)a.wfix[ja] + (long long)a.wfix[jb];
                            if (d < best) { best = d; bcls = 1; btb = tb; bja = ja; bjb = jb; }
                        }
                } else {
                    for (unsigned pat = 1; pat < (1u << w); ++pat) {
                        uint32_t tb = 0;
                        long long ws = 0;
                        for (int b = 0; b < w; ++b)
                            if ((pat >> b) & 1u) { tb ^= tv[b * 64 + lane]; ws += (long long)a.wfix[tv[QL_TV_MAX * 64 + b]]; }
                        const long long d = delta_of(tb) + ws;
                        if (d < best) { best = d; bcls = 1; btb = tb; bpat = pat; }
                    }
                }
                if (best < 0) {
                    spb ^= btb & mp;
#pragma unroll
                    for (int k = 0; k < NR; ++k) fl.v[k] = (uint8_t)((fl.v[k] & ~QL_F_SP) | ((spb >> k) & 1u));
                    fl_blk = fl;
                    if (lane == 0) {
                        uint32_t *eb = a.err_bits + shot * a.out_words;
                        if (a.lsd_w == 1) {
                            atomicOr(&eb[bja >> 5], 1u << (bja & 31));
                            if (bjb >= 0) atomicOr(&eb[bjb >> 5], 1u << (bjb & 31));
                        } else
                            for (int b = 0; b < w; ++b)
                                if ((bpat >> b) & 1u) { const uint32_t jj = tv[QL_TV_MAX * 64 + b]; atomicOr(&eb[jj >> 5], 1u << (jj & 31u)); }
                    }
                }
                __syncthreads();
            }
        }
        {                                                                      // err[pivot column] = transformed syndrome at the pivot row
            const BlkU8 fl = fl_blk;
#pragma unroll
            for (int k = 0; k < NR; ++k)
                if ((fl.v[k] & (QL_F_SP | QL_F_PIV)) == (QL_F_SP | QL_F_PIV)) { const uint32_t j = pcol[base + k]; atomicOr(&a.err_bits[shot * a.out_words + (j >> 5)], 1u << (j & 31u)); }
        }
        __syncthreads();
        if (lane == 0)
            a.status[shot] = (a.status[shot] & 0xFFFF) | QD_STATUS_OSD | (inconsistent ? QD_STATUS_INCONSISTENT : 0) | (min(npiv, 4095) << 20);
        __syncthreads();
        QL_T(10);
    }
#ifdef QD_LSD_TIMING
    QL_T(11);
    if (lane == 0)
        for (int k = 0; k < 16; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(a.next_slot) + 1 + k, (unsigned long long)tacc[k]);
#endif
}

#ifdef QD_LSD_TIMING
__global__ void qd_lsd_timing_print(unsigned long long *t)
{
    static const char *nm[16] = {"init", "seed scan", "round list", "select", "key scan", "join", "rescan", "elim t", "rescan end", "pivot+valid",
                                 "output", "idle tail", "shots", "rounds", "steps", "-"};
    unsigned long long tot = 0;
    for (int k = 0; k < 12; ++k) tot += t[k];
    for (int k = 0; k < 16; ++k) printf("lsd %d %s %llu permille %llu\n", k, nm[k], t[k], k < 12 ? t[k] * 1000ull / (tot ? tot : 1ull) : 0ull);
    for (int k = 0; k < 16; ++k) t[k] = 0;
}
#endif

// rows per lane for a window of m checks (0: too many), and the LDS footprint of one shot
static int lsd_nr(int m) { return m <= 512 ? 8 : m <= 1024 ? 16 : m <= 1536 ? 24 : m <= 2048 ? 32 : 0; }
static int lsd_lds(int nr, int n, int out_words)
{
    const int fixed = nr == 8 ? QlLay<8>::o_added : nr == 16 ? QlLay<16>::o_added : nr == 24 ? QlLay<24>::o_added : QlLay<32>::o_added;
    (void)out_words;
    return fixed + (((n + 31) / 32 + 3) & ~3) * 4;
}

int qd_lsd_lds_bytes(int m, int n, int out_words)
{
    if (n > 65535 || lsd_nr(m) == 0) return 1 << 30;                           // candidate keys carry the fault in 16 bits
    return lsd_lds(lsd_nr(m), n, out_words);
}

// rows of one Q plane in the HBM workspace
int qd_lsd_plane_rows(int m) { return 64 * lsd_nr(m); }

// bytes of the per-slot workspace behind the Q planes: work counter (+ debug timers), pivot columns, and for the higher orders
// the non-pivot list and the image scratch
size_t qd_lsd_ws_bytes(int m, int n, int blocks, int lsd_w)
{
    const size_t rows = (size_t)qd_lsd_plane_rows(m);
    size_t b = sizeof(uint64_t) * ((size_t)blocks * ((m + 63) / 64) * rows + 32 + ((size_t)blocks * rows + 3) / 4);
    if (lsd_w) b += ((size_t)blocks * ((n + 3) & ~3) * sizeof(uint16_t) + 15) / 16 * 16 * 2 + (size_t)blocks * ((n + 3) & ~3) * sizeof(uint32_t)
                    + (size_t)blocks * QL_TV_WORDS * sizeof(uint32_t) + 64;
    return b;
}

hipError_t qd_launch_lsd0(const GenGraphDev &gg, const BpGraphDev &bg, const DecodeArgs &d, uint64_t *q_ws, int blocks_alloc, int blocks,
                          int lsd_w, int lsd_order, const uint32_t *wfix, hipStream_t s)
{
    LsdArgs a{};
    const int nr = lsd_nr(gg.m);
    if (nr == 0 || gg.n > 65535) return hipErrorInvalidValue;
    a.m = gg.m; a.n = gg.n; a.n_pad = bg.n_pad; a.mw = (gg.m + 63) / 64; a.out_words = bg.out_words;
    a.upd_rows = d.upd_rows;
    a.cp = gg.cp; a.ri = gg.ri;
    a.ell = reinterpret_cast<const int2 *>(gg.ell); a.ell_w = gg.ell_w;
    a.det = d.det; a.upd = d.upd; a.det_stride = d.det_stride; a.det_offset = d.det_offset; a.upd_stride = d.upd_stride;
    a.llr_ws = d.llr_ws; a.fail_list = d.fail_list; a.fail_count = d.fail_count; a.q_ws = q_ws;
    a.err_bits = d.err_bits; a.status = d.status;
    const int lds = lsd_lds(nr, gg.n, bg.out_words);
    a.next_slot = reinterpret_cast<int32_t *>(q_ws + (size_t)blocks_alloc * a.mw * 64 * nr);
    a.pcol_ws = reinterpret_cast<uint16_t *>(q_ws + (size_t)blocks_alloc * a.mw * 64 * nr + 32);
    a.lsd_w = lsd_order > 0 ? lsd_w : 0; a.order = lsd_order; a.wfix = wfix; a.slot_of = bg.bit_slot_of;
    a.npl_cap = (gg.n + 3) & ~3;
    {
        unsigned char *tail = reinterpret_cast<unsigned char *>(q_ws + (size_t)blocks_alloc * a.mw * 64 * nr + 32 + ((size_t)blocks_alloc * 64 * nr + 3) / 4);
        const size_t l16 = ((size_t)blocks_alloc * a.npl_cap * sizeof(uint16_t) + 15) / 16 * 16;
        a.npl_ws = reinterpret_cast<uint16_t *>(tail);
        a.nown_ws = reinterpret_cast<uint16_t *>(tail + l16);
        a.nkey_ws = reinterpret_cast<uint32_t *>(tail + 2 * l16);
        a.tv_ws = reinterpret_cast<uint32_t *>(tail + 2 * l16 + (size_t)blocks_alloc * a.npl_cap * sizeof(uint32_t));
    }
#ifdef QD_LSD_TIMING
    hipError_t e = hipMemsetAsync(a.next_slot, 0, sizeof(uint64_t) * 17, s);
#else
    hipError_t e = hipMemsetAsync(a.next_slot, 0, sizeof(uint64_t), s);
#endif
    if (e != hipSuccess) return e;
    const void *fn = nr == 8 ? (const void *)qd_lsd0_kernel<8> : nr == 16 ? (const void *)qd_lsd0_kernel<16>
                   : nr == 24 ? (const void *)qd_lsd0_kernel<24> : (const void *)qd_lsd0_kernel<32>;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    switch (nr) {
    case 8: hipLaunchKernelGGL(qd_lsd0_kernel<8>, dim3((unsigned)blocks), dim3(64), lds, s, a); break;
    case 16: hipLaunchKernelGGL(qd_lsd0_kernel<16>, dim3((unsigned)blocks), dim3(64), lds, s, a); break;
    case 24: hipLaunchKernelGGL(qd_lsd0_kernel<24>, dim3((unsigned)blocks), dim3(64), lds, s, a); break;
    default: hipLaunchKernelGGL(qd_lsd0_kernel<32>, dim3((unsigned)blocks), dim3(64), lds, s, a); break;
    }
#ifdef QD_LSD_TIMING
    hipLaunchKernelGGL(qd_lsd_timing_print, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned long long *>(a.next_slot) + 1);
#endif
    return hipGetLastError();
}
