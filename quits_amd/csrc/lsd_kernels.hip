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
// for a workgroup, so a shot gets one wavefront (lane = check for the scans over the m rows, lane = row entry when a check's
// best candidate is recomputed) and a CU runs several shots side by side (about 40 KB of LDS per shot at the headline
// window).  Each check caches its best candidate (lowest (LLR, index) among its unused faults) in LDS, so a growth step is
// a wave-minimum over the cluster's checks, plus a rescan of the checks that joined or whose cached candidate was just used.
// A step is a chain of dependent memory round trips, so everything a step touches sits in LDS -- including the first two Q
// planes (pivot orders 0..127; later planes, rarely reached, live in a per-slot HBM workspace) -- and the only global loads
// left, the rescans, are batched: rows in ELL form {fault, posterior column}, QL_NB rows in flight at a time, two latencies
// (entries, then LLRs) per batch.
#include "qd_internal.h"
#include "../../include/quits_amd.h"

#define QL_NONE 0xFFFFu
#define QL_NOKEY64 0xFFFFFFFFFFFFFFFFull

struct LsdArgs {
    int m, n, m_pad, n_pad, mw, out_words, upd_rows, max_cdeg;
    const int32_t *rp, *ci;        // CSR of the window matrix (fault indices)
    const int32_t *cp, *ri;        // CSC
    const uint32_t *bit_slot_of;   // fault -> column of the posterior rows
    const uint8_t *det, *upd;
    int64_t det_stride, det_offset, upd_stride;
    const float *llr_ws;           // [fail slot][n_pad]
    const int32_t *fail_list, *fail_count;
    uint64_t *q_ws;                // [blocks][mw][m_pad]
    uint32_t *err_bits;
    int32_t *status;
    // LDS carve-up (bytes)
    const int2 *ell;               // [m][ell_w] rows in ELL form: {fault, its posterior column}, {-1, 0} padding
    int ell_w;                     // multiple of 64
    int32_t *next_slot;            // work counter, zero at launch
    int chunk_shift;               // member masks: chunk = 2^chunk_shift checks, at most 32 chunks
    int off_owner, off_best, off_q, off_added, off_sp, off_rowpiv, off_pcol, off_cstate, off_cnbits, off_cmask, off_rl, off_rs, off_t, off_out;
};

__device__ __forceinline__ uint32_t ql_mono_key(float llr)
{
    const float f = llr + 0.0f;                    // -0 -> +0: they tie on the index like the oracle's '<' on doubles
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// minimum of 48-bit keys (32-bit LLR key << 16 | 16-bit index) over the wavefront; uniform result
__device__ __forceinline__ uint64_t ql_wave_min48(uint64_t key)
{
    const uint32_t hi = (uint32_t)(key >> 16), mh = qd_wave_umin(hi);
    const uint32_t lo = (hi == mh) ? (uint32_t)(key & 0xFFFFu) : 0xFFFFFFFFu;
    const uint32_t ml = qd_wave_umin(lo);
    return (mh == 0xFFFFFFFFu && ml >= 0xFFFFu) ? QL_NOKEY64 : (((uint64_t)mh << 16) | (ml & 0xFFFFu));
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
#define QL_NB 8            // rows per batch of the candidate scan: their loads are in flight together

// rows of the chunks named by `mask` (uniform), lane-strided; chunk = 2^sh consecutive checks
#define QL_FOR_ROWS(mask, r)                                                                                      \
    for (uint32_t mm_ = (mask); mm_; mm_ &= mm_ - 1u)                                                             \
        for (int r = ((int)__builtin_ctz(mm_) << sh) + lane, re_ = min(((int)__builtin_ctz(mm_) + 1) << sh, m); r < re_; r += 64)

__global__ void __launch_bounds__(64) qd_lsd0_kernel(LsdArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *owner = reinterpret_cast<uint16_t *>(smem + a.off_owner);      // check -> cluster id (= seed check), QL_NONE = free
    uint64_t *best = reinterpret_cast<uint64_t *>(smem + a.off_best);        // check -> best unused fault as a 48-bit key
    uint64_t *ql = reinterpret_cast<uint64_t *>(smem + a.off_q);             // Q planes 0 and 1 (pivot orders 0..127)
    uint32_t *added = reinterpret_cast<uint32_t *>(smem + a.off_added);      // fault bitmap
    uint8_t *sp = smem + a.off_sp;
    int16_t *rowpiv = reinterpret_cast<int16_t *>(smem + a.off_rowpiv);      // check -> pivot order, -1 = not a pivot row
    uint16_t *pcol = reinterpret_cast<uint16_t *>(smem + a.off_pcol);        // pivot row -> its fault
    uint8_t *cstate = smem + a.off_cstate;                                   // per cluster id: 0 none, 1 invalid, 2 valid, 3 gone
    uint16_t *cnbits = reinterpret_cast<uint16_t *>(smem + a.off_cnbits);
    uint32_t *cmask = reinterpret_cast<uint32_t *>(smem + a.off_cmask);      // per cluster id: chunks of checks holding its members
    uint32_t *rl = reinterpret_cast<uint32_t *>(smem + a.off_rl);            // the round: size at its start << 16 | cluster id
    uint16_t *rs = reinterpret_cast<uint16_t *>(smem + a.off_rs);            // checks whose candidate has to be (re)computed
    uint8_t *tb = smem + a.off_t;                                            // image of the column being eliminated, per row
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + a.off_out);
    const int lane = threadIdx.x;
    const int m = a.m, m_pad = a.m_pad, sh = a.chunk_shift;
    const int nfail = *a.fail_count;
    uint64_t *Q = a.q_ws + (size_t)blockIdx.x * (size_t)a.mw * m_pad;         // planes >= 2 (rare): HBM, per resident slot
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
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

        // best unused fault of the checks rs[0..cnt): lane = entry of the row, QL_NB rows per trip so that a trip costs two
        // memory latencies (row entries, then their LLRs) whatever the number of rows
        auto scan_list = [&](int cnt) {
            for (int b0 = 0; b0 < cnt; b0 += QL_NB) {
                int row[QL_NB];
                uint64_t key[QL_NB];
#pragma unroll
                for (int b = 0; b < QL_NB; ++b) {
                    row[b] = b0 + b < cnt ? __builtin_amdgcn_readfirstlane((int)rs[b0 + b]) : -1;
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
                    const uint64_t k = ql_wave_min48(key[b]);
                    if (lane == 0) best[row[b]] = k;
                }
            }
            __syncthreads();
        };
        auto q_ld = [&](int w, int r) -> uint64_t { return w < 2 ? ql[w * m_pad + r] : Q[(size_t)w * m_pad + r]; };
        auto q_st = [&](int w, int r, uint64_t v) { if (w < 2) ql[w * m_pad + r] = v; else Q[(size_t)w * m_pad + r] = v; };

        int nseed = 0;
        for (int r0 = 0; r0 < m_pad; r0 += 64) {
            const int r = r0 + lane;
            uint32_t s = 0;
            if (r < m) {
                s = det[r] & 1u;
                if (upd && r < a.upd_rows) s ^= upd[r] & 1u;
            }
            sp[r] = (uint8_t)s; rowpiv[r] = -1; owner[r] = s ? (uint16_t)r : (uint16_t)QL_NONE;
            cstate[r] = s ? 1 : 0; cnbits[r] = 0; best[r] = QL_NOKEY64; tb[r] = 0; cmask[r] = 1u << (r >> sh);
            ql[r] = 0ull; ql[m_pad + r] = 0ull;                                // later planes are cleared when first used
            const unsigned long long bs = __ballot(s != 0u);
            if (s) {
                const int at = nseed + __popcll(bs & lt_mask);
                rs[at] = (uint16_t)r; rl[at] = (uint32_t)r;
            }
            nseed += __popcll(bs);
        }
        for (int w = lane; w < (a.n + 31) / 32; w += 64) added[w] = 0u;
        for (int w = lane; w < a.out_words; w += 64) outw[w] = 0u;
        __syncthreads();
        QL_T(0);
        scan_list(nseed);                                                      // seeds: the candidate of every unsatisfied check
        QL_T(1); QL_CNT(12, 1);

        int npiv = 0, inconsistent = 0, nrl = nseed;
        for (;;) {
            // ---- this round: the clusters still invalid, by (size now, id); they all were in the previous round's list
            int out = 0;
            for (int b0 = 0; b0 < nrl; b0 += 64) {
                const int idx = b0 + lane;
                const int c = idx < nrl ? (int)(rl[idx] & 0xFFFFu) : 0;
                const bool inv = idx < nrl && cstate[c] == 1;
                const uint32_t v = ((uint32_t)cnbits[c] << 16) | (uint32_t)c;
                const unsigned long long bi = __ballot(inv);
                if (inv) rl[out + __popcll(bi & lt_mask)] = v;                 // out <= b0: never ahead of the reads
                out += __popcll(bi);
            }
            nrl = out;
            __syncthreads();
            QL_T(2); QL_CNT(13, 1);
            if (nrl == 0) break;
            long long last = -1;
            for (;;) {
                uint32_t k = 0xFFFFFFFFu;
                for (int idx = lane; idx < nrl; idx += 64) { const uint32_t v = rl[idx]; if ((long long)v > last) k = min(k, v); }
                k = qd_wave_umin(k);
                if (k == 0xFFFFFFFFu) break;
                last = (long long)k;
                const int c = (int)(k & 0xFFFFu);
                QL_T(3);
                if (cstate[c] != 1) continue;                                  // became valid or was absorbed earlier in the round
                QL_CNT(14, 1);
                uint32_t cm = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmask[c]);
                // ---- the fault that joins: lowest (LLR, index) among the cached candidates of the cluster's checks
                uint64_t key = QL_NOKEY64;
                QL_FOR_ROWS(cm, r)
                    if (owner[r] == (uint16_t)c) { const uint64_t b = best[r]; key = b < key ? b : key; }
                key = ql_wave_min48(key);
                if (key == QL_NOKEY64) {                                       // nothing left to add: the syndrome is outside the column space
                    if (lane == 0) cstate[c] = 3;
                    inconsistent = 1;
                    __syncthreads();
                    continue;
                }
                const int j = (int)(key & 0xFFFFu);
                QL_T(4);
                if (lane == 0) { added[j >> 5] |= 1u << (j & 31); cnbits[c] = (uint16_t)(cnbits[c] + 1); }
                const int c0 = a.cp[j], c1 = a.cp[j + 1], deg = c1 - c0;
                // the column's rows, once, into the first lanes (one vector load instead of a dependent scalar load per row)
                const int myrow = lane < deg ? a.ri[c0 + lane] : -1;
                __syncthreads();
                // ---- its checks join; clusters owning one of them are absorbed; cached candidates that were this fault are redone
                int nrs = 0;
                for (int x = 0; x < deg; ++x) {
                    const int i = __builtin_amdgcn_readlane(myrow, x);
                    const int d = owner[i];
                    if (d == QL_NONE) {
                        if (lane == 0) owner[i] = (uint16_t)c;
                        cm |= 1u << (i >> sh);
                    } else if (d != c) {
                        const uint32_t dm = (uint32_t)__builtin_amdgcn_readfirstlane((int)cmask[d]);
                        QL_FOR_ROWS(dm, r) if (owner[r] == (uint16_t)d) owner[r] = (uint16_t)c;
                        if (lane == 0) { cnbits[c] = (uint16_t)(cnbits[c] + cnbits[d]); cstate[d] = 3; }
                        cm |= dm;
                    }
                    if (d == QL_NONE || (int)(best[i] & 0xFFFFu) == j) { if (lane == 0) rs[nrs] = (uint16_t)i; ++nrs; }
                    __syncthreads();
                }
                if (lane == 0) cmask[c] = cm;
                QL_T(5);
                scan_list(nrs);
                QL_T(6);
                // ---- the column through the elimination
                const int mypk = lane < deg ? (int)rowpiv[myrow] : -1;
                if (lane < deg) tb[myrow] = 1;
                uint64_t mk[2] = {0ull, 0ull};                                 // pivot orders of the column's pivoted rows, planes 0 and 1
                int hi_planes = 0;                                             // a pivot order beyond plane 1: take the general loop
                const unsigned long long bpk = __ballot(mypk >= 0);
                for (unsigned long long bb = bpk; bb; bb &= bb - 1ull) {
                    const int pk = __builtin_amdgcn_readlane(mypk, (int)__builtin_ctzll(bb));
                    if (pk < 128) mk[pk >> 6] |= 1ull << (pk & 63); else hi_planes = 1;
                }
                __syncthreads();
                uint32_t pkey = 0xFFFFFFFFu;
                QL_FOR_ROWS(cm, r) {
                    if (owner[r] != (uint16_t)c) continue;
                    uint32_t t = tb[r];
                    if (!hi_planes) {
                        if (mk[0]) t ^= (uint32_t)__popcll(ql[r] & mk[0]) & 1u;
                        if (mk[1]) t ^= (uint32_t)__popcll(ql[m_pad + r] & mk[1]) & 1u;
                    } else {
                        for (unsigned long long bb = bpk; bb; bb &= bb - 1ull) {            // (v_readlane reads any lane, active or not)
                            const int pk = __builtin_amdgcn_readlane(mypk, (int)__builtin_ctzll(bb));
                            t ^= (uint32_t)((q_ld(pk >> 6, r) >> (pk & 63)) & 1ull);
                        }
                    }
                    tb[r] = (uint8_t)t;
                    if (t && rowpiv[r] < 0) pkey = min(pkey, (uint32_t)r);
                }
                pkey = qd_wave_umin(pkey);
                __syncthreads();
                QL_T(7);
                if (pkey != 0xFFFFFFFFu) {
                    const int p = (int)pkey, K = npiv, kw = K >> 6;
                    const uint64_t kbit = 1ull << (K & 63);
                    if ((K & 63) == 0 && kw >= 2) {                            // a new HBM plane comes into use: clear it
                        for (int r = lane; r < m_pad; r += 64) Q[(size_t)kw * m_pad + r] = 0ull;
                        __syncthreads();
                    }
                    const uint32_t spp = sp[p];
                    QL_FOR_ROWS(cm, r) {
                        if (owner[r] != (uint16_t)c || !tb[r] || r == p) continue;
                        for (int w = 0; w <= kw; ++w) {
                            uint64_t v = q_ld(w, r) ^ q_ld(w, p);
                            if (w == kw) v ^= kbit;
                            q_st(w, r, v);
                        }
                        if (spp) sp[r] ^= 1;
                    }
                    if (lane == 0) { rowpiv[p] = (int16_t)K; pcol[p] = (uint16_t)j; }
                    npiv = K + 1;
                }
                __syncthreads();
                QL_T(8);
                // ---- valid once no unpivoted check of the cluster carries syndrome
                int bad = 0;
                QL_FOR_ROWS(cm, r) {
                    if (owner[r] == (uint16_t)c) { if (rowpiv[r] < 0 && sp[r]) bad = 1; tb[r] = 0; }
                }
                const bool anybad = __ballot(bad) != 0ull;
                if (lane == 0) cstate[c] = anybad ? 1 : 2;
                __syncthreads();
                QL_T(9);
            }
            QL_T(3);
            __syncthreads();
        }
        for (int r = lane; r < m; r += 64)                                     // err[pivot column] = transformed syndrome at the pivot row
            if (rowpiv[r] >= 0 && sp[r]) { const uint32_t j = pcol[r]; atomicOr(&outw[j >> 5], 1u << (j & 31u)); }
        __syncthreads();
        for (int w = lane; w < a.out_words; w += 64) a.err_bits[shot * a.out_words + w] = outw[w];
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
    static const char *nm[16] = {"init", "seed scan", "round list", "select", "key scan", "join", "rescan", "elim t", "pivot upd", "validity",
                                 "output", "idle tail", "shots", "rounds", "steps", "-"};
    unsigned long long tot = 0;
    for (int k = 0; k < 12; ++k) tot += t[k];
    for (int k = 0; k < 16; ++k) printf("lsd %d %s %llu permille %llu\n", k, nm[k], t[k], k < 12 ? t[k] * 1000ull / (tot ? tot : 1ull) : 0ull);
    for (int k = 0; k < 16; ++k) t[k] = 0;
}
#endif

// LDS footprint of one shot; fills the offsets of `a`
static int lsd_layout(LsdArgs &a)
{
    auto al = [](int x) { return (x + 15) & ~15; };
    int o = 0;
    a.off_best = o; o += al(a.m_pad * 8);
    a.off_q = o; o += al(a.m_pad * 16);
    a.off_cmask = o; o += al(a.m_pad * 4);
    a.off_rl = o; o += al(a.m_pad * 4);
    a.off_added = o; o += al(((a.n + 31) / 32) * 4);
    a.off_out = o; o += al(a.out_words * 4);
    a.off_owner = o; o += al(a.m_pad * 2);
    a.off_rowpiv = o; o += al(a.m_pad * 2);
    a.off_pcol = o; o += al(a.m_pad * 2);
    a.off_cnbits = o; o += al(a.m_pad * 2);
    a.off_rs = o; o += al(a.m_pad * 2);
    a.off_sp = o; o += al(a.m_pad);
    a.off_cstate = o; o += al(a.m_pad);
    a.off_t = o; o += al(a.m_pad);
    a.chunk_shift = 6;
    while ((a.m_pad >> a.chunk_shift) > 32) ++a.chunk_shift;
    return o;
}

int qd_lsd_lds_bytes(int m_pad, int n, int out_words)
{
    LsdArgs a{};
    a.m_pad = m_pad; a.n = n; a.out_words = out_words;
    if (n > 65535) return 1 << 30;                  // candidate keys carry the fault in 16 bits
    return lsd_layout(a);
}

hipError_t qd_launch_lsd0(const GenGraphDev &gg, const BpGraphDev &bg, const DecodeArgs &d, uint64_t *q_ws, int blocks_alloc, int blocks, hipStream_t s)
{
    LsdArgs a{};
    a.m = gg.m; a.n = gg.n; a.m_pad = bg.m_pad; a.n_pad = bg.n_pad; a.mw = (gg.m + 63) / 64; a.out_words = bg.out_words;
    a.upd_rows = d.upd_rows; a.max_cdeg = bg.max_cdeg;
    a.rp = gg.rp; a.ci = gg.ci; a.cp = gg.cp; a.ri = gg.ri; a.bit_slot_of = bg.bit_slot_of;
    a.ell = reinterpret_cast<const int2 *>(gg.ell); a.ell_w = gg.ell_w;
    a.det = d.det; a.upd = d.upd; a.det_stride = d.det_stride; a.det_offset = d.det_offset; a.upd_stride = d.upd_stride;
    a.llr_ws = d.llr_ws; a.fail_list = d.fail_list; a.fail_count = d.fail_count; a.q_ws = q_ws;
    a.err_bits = d.err_bits; a.status = d.status;
    const int lds = lsd_layout(a);
    a.next_slot = reinterpret_cast<int32_t *>(q_ws + (size_t)blocks_alloc * a.mw * a.m_pad);
#ifdef QD_LSD_TIMING
    hipError_t e = hipMemsetAsync(a.next_slot, 0, sizeof(uint64_t) * 17, s);
#else
    hipError_t e = hipMemsetAsync(a.next_slot, 0, sizeof(uint64_t), s);
#endif
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)qd_lsd0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(qd_lsd0_kernel, dim3((unsigned)blocks), dim3(64), lds, s, a);
#ifdef QD_LSD_TIMING
    hipLaunchKernelGGL(qd_lsd_timing_print, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned long long *>(a.next_slot) + 1);
#endif
    return hipGetLastError();
}
