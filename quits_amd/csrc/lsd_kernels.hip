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
// best candidate is recomputed) and a CU runs several shots side by side (about 28 KB of LDS per shot at the headline
// window).  Each check caches its best candidate (lowest (LLR, index) among its unused faults) in LDS, so a growth step is
// a wave-minimum over the cluster's checks, plus a rescan of the one or two checks whose cached candidate was just used.
// Q lives in a per-slot HBM workspace (only the rows of the cluster at hand are touched; L2-resident).
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
    int off_owner, off_best, off_added, off_sp, off_rowpiv, off_prow, off_pcol, off_cstate, off_cnbits, off_rkey, off_t, off_out;
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

__global__ void __launch_bounds__(64) qd_lsd0_kernel(LsdArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *owner = reinterpret_cast<uint16_t *>(smem + a.off_owner);      // check -> cluster id (= seed check), QL_NONE = free
    uint64_t *best = reinterpret_cast<uint64_t *>(smem + a.off_best);        // check -> best unused fault as a 48-bit key
    uint32_t *added = reinterpret_cast<uint32_t *>(smem + a.off_added);      // fault bitmap
    uint8_t *sp = smem + a.off_sp;
    int16_t *rowpiv = reinterpret_cast<int16_t *>(smem + a.off_rowpiv);
    uint16_t *prow = reinterpret_cast<uint16_t *>(smem + a.off_prow);
    uint32_t *pcol = reinterpret_cast<uint32_t *>(smem + a.off_pcol);
    uint8_t *cstate = smem + a.off_cstate;                                   // per cluster id: 0 none, 1 invalid, 2 valid, 3 gone
    uint16_t *cnbits = reinterpret_cast<uint16_t *>(smem + a.off_cnbits);
    uint32_t *rkey = reinterpret_cast<uint32_t *>(smem + a.off_rkey);        // order of the round: size << 16 | id, sizes as at its start
    uint8_t *tb = smem + a.off_t;                                            // image of the column being eliminated, per row
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + a.off_out);
    const int lane = threadIdx.x;
    const int m = a.m, m_pad = a.m_pad;
    const int nfail = *a.fail_count;
    uint64_t *Q = a.q_ws + (size_t)blockIdx.x * (size_t)a.mw * m_pad;

    for (int slot = blockIdx.x; slot < nfail; slot += gridDim.x) {
        const int64_t shot = a.fail_list[slot];
        const float *llr = a.llr_ws + (int64_t)slot * a.n_pad;
        const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
        const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;

        // best unused fault of check i: lane = entry of the row
        auto scan_row = [&](int i) -> uint64_t {
            uint64_t key = QL_NOKEY64;
            const int e0 = a.rp[i], e1 = a.rp[i + 1];
            for (int e = e0 + lane; e < e1; e += 64) {
                const uint32_t j = (uint32_t)a.ci[e];
                if (!((added[j >> 5] >> (j & 31u)) & 1u)) {
                    const uint64_t k = ((uint64_t)ql_mono_key(llr[a.bit_slot_of[j]]) << 16) | j;
                    key = k < key ? k : key;
                }
            }
            return ql_wave_min48(key);
        };

        for (int r = lane; r < m_pad; r += 64) {
            uint32_t s = 0;
            if (r < m) {
                s = det[r] & 1u;
                if (upd && r < a.upd_rows) s ^= upd[r] & 1u;
            }
            sp[r] = (uint8_t)s; rowpiv[r] = -1; owner[r] = s ? (uint16_t)r : (uint16_t)QL_NONE;
            cstate[r] = s ? 1 : 0; cnbits[r] = 0; best[r] = QL_NOKEY64; tb[r] = 0;
            Q[r] = 0ull;                                                       // plane 0; later planes are cleared when first used
        }
        for (int w = lane; w < (a.n + 31) / 32; w += 64) added[w] = 0u;
        for (int w = lane; w < a.out_words; w += 64) outw[w] = 0u;
        __syncthreads();
        for (int i = 0; i < m; ++i)                                            // seeds: one row scan each
            if (sp[i]) { const uint64_t k = scan_row(i); if (lane == 0) best[i] = k; }
        __syncthreads();

        int npiv = 0, inconsistent = 0;
        for (;;) {
            // ---- order of this round
            int any = 0;
            for (int c = lane; c < m_pad; c += 64) {
                const bool inv = c < m && cstate[c] == 1;
                rkey[c] = inv ? (((uint32_t)cnbits[c] << 16) | (uint32_t)c) : 0xFFFFFFFFu;
                any |= inv ? 1 : 0;
            }
            __syncthreads();
            if (__ballot(any) == 0ull) break;
            long long last = -1;
            for (;;) {
                uint32_t k = 0xFFFFFFFFu;
                for (int c = lane; c < m; c += 64) { const uint32_t v = rkey[c]; if ((long long)v > last) k = min(k, v); }
                k = qd_wave_umin(k);
                if (k == 0xFFFFFFFFu) break;
                last = (long long)k;
                const int c = (int)(k & 0xFFFFu);
                if (cstate[c] != 1) continue;                                  // became valid or was absorbed earlier in the round
                // ---- the fault that joins: lowest (LLR, index) among the cached candidates of the cluster's checks
                uint64_t key = QL_NOKEY64;
                for (int r = lane; r < m; r += 64)
                    if (owner[r] == (uint16_t)c) { const uint64_t b = best[r]; key = b < key ? b : key; }
                key = ql_wave_min48(key);
                if (key == QL_NOKEY64) {                                       // nothing left to add: the syndrome is outside the column space
                    if (lane == 0) cstate[c] = 3;
                    inconsistent = 1;
                    __syncthreads();
                    continue;
                }
                const int j = (int)(key & 0xFFFFu);
                if (lane == 0) { added[j >> 5] |= 1u << (j & 31); cnbits[c] = (uint16_t)(cnbits[c] + 1); }
                __syncthreads();
                const int c0 = a.cp[j], c1 = a.cp[j + 1];
                // ---- its checks join; clusters owning one of them are absorbed; cached candidates that were this fault are redone
                for (int e = c0; e < c1; ++e) {
                    const int i = a.ri[e];
                    const int d = owner[i];
                    if (d == QL_NONE) {
                        if (lane == 0) owner[i] = (uint16_t)c;
                    } else if (d != c) {
                        for (int r = lane; r < m; r += 64) if (owner[r] == (uint16_t)d) owner[r] = (uint16_t)c;
                        if (lane == 0) { cnbits[c] = (uint16_t)(cnbits[c] + cnbits[d]); cstate[d] = 3; }
                    }
                    __syncthreads();
                    if (d == QL_NONE || (int)(best[i] & 0xFFFFu) == j) {
                        const uint64_t kb = scan_row(i);
                        if (lane == 0) best[i] = kb;
                        __syncthreads();
                    }
                }
                // ---- the column through the elimination
                int maskk[QD_MAX_COL_DEG], nmask = 0;
                for (int e = c0; e < c1; ++e) {
                    const int r = a.ri[e];
                    if (lane == 0) tb[r] = 1;
                    const int pk = rowpiv[r];
                    if (pk >= 0 && nmask < QD_MAX_COL_DEG) maskk[nmask++] = pk;
                }
                __syncthreads();
                uint32_t pkey = 0xFFFFFFFFu;
                for (int r = lane; r < m; r += 64) {
                    if (owner[r] != (uint16_t)c) continue;
                    uint32_t t = tb[r];
                    for (int x = 0; x < nmask; ++x) t ^= (uint32_t)((Q[(size_t)(maskk[x] >> 6) * m_pad + r] >> (maskk[x] & 63)) & 1ull);
                    tb[r] = (uint8_t)t;
                    if (t && rowpiv[r] < 0) pkey = min(pkey, (uint32_t)r);
                }
                pkey = qd_wave_umin(pkey);
                __syncthreads();
                if (pkey != 0xFFFFFFFFu) {
                    const int p = (int)pkey, K = npiv, kw = K >> 6;
                    const uint64_t kbit = 1ull << (K & 63);
                    if ((K & 63) == 0 && K > 0) {                              // a new plane comes into use: clear it
                        for (int r = lane; r < m_pad; r += 64) Q[(size_t)kw * m_pad + r] = 0ull;
                        __syncthreads();
                    }
                    const uint32_t spp = sp[p];
                    for (int r = lane; r < m; r += 64) {
                        if (owner[r] != (uint16_t)c || !tb[r] || r == p) continue;
                        for (int w = 0; w <= kw; ++w) {
                            uint64_t v = Q[(size_t)w * m_pad + r] ^ Q[(size_t)w * m_pad + p];
                            if (w == kw) v ^= kbit;
                            Q[(size_t)w * m_pad + r] = v;
                        }
                        if (spp) sp[r] ^= 1;
                    }
                    if (lane == 0) { rowpiv[p] = (int16_t)K; prow[K] = (uint16_t)p; pcol[K] = (uint32_t)j; }
                    npiv = K + 1;
                }
                __syncthreads();
                // ---- valid once no unpivoted check of the cluster carries syndrome
                int bad = 0;
                for (int r = lane; r < m; r += 64) {
                    if (owner[r] == (uint16_t)c) { if (rowpiv[r] < 0 && sp[r]) bad = 1; tb[r] = 0; }
                }
                const bool anybad = __ballot(bad) != 0ull;
                if (lane == 0) cstate[c] = anybad ? 1 : 2;
                __syncthreads();
            }
            __syncthreads();
        }
        for (int k = lane; k < npiv; k += 64)
            if (sp[prow[k]]) { const uint32_t j = pcol[k]; atomicOr(&outw[j >> 5], 1u << (j & 31u)); }
        __syncthreads();
        for (int w = lane; w < a.out_words; w += 64) a.err_bits[shot * a.out_words + w] = outw[w];
        if (lane == 0)
            a.status[shot] = (a.status[shot] & 0xFFFF) | QD_STATUS_OSD | (inconsistent ? QD_STATUS_INCONSISTENT : 0) | (min(npiv, 4095) << 20);
        __syncthreads();
    }
}

// LDS footprint of one shot; fills the offsets of `a`
static int lsd_layout(LsdArgs &a)
{
    auto al = [](int x) { return (x + 15) & ~15; };
    int o = 0;
    a.off_best = o; o += al(a.m_pad * 8);
    a.off_pcol = o; o += al(a.m_pad * 4);
    a.off_rkey = o; o += al(a.m_pad * 4);
    a.off_added = o; o += al(((a.n + 31) / 32) * 4);
    a.off_out = o; o += al(a.out_words * 4);
    a.off_owner = o; o += al(a.m_pad * 2);
    a.off_rowpiv = o; o += al(a.m_pad * 2);
    a.off_prow = o; o += al(a.m_pad * 2);
    a.off_cnbits = o; o += al(a.m_pad * 2);
    a.off_sp = o; o += al(a.m_pad);
    a.off_cstate = o; o += al(a.m_pad);
    a.off_t = o; o += al(a.m_pad);
    return o;
}

int qd_lsd_lds_bytes(int m_pad, int n, int out_words)
{
    LsdArgs a{};
    a.m_pad = m_pad; a.n = n; a.out_words = out_words;
    return lsd_layout(a);
}

hipError_t qd_launch_lsd0(const GenGraphDev &gg, const BpGraphDev &bg, const DecodeArgs &d, uint64_t *q_ws, int blocks, hipStream_t s)
{
    LsdArgs a{};
    a.m = gg.m; a.n = gg.n; a.m_pad = bg.m_pad; a.n_pad = bg.n_pad; a.mw = (gg.m + 63) / 64; a.out_words = bg.out_words;
    a.upd_rows = d.upd_rows; a.max_cdeg = bg.max_cdeg;
    a.rp = gg.rp; a.ci = gg.ci; a.cp = gg.cp; a.ri = gg.ri; a.bit_slot_of = bg.bit_slot_of;
    a.det = d.det; a.upd = d.upd; a.det_stride = d.det_stride; a.det_offset = d.det_offset; a.upd_stride = d.upd_stride;
    a.llr_ws = d.llr_ws; a.fail_list = d.fail_list; a.fail_count = d.fail_count; a.q_ws = q_ws;
    a.err_bits = d.err_bits; a.status = d.status;
    const int lds = lsd_layout(a);
    hipError_t e = hipFuncSetAttribute((const void *)qd_lsd0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(qd_lsd0_kernel, dim3((unsigned)blocks), dim3(64), lds, s, a);
    return hipGetLastError();
}
