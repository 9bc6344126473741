// osd_kernels.hip -- OSD-0 post-processing for the shots whose BP did not converge: one workgroup per shot.
//
// Replaces ldpc.BpOsdDecoder.decode -> OsdDecoder::decode (osd.hpp) with osd_method = OSD_0 / osd_order = 0, as the
// reference reaches it through quits/decoder/sliding_window.py:171,182.  CPU restatement: oracle/qd_oracle.c
// (oq_osd_column_order + elim_run + oq_osd0); identical results bit for bit (tests/test_gpu_parity.py).
//
//   1. column order: ascending posterior LLR, ties by ascending fault index  (ldpc soft_decision_col_sort; std::sort
//      leaves ties open, this build fixes them).  Bitonic sort of 64-bit (monotone-key << 32 | index) words in LDS.
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
#include "qd_internal.h"

#define QD_NOKEY 0xFFFFFFFFu

__device__ __forceinline__ uint64_t &qd_qword(uint64_t *q_lds, uint64_t *q_glb, int kw_lds, int m_pad, int w, int r)
{
    return (w < kw_lds) ? q_lds[(size_t)w * m_pad + r] : q_glb[(size_t)(w - kw_lds) * m_pad + r];
}

template <int T>
__global__ void __launch_bounds__(T) qd_osd0_kernel(OsdGraphDev g, BpGraphDev bg, DecodeArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int nfail = *a.fail_count;
    constexpr int NW = T / 64;
  // persistent: a fixed grid walks the list of non-converged shots; order/spill workspace is per workgroup
  for (int slot = blockIdx.x; slot < nfail; slot += gridDim.x) {
    const int64_t shot = a.fail_list[slot];

    uint64_t *sortbuf = reinterpret_cast<uint64_t *>(smem + g.off_q);      // phase 1; phase 2 reuses it as Q planes
    uint64_t *qlds = sortbuf;
    uint64_t *tb = reinterpret_cast<uint64_t *>(smem + g.off_tb);          // [m_pad] image of the 64 batch columns
    uint8_t *sp = smem + g.off_sp;                                         // [m_pad] transformed syndrome
    int16_t *rowpiv = reinterpret_cast<int16_t *>(smem + g.off_rowpiv);    // [m_pad] row -> pivot order or -1
    uint16_t *prow = reinterpret_cast<uint16_t *>(smem + g.off_prow);      // [m_pad] pivot order -> row
    uint32_t *pcol = reinterpret_cast<uint32_t *>(smem + g.off_pcol);      // [m_pad] pivot order -> fault
    uint32_t *pairs = reinterpret_cast<uint32_t *>(smem + g.off_pairs);    // [64 * max_cdeg] column-in-batch | pivot order << 8
    uint32_t *bcols = reinterpret_cast<uint32_t *>(smem + g.off_cols);     // [64]
    volatile uint32_t *red = reinterpret_cast<volatile uint32_t *>(smem + g.off_red); // [0..15] keys A, [16..31] keys B, [32..63] flags, [64] npairs
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + g.off_out);
    uint16_t *order = a.order_ws + (int64_t)blockIdx.x * g.n;
    uint64_t *qglb = a.q_spill ? a.q_spill + (int64_t)blockIdx.x * (int64_t)(g.mw - g.kw_lds) * g.m_pad : nullptr;

    // ---------------- 1. column order
    const float *llr = a.llr_ws + (int64_t)slot * bg.n_pad;
    for (int i = tid; i < g.npow2; i += T) sortbuf[i] = ~0ull;
    __syncthreads();
    for (int b = tid; b < g.n; b += T) {
        const float f = llr[b] + 0.0f;                    // -0 -> +0, so that +-0 tie on the index like the oracle's '<'
        uint32_t u = __float_as_uint(f);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone float -> unsigned
        const uint32_t j = bg.bit_orig[b];
        sortbuf[j] = ((uint64_t)u << 32) | j;
    }
    __syncthreads();
    for (int k = 2; k <= g.npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int pi = tid; pi < (g.npow2 >> 1); pi += T) {
                const int i = ((pi & ~(j - 1)) << 1) | (pi & (j - 1));
                const int l = i | j;
                const uint64_t x = sortbuf[i], y = sortbuf[l];
                const bool up = ((i & k) == 0);
                if ((x > y) == up) { sortbuf[i] = y; sortbuf[l] = x; }
            }
            __syncthreads();
        }
    for (int i = tid; i < g.n; i += T) order[i] = (uint16_t)(sortbuf[i] & 0xFFFFu);
    __syncthreads();   // order[] is re-read by this workgroup only (global, same CU -> L1/L2 coherent for own stores after the barrier's vmcnt drain)

    // ---------------- 2. elimination state
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    for (int r = tid; r < g.m_pad; r += T) {
        uint8_t s = 0;
        if (r < g.m) {
            s = det[r] & 1u;
            if (upd && r < a.upd_rows) s ^= upd[r] & 1u;
        }
        sp[r] = s;
        rowpiv[r] = -1;
    }
    for (int i = tid; i < g.kw_lds * g.m_pad; i += T) qlds[i] = 0ull;
    if (qglb)
        for (int i = tid; i < (g.mw - g.kw_lds) * g.m_pad; i += T) qglb[i] = 0ull;
    for (int w = tid; w < bg.out_words; w += T) outw[w] = 0u;
    __syncthreads();

    int npiv = 0, done = 0, inconsistent = 0;
    for (int base = 0; base < g.n && !done; base += 64) {
        // ---- transform the next 64 columns: tb[r] bit c = (T * column_c)[r]
        for (int r = tid; r < g.m_pad; r += T) tb[r] = 0ull;
        if (tid == 0) red[64] = 0u;
        if (tid < 64) bcols[tid] = (base + tid < g.n) ? (uint32_t)order[base + tid] : 0xFFFFFFFFu;
        __syncthreads();
        for (int x = tid; x < 64 * g.max_cdeg; x += T) {
            const int c = x / g.max_cdeg, q = x - c * g.max_cdeg;
            const uint32_t col = bcols[c];
            if (col != 0xFFFFFFFFu) {
                const uint32_t e0 = g.csc_ptr[col], e1 = g.csc_ptr[col + 1];
                if (e0 + q < e1) {
                    const int r = g.csc_row[e0 + q];
                    atomicXor(reinterpret_cast<unsigned long long *>(&tb[r]), 1ull << c);
                    const int k = rowpiv[r];
                    if (k >= 0) pairs[atomicAdd(const_cast<uint32_t *>(&red[64]), 1u)] = (uint32_t)c | ((uint32_t)k << 8);
                }
            }
        }
        __syncthreads();
        const int np = (int)red[64];
        for (int r = tid; r < g.m; r += T) {
            uint64_t x = tb[r];
            for (int i = 0; i < np; ++i) {
                const uint32_t pr = pairs[i];
                const int k = (int)(pr >> 8);
                const uint64_t qw = qd_qword(qlds, qglb, g.kw_lds, g.m_pad, k >> 6, r);
                x ^= ((qw >> (k & 63)) & 1ull) << (pr & 63u);
            }
            tb[r] = x;
        }
        // ---- take pivots out of the batch, in column order
        int phase = 0;
        for (;;) {
            uint32_t key = QD_NOKEY;
            int resid = 0;
            for (int r = tid; r < g.m; r += T)
                if (rowpiv[r] < 0) {
                    const uint64_t x = tb[r];
                    if (x) key = min(key, ((uint32_t)__builtin_ctzll(x) << 16) | (uint32_t)r);
                    resid |= sp[r];
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o));
            const unsigned long long bal = __ballot(resid);
            if ((tid & 63) == 0) { red[phase * 16 + (tid >> 6)] = key; red[32 + phase * 16 + (tid >> 6)] = (bal != 0ull); }
            __syncthreads();
            key = QD_NOKEY;
            int anyres = 0;
            for (int w = 0; w < NW; ++w) { key = min(key, (uint32_t)red[phase * 16 + w]); anyres |= (int)red[32 + phase * 16 + w]; }
            phase ^= 1;
            if (!anyres) { done = 1; break; }            // syndrome already in the span of the pivots found
            if (key == QD_NOKEY) break;                   // rest of the batch depends on earlier pivots
            const int c = (int)(key >> 16), p = (int)(key & 0xFFFFu);
            const int K = npiv, kw = K >> 6;
            const uint64_t kb = 1ull << (K & 63);
            const uint64_t tp = tb[p];
            const uint8_t spp = sp[p];
            for (int r = tid; r < g.m; r += T)
                if (r != p && ((tb[r] >> c) & 1ull)) {
                    tb[r] ^= tp;
                    sp[r] ^= spp;
                    for (int w = 0; w <= kw; ++w) {
                        uint64_t &dst = qd_qword(qlds, qglb, g.kw_lds, g.m_pad, w, r);
                        uint64_t v = dst ^ qd_qword(qlds, qglb, g.kw_lds, g.m_pad, w, p);
                        if (w == kw) v ^= kb;
                        dst = v;
                    }
                }
            // nobody reads rowpiv/prow/pcol during the update, so the new pivot can be recorded alongside it
            if (tid == 0) { rowpiv[p] = (int16_t)K; prow[K] = (uint16_t)p; pcol[K] = bcols[c]; }
            npiv = K + 1;
            __syncthreads();                              // updated rows + the pivot record, before the next round
        }
    }
    // residual left on a non-pivot row <=> syndrome outside the column space
    {
        int resid = 0;
        for (int r = tid; r < g.m; r += T)
            if (rowpiv[r] < 0) resid |= sp[r];
        const unsigned long long bal = __ballot(resid);
        __syncthreads();
        if ((tid & 63) == 0) red[32 + (tid >> 6)] = (bal != 0ull);
        __syncthreads();
        for (int w = 0; w < NW; ++w) inconsistent |= (int)red[32 + w];
    }
    // ---------------- 3. OSD-0 solution
    for (int k = tid; k < npiv; k += T)
        if (sp[prow[k]]) {
            const uint32_t j = pcol[k];
            atomicOr(&outw[j >> 5], 1u << (j & 31u));
        }
    __syncthreads();
    for (int w = tid; w < bg.out_words; w += T) a.err_bits[shot * bg.out_words + w] = outw[w];
    if (tid == 0) a.status[shot] = (a.status[shot] & 0xFFFF) | (1 << 17) | (inconsistent ? (1 << 18) : 0) | (min(npiv, 4095) << 20);
    __syncthreads();   // LDS is recycled by the next shot
  }
}

hipError_t qd_launch_osd0(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int64_t cap /* workgroups */, hipStream_t s)
{
    hipError_t e;
    switch (g.threads) {
    case 256: {
        auto k = qd_osd0_kernel<256>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)cap), dim3(256), g.lds_bytes, s, g, bg, a);
        break;
    }
    case 512: {
        auto k = qd_osd0_kernel<512>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)cap), dim3(512), g.lds_bytes, s, g, bg, a);
        break;
    }
    default: {
        auto k = qd_osd0_kernel<1024>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)cap), dim3(1024), g.lds_bytes, s, g, bg, a);
        break;
    }
    }
    return hipGetLastError();
}
