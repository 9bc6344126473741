// bp_kernels.hip -- flooding min-sum belief propagation, one workgroup per shot, all message state in LDS.
//
// Replaces ldpc.BpOsdDecoder.decode -> BpDecoder::bp_decode_parallel (MINIMUM_SUM) as the reference calls it at
// quits/decoder/sliding_window.py:171,182.  CPU restatement with the same arithmetic: oracle/bp_core.inc,
// bp_minsum_compressed (REAL=float) -- the two agree bit for bit (tests/test_gpu_parity.py).
//
// Design (gfx950):
//   * A check keeps only (min1, min2, argmin, parity, per-edge sign bits) of its incoming messages: 16 bytes instead
//     of one float per edge; a fault keeps its posterior LLR and one sign bit per edge.  For the [[144,12,12]] R=12
//     window (1008 x 9504, 33192 edges) that is 68 KB of LDS per shot, so two shots are resident per CU and the
//     message traffic never leaves the CU.  HBM only sees the syndrome bytes in and the packed decision bits out.
//   * lane = node.  Check pass: lane owns a check, walks its faults (ELL-transposed adjacency -> coalesced index
//     loads, LDS gathers of the posteriors).  Bit pass: lane owns a fault, gathers the <= 16 checks' 16-byte states
//     with one ds_read_b128 each.  Nodes are degree-sorted so a wavefront's lanes share a trip count.
//   * The syndrome test of iteration t rides on check pass t+1 (it needs the same gathers), so an iteration costs
//     one check pass + one bit pass + two barriers.
//   * No atomics on floats, fixed summation order (ascending detector index) -> bit-reproducible.
#include "qd_internal.h"
#include "../../include/quits_amd.h"
#include <float.h>

// Four LDS byte offsets per vector load: uint2 = 4 x uint16 (windows up to 16379 fault slots), uint4 = 4 x uint32.
template <int I> __device__ __forceinline__ uint32_t qd_adj_get(const uint2 &v)
{
    const uint32_t w = (I < 2) ? v.x : v.y;
    return (I & 1) ? (w >> 16) : (w & 0xFFFFu);
}
template <int I> __device__ __forceinline__ uint32_t qd_adj_get(const uint4 &v)
{
    return I == 0 ? v.x : (I == 1 ? v.y : (I == 2 ? v.z : v.w));
}

#ifdef QD_BP_TIMING   // phase cycle counters of wavefront 0 (tools/bp_timing.py); slots: 0 check pass, 1 block-OR, 2 bit pass, 3 barrier, 4 prologue, 5 epilogue
#define QD_BP_TICK(slot) { const unsigned long long now_ = clock64(); acc_[slot] += now_ - tick_; tick_ = now_; }
#else
#define QD_BP_TICK(slot)
#endif
#ifndef QD_BP_MINWAVES
#define QD_BP_MINWAVES 8   // waves per SIMD the register allocator must leave room for (8 = two 1024-thread workgroups per CU)
#endif
// round 6: at 8 wavefronts per SIMD (64 registers) every instantiation spilled 12-16 bytes per lane; K1 is a fall-back now, so the budget is the
// one that keeps it out of scratch (tests/test_api.py asserts it)
#ifndef QD_BP_MINWAVES_T
#define QD_BP_MINWAVES_T(T) ((T) == 1024 ? 4 : 6)
#endif

// min(a, |b|) as the single instruction it is: fminf() makes the compiler quiet a possible signalling NaN first
// (v_max_f32 x, x, x -- one more 4-clock instruction per four edges); neither operand can be a NaN here.
__device__ __forceinline__ float qd_min_abs(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One edge of the check pass.
//   off  = LDS address of the fault's posterior (doubles as the edge's label for the argmin bookkeeping); the kernel's
//          dynamic LDS starts at address 0 (checked on entry), so the packed 16-bit offsets are used as addresses as they are
//   sb   = bit of `sgnw` that holds the sign of the previous check->bit message on this edge
// Branch-free: the second minimum is the median of (min1, min2, |b|); the new sign bits are shifted in from bit 0.
// (b <= 0) is taken as the sign bit of (bits(b) - 1): exact for every float except -0.0, which cannot occur here: a prior that
// rounds to zero is uploaded as +0 (qd_api.hip, on_grid), sums and differences of values that are not -0 only yield -0 from
// (-0) - (+0), and a message's sign is ORed onto a magnitude only where it is read, never stored as a float.
#define QD_CHECK_EDGE(off, sb) QD_CHECK_EDGE_L(off, sb, (*(const __attribute__((address_space(3))) float *)(uintptr_t)(uint32_t)(off)))
#define QD_CHECK_EDGE_L(off, sb, Lval)                                                                       \
    {                                                                                                        \
        const float L_ = (Lval);                                                                             \
        const float mag_ = ((off) == idx_old) ? st.y : st.x;                                                 \
        const float prev_ = __uint_as_float(((sgnw >> (sb)) & 1u) << 31 | __float_as_uint(mag_));            \
        const float bm_ = L_ - prev_;                  /* bit->check message, "total minus own" */            \
        const float ab_ = fabsf(bm_);                                                                        \
        neww = __builtin_amdgcn_alignbit(neww, __float_as_uint(bm_) - 1u, 31);   /* neww = neww << 1 | (bm_ <= 0) */ \
        idx = (ab_ < a1) ? (off) : idx;                                                                      \
        a2 = __builtin_amdgcn_fmed3f(a1, a2, ab_);                                                           \
        a1 = qd_min_abs(a1, bm_);                                                                            \
    }

// Bit pass, one edge.  rec = (LDS byte offset of the check state) << 16 | where its sign lives.
//   load:  gather the 16-byte check state (all of a fault's gathers are issued before any is used)
//   use :  magnitude = min2 if this fault is the check's argmin, else min1; sign = the check's outgoing sign bit.
//          {z, w} is laid out as one 64-bit value (signs 0..31 | argmin slot | signs 32..46 | syndrome), so the sign
//          is a single 64-bit shift by the bit index the record carries.
// The gather is issued as explicit ds_read_b128: left to the compiler, the {z, w} pair is peeled off into its own
// 64-bit load and the state arrives as ds_read2_b64, which measured 40% slower for the whole kernel.  QD_BIT_WAIT*
// is the matching s_waitcnt; it names the gathered registers so that nothing reads them early.
typedef uint32_t qd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ qd_u32x4 qd_lds_gather16(uint32_t lds_addr)
{
    qd_u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
#define QD_BIT_LOAD(rec) qd_lds_gather16((rec) >> 16)
#define QD_BIT_WAIT1(a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory")
#define QD_BIT_WAIT2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory")
#define QD_BIT_WAIT3(a, b, c) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) : : "memory")
#define QD_BIT_WAIT4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory")
#define QD_BIT_FLIP(rec) __hip_atomic_fetch_xor((__attribute__((address_space(3))) uint32_t *)(uintptr_t)(((rec) >> 16) + 12u), 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define QD_BIT_USE(rec, st_)                                                                                 \
    {                                                                                                        \
        const uint32_t meta_ = (st_).w;                                                     \
        const uint32_t mag_ = ((uint16_t)meta_ == mylabel) ? (st_).y : (st_).x;                                 \
        uint32_t sg_;                                                                                        \
        if (SM == 2) {                                                                                       \
            uint32_t sw_ = (st_).z;                                                         \
            if ((rec) & 0xE0u) sw_ = csgn_hi[((((rec) >> 5) & 7u) - 1u) * m_pad + ((rec) >> 20)];            \
            sg_ = sw_ >> ((rec) & 31u);                                                                      \
        } else if (SM == 1) {                                                                                \
            sg_ = (uint32_t)((((uint64_t)meta_ << 32) | (st_).z) >> ((rec) & 63u));         \
        } else {                                                                                             \
            sg_ = (st_).z >> ((rec) & 31u);                                                 \
        }                                                                                                    \
        acc += __uint_as_float(sg_ << 31 | mag_);                                           \
        sabs += __uint_as_float(mag_);                                                      \
    }

// State of a check in LDS (16 bytes, one ds_read_b128):  x = min1 * alpha,  y = min2 * alpha,
//   z = SIGN bits of the outgoing messages on edges 0..31; inside a word, edge k sits at bit (edges_in_word - 1 - k)
//       (sign of check->bit message k = syndrome ^ parity of all incoming signs ^ incoming sign k),
//   w = slot of the argmin fault (bits 0..15) | (sign mode 1: sign bits of edges 32..43) << 16 | parity of the hard decisions of
//       the check's faults << 30 (flipped by the bit pass) | syndrome bit << 31.
// Sign mode 2 (checks wider than 44): edges 32.. keep their sign words in `csgn_hi`.
template <int T, int NCH, int SM, typename ADJ4, int MW>
__global__ void __launch_bounds__(T, MW) qd_bp_minsum_kernel(BpGraphDev g, DecodeArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;   // LDS address of smem[0]
    if (lds_base != 0u) __builtin_trap();   // no static LDS in this kernel: byte offsets into smem are LDS addresses
    float4 *chk = reinterpret_cast<float4 *>(smem + g.off_chk);
    uint32_t *csgn_hi = reinterpret_cast<uint32_t *>(smem + g.off_cneg);
    float *llr = reinterpret_cast<float *>(smem + g.off_llr);
    uint32_t *outw = reinterpret_cast<uint32_t *>(smem + g.off_out);
    int *misc = reinterpret_cast<int *>(smem + g.off_misc);   // [0..31] OR flags, [32] fail slot
    constexpr int NW = T / 64;

    const int tid = threadIdx.x;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid & ~63);   // first thread of this wavefront
    int64_t shot = blockIdx.x;
    if (a.shot_list) {                                             // redo pass: one workgroup per parked shot
        if ((int)blockIdx.x >= *a.shot_count) return;
        shot = a.shot_list[blockIdx.x];
    }
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    const int m_pad = g.m_pad, n_pad = g.n_pad;
    const uint4 *rec4 = reinterpret_cast<const uint4 *>(g.bit_rec);
    const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.bit_rec, 0, g.rec_words * n_pad * 4, 0x00020000);
    const int rec_voff = tid * 16;
    const ADJ4 *adj4 = reinterpret_cast<const ADJ4 *>(g.chk_adj);
    const uint32_t llr_base = (uint32_t)g.off_llr;

#ifdef QD_BP_TIMING
    unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = clock64();
#endif
    // ---- load the window syndrome (sliding_window.py:168-169) and reset the state
    int any = 0;
    if (tid < 64) misc[tid] = 0;
    for (int c = tid; c < g.m; c += T) {
        const uint32_t o = g.chk_orig[c];
        uint32_t s = det[o] & 1u;
        if (upd && (int)o < a.upd_rows) s ^= upd[o] & 1u;
        any |= (int)s;
        chk[c] = make_float4(0.f, 0.f, __uint_as_float(0u), __uint_as_float(0xFFFFu | (s << 31)));   // no message yet
        if (SM == 2)
            for (int w = 1; w < g.neg_words; ++w) csgn_hi[(w - 1) * m_pad + c] = 0u;
    }
    for (int b = tid; b < g.n; b += T) llr[b] = __uint_as_float(rec4[b].x);
    for (int w = tid; w < g.out_words; w += T) outw[w] = 0u;
    if (tid == 0) {
        llr[g.dummy_bit] = __builtin_inff();                                            // padding edge of a short row: |b| = inf, never a minimum, never negative
        chk[g.dummy_chk] = make_float4(0.f, 0.f, __uint_as_float(0u), __uint_as_float(0xFFFFu));     // padding edge of a short column: message +0
    }
    __syncthreads();
    any = qd_block_or(any, misc, NW, 0);
    if (!any) {   // bposd_decoder.pyx: an all-zero syndrome returns the zero vector without running BP
        for (int w = tid; w < g.out_words; w += T) a.err_bits[shot * g.out_words + w] = 0u;
        if (tid == 0) a.status[shot] = (1 << 16) | (1 << 19) | a.status_or;
        return;
    }

    // Exactness bound of the grid arithmetic.  The channel LLRs are multiples of q = 2^-k and ms_scaling is 1, so every
    // message is a multiple of q; a float sum or difference of such values is exact while its magnitude stays below
    // 2^24 q.  S_j = |llr0_j| + sum_k |c2b_k| bounds every partial sum of fault j's posterior, and a bit->check message
    // |L_j - c2b| <= S_j as well; S_j itself is a sum of non-negative terms (monotone: exact whenever the total is).  So
    // max_j S_j < 2^23 q over the whole run (a.s_limit) certifies that no operation rounded: the result is then what
    // ldpc's double-precision arithmetic returns for the same LLRs.  One extra add per edge, one compare per fault.
    bool trip = false;             // a lane mask in scalar registers: costs no vector register
    const float s_lim = a.s_limit > 0.f ? a.s_limit : __builtin_inff();
    int t = 0, converged = 0, phase = 1;
    QD_BP_TICK(4)
    for (;;) {
        const float alpha = (a.ms_scale == 0.f) ? (1.0f - ldexpf(1.0f, -(t + 1))) : a.ms_scale;
        // ---- check pass t+1; its parity test is the convergence test of iteration t
        bool unsat = false;
        for (int c = tid; c < g.m; c += T) {
            const float4 st = chk[c];
            const uint32_t meta = __float_as_uint(st.w);
            const uint32_t idx_old = llr_base + ((meta & 0xFFFFu) << 2);   // label = LDS offset of the argmin fault's posterior
            const uint32_t synd = meta >> 31;
            const int dw = g.chk_degp_w[__builtin_amdgcn_readfirstlane(c) >> 6];       // scalar load
            const int degp = dw & 0xFFFF, wmax = dw >> 16;                             // trip count (multiple of 4), largest degree in this wavefront
            // syndrome of the hard decision: the bit pass flips bit 30 of this word once for every fault on this check that it
            // decided to be 1 (a few dozen faults per shot), so the parity test costs nothing per edge here
            const bool us = (((meta >> 31) ^ (meta >> 30)) & 1u) != 0u;
            uint32_t idx = llr_base + (0xFFFFu << 2);
            float a1 = FLT_MAX, a2 = FLT_MAX;
            uint32_t neg0 = 0u, neg1 = 0u, npar = 0u;
            for (int k0 = 0; k0 < degp; k0 += 32) {
                uint32_t sgnw = __float_as_uint(st.z);
                if (SM == 1 && k0 != 0) sgnw = (meta >> 16) & 0x0FFFu;
                if (SM == 2 && k0 != 0) sgnw = csgn_hi[((k0 >> 5) - 1) * m_pad + c];
                uint32_t neww = 0u;
                const int kend = min(degp - k0, 32);                // multiple of 4
                const ADJ4 *ap = adj4 + (size_t)(k0 >> 2) * m_pad + c;
                ADJ4 nx = ap[0];
                int kk = 0;
#pragma unroll 1
                for (; kk + 4 < kend; kk += 4) {
                    const ADJ4 cur = nx;
                    nx = ap[(size_t)((kk >> 2) + 1) * m_pad];                       // next four fault offsets while these are processed
                    const int sb = kend - 1 - kk;
                    if constexpr (SM == 2) {
                    // wide checks (> 44 faults: the QLP windows, one workgroup per CU, four wavefronts per SIMD): the four
                    // posteriors of the trip are gathered before any is used.  +3 % there; -3 % at the headline window, where
                    // eight wavefronts per SIMD hide the gather's latency and the early loads only add register pressure.
                    float g0_, g1_, g2_, g3_;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(g0_) : "v"((uint32_t)qd_adj_get<0>(cur)) : "memory");
                    asm volatile("ds_read_b32 %0, %1" : "=v"(g1_) : "v"((uint32_t)qd_adj_get<1>(cur)) : "memory");
                    asm volatile("ds_read_b32 %0, %1" : "=v"(g2_) : "v"((uint32_t)qd_adj_get<2>(cur)) : "memory");
                    asm volatile("ds_read_b32 %0, %1" : "=v"(g3_) : "v"((uint32_t)qd_adj_get<3>(cur)) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g0_), "+v"(g1_), "+v"(g2_), "+v"(g3_) : : "memory");
                    QD_CHECK_EDGE_L(qd_adj_get<0>(cur), sb, g0_)
                    QD_CHECK_EDGE_L(qd_adj_get<1>(cur), sb - 1, g1_)
                    QD_CHECK_EDGE_L(qd_adj_get<2>(cur), sb - 2, g2_)
                    QD_CHECK_EDGE_L(qd_adj_get<3>(cur), sb - 3, g3_)
                    } else {
                    QD_CHECK_EDGE(qd_adj_get<0>(cur), sb)
                    QD_CHECK_EDGE(qd_adj_get<1>(cur), sb - 1)
                    QD_CHECK_EDGE(qd_adj_get<2>(cur), sb - 2)
                    QD_CHECK_EDGE(qd_adj_get<3>(cur), sb - 3)
                    }
                }
                {
                    // last group: edges beyond the wavefront's largest degree are padding for every lane (posterior +inf:
                    // never a minimum, never negative) -- only their zero sign bit is shifted in
                    const ADJ4 cur = nx;
                    const int real = wmax - (k0 + kk);                              // 1..4 (or more in a non-final word)
                    QD_CHECK_EDGE(qd_adj_get<0>(cur), 3)
                    if (real > 1) QD_CHECK_EDGE(qd_adj_get<1>(cur), 2) else neww <<= 1;
                    if (real > 2) QD_CHECK_EDGE(qd_adj_get<2>(cur), 1) else neww <<= 1;
                    if (real > 3) QD_CHECK_EDGE(qd_adj_get<3>(cur), 0) else neww <<= 1;
                }
                npar ^= neww;
                if (k0 == 0) neg0 = neww;
                else if (SM == 1) neg1 = neww;
                else if (SM == 2) csgn_hi[((k0 >> 5) - 1) * m_pad + c] = neww;   // fixed up below once the parity is known
            }
            unsat |= us;
            // outgoing sign on edge k = syndrome ^ (parity of all incoming signs) ^ incoming sign k
            const uint32_t flip = 0u - ((synd ^ (uint32_t)__popc(npar)) & 1u);
            if (SM == 2)
                for (int k0 = 32; k0 < degp; k0 += 32) csgn_hi[((k0 >> 5) - 1) * m_pad + c] ^= flip;
            uint32_t nmeta = ((idx - llr_base) >> 2) | (synd << 31);
            if (SM == 1) nmeta |= ((neg1 ^ flip) & 0x0FFFu) << 16;      // at most 44 edges: signs 32..43 in bits 16..27; bit 30 is the decision parity
            chk[c] = make_float4(a1 * alpha, a2 * alpha, __uint_as_float(neg0 ^ flip), __uint_as_float(nmeta));
        }
        QD_BP_TICK(0)
        const int anyun = qd_block_or(unsat ? 1 : 0, misc, NW, phase);
        QD_BP_TICK(1)
        phase ^= 1;
        if (t >= 1 && !anyun) { converged = 1; break; }
        if (t == a.max_iter) break;
        // ---- bit pass t+1: posterior = prior + sum of check->bit messages, in ascending detector order.
        // Records beyond a fault's degree point at the dummy check (message +0), so edges are handled in fixed groups
        // (3 | 2 | 2 | 4 | 4 | 1) with one wave-uniform test per group instead of one per edge.
        // (the trip count and everything derived from it stay in scalar registers; only the last trip is partial)
        for (int base = 0; base < g.n; base += T) {
            const int b = base + tid;
            if (base + T > g.n && b >= g.n) break;
            const int b0 = base + wave0;                                // first slot of this wavefront
            const uint16_t mylabel = (uint16_t)b;                       // my slot; the check keeps its argmin slot in the low 16 bits of w
            // records through buffer loads: per-lane offset tid * 16 fixed, everything that moves is a scalar offset
            const qd_u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, rec_voff, base * 16, 0);
            qd_u32x4 r1, r2, r3, r4;
            if (NCH > 1 && b0 < g.bit_thr[3]) r1 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, rec_voff, (n_pad + base) * 16, 0);
            if (NCH > 2 && b0 < g.bit_thr[7]) r2 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, rec_voff, (2 * n_pad + base) * 16, 0);
            if (NCH > 3 && b0 < g.bit_thr[11]) r3 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, rec_voff, (3 * n_pad + base) * 16, 0);
            if (NCH > 4 && b0 < g.bit_thr[15]) r4 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, rec_voff, (4 * n_pad + base) * 16, 0);
            float acc = __uint_as_float(r0.x);
            float sabs = fabsf(acc);
            {
                qd_u32x4 s0 = QD_BIT_LOAD(r0.y), s1 = QD_BIT_LOAD(r0.z), s2 = QD_BIT_LOAD(r0.w);
                QD_BIT_WAIT3(s0, s1, s2);
                QD_BIT_USE(r0.y, s0) QD_BIT_USE(r0.z, s1)
                if (b0 < g.bit_thr[2]) QD_BIT_USE(r0.w, s2)          // (a wavefront of weight-2 faults gathers the dummy but skips its arithmetic)
            }
            if (NCH > 1 && b0 < g.bit_thr[3]) {
                qd_u32x4 s0 = QD_BIT_LOAD(r1.x), s1 = QD_BIT_LOAD(r1.y);
                QD_BIT_WAIT2(s0, s1);
                QD_BIT_USE(r1.x, s0)
                if (b0 < g.bit_thr[4]) QD_BIT_USE(r1.y, s1)
                if (b0 < g.bit_thr[5]) {
                    qd_u32x4 s2 = QD_BIT_LOAD(r1.z), s3 = QD_BIT_LOAD(r1.w);
                    QD_BIT_WAIT2(s2, s3);
                    QD_BIT_USE(r1.z, s2)
                    if (b0 < g.bit_thr[6]) QD_BIT_USE(r1.w, s3)
                }
            }
            if (NCH > 2 && b0 < g.bit_thr[7]) {
                qd_u32x4 s0 = QD_BIT_LOAD(r2.x), s1 = QD_BIT_LOAD(r2.y), s2 = QD_BIT_LOAD(r2.z), s3 = QD_BIT_LOAD(r2.w);
                QD_BIT_WAIT4(s0, s1, s2, s3);
                QD_BIT_USE(r2.x, s0) QD_BIT_USE(r2.y, s1) QD_BIT_USE(r2.z, s2) QD_BIT_USE(r2.w, s3)
            }
            if (NCH > 3 && b0 < g.bit_thr[11]) {
                qd_u32x4 s0 = QD_BIT_LOAD(r3.x), s1 = QD_BIT_LOAD(r3.y), s2 = QD_BIT_LOAD(r3.z), s3 = QD_BIT_LOAD(r3.w);
                QD_BIT_WAIT4(s0, s1, s2, s3);
                QD_BIT_USE(r3.x, s0) QD_BIT_USE(r3.y, s1) QD_BIT_USE(r3.z, s2) QD_BIT_USE(r3.w, s3)
            }
            if (NCH > 4 && b0 < g.bit_thr[15]) {
                qd_u32x4 s0 = QD_BIT_LOAD(r4.x);
                QD_BIT_WAIT1(s0);
                QD_BIT_USE(r4.x, s0)
            }
            llr[b] = acc;
            trip |= (sabs >= s_lim);
            if (acc <= 0.f) {
                // hard decision 1 (rare): tell the fault's checks -- one LDS atomic per edge on bit 30 of the state's last word
                // (records beyond the degree point at the dummy check; readers of the word ignore that bit)
                QD_BIT_FLIP(r0.y) QD_BIT_FLIP(r0.z) QD_BIT_FLIP(r0.w)
                if (NCH > 1 && b0 < g.bit_thr[3]) { QD_BIT_FLIP(r1.x) QD_BIT_FLIP(r1.y) QD_BIT_FLIP(r1.z) QD_BIT_FLIP(r1.w) }
                if (NCH > 2 && b0 < g.bit_thr[7]) { QD_BIT_FLIP(r2.x) QD_BIT_FLIP(r2.y) QD_BIT_FLIP(r2.z) QD_BIT_FLIP(r2.w) }
                if (NCH > 3 && b0 < g.bit_thr[11]) { QD_BIT_FLIP(r3.x) QD_BIT_FLIP(r3.y) QD_BIT_FLIP(r3.z) QD_BIT_FLIP(r3.w) }
                if (NCH > 4 && b0 < g.bit_thr[15]) { QD_BIT_FLIP(r4.x) }
            }
        }
        QD_BP_TICK(2)
        __syncthreads();
        QD_BP_TICK(3)
        ++t;
    }

    // ---- grid arithmetic: did the bound hold?  (a.s_limit = 0: LLRs are not on a grid, nothing to certify)
    int status_or = a.status_or;
    if (a.s_limit > 0.f) {
        const int tripped = qd_block_or(trip ? 1 : 0, misc, NW, phase);
        if (tripped) {
            if (a.redo_list) {                                     // park the shot for the coarse-grid pass, if there is room
                if (tid == 0) misc[33] = atomicAdd(a.redo_count, 1);
                __syncthreads();
                const int at = misc[33];
                if (at < a.redo_cap) {
                    if (tid == 0) a.redo_list[at] = (int32_t)shot;
                    return;
                }
            }
            status_or |= QD_STATUS_INEXACT;
        }
    }
    // ---- hard decision, packed by fault index
    for (int b = tid; b < g.n; b += T)
        if (llr[b] <= 0.f) {
            const uint32_t j = g.bit_orig[b];
            atomicOr(&outw[j >> 5], 1u << (j & 31u));
        }
    if (!converged && a.want_llr && tid == 0) misc[32] = atomicAdd(a.fail_count, 1);
    __syncthreads();
    for (int w = tid; w < g.out_words; w += T) a.err_bits[shot * g.out_words + w] = outw[w];
    if (!converged && a.want_llr) {
        const int slot = misc[32];
        float *dst = a.llr_ws + (int64_t)slot * n_pad;
        for (int b = tid; b < g.n; b += T) dst[b] = llr[b];
        if (tid == 0) a.fail_list[slot] = (int32_t)shot;
    }
    if (tid == 0) a.status[shot] = t | (converged << 16) | status_or;
#ifdef QD_BP_TIMING
    QD_BP_TICK(5)
    if (tid == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&a.dbg[i], acc_[i]);
        atomicAdd(&a.dbg[6], 1ull); atomicAdd(&a.dbg[7], (unsigned long long)t);
    }
#endif
}

// ---- launch wrappers -------------------------------------------------------------------------------------------------
template <int T, int NCH, int SM, typename ADJ4, int MW>
static hipError_t launch_bp_m(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    auto k = qd_bp_minsum_kernel<T, NCH, SM, ADJ4, MW>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)B), dim3(T), g.lds_bytes, s, g, a);
    return hipGetLastError();
}

// Round 6: K1 is the recheck / coarse-grid pass of the scatter kernels and the flooding min-sum kernel of the windows they do not take; it is on
// no timed path.  Its 96 instantiations (16-bit adjacency, a 128-register build for QLP-size windows) are down to the 36 a graph can select:
// threads x record words x sign mode.
template <int T, int NCH>
static hipError_t launch_bp_n(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    if (g.sign_mode == 0) return launch_bp_m<T, NCH, 0, uint4, QD_BP_MINWAVES_T(T)>(g, a, B, s);
    if (g.sign_mode == 1) return launch_bp_m<T, NCH, 1, uint4, QD_BP_MINWAVES_T(T)>(g, a, B, s);
    return launch_bp_m<T, NCH, 2, uint4, QD_BP_MINWAVES_T(T)>(g, a, B, s);
}

template <int T>
static hipError_t launch_bp_t(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    switch (g.rec_words / 4) {
    case 1: return launch_bp_n<T, 1>(g, a, B, s);
    case 2: return launch_bp_n<T, 2>(g, a, B, s);
    case 3: return launch_bp_n<T, 3>(g, a, B, s);
    default: return launch_bp_n<T, 5>(g, a, B, s);
    }
}

hipError_t qd_launch_bp(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    switch (g.threads) {
    case 256: return launch_bp_t<256>(g, a, B, s);
    case 512: return launch_bp_t<512>(g, a, B, s);
    default: return launch_bp_t<1024>(g, a, B, s);
    }
}
