// qd_internal.h -- device-side views shared by the kernels and the C-ABI host code (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QD_WAVE 64
#define QD_MAX_COL_DEG 16      // bit-side sign copy is a uint16 per fault
#define QD_MAX_ROW_DEG 255     // edge position inside a check is a byte
#define QD_LDS_BYTES (160 * 1024)
#ifndef QD_SR_KWR
#define QD_SR_KWR 4            // Q planes (64 pivots each) qd_osd0_sr_kernel keeps in registers (variant builds: -DQD_SR_KWR=8 -DQD_SR_WPS12=3)
#endif
#ifndef QD_SR_WPS12
#define QD_SR_WPS12 4          // wavefronts per SIMD the register budget of its one- and two-rows-per-thread shapes is cut for
#endif
#ifndef QD_SR_KWR_OF
#define QD_SR_KWR_OF(rpt) QD_SR_KWR
#endif
#ifndef QD_SR_WPS_OF
#define QD_SR_WPS_OF(rpt) ((rpt) <= 2 ? QD_SR_WPS12 : 2)   // wavefronts per SIMD its register budget is cut for: 128 / 256 registers -- no instantiation may
                                                 // spill vector registers (scalar registers it cannot keep live in lanes of vector registers)
#endif
#ifndef QD_SR_TSMALL
#define QD_SR_TSMALL 512       // its workgroup size for windows of <= 1024 detectors
#endif
//      QD_SR_KWR_OF(rpt)       // Q planes (64 pivots each) the many-pivots-per-round OSD-0 kernel keeps in registers at rpt rows per thread (osd_sr.hip)
#define QD_SR_KWR_MAX (QD_SR_KWR > 4 ? QD_SR_KWR : 4)

// One window's Tanner graph as the BP kernel wants it.
//   check slots: checks sorted by degree (descending); bit slots: faults sorted by degree (descending), so that a
//   wavefront's lanes run the same trip count.  Indices refer to SLOTS; the LDS arrays are indexed by slot.
//   Everything is pre-scaled to LDS byte offsets so the kernel does no address arithmetic on the gathers.
struct BpGraphDev {
    int m, n, m_pad, n_pad;
    int max_rdeg, max_cdeg, neg_words, out_words;
    int max_rdeg_pad;           // max_rdeg rounded up to a multiple of 4
    int dummy_bit, dummy_chk;   // LDS slots that pad short rows/columns: llr[dummy_bit] = +inf, chk[dummy_chk] = zero message
    int rec_words;              // uint32 words per fault record (multiple of 4)
    int sign_mode;              // 0: every check has <= 32 (padded) edges; 1: <= 44, signs 32..46 ride in the state's meta word;
                                // 2: wider, signs 32.. live in separate LDS words
    int adj32;                  // 1: chk_adj holds uint32 entries (windows with more than 16379 fault slots), else uint16
    const void *chk_adj;        // [max_rdeg_pad/4][m_pad][4] absolute LDS byte offset (off_llr + slot * 4) of the posterior of the k-th
                                //                        fault of the check, k ascending = original column order; the dummy bit beyond the degree
    const int32_t *chk_degp_w;  // [m_pad / 64]           low 16 bits: trip count of a wavefront of check slots (its max degree rounded up
                                //                        to 4); high 16 bits: that max degree itself
    const uint32_t *chk_orig;   // [m_pad]                detector index of the check slot
    const uint32_t *bit_rec;    // [rec_words/4][n_pad][4] word 0 = prior LLR (float bits: log((1-p)/p) computed in double, rounded once);
                                //                        word 1+q = (LDS byte offset of the check state) << 16 | where the check keeps this edge's
                                //                        sign (mode 0/1: bit index 0..63 into {w, z}; mode 2: word index << 5 | bit index)
                                //                        for the q-th check of the fault (q ascending = original row order)
    const uint32_t *bit_orig;   // [n_pad]                fault index of the bit slot
    const uint32_t *bit_slot_of;// [n]                    bit slot of a fault (inverse of bit_orig)
    int bit_thr[QD_MAX_COL_DEG];// bit_thr[q] = number of bit slots (multiple of 64) whose wavefront has a fault of degree > q
    // LDS carve-up (byte offsets, 16-byte aligned)
    int off_chk, off_cneg, off_llr, off_out, off_misc, lds_bytes;   // off_misc: 64 ints of reduction scratch
    int threads;
};


// The scatter form of the flooding min-sum kernel (bp_scatter.hip): a check keeps its state in the registers of its lane and ADDS its
// messages to the faults' integer accumulators in LDS (ds_add_u32), so there is no bit pass.  Two posterior buffers A / B alternate
// (one is read while the other accumulates); the adjacency exists once per buffer so that a gather / scatter needs no address add.
struct ScatGraphDev {
    int ok;                     // 1: this window can run in the scatter kernel (m <= threads, rows of 2..64 faults, the LDS holds both buffers)
    const uint32_t *adjA, *adjB;// [max_rdeg_pad/4][m_pad][4] LDS byte offset of the fault's accumulator in buffer A / B, in the walk order of
                                //                        BpGraphDev::chk_adj; the trash slot beyond a check's degree
    const uint32_t *deg_w;      // [m_pad/64]             per wavefront of check slots: trip count (multiple of 4) | largest degree << 8 | smallest << 16
    const uint8_t *chk_deg;     // [m_pad]                degree of the check slot (0 beyond m)
    int offA, offB, off_out, off_bmap, off_misc, lds_bytes;
    // The accumulators have their own slot order (not BpGraphDev's bit slots): bank = slot mod 32 is chosen per fault so that the 32 lanes of a
    // half-wavefront can meet 32 different banks at every step of the walk (qd_graph_create: scatter_banks / scatter_walk).
    int nslots;                 // accumulators incl. unused slots and 32 trash slots (one per bank) for the steps beyond a check's degree; multiple of 4
    const uint32_t *slot_fault; // [nslots] fault of the slot, 0xFFFFFFFF: none (its accumulator stays 0)
    const uint32_t *k1_slot;    // [n] BpGraphDev's bit slot (the OSD workspace's row layout) -> accumulator slot of that fault
    const int32_t *wave_map;    // [wide_cpl][wide_threads / 64] the 64 consecutive check slots (index / 64) a wavefront takes in its j-th round, -1:
                                //                        none; chosen so that the wavefronts of a workgroup walk equally many edges (slots are sorted by degree)
    int wide_threads, wide_cpl; // 0: one check per lane (bp_scatter.hip); else the workgroup size and the checks per lane of
                                //    qd_bp_scatter_wide_kernel (bp_scatter_wide.hip)
};
struct ScatArgs {
    const int32_t *prior_g;     // [n_pad] channel LLR of the bit slot in grid units (llr * 2^k, an integer) MINUS ONE: the accumulators hold L - 1
    float grid_inv;             // 2^-k
    float m2_limit;             // the run is certified exact while every check's second minimum stays below it (grid units)
    int32_t *recheck_list;      // shots the cheap bound could not certify: decoded again by qd_bp_minsum_kernel, which carries the per-fault bound
    int32_t *recheck_count;
    int recheck_cap;
};

// The general (one message per edge) BP kernel's view: plain CSR + CSC in fault / detector order, prior LLRs in float.
// column-weight bound the per-edge kernel's serial schedule unrolls for (register arrays of that length, record width)
#ifndef QD_GEN_D6
#define QD_GEN_D6 0           // 1: a weight-6 instantiation between 4 and 8 (A/B, profiles/r05_k1g_*)
#endif
static inline int qd_gen_unroll(int max_cdeg) { return max_cdeg <= 4 ? 4 : ((QD_GEN_D6 && max_cdeg <= 6) ? 6 : (max_cdeg <= 8 ? 8 : QD_MAX_COL_DEG)); }
#ifndef QD_GEN_GS
#define QD_GEN_GS 4           // wavefronts per 64 shots in the serial schedule (faults of one dependency level in parallel)
#endif
struct GenGraphDev {
    int m, n, nnz, out_words;
    const int32_t *rp, *ci;     // [m + 1], [nnz]   CSR, columns ascending in a row
    const int32_t *cp, *ri;     // [n + 1], [nnz]   CSC, rows ascending in a column
    const int32_t *c2r;         // [nnz]            CSC edge -> CSR edge
    const uint16_t *erow;       // [nnz]            CSR edge -> its row
    const uint16_t *frec;       // [n][16]          the CSR edges of a fault's column (8 entries), then their rows (8), 0xFFFF beyond its weight
                                //                  (null if a column is heavier than 8 or nnz > 65534)
    const float *llr0;          // [n]              (float)log((1 - p) / p), the log in double
    const int32_t *ell;         // [m][ell_w][2]    rows in ELL form for BP-LSD: {fault, its posterior column}, {-1, 0} padding
    int ell_w;                  //                  max row weight rounded up to a multiple of 64
    // serial schedule: faults grouped into dependency levels.  Two faults that share no check commute, so natural order
    // is reproduced by any order that keeps every pair of faults with a common check in index order; level(j) = 1 + the
    // highest level among earlier faults on j's checks.  Faults of one level are mutually independent.
    // The G = QD_GEN_GS wavefronts that share 64 shots take the faults of a level in parallel, one barrier per level.  What a
    // wavefront needs to know about its next fault -- index, weight, prior LLR, the rows and CSR edges of its column -- is ONE record
    // of srec_w dwords, read with one scalar load a whole step ahead (the adjacency arrays cost 3 + 2 * weight dependent scalar
    // round trips per fault):  srec[(step * G + wavefront) * srec_w + ...] =
    //   [0] fault | weight << 24 | (barrier after this step) << 31     (weight 0: nothing to do in this step)
    //   [1] prior LLR (float bits)      [2 .. 2 + D) rows      [2 + D .. 2 + 2 D) CSR edges      D = 4, 8 or QD_MAX_COL_DEG
    int nlev, nstep, srec_w;
    const uint32_t *srec;       // [nstep][G][srec_w]
    // Row i's running prefix is live from the level of its first fault to the level of its last; rows whose intervals do not
    // overlap share one of `nslots` LDS slots (64 floats each; greedy interval colouring = the minimum), so the 8 bytes per
    // edge and sweep the prefixes would move through HBM stay on the CU.  Then a row entry of a record is
    //   row | (first entry of its row: the prefix starts at +-1 / +-max, sign = syndrome bit) << 23 | slot << 24.
    // nslots = 0: too many slots for the LDS budget, the prefixes live in the [m][S] plane GenWs::pre and the entry is the row.
    int nslots;
};
// ... and its per-chunk workspace, [index][shot] with S shots per row
struct GenWs {
    float *b2c, *c2b, *th, *llr;    // [nnz][S] x 3 (th: product-sum only; b2c: not for serial product-sum, which keeps tanh(b2c/2) alone), [n][S]
    float *pre;                     // [m][S]  serial schedule: running prefix of each row (product / minimum + sign parity)
    uint8_t *syn;                   // [m][S]
    int32_t *slot;                  // [S]  fail-list slot of a shot BP could not finish, else -1
    int64_t S;
};
// One launch of the serial schedule = iterations it0 + 1 .. it_end of the shots in its columns.  A launch that is not the last hands the
// shots that have not converged on, PACKED: their state between two sweeps is the message plane and the syndrome plane, nothing else (suffixes,
// prefixes and posteriors are rebuilt by every sweep), so the survivors' columns of those two planes are copied to consecutive columns of `next`
// and the next launch runs on full wavefronts (bp.hpp's shot loop has no such problem: one shot, one thread; here a lane that has converged
// idles until the slowest of its 64 shots is done -- 31 % of the lanes of the reference-settings windows, profiles/r05_k1g_load_curve.txt).
// Wavefront priority of qd_osd0_sr_kernel.  In the pipelined driver it runs beside the BP kernel of the other lane, whose workgroups fill every wavefront
// slot: an OSD workgroup takes the place of one BP workgroup on its CU for as long as it lives, and at equal priority it lives 2-5 times longer than alone.
#ifndef QD_SR_PRIO
#define QD_SR_PRIO 0
#endif
// hipEventQuery without side effects: "not ready" is an answer, not an error -- it must not be what a later hipGetLastError() behind a kernel launch reports
static inline bool qd_event_done(hipEvent_t e)
{
    const hipError_t r = hipEventQuery(e);
    if (r != hipSuccess) (void)hipGetLastError();
    return r == hipSuccess;
}
#define QD_GEN_MAX_STAGES 12
struct GenStagePlan;
struct GenStage {
    const int32_t *in_shot;     // [columns] the shot (index into the batch) of each column; nullptr: column c holds shot shot0 + c (first launch)
    const int32_t *in_count;    // columns in use (device); nullptr: nshots
    int32_t *out_shot;          // the survivors' shots, by column of `next`
    int32_t *out_count;
    int it0, it_end, last;      // last: it_end is max_iter -- shots still running are BP failures, there is no next launch
    GenWs next;
};
struct GenStagePlan {           // host side: the iteration bounds between the launches and what they hand over
    int nbounds;
    int bounds[QD_GEN_MAX_STAGES];
    GenWs w2;                   // the workspace of the odd launches: its own message and syndrome planes (msg2, syn2), everything else shared with the first
    float *msg2;
    uint8_t *syn2;
    int32_t *lists[2];          // [S] each
    int32_t *counts;            // [QD_GEN_MAX_STAGES + 1]
    // what the last staged call packed, read back without waiting (pinned copy + event): a decoder whose shots do not converge -- a window far above
    // threshold -- gains nothing from packing and pays the copies; it then runs in one launch and tries again every QD_GEN_PROBE calls.  Same results either way.
    int32_t *host_counts;
    hipEvent_t counts_ready;
    int pending, pending_shots, pending_nb;
    int one_launch_calls;
};
#define QD_GEN_PROBE 16

// Elimination (OSD) view: original indexing.
struct OsdGraphDev {
    int m, n, m_pad, max_cdeg;
    int mw;                     // words per Q row = ceil(m / 64)
    int npow2;                  // bitonic sort size of the full kernel
    int kw_lds;                 // full kernel: Q word-planes that live in LDS; planes >= kw_lds spill to global
    const uint32_t *csc_ptr;    // [n + 1]
    const uint16_t *csc_row;    // [nnz]  detector index, ascending inside a column
    // LDS carve-up of the full kernel: off[] = q, tb, sp, rowpiv, prow, pcol, pairs, cols, red, out
    int off[10], lds_bytes;
    // ... of the register kernel for OSD-0 (aims at two workgroups per CU; f_lds_bytes = 0 disables it)
    int f_off[10], f_off_hist, f_off_sort, f_off_order, f_off_pivmask, f_off_npl, f_kw, f_lds_bytes, f_threads;
    // ... and of the register kernel for OSD-CS / OSD-E (one workgroup per CU, as many Q planes in LDS as fit: the
    //     candidate sweep reads arbitrary Q bits of every pivot row)
    int w_off[10], w_off_sort, w_off_order, w_off_pivmask, w_off_npl, w_kw, w_lds_bytes;
    const uint32_t *wfix;       // [n] integer candidate costs round(log(1/p) * 2^18) for OSD-CS / OSD-E
    uint32_t max_wfix;          // largest of them (the rebuilt OSD-CS / OSD-E kernel adds 64 of them in 32 bits)
    int threads;
    // OSD-0 with simultaneous singleton pivots (osd_sr.hip, qd_osd0_sr_kernel): its LDS layout (s_lds_bytes = 0: not taken),
    // instantiation and the columns in ELL form
    int s_off[13], s_lds_bytes, s_threads, s_rpt, s_per_cu;
    const uint16_t *csc_ell;    // [n][1 << ell_log2] detector indices of a fault, ascending, 0xFFFF beyond its weight
    int ell_log2;
};

struct DecodeArgs {
    const uint8_t *det;         // [B][det_stride] one byte per detector
    int64_t det_stride, det_offset;
    const uint8_t *upd;         // [B][upd_stride] or null
    int64_t upd_stride;
    int upd_rows;
    int max_iter;
    float ms_scale;             // 0 -> 1 - 2^-it
    int want_llr;               // 1: non-converged shots publish their posteriors for OSD
    uint32_t *err_bits;         // [B][out_words]
    int32_t *status;            // [B]
    // workspace
    float *llr_ws;              // [cap][n_pad]  posterior by bit slot, one row per non-converged shot
    int32_t *fail_list;         // [cap]
    int32_t *fail_count;        // [1]
    uint16_t *order_ws;         // [blocks][n]   sorted column order (full OSD kernel)
    uint64_t *q_spill;          // [blocks][(mw - kw_lds)][m_pad]
    uint64_t *q_spill_fast;     // [blocks_fast][(mw - f_kw)][m_pad]   register kernel
    uint64_t *q_spill_sr;       // [blocks_sr][...] spilled Q planes and parked state of qd_osd0_sr_kernel (osd_sr.hip, qd_osd_sr_ws_words)
    uint64_t *mt_ws;            // [blocks_fast][mw * m_pad + 2048]    higher-order OSD: transposed Q + candidate vectors
    int32_t *hard_list;         // [cap]         fail-list slots the first fast OSD pass could not finish
    int32_t *hard_list2;        // [cap]         ... and the second
    int32_t *hard_count;        // [2]
    unsigned long long *dbg;    // [16] phase cycle counters (only written by -DQD_OSD_TIMING builds)
    int osd_w, osd_order, rank; // higher-order OSD: 0 = OSD-0, 1 = combination sweep, 2 = exhaustive; GF(2) rank of the window matrix
    // ---- grid arithmetic of the LDS min-sum kernel (channel LLRs are multiples of 2^-k, ms_scaling = 1: see bp_kernels.hip)
    float s_limit;              // 2^(23-k): the run is exact while every S_j = |llr0_j| + sum |c2b| stays below it; 0 = not on a grid
    const int32_t *shot_list;   // redo pass: workgroup x decodes shot shot_list[x] (x < *shot_count); null = shot x
    const int32_t *shot_count;
    int32_t *redo_list;         // first pass: shots whose bound tripped are parked here for the coarse-grid pass; null = last pass
    int32_t *redo_count;
    int redo_cap;
    int status_or;              // ORed into the status word (QD_STATUS_COARSE_GRID in the redo pass)
};

// Workgroup-wide OR without static LDS (a static __shared__ object in front of the dynamic region can knock the
// 16-byte alignment the ds_read_b128 gathers rely on).  `red` = 32 ints of dynamic LDS, 16-byte aligned; `phase`
// alternates 0/1 between successive calls so a fast wave cannot overwrite flags a slow wave still reads.  One barrier.
__device__ __forceinline__ int qd_block_or(int pred, int *red, int nwaves, int phase)
{
    const unsigned long long bal = __ballot(pred);
    if ((threadIdx.x & 63) == 0) red[phase * 16 + (threadIdx.x >> 6)] = (bal != 0ull);
    __syncthreads();
    const int4 *r4 = reinterpret_cast<const int4 *>(red + phase * 16);
    int r = 0;
    for (int w = 0; w < (nwaves + 3) / 4; ++w) {
        const int4 v = r4[w];                   // slots beyond nwaves are zero (cleared once at kernel start)
        r |= v.x | v.y | v.z | v.w;
    }
    return r;
}

// Wave-wide reductions on the DPP path (row shifts + row broadcasts: a handful of cycles per step, no LDS traffic);
// the result is uniform (taken from lane 63).
#define QD_DPP(v, old, ctrl, rowmask) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rowmask), 0xf, false))
__device__ __forceinline__ uint32_t qd_wave_umin(uint32_t v)
{
    v = min(v, QD_DPP(v, 0xFFFFFFFFu, 0x111, 0xf));   // row_shr:1
    v = min(v, QD_DPP(v, 0xFFFFFFFFu, 0x112, 0xf));   // row_shr:2
    v = min(v, QD_DPP(v, 0xFFFFFFFFu, 0x114, 0xf));   // row_shr:4
    v = min(v, QD_DPP(v, 0xFFFFFFFFu, 0x118, 0xf));   // row_shr:8   -> lane 15 of each row holds the row minimum
    v = min(v, QD_DPP(v, 0xFFFFFFFFu, 0x142, 0xa));   // row_bcast:15 into rows 1 and 3
    v = min(v, QD_DPP(v, 0xFFFFFFFFu, 0x143, 0xc));   // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t qd_wave_or(uint32_t v)
{
    v |= QD_DPP(v, 0u, 0x111, 0xf);
    v |= QD_DPP(v, 0u, 0x112, 0xf);
    v |= QD_DPP(v, 0u, 0x114, 0xf);
    v |= QD_DPP(v, 0u, 0x118, 0xf);
    v |= QD_DPP(v, 0u, 0x142, 0xa);
    v |= QD_DPP(v, 0u, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t qd_wave_add(uint32_t v)
{
    v += QD_DPP(v, 0u, 0x111, 0xf);
    v += QD_DPP(v, 0u, 0x112, 0xf);
    v += QD_DPP(v, 0u, 0x114, 0xf);
    v += QD_DPP(v, 0u, 0x118, 0xf);
    v += QD_DPP(v, 0u, 0x142, 0xa);
    v += QD_DPP(v, 0u, 0x143, 0xc);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct SpmatDev {
    int nrows, ncols, nnz;
    const uint32_t *row_ptr;
    const uint32_t *col_idx;
    const uint32_t *colmask;    // [ncols][mask_words] the rows of each column as a bit mask (null when nrows > 512)
    int mask_words;
};
