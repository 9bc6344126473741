/* quits_amd.h -- C ABI of libquits_amd.so, the MI355X (gfx950) sliding-window BP-OSD decoder.
 *
 * This is the drop-in boundary for the decoding hot path of mkangquantum/quits.  Every entry point names the
 * reference interface it stands in for (paths relative to /root/reference/src/quits/).  The reference reaches its
 * inner decoder through the Python extension `ldpc.bposd_decoder.BpOsdDecoder`; a maintainer would bind this
 * library with ctypes exactly as quits_amd/_lib.py does (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary; return 0 on success, a negative QD_E* code on error,
 *     with a thread-local message behind qd_last_error();
 *   - pointers named d_* are DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr()); the library
 *     allocates only its own graph and workspace memory and frees it in *_destroy;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); NULL = default stream;
 *   - all launches are asynchronous; nothing here synchronises the device unless it says so;
 *   - one process per GPU; distinct decoders are independent, one decoder must not be used concurrently.
 */
#ifndef QUITS_AMD_H
#define QUITS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_OK 0
#define QD_EINVAL (-1)      /* bad argument                                             */
#define QD_EUNSUPPORTED (-2)/* legal for ldpc, not implemented on the device path      */
#define QD_EHIP (-3)        /* HIP runtime error                                        */
#define QD_ECAPACITY (-4)   /* graph does not fit the kernels' on-chip layout          */

/* bp_method / schedule / osd_method: the string options of quits/decoder/bposd.py:27-29 as integers */
#define QD_BP_PRODUCT_SUM 0
#define QD_BP_MINIMUM_SUM 1
#define QD_SCHEDULE_PARALLEL 0
#define QD_SCHEDULE_SERIAL 1
#define QD_OSD_OFF 0
#define QD_OSD_0 1
#define QD_OSD_E 2
#define QD_OSD_CS 3
#define QD_LSD_0 4            /* BP-LSD (ldpc.bplsd_decoder.BpLsdDecoder, quits/decoder/bplsd.py:5) as values of osd_method:       */
#define QD_LSD_E 5            /*   lsd_method 'lsd_0' / 'lsd_e' / 'lsd_cs' (bplsd.py:10,54 forward lsd_method, lsd_order);        */
#define QD_LSD_CS 6           /*   osd_order carries lsd_order (0: the three are the same decoder)                                */

/* status word written per shot by qd_decode_batch */
#define QD_STATUS_ITER_MASK 0x3FFF      /* BP iterations used (max_iter is capped at 16383)           */
#define QD_STATUS_COARSE_GRID (1 << 14) /* flooding min-sum: decoded on the coarse LLR grid (see qd_decoder_info)          */
#define QD_STATUS_INEXACT (1 << 15)     /* ... and even there the exactness bound tripped: float rounding may have occurred */
#define QD_STATUS_CONVERGED (1 << 16)   /* BP reproduced the syndrome                                 */
#define QD_STATUS_OSD (1 << 17)         /* OSD post-processing produced the output                    */
#define QD_STATUS_INCONSISTENT (1 << 18)/* OSD: syndrome outside the column space of the window matrix */
#define QD_STATUS_ZERO (1 << 19)        /* all-zero syndrome short-circuit                            */

typedef struct qd_graph qd_graph;       /* one window's Tanner graph + priors, resident on one device */
typedef struct qd_decoder qd_decoder;   /* graph + parameters + device workspace                      */
typedef struct qd_spmat qd_spmat;       /* sparse GF(2) matrix on the device (L_k, U_k, H for sampling) */

/* qd_params.reserved: run flooding min-sum in the one-message-per-edge kernel too (ldpc's own update order, prefix sums
 * instead of "total minus own").  On the LLR grid both kernels compute exactly and agree bit for bit.  Validation aid. */
#define QD_FLAG_EDGE_MESSAGES 1
/* qd_params.reserved: keep the channel LLRs as (float)log((1-p)/p) instead of putting them on a binary grid (round-1
 * arithmetic: float rounding in every sum).  Validation aid; see qd_decoder_info. */
#define QD_FLAG_RAW_LLR 2

/* Keyword arguments the reference hands to BpOsdDecoder (decoder/bposd.py:38-49,74-83). */
typedef struct qd_params {
    int32_t bp_method;          /* QD_BP_*        ; MINIMUM_SUM + PARALLEL runs in the compressed LDS kernel,        */
    int32_t schedule;           /* QD_SCHEDULE_*  ; every other pair in the one-message-per-edge kernel (HBM)        */
    int32_t max_iter;           /* 0 -> number of faults n (ldpc convention)                     */
    int32_t osd_method;         /* QD_OSD_* | QD_LSD_* ; device path: OFF, 0, CS (order <= 64), E (order <= 15), LSD-0 / -CS / -E (same limits) */
    int32_t osd_order;
    int32_t reserved;           /* flag bits, QD_FLAG_*; 0 for the reference's behaviour                           */
    double ms_scaling_factor;   /* not exposed by the reference wrapper -> ldpc default 1.0; 0 = 1-2^-it */
} qd_params;

int qd_version(void);                 /* 103 (103: qd_decoder_post_head_start; 101: qd_graph_info wrote 12 entries; 102: 10 again + qd_graph_info_ex) */
const char *qd_last_error(void);
/* Number of visible HIP devices (0 if none): lets a host fail loudly before building anything. */
int qd_device_count(void);

/* ---- graph: replaces the sparse-matrix half of BpOsdDecoder.__init__ (call sites
 *      decoder/sliding_window.py:61,69,149,152).  CSR of the window check matrix (m detectors x n faults, column
 *      indices ascending in each row) and the n channel probabilities (`channel_probs`, sliding_window.py:148,151;
 *      pass n copies of `error_rate` for the phenomenological variant, bposd.py:43,49). */
int qd_graph_create(int32_t m, int32_t n, const int32_t *row_ptr, const int32_t *col_idx, const double *priors,
                    int32_t device, qd_graph **out);
void qd_graph_destroy(qd_graph *g);
/* info[0..9] = m, n, nnz, max row weight, max column weight, BP block threads, BP LDS bytes, OSD block threads,
 *              OSD LDS bytes, GF(2) rank of the matrix.  Exactly 10 entries are written (as in library version 100; version 101
 *              wrote 12 -- callers built against that header should move to qd_graph_info_ex). */
int qd_graph_info(const qd_graph *g, int32_t *info);
/* The same with the caller saying how many int32 entries `info` has room for; entries 10, 11 = modelled LDS cycles of one pass of
 * the scatter kernels' walk over the accumulators (bank conflicts included) and the same without any conflict (0, 0: the window
 * does not run there); entries beyond the ones this version knows are set to 0. */
int qd_graph_info_ex(const qd_graph *g, int32_t *info, int32_t n_entries);

/* ---- decoder: replaces BpOsdDecoder.__init__'s parameter half. */
int qd_decoder_create(const qd_graph *g, const qd_params *params, qd_decoder **out);
void qd_decoder_destroy(qd_decoder *d);
/* Arithmetic of this decoder.  info[0] = k: the channel LLRs log((1-p)/p) (computed in double, as ldpc does) are rounded to
 * the nearest multiple of 2^-k before BP starts; -1 = not rounded (product-sum, ms_scaling_factor != 1, QD_FLAG_RAW_LLR).
 * Flooding min-sum with ms_scaling_factor = 1 only adds, subtracts, negates and compares messages, so on such a grid
 * single-precision arithmetic is EXACT while magnitudes stay below 2^(24-k): the kernels then return, bit for bit, what
 * ldpc's double-precision BpDecoder returns for those LLRs, in any summation order.  The LDS kernel proves this per shot
 * (bound S < 2^(23-k), bp_kernels.hip); a shot whose bound trips is decoded again on the coarser grid info[1]
 * (QD_STATUS_COARSE_GRID), and flagged QD_STATUS_INEXACT if that trips too.  k = max(10, r), r = 23 - ceil(log2(8 * max|llr| * max_iter))
 * clamped to [2, 20]; info[1] = max(r - 4, 0) if r >= 10, else r (the rule's own grid takes the shots that outgrow 2^-10).  info[2] = 1 if BP runs in the one-message-per-edge kernel.  info[3] = 1 if the fine-grid pass of flooding min-sum runs in
 * the scatter kernel (bp_scatter.hip: same results, bit for bit; QD_NO_SCATTER=1 in the environment keeps the gather kernel),
 * 2 if in its several-checks-per-lane form (bp_scatter_wide.hip: the default shape -- two checks per lane on half the lanes; three per
 * lane for windows of more than 1024 checks or rows of 65..96 faults).
 * No reference counterpart (ldpc computes in double). */
int qd_decoder_info(const qd_decoder *d, int32_t *info);
/* Which kernel post-processes the shots BP leaves unconverged (decoder/device.py reports it; bench.py labels its roofline.osd object
 * with it).  No reference counterpart: ldpc has one OsdDecoder / LsdDecoder. */
#define QD_POST_NONE 0          /* osd_method = osd_off                                                                          */
#define QD_POST_OSD0_SR 1       /* OSD-0, many pivots per barrier round: qd_osd0_sr_kernel (osd_sr.hip)                          */
#define QD_POST_OSD0_REG 2      /* OSD-0, one pivot per round: qd_osd0_reg_kernel (osd_kernels.hip)                              */
#define QD_POST_OSD_W_OLD 3     /* OSD-CS / OSD-E by row (qd_osd0_reg_kernel<.., true>, osd_kernels.hip): windows the panel kernel does not take */
#define QD_POST_OSD_CS_PANEL 4  /* OSD-CS / OSD-E, round 5: qd_osdcs_kernel (osd_cs.hip), one panel of 64 sorted columns at a time */
#define QD_POST_LSD 5           /* BP-LSD: qd_lsd0_kernel (lsd_kernels.hip)                                                      */
int qd_decoder_postproc_kernel(const qd_decoder *d);
/* Pre-size the device workspace for batches of up to max_batch shots (otherwise grown on demand, which
 * synchronises). */
int qd_decoder_reserve(qd_decoder *d, int64_t max_batch);
/* Cap the message workspace of the one-message-per-edge BP kernel (bytes; default 48 GB or QD_GENERAL_WS_GB).  Larger
 * batches are decoded in equal chunks that fit.  A sliding-window plan holds one decoder per window: the host divides
 * the budget among them.  No reference counterpart (memory management). */
int qd_decoder_set_workspace_limit(qd_decoder *d, int64_t bytes);
/* Hand the device workspace back (posteriors, fail lists, message planes, elimination scratch); the decoder stays usable and
 * sizes it again at the next decode.  The host keeps sliding-window plans between calls (the reference is called once per
 * experiment point, bposd.py:54-86) and only the plan in use keeps its workspace.  No reference counterpart (memory management). */
int qd_decoder_release_workspace(qd_decoder *d);

/* ---- decode: replaces the per-shot `decoder.decode(syndrome)` calls (sliding_window.py:85,95,171,182) for a
 *      whole batch of shots, including the syndrome preparation in front of them (:168-169,179-180):
 *        syndrome[b][i] = d_det[b * det_stride + det_offset + i]  (i < m)   XOR   d_upd[b * upd_stride + i] (i < upd_rows)
 *      d_det: one byte per detector (0/1), i.e. the `zcheck_samples` array; d_upd may be NULL.
 *      Outputs: d_err_bits[b][w], w < ceil(n/32): bit (j & 31) of word (j >> 5) = decoded fault j;
 *               d_status[b]: QD_STATUS_* flags | iterations. */
int qd_decode_batch(qd_decoder *d, const uint8_t *d_det, int64_t det_stride, int64_t det_offset,
                    const uint8_t *d_upd, int64_t upd_stride, int32_t upd_rows, int64_t B,
                    uint32_t *d_err_bits, int32_t *d_status, void *stream);

/* The two stages of qd_decode_batch separately: stage 1 = BP (non-converged shots and their posteriors are parked in
 * the decoder's workspace), stage 2 = OSD over the shots parked by the preceding stage-1 call with the SAME
 * arguments, stage 3 = both.  Lets a host overlap the (latency-bound) OSD of one batch with the (ALU-bound) BP of the
 * next on a second stream, using one decoder per batch in flight. */
int qd_decode_stage(qd_decoder *d, const uint8_t *d_det, int64_t det_stride, int64_t det_offset, const uint8_t *d_upd,
                    int64_t upd_stride, int32_t upd_rows, int64_t B, uint32_t *d_err_bits, int32_t *d_status,
                    int32_t stage, void *stream);

/* The decoder's OSD stage alone (OSD-0, or OSD-CS / OSD-E when the decoder was created with them), on caller-supplied soft information: every shot of the batch is post-processed as if BP had failed with
 * posterior LLRs d_llr[b][j] (float, fault order, row stride n).  Same syndrome arguments as qd_decode_batch.  This is
 * ldpc's OsdDecoder.decode(syndrome, log_prob_ratios) (osd.hpp), which BpOsdDecoder.decode calls after a failed BP. */
int qd_osd0_batch(qd_decoder *d, const uint8_t *d_det, int64_t det_stride, int64_t det_offset, const uint8_t *d_upd,
                  int64_t upd_stride, int32_t upd_rows, int64_t B, const float *d_llr, uint32_t *d_err_bits,
                  int32_t *d_status, void *stream);

/* Posterior LLRs of the last qd_decode_batch call for shots whose BP did not converge are kept in the decoder's
 * workspace; this copies shot b's (float[n], fault order) to d_out, or returns QD_EINVAL if b converged.
 * Synchronises.  Test/diagnostic hook (ldpc exposes `log_prob_ratios` the same way). */
int qd_decoder_failed_llr(qd_decoder *d, int64_t b, float *d_out, void *stream);

/* Diagnostic: 16 cycle counters the OSD kernels accumulate per phase when the library is built with -DQD_OSD_TIMING
 * (all zero otherwise); reading clears them.  Synchronises. */
int qd_decoder_debug_counters(qd_decoder *d, uint64_t *out16);

/* Two-stream drivers (qd_decode_stage 1 on one stream, 2 on another; no counterpart in the reference, whose shot loop is sequential,
 * sliding_window.py:162-186): call this on the BP stream right after a stage-1 call.  If that batch's post-processing is heavy (OSD-CS / OSD-E, BP-LSD, or
 * OSD-0 over at least three quarters of the batch -- launched only when an earlier call's failure count, read back without waiting, says so, then decided on the
 * device from the batch's own count) the stream is held for
 * `microseconds` (< 0: the default, 50; at most 5000), so that the post-processor started on the other stream at that moment is on the CUs before the next
 * BP kernel fills them; otherwise it costs one empty launch.  Changes no result. */
int qd_decoder_post_head_start(qd_decoder *d, int32_t microseconds, void *stream);

/* Per-kernel device time of this decoder accumulated between calls (milliseconds, HIP events on `stream`):
 * out[0] BP kernel, out[1] OSD kernel, out[2] number of BP launches, out[3] number of OSD launches.
 * enable != 0 turns event recording on.  qd_decoder_profile synchronises on the recorded events. */
int qd_decoder_set_profiling(qd_decoder *d, int32_t enable);
int qd_decoder_profile(qd_decoder *d, double *out, int32_t reset);

/* ---- GF(2) sparse matrices and the window hand-off: replace `window_observable_set[k] @ e % 2` and
 *      `window_update[k] @ e % 2` (sliding_window.py:172,174,183).  CSR, nrows x ncols. */
int qd_spmat_create(int32_t nrows, int32_t ncols, const int32_t *row_ptr, const int32_t *col_idx, int32_t device,
                    qd_spmat **out);
void qd_spmat_destroy(qd_spmat *s);
/* d_out[b * out_stride + r] (one byte per row) = (accumulate ? old : 0) XOR parity(row r of A AND e_b),
 * where e_b are the packed error bits of shot b (err_stride_words words per shot). */
int qd_gf2_spmv_batch(const qd_spmat *A, const uint32_t *d_err_bits, int64_t err_stride_words, int64_t B,
                      uint8_t *d_out, int64_t out_stride, int32_t accumulate, void *stream);

/* Packed error bits -> one byte per fault (the array ldpc's decode() returns; class-level plug-in). */
int qd_unpack_bits(const uint32_t *d_bits, int64_t stride_words, int32_t nbits, int64_t B, uint8_t *d_out,
                   int64_t out_stride, void *stream);

/* Number of shots whose prediction differs from the observable flips on any of k bits: the `pL` numerator of
 * tests/test_sliding_window.py:83.  d_count (int64 on device) is incremented. */
int qd_count_mismatch(const uint8_t *d_pred, const uint8_t *d_obs, int32_t k, int64_t B, int64_t *d_count,
                      void *stream);

/* ---- synthetic input: stands in for stim's detector sampler (simulation.py:23-27), absent here.
 *      e_j ~ Bernoulli(priors_j) (Philox4x32-10, key = seed, counter = (shot0 + b, j / 4)), s = H e, o = L e.
 *      Ht / Lt are the TRANSPOSES as CSR (row j = fault j -> detectors / observables it flips).
 *      d_det: B x det_stride bytes (first m columns written), d_obs: B x obs_stride bytes. */
int qd_sample_dem(const qd_spmat *Ht, const qd_spmat *Lt, const double *priors, uint64_t seed, int64_t shot0,
                  int64_t B, uint8_t *d_det, int64_t det_stride, uint8_t *d_obs, int64_t obs_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QUITS_AMD_H */
