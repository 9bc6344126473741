/* qd_oracle.c -- CPU restatement of the BP-OSD inner decoder and the sliding-window loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under quits_amd/ may import, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and there only as the checker / the reported
 * CPU baseline.  The product path is the HIP library (quits_amd/csrc) and fails loudly without it.
 *
 * PARITY STATUS: **parity unpinned** for the BP-OSD arithmetic.  The reference (mkangquantum/quits 1.1.0) does
 * this arithmetic in a third-party dependency, `ldpc>=2.1.2` (pyproject.toml:35; no exact pin, no lock file),
 * class `ldpc.bposd_decoder.BpOsdDecoder`, which is neither vendored under /root/reference nor installable
 * here (no network).  The reference's own tests hold no golden vectors for it (only 50-shot statistical
 * thresholds, tests/test_sliding_window.py:102-103).  What follows restates ldpc 2.x's published algorithm
 * (src_cpp/bp.hpp, osd.hpp, rref.hpp; Python wrapper bposd_decoder.pyx) anchored on the reference's call sites:
 *   construction  quits/decoder/sliding_window.py:61,69,149,152   (pcm, bp_method, max_iter, schedule,
 *                                                                  osd_method, osd_order, error_rate|channel_probs)
 *   decode        quits/decoder/sliding_window.py:85,95,171,182
 * The orchestration around it (window bookkeeping, syndrome hand-off) IS pinned: it is checked bit for bit
 * against the reference's own Python loop run with a deterministic plug-in decoder (tests/golden/, G5).
 *
 * Where ldpc leaves behaviour unspecified this file makes it deterministic and says so:
 *   - OSD column order: ldpc uses std::sort on the posterior LLRs (ties unspecified); here ties break by
 *     ascending column index (stable).
 *   - OSD pivot row: ldpc's RowReduce picks the lightest candidate row (a sparsity heuristic); the OSD-0
 *     solution does not depend on that choice whenever the syndrome lies in the column space.  Here the pivot
 *     is the lowest-index candidate row; for syndromes outside the column space the result is therefore
 *     defined by this file, not by ldpc.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "oq_math.h"   /* the oracle's own float tanh(x/2) and log((1+c)/(1-c)): same arithmetic as quits_amd/csrc/qd_math.h, no shared code */

#define OQ_PRODUCT_SUM 0
#define OQ_MINIMUM_SUM 1
#define OQ_PARALLEL 0
#define OQ_SERIAL 1
#define OQ_OSD_OFF 0
#define OQ_OSD_0 1
#define OQ_OSD_E 2
#define OQ_OSD_CS 3
#define OQ_LSD_0 4            /* BP-LSD, lsd_order 0 (ldpc.bplsd_decoder.BpLsdDecoder) */
#define OQ_LSD_E 5            /* BP-LSD, lsd_method 'lsd_e',  osd_order = lsd_order */
#define OQ_LSD_CS 6           /* BP-LSD, lsd_method 'lsd_cs', osd_order = lsd_order */
#define OQ_FORM_LDPC_F64 0      /* per-edge messages, double, ldpc's update order              */
#define OQ_FORM_COMPRESSED_F32 1 /* compressed min-sum state, float: bit-exact mirror of the HIP kernel */
#define OQ_FORM_COMPRESSED_F64 2
#define OQ_FORM_LDPC_F32 3       /* per-edge messages, float: bit-exact mirror of the general HIP kernel (bp_general.hip); the serial
                                 * product-sum product is taken as prefix * suffix there (bp_core.inc, bp_serial_ps_presuf) */
#define OQ_MAX_COL_DEG 64

/* Workspace of the BP forms: a few per-thread buffers that grow to the largest request and are kept between calls, so that the
 * CPU baseline bench.py times does not pay a malloc/free pair per array and decode (VERDICT r4 #8a; SURVEY.md 8d "workspace once
 * per worker": bench.py's workers are processes, each owns its slots).  `zero` clears the requested bytes (calloc's contract). */
#define OQ_WS_SLOTS 8
static __thread void *oq_ws_buf[OQ_WS_SLOTS];
static __thread size_t oq_ws_cap[OQ_WS_SLOTS];
static void *oq_ws(int slot, size_t bytes, int zero)
{
    if (bytes == 0) bytes = 1;
    if (oq_ws_cap[slot] < bytes) {
        free(oq_ws_buf[slot]);
        oq_ws_buf[slot] = malloc(bytes);
        oq_ws_cap[slot] = oq_ws_buf[slot] ? bytes : 0;
    }
    if (zero && oq_ws_buf[slot]) memset(oq_ws_buf[slot], 0, bytes);
    return oq_ws_buf[slot];
}

typedef struct {
    int bp_method;           /* OQ_PRODUCT_SUM | OQ_MINIMUM_SUM                                   */
    int schedule;            /* OQ_PARALLEL | OQ_SERIAL                                           */
    int max_iter;            /* 0 -> n (ldpc convention)                                          */
    int osd_method;          /* OQ_OSD_*                                                          */
    int osd_order;
    int form;                /* OQ_FORM_*                                                         */
    double ms_scaling_factor;/* ldpc default 1.0; 0 -> 1 - 2^-it                                  */
} oq_params;

typedef struct {
    int m, n, nnz;
    int *rp, *ci;            /* CSR, columns ascending in a row   */
    int *cp, *ri;            /* CSC, rows ascending in a column   */
    int *csc2csr;            /* CSC edge -> CSR edge              */
    int *csr_pos;            /* CSC edge -> position of that edge inside its row */
    double *prior;
    double *llr0;            /* log((1-p)/p) in double; cast to float by the f32 forms */
    int rank;                /* -1 until computed */
    int llr_frac_bits;       /* -1: llr0 exact; k >= 0: llr0 rounded to the nearest multiple of 2^-k (oq_graph_quantize_llr) */
    int llr_coarse_bits;     /* -1: none; else the grid a shot is re-decoded on when the fine grid's exactness bound trips */
} oq_graph;

/* largest |posterior LLR| any BP call of this process has produced since the last reset (tools/ler_forms.py uses it to
 * show how far the exact-arithmetic range of the quantised forms is from being exhausted) */
static double g_max_abs_llr = 0.0;
double oq_max_abs_llr(int reset) { double v = g_max_abs_llr; if (reset) g_max_abs_llr = 0.0; return v; }
#define OQ_TRACK_LLR(x) do { double ax_ = fabs((double)(x)); if (ax_ > g_max_abs_llr) g_max_abs_llr = ax_; } while (0)
/* Exactness bound of the grid arithmetic (mirrors quits_amd/csrc/bp_kernels.hip): for every fault and iteration
 * S_j = |llr0_j| + sum_k |c2b_k|.  Every partial sum of the posterior and every bit->check message is bounded by S_j, so
 * S_j < 2^(23-k) for all j guarantees that single-precision arithmetic on multiples of 2^-k never rounded.  The largest
 * S of the current BP call is kept here; oq_bposd_decode compares it with the limit. */
static double g_max_s = 0.0;
#define OQ_TRACK_S(x) do { if ((double)(x) > g_max_s) g_max_s = (double)(x); } while (0)

/* the oracle's float functions, exposed so that tests can compare them with libm in double and with the product's */
void oq_math_f32(int kind, const float *x, float *y, int64_t count)
{
    for (int64_t i = 0; i < count; i++) y[i] = kind == 0 ? oq_exp_neg_f32(x[i]) : oq_neg_log_f32(x[i]);
}
void oq_math_ucomb_f32(const float *a, const float *b, float *y, int64_t count)
{
    for (int64_t i = 0; i < count; i++) y[i] = oq_ucomb_f32(a[i], b[i]);
}

/* signed u values: magnitude in [0, 1], the sign bit is the sign of the tanh it stands for */
static inline float oq_u_comb_signed(float a, float b)
{
    const uint32_t sg = (oq_bits(a) ^ oq_bits(b)) & 0x80000000u;
    return oq_float(oq_bits(oq_ucomb_f32(fabsf(a), fabsf(b))) | sg);
}
static inline float oq_u_llr(float z, int syndrome_bit)      /* the check->bit message of a row product z: +-(-log|z|) */
{
    const uint32_t sg = (oq_bits(z) & 0x80000000u) ^ ((uint32_t)(syndrome_bit != 0) << 31);
    return oq_float(oq_bits(oq_neg_log_f32(fabsf(z))) ^ sg);
}

/* ---------------------------------------------------------------------------------------------------------- */
oq_graph *oq_graph_create(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *priors)
{
    oq_graph *g = (oq_graph *)calloc(1, sizeof(oq_graph));
    int nnz = row_ptr[m];
    g->m = m; g->n = n; g->nnz = nnz; g->rank = -1; g->llr_frac_bits = -1; g->llr_coarse_bits = -1;
    g->rp = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    g->ci = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    g->cp = (int *)calloc((size_t)(n + 1), sizeof(int));
    g->ri = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    g->csc2csr = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    g->csr_pos = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    g->prior = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    g->llr0 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    memcpy(g->rp, row_ptr, sizeof(int) * (size_t)(m + 1));
    memcpy(g->ci, col_idx, sizeof(int) * (size_t)nnz);
    for (int e = 0; e < nnz; e++) {
        if (col_idx[e] < 0 || col_idx[e] >= n) { free(g); return NULL; }
        g->cp[col_idx[e] + 1]++;
    }
    for (int j = 0; j < n; j++) g->cp[j + 1] += g->cp[j];
    int *fill = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
    for (int i = 0; i < m; i++)
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) {
            int j = col_idx[e], dst = g->cp[j] + fill[j]++;
            g->ri[dst] = i; g->csc2csr[dst] = e; g->csr_pos[dst] = e - row_ptr[i];
        }
    free(fill);
    for (int j = 0; j < n; j++) {
        if (g->cp[j + 1] - g->cp[j] > OQ_MAX_COL_DEG) { free(g); return NULL; }
        g->prior[j] = priors[j];
        g->llr0[j] = log((1.0 - priors[j]) / priors[j]);
    }
    return g;
}

/* Channel LLRs on a binary grid.  frac_bits = k >= 0 replaces llr0[j] = log((1-p_j)/p_j) by the nearest multiple of
 * 2^-k (ties to even); k < 0 restores the exact doubles.  Min-sum BP with ms_scaling_factor = 1 only ever adds,
 * subtracts, negates and compares such values, so while every intermediate stays below 2^(24-k) in magnitude the float
 * forms -- and below 2^(53-k) the double forms -- perform EXACT arithmetic: all four forms (ldpc's prefix sums or
 * "total minus own", float or double) then return identical bits.  That is the arithmetic of the headline HIP kernel
 * (qd_graph_create does the same rounding); what separates it from ldpc is therefore this one rounding of the inputs. */
void oq_graph_set_coarse_grid(oq_graph *g, int coarse_bits) { g->llr_coarse_bits = coarse_bits; }

/* The rule by which the HIP library picks the grid (qd_decoder_create; restated here so that tests can hold the two
 * against each other): the bound S grows by a few prior-LLRs per iteration (about 6 * max|llr0| * max_iter at the headline),
 * so the grid leaves room for 8 * max|llr0| * max_iter below 2^(23-k); the coarse grid is 16 times wider. */
int oq_grid_bits_rule(double max_abs_llr0, int max_iter, int *coarse)
{
    if (max_iter > 16383) max_iter = 16383;      /* the device keeps the iteration count in 14 status bits (QD_STATUS_ITER_MASK) */
    double need = 8.0 * max_abs_llr0 * (double)(max_iter > 0 ? max_iter : 1);
    int e = 0;
    while (ldexp(1.0, e) < need && e < 40) e++;
    int kr = 23 - e;
    if (kr > 20) kr = 20;
    if (kr < 2) kr = 2;
    /* the fine grid is never coarser than 2^-10: with a large max_iter the rule's grid becomes the coarse (redo) grid */
    int k = kr < 10 ? 10 : kr;
    if (coarse) *coarse = kr >= 10 ? (kr - 4 < 0 ? 0 : kr - 4) : kr;
    return k;
}

void oq_graph_quantize_llr(oq_graph *g, int frac_bits)
{
    g->llr_frac_bits = frac_bits;
    for (int j = 0; j < g->n; j++) {
        double l = log((1.0 - g->prior[j]) / g->prior[j]);
        if (frac_bits >= 0) l = ldexp(nearbyint(ldexp(l, frac_bits)), -frac_bits);
        g->llr0[j] = l;
    }
}

void oq_graph_destroy(oq_graph *g)
{
    if (!g) return;
    free(g->rp); free(g->ci); free(g->cp); free(g->ri); free(g->csc2csr); free(g->csr_pos);
    free(g->prior); free(g->llr0); free(g);
}

/* ---------------------------------------------------------------------------------------------------------- */
#define REAL double
#define SFX _f64
#define REAL_MAX DBL_MAX
#define REAL_ABS fabs
#define REAL_TANH_HALF(x) tanh((x) / 2)
#define REAL_LOG_RATIO(c) log((1 + (c)) / (1 - (c)))
#include "bp_core.inc"
#undef REAL
#undef SFX
#undef REAL_MAX
#undef REAL_ABS
#undef REAL_TANH_HALF
#undef REAL_LOG_RATIO

#define REAL float
#define SFX _f32
#define OQ_SERIAL_PRESUF
#define REAL_MAX FLT_MAX
#define REAL_ABS fabsf
#define OQ_UDOMAIN
#define U_OF_LLR(x) oq_exp_neg_f32(x)
#define U_COMB(a, b) oq_u_comb_signed(a, b)
#define U_LLR(z, sy) oq_u_llr(z, sy)
#include "bp_core.inc"
#undef OQ_UDOMAIN
#undef REAL
#undef SFX
#undef REAL_MAX
#undef REAL_ABS
#undef REAL_TANH_HALF
#undef REAL_LOG_RATIO

/* BP only.  llr_out receives the posterior LLRs as double (exact widening for the float forms).
 * Returns 1 if converged.  ldpc's BpOsdDecoder.decode short-circuits the all-zero syndrome
 * (bposd_decoder.pyx: returns zeros, converge=True); mirrored here with iters = 0. */
int oq_bp_decode(const oq_graph *g, const oq_params *prm_in, const uint8_t *synd,
                 uint8_t *dec, double *llr_out, int *iters)
{
    oq_params prm = *prm_in;
    if (prm.max_iter <= 0) prm.max_iter = g->n;
    int zero = 1;
    for (int i = 0; i < g->m; i++) if (synd[i]) { zero = 0; break; }
    if (zero) {
        memset(dec, 0, (size_t)g->n);
        for (int j = 0; j < g->n; j++) llr_out[j] = g->llr0[j];
        *iters = 0;
        return 1;
    }
    int conv;
    int use_f32 = (prm.form == OQ_FORM_COMPRESSED_F32 || prm.form == OQ_FORM_LDPC_F32);
    int compressed = (prm.form == OQ_FORM_COMPRESSED_F32 || prm.form == OQ_FORM_COMPRESSED_F64);
    if (compressed && (prm.bp_method != OQ_MINIMUM_SUM || prm.schedule != OQ_PARALLEL)) return -1;
    if (use_f32) {
        float *l = (float *)malloc(sizeof(float) * (size_t)g->n);
        if (compressed) conv = bp_minsum_compressed_f32(g, &prm, synd, dec, l, iters);
        else if (prm.schedule == OQ_SERIAL && prm.bp_method == OQ_PRODUCT_SUM) conv = bp_serial_ps_presuf_f32(g, &prm, synd, dec, l, iters);
        else if (prm.schedule == OQ_SERIAL) conv = bp_serial_edge_f32(g, &prm, synd, dec, l, iters);
        else conv = bp_parallel_edge_f32(g, &prm, synd, dec, l, iters);
        for (int j = 0; j < g->n; j++) llr_out[j] = (double)l[j];
        free(l);
    } else {
        if (compressed) conv = bp_minsum_compressed_f64(g, &prm, synd, dec, llr_out, iters);
        else if (prm.schedule == OQ_SERIAL) conv = bp_serial_edge_f64(g, &prm, synd, dec, llr_out, iters);
        else conv = bp_parallel_edge_f64(g, &prm, synd, dec, llr_out, iters);
    }
    return conv;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* OSD (ldpc osd.hpp: OsdDecoder::decode).
 *
 * Column order: ascending posterior LLR (ldpc `soft_decision_col_sort`), ties by ascending index.
 * Elimination: columns are taken in that order; a column whose reduced image has a 1 on a not-yet-pivoted
 * row becomes a pivot column (so the pivot set S is the lexicographically first independent set, exactly what
 * ldpc's `rref(false, true, column_ordering)` finds); OSD-0 solution: e_S = H_S^-1 s, e elsewhere 0.
 *
 * Bookkeeping ("T-form"): rows are only ever modified by adding a pivot row, so the accumulated row
 * transformation T (m x m) is the identity plus columns belonging to pivot rows.  Q[r] holds those bits indexed
 * by pivot ORDER k (bit k <=> T[r][p_k], r != p_k).  (T c)[r] = [r in c] xor parity(Q[r] & {k : p_k in c}).
 * The HIP kernel (quits_amd/csrc/osd_kernels.hip) uses the same bookkeeping.
 *
 * Early stop: once the transformed syndrome is zero on every non-pivot row, s lies in the span of the pivot
 * columns found so far, and because S is an independent set the solution restricted to them is already the
 * final OSD-0 solution (later pivots get coefficient 0).  `stop_early` = 0 disables it (full rank is then
 * reached, as ldpc does); the result is identical, which tests/ verify.
 */
typedef struct { double key; int idx; } oq_sortrec;

static int cmp_sortrec(const void *a, const void *b)
{
    const oq_sortrec *x = (const oq_sortrec *)a, *y = (const oq_sortrec *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

void oq_osd_column_order(int n, const double *llr, int32_t *order)
{
    oq_sortrec *r = (oq_sortrec *)malloc(sizeof(oq_sortrec) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; j++) { r[j].key = llr[j]; r[j].idx = j; }
    qsort(r, (size_t)n, sizeof(oq_sortrec), cmp_sortrec);
    for (int j = 0; j < n; j++) order[j] = r[j].idx;
    free(r);
}

typedef struct {
    int mw;                 /* words per Q row = ceil(m/64) (rank <= m) */
    uint64_t *Q;            /* m x mw                                    */
    uint8_t *sp;            /* transformed syndrome, per row             */
    int *rowpiv;            /* row -> pivot order or -1                  */
    int *prow, *pcol;       /* pivot order -> row / column               */
    int npiv;
    int ncols_examined;
} oq_elim;

static void elim_free(oq_elim *E) { free(E->Q); free(E->sp); free(E->rowpiv); free(E->prow); free(E->pcol); }

/* Eliminates in `order`; stops at rank `max_rank` (or when columns run out), or early when the residual syndrome
 * vanishes (stop_early).  Gauss-Jordan: every row, pivoted or not, is updated, so on exit e[pcol[k]] = sp[prow[k]]. */
static void elim_run(const oq_graph *g, const int32_t *order, const uint8_t *synd, int stop_early, int max_rank,
                     oq_elim *E)
{
    const int m = g->m, n = g->n;
    const int mw = (m + 63) / 64;
    E->mw = mw;
    E->Q = (uint64_t *)calloc((size_t)m * (size_t)mw + 1, sizeof(uint64_t));
    E->sp = (uint8_t *)malloc((size_t)m + 1);
    E->rowpiv = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    E->prow = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    E->pcol = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    E->npiv = 0; E->ncols_examined = 0;
    uint8_t *t = (uint8_t *)malloc((size_t)m + 1);
    int resid = 0;
    for (int r = 0; r < m; r++) { E->sp[r] = synd[r] & 1; E->rowpiv[r] = -1; resid += E->sp[r]; }
    for (int c = 0; c < n; c++) {
        if (E->npiv >= max_rank) break;
        if (stop_early && resid == 0) break;
        int col = order[c];
        E->ncols_examined = c + 1;
        /* t = T * column */
        memset(t, 0, (size_t)m);
        int nmask = 0, maskk[OQ_MAX_COL_DEG];
        for (int e = g->cp[col]; e < g->cp[col + 1]; e++) {
            int r = g->ri[e];
            t[r] ^= 1;
            if (E->rowpiv[r] >= 0) maskk[nmask++] = E->rowpiv[r];
        }
        if (nmask)
            for (int r = 0; r < m; r++) {
                const uint64_t *q = E->Q + (size_t)r * mw;
                int b = 0;
                for (int x = 0; x < nmask; x++) b ^= (int)((q[maskk[x] >> 6] >> (maskk[x] & 63)) & 1);
                t[r] ^= (uint8_t)b;
            }
        int p = -1;
        for (int r = 0; r < m; r++) if (t[r] && E->rowpiv[r] < 0) { p = r; break; }
        if (p < 0) continue;                       /* dependent on earlier pivot columns */
        int k = E->npiv++;
        const uint64_t *qp = E->Q + (size_t)p * mw;
        int kw = k >> 6; uint64_t kb = 1ull << (k & 63);
        int nw = kw + 1;                           /* words that can be non-zero so far */
        for (int r = 0; r < m; r++) {
            if (!t[r] || r == p) continue;
            uint64_t *q = E->Q + (size_t)r * mw;
            for (int w = 0; w < nw; w++) q[w] ^= qp[w];
            q[kw] ^= kb;
            if (E->sp[p]) {
                if (E->rowpiv[r] < 0) resid += E->sp[r] ? -1 : 1;
                E->sp[r] ^= 1;
            }
        }
        if (E->sp[p]) resid -= 1;                  /* p leaves the non-pivot set */
        E->rowpiv[p] = k; E->prow[k] = p; E->pcol[k] = col;
    }
    free(t);
}

int oq_gf2_rank(oq_graph *g)
{
    if (g->rank >= 0) return g->rank;
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->n > 0 ? g->n : 1));
    uint8_t *z = (uint8_t *)calloc((size_t)g->m + 1, 1);
    for (int j = 0; j < g->n; j++) order[j] = j;
    oq_elim E;
    elim_run(g, order, z, 0, g->m, &E);
    g->rank = E.npiv;
    elim_free(&E); free(order); free(z);
    return g->rank;
}

/* OSD-0.  stats (optional, int[4]): pivots used, columns examined, residual-nonzero flag (syndrome not in the
 * column space), 0. */
int oq_osd0(oq_graph *g, const uint8_t *synd, const double *llr, int stop_early, uint8_t *err, int32_t *stats)
{
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->n > 0 ? g->n : 1));
    oq_osd_column_order(g->n, llr, order);
    oq_elim E;
    elim_run(g, order, synd, stop_early, stop_early ? g->m : oq_gf2_rank(g), &E);
    memset(err, 0, (size_t)g->n);
    for (int k = 0; k < E.npiv; k++) err[E.pcol[k]] = E.sp[E.prow[k]];
    int inconsistent = 0;
    for (int r = 0; r < g->m; r++) if (E.rowpiv[r] < 0 && E.sp[r]) inconsistent = 1;
    if (stats) { stats[0] = E.npiv; stats[1] = E.ncols_examined; stats[2] = inconsistent; stats[3] = 0; }
    elim_free(&E); free(order);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* BP-LSD: localized statistics decoding after a failed BP (Hillmann, Berent, Quintavalle, Eisert, Wille, Roffe 2024;
 * ldpc 2.x src_cpp/lsd.hpp, LsdDecoder::lsd_decode, as the reference reaches it through ldpc.bplsd_decoder.BpLsdDecoder:
 * quits/decoder/bplsd.py:5,51,86).  PARITY UNPINNED like the rest of the ldpc arithmetic (no ldpc here): restated from the
 * published algorithm with ldpc's defaults bits_per_step = 1, lsd_order = 0 (LSD-0), deterministic where ldpc is not:
 *
 *   - every unsatisfied check seeds a cluster (id = its index);
 *   - while invalid clusters exist: the clusters that are invalid at the start of the round grow by ONE fault each, in order
 *     of (number of faults, id) ascending (ldpc: std::sort by bit_nodes.size(), ties unspecified); a cluster that became
 *     valid or was absorbed earlier in the round is skipped;
 *   - growth: among the faults not yet in any cluster that touch a check of the cluster, the one with the LOWEST posterior
 *     LLR joins (ties: lowest index; ldpc sorts its candidate list with std::sort); all its checks join the cluster, and a
 *     cluster owning one of them is absorbed (its faults keep their order);
 *   - a cluster is valid when its part of the syndrome lies in the span of its faults' columns, decided by on-the-fly
 *     elimination: each new column is reduced against the pivots found so far; clusters own disjoint checks, so one
 *     elimination over all rows serves every cluster (T-form bookkeeping as in elim_run; pivot row = lowest candidate);
 *   - the correction is, cluster by cluster, the solution on the pivot columns in order of addition (what ldpc's PLU
 *     lu_solve returns): err[pcol[k]] = transformed syndrome at pivot row k.
 * A cluster whose checks have no fault left to add while still invalid can never become valid (syndrome outside the column
 * space): it is retired and the shot flagged.  stats: pivots, faults added, inconsistent flag, growth rounds. */
static void elim_init(const oq_graph *g, const uint8_t *synd, oq_elim *E)
{
    const int m = g->m, mw = (m + 63) / 64;
    E->mw = mw;
    E->Q = (uint64_t *)calloc((size_t)m * (size_t)mw + 1, sizeof(uint64_t));
    E->sp = (uint8_t *)malloc((size_t)m + 1);
    E->rowpiv = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    E->prow = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    E->pcol = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    E->npiv = 0; E->ncols_examined = 0;
    for (int r = 0; r < m; r++) { E->sp[r] = synd[r] & 1; E->rowpiv[r] = -1; }
}

/* one column through the elimination; returns 1 if it became a pivot column */
static int elim_add_column(const oq_graph *g, oq_elim *E, int col, uint8_t *t /* m bytes of scratch */)
{
    const int m = g->m, mw = E->mw;
    memset(t, 0, (size_t)m);
    int nmask = 0, maskk[OQ_MAX_COL_DEG];
    for (int e = g->cp[col]; e < g->cp[col + 1]; e++) {
        int r = g->ri[e];
        t[r] ^= 1;
        if (E->rowpiv[r] >= 0) maskk[nmask++] = E->rowpiv[r];
    }
    if (nmask)
        for (int r = 0; r < m; r++) {
            const uint64_t *q = E->Q + (size_t)r * mw;
            int b = 0;
            for (int x = 0; x < nmask; x++) b ^= (int)((q[maskk[x] >> 6] >> (maskk[x] & 63)) & 1);
            t[r] ^= (uint8_t)b;
        }
    int p = -1;
    for (int r = 0; r < m; r++) if (t[r] && E->rowpiv[r] < 0) { p = r; break; }
    if (p < 0) return 0;
    int k = E->npiv++;
    const uint64_t *qp = E->Q + (size_t)p * mw;
    int kw = k >> 6; uint64_t kb = 1ull << (k & 63);
    for (int r = 0; r < m; r++) {
        if (!t[r] || r == p) continue;
        uint64_t *q = E->Q + (size_t)r * mw;
        for (int w = 0; w <= kw; w++) q[w] ^= qp[w];
        q[kw] ^= kb;
        if (E->sp[p]) E->sp[r] ^= 1;
    }
    E->rowpiv[p] = k; E->prow[k] = p; E->pcol[k] = col;
    return 1;
}

uint32_t oq_fixed_weight(double p);

/* Higher-order LSD (lsd_method 'lsd_cs' / 'lsd_e' with lsd_order > 0; every BP-LSD call the reference itself makes passes
 * lsd_order = 1: /root/reference/tests/test_decoders.py:136, doc/05_decoder_variants.ipynb cell 9).  Restated from the published
 * algorithm (Hillmann et al. 2024, "higher-order reprocessing" applied to each cluster's own factorisation; ldpc lsd.hpp
 * LsdDecoder::apply_lsdw), PARITY UNPINNED, with these choices where the paper / ldpc leave room:
 *   1. growth stage: once every cluster is valid, each active cluster, in ascending id, whose number of non-pivot faults
 *      ("dimension") is below lsd_order grows by further faults -- same growth rule, merges included -- until the dimension
 *      reaches lsd_order or lsd_order faults have been added (ldpc: "the number of bits added is limited to be at most the
 *      size of the lsd order");
 *   2. sweep: per valid cluster, ascending id, OSD-CS / OSD-E of that order on the cluster's own faults: its non-pivot faults
 *      sorted by (posterior LLR, index) ascending are the candidate positions (combination sweep: every single one, plus the
 *      pairs among the first min(order, k_c); exhaustive: the 2^min(order, k_c) - 1 patterns on the first ones), the pivot
 *      coefficients follow from the cluster's elimination, cost = sum of log(1 / p_j) over the cluster's flipped faults (the
 *      cost osd.hpp uses), strict '<' keeps the earliest minimum, the LSD-0 solution first.
 * Clusters own disjoint checks, so the shared elimination is block diagonal and each cluster's sweep only reads its block.
 * `fixed` != 0: integer costs round(log(1/p) * 2^18), the device's arithmetic (exact sums). */
typedef struct {
    oq_graph *g;
    const double *llr;
    oq_elim E;
    uint8_t *t, *added, *state, *ispiv;
    int *owner, *nbits, *addlist;
    int nadded, inconsistent;
} oq_lsd_ctx;

/* one growth step of cluster c; 0 if the cluster has no fault left to add (an invalid cluster is then retired: it can
 * never become valid; a valid one, in the growth stage of the higher orders, just stays as it is) */
static int lsd_grow(oq_lsd_ctx *x, int c, int retire)
{
    const oq_graph *g = x->g;
    const int m = g->m;
    int best = -1;
    for (int i = 0; i < m; i++) {
        if (x->owner[i] != c) continue;
        for (int e = g->rp[i]; e < g->rp[i + 1]; e++) {
            int j = g->ci[e];
            if (x->added[j]) continue;
            if (best < 0 || x->llr[j] < x->llr[best] || (x->llr[j] == x->llr[best] && j < best)) best = j;
        }
    }
    if (best < 0) {
        if (retire) { x->state[c] = 3; x->inconsistent = 1; }
        return 0;
    }
    x->added[best] = 1; x->addlist[x->nadded++] = best; x->nbits[c]++;
    for (int e = g->cp[best]; e < g->cp[best + 1]; e++) {
        int i = g->ri[e], d = x->owner[i];
        if (d == c) continue;
        if (d < 0) { x->owner[i] = c; continue; }
        for (int r = 0; r < m; r++) if (x->owner[r] == d) x->owner[r] = c;     /* absorb cluster d */
        x->nbits[c] += x->nbits[d]; x->state[d] = 3;
    }
    x->ispiv[best] = (uint8_t)elim_add_column(g, &x->E, best, x->t);
    int bad = 0;
    for (int r = 0; r < m; r++) if (x->owner[r] == c && x->E.rowpiv[r] < 0 && x->E.sp[r]) { bad = 1; break; }
    x->state[c] = bad ? 1 : 2;
    return 1;
}

static int lsd_cluster_dimension(const oq_lsd_ctx *x, int c)
{
    int piv = 0;
    for (int r = 0; r < x->g->m; r++) if (x->owner[r] == c && x->E.rowpiv[r] >= 0) piv++;
    return x->nbits[c] - piv;
}

/* stats: pivots, faults added, inconsistent flag, growth rounds; [4..6] (if order > 0): faults added by the growth stage,
 * clusters swept, clusters whose LSD-0 solution was replaced */
int oq_lsd(oq_graph *g, const uint8_t *synd, const double *llr, int lsd_method, int lsd_order, int fixed, uint8_t *err,
           int32_t *stats)
{
    const int m = g->m, n = g->n;
    oq_lsd_ctx x;
    x.g = g; x.llr = llr; x.nadded = 0; x.inconsistent = 0;
    elim_init(g, synd, &x.E);
    x.t = (uint8_t *)malloc((size_t)m + 1);
    x.added = (uint8_t *)calloc((size_t)n + 1, 1);
    x.ispiv = (uint8_t *)calloc((size_t)n + 1, 1);
    x.owner = (int *)malloc(sizeof(int) * (size_t)(m + 1));      /* check -> cluster id, -1 = free */
    x.nbits = (int *)calloc((size_t)m + 1, sizeof(int));          /* per cluster id */
    x.state = (uint8_t *)calloc((size_t)m + 1, 1);                /* per cluster id: 0 none, 1 active+invalid, 2 active+valid, 3 gone */
    x.addlist = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int *round = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    int rounds = 0, grown = 0, swept = 0, replaced = 0;
    for (int i = 0; i < m; i++) { x.owner[i] = (synd[i] & 1) ? i : -1; x.state[i] = (synd[i] & 1) ? 1 : 0; }
    for (;;) {
        int nr = 0;
        for (int c = 0; c < m; c++) if (x.state[c] == 1) round[nr++] = c;
        if (!nr) break;
        rounds++;
        /* (size, id) ascending, sizes as they are at the start of the round: insertion sort, the lists are short */
        for (int a = 1; a < nr; a++) {
            int c = round[a], b = a - 1;
            while (b >= 0 && (x.nbits[round[b]] > x.nbits[c])) { round[b + 1] = round[b]; b--; }
            round[b + 1] = c;
        }
        for (int q = 0; q < nr; q++)
            if (x.state[round[q]] == 1) lsd_grow(&x, round[q], 1);
    }
    const int order = (lsd_method == OQ_LSD_CS || lsd_method == OQ_LSD_E) ? lsd_order : 0;
    if (order > 0) {
        /* 1. growth stage */
        for (int c = 0; c < m; c++) {
            if (x.state[c] != 2) continue;
            int cnt = 0;
            while (x.state[c] == 2 && lsd_cluster_dimension(&x, c) < order && cnt < order) {
                if (!lsd_grow(&x, c, 0)) break;
                cnt++; grown++;
            }
        }
    }
    memset(err, 0, (size_t)n);
    for (int k = 0; k < x.E.npiv; k++) err[x.E.pcol[k]] = x.E.sp[x.E.prow[k]];
    if (order > 0) {
        /* 2. sweep, cluster by cluster */
        double *wd = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        for (int j = 0; j < n; j++) wd[j] = fixed ? (double)oq_fixed_weight(g->prior[j]) : log(1.0 / g->prior[j]);
        oq_sortrec *np = (oq_sortrec *)malloc(sizeof(oq_sortrec) * (size_t)(x.nadded > 0 ? x.nadded : 1));
        int *cpk = (int *)malloc(sizeof(int) * (size_t)(m + 1));         /* the cluster's pivots (orders) */
        uint8_t *flip = (uint8_t *)malloc((size_t)m + 1);                /* per pivot order of the cluster: coefficient flipped by the candidate */
        uint8_t *bestflip = (uint8_t *)malloc((size_t)m + 1);
        for (int c = 0; c < m; c++) {
            if (x.state[c] != 2) continue;
            int kk = 0, npk = 0;
            for (int a = 0; a < x.nadded; a++) {
                int j = x.addlist[a];
                if (x.owner[g->ri[g->cp[j]]] != c || x.ispiv[j]) continue;   /* all checks of an added fault are in its cluster */
                np[kk].key = llr[j]; np[kk].idx = j; kk++;
            }
            if (!kk) continue;
            swept++;
            qsort(np, (size_t)kk, sizeof(oq_sortrec), cmp_sortrec);
            for (int k = 0; k < x.E.npiv; k++) if (x.owner[x.E.prow[k]] == c) cpk[npk++] = k;
            double base = 0;
            for (int q = 0; q < npk; q++) if (x.E.sp[x.E.prow[cpk[q]]]) base += wd[x.E.pcol[cpk[q]]];
            int w = order > kk ? kk : order;
            long ncand = lsd_method == OQ_LSD_E ? (1L << w) - 1 : (long)kk + (long)w * (w - 1) / 2;
            double best = base;
            int bsel[64], bnsel = 0;
            for (long ic = 0; ic < ncand; ic++) {
                int sel[64], nsel = 0;
                if (lsd_method == OQ_LSD_E) {
                    long pat = ic + 1;
                    for (int b = 0; b < w; b++) if ((pat >> b) & 1) sel[nsel++] = b;
                } else if (ic < kk) {
                    sel[nsel++] = (int)ic;
                } else {
                    long q = ic - kk; int a = 0;
                    while (q >= w - 1 - a) { q -= w - 1 - a; a++; }
                    sel[0] = a; sel[1] = a + 1 + (int)q; nsel = 2;
                }
                memset(flip, 0, (size_t)npk + 1);
                double wgt = 0;
                for (int s = 0; s < nsel; s++) {
                    int col = np[sel[s]].idx;
                    wgt += wd[col];
                    int nmask = 0, maskk[OQ_MAX_COL_DEG];
                    memset(x.t, 0, (size_t)m);
                    for (int e = g->cp[col]; e < g->cp[col + 1]; e++) {
                        int r = g->ri[e];
                        x.t[r] ^= 1;
                        if (x.E.rowpiv[r] >= 0) maskk[nmask++] = x.E.rowpiv[r];
                    }
                    for (int q = 0; q < npk; q++) {
                        int r = x.E.prow[cpk[q]];
                        const uint64_t *Q = x.E.Q + (size_t)r * x.E.mw;
                        int b = x.t[r];
                        for (int z = 0; z < nmask; z++) b ^= (int)((Q[maskk[z] >> 6] >> (maskk[z] & 63)) & 1);
                        flip[q] ^= (uint8_t)b;
                    }
                }
                for (int q = 0; q < npk; q++)
                    if (x.E.sp[x.E.prow[cpk[q]]] ^ flip[q]) wgt += wd[x.E.pcol[cpk[q]]];
                if (wgt < best) { best = wgt; bnsel = nsel; memcpy(bsel, sel, sizeof(int) * (size_t)nsel); memcpy(bestflip, flip, (size_t)npk); }
            }
            if (bnsel) {
                replaced++;
                for (int q = 0; q < npk; q++) if (bestflip[q]) err[x.E.pcol[cpk[q]]] ^= 1;
                for (int s = 0; s < bnsel; s++) err[np[bsel[s]].idx] = 1;
            }
        }
        free(wd); free(np); free(cpk); free(flip); free(bestflip);
    }
    if (stats) {
        stats[0] = x.E.npiv; stats[1] = x.nadded; stats[2] = x.inconsistent; stats[3] = rounds;
        if (order > 0) { stats[4] = grown; stats[5] = swept; stats[6] = replaced; }
    }
    elim_free(&x.E); free(x.t); free(x.added); free(x.ispiv); free(x.owner); free(x.nbits); free(x.state); free(x.addlist); free(round);
    return 0;
}

int oq_lsd0(oq_graph *g, const uint8_t *synd, const double *llr, uint8_t *err, int32_t *stats)
{
    return oq_lsd(g, synd, llr, OQ_LSD_0, 0, 0, err, stats);
}

/* Candidate cost.  ldpc sums log(1/p_j) in double (osd.hpp).  `fixed` != 0 switches to the integer weights the HIP
 * kernel uses -- round(log(1/p_j) * 2^18) -- so that sums are exact and independent of summation order; the two only
 * differ on candidates whose costs tie to within ~1e-5. */
uint32_t oq_fixed_weight(double p)
{
    double w = log(1.0 / p) * 262144.0;
    if (w < 0) w = 0;
    if (w > 4294967295.0) w = 4294967295.0;
    return (uint32_t)llround(w);
}

/* OSD-CS / OSD-E of order `osd_order` (osd.hpp: osd_setup + decode).  Candidate strings live on the k = n - rank
 * non-pivot columns taken in the sorted order; cost of a candidate = sum over its 1s of log(1/p_j) (channel
 * probabilities, not Hamming weight); strict '<' keeps the earliest minimum, OSD-0 first. */
static int osd_w_impl(oq_graph *g, const uint8_t *synd, const double *llr, int osd_method, int osd_order, int fixed,
                      uint8_t *err, int32_t *stats)
{
    const int m = g->m, n = g->n;
    int rank = oq_gf2_rank(g);
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    oq_osd_column_order(n, llr, order);
    oq_elim E;
    elim_run(g, order, synd, 0, rank, &E);
    memset(err, 0, (size_t)n);
    for (int k = 0; k < E.npiv; k++) err[E.pcol[k]] = E.sp[E.prow[k]];
    if (stats) { stats[0] = E.npiv; stats[1] = E.ncols_examined; stats[2] = 0; stats[3] = 0; }
    if (osd_order <= 0 || osd_method == OQ_OSD_0 || osd_method == OQ_OSD_OFF) { elim_free(&E); free(order); return 0; }
    /* non-pivot columns in sorted order */
    uint8_t *ispiv = (uint8_t *)calloc((size_t)n + 1, 1);
    for (int k = 0; k < E.npiv; k++) ispiv[E.pcol[k]] = 1;
    int kk = n - E.npiv;
    int *npc = (int *)malloc(sizeof(int) * (size_t)(kk > 0 ? kk : 1));
    for (int c = 0, q = 0; c < n; c++) if (!ispiv[order[c]]) npc[q++] = order[c];
    double *wd = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; j++) wd[j] = fixed ? (double)oq_fixed_weight(g->prior[j]) : log(1.0 / g->prior[j]);
    /* (T c)[pivot rows] for a non-pivot column c, via the final Q: flipping column c on changes the pivot
     * coefficients by exactly that vector.  With `fixed` the costs are integers < 2^53, exact in double. */
    double best = 0;
    for (int j = 0; j < n; j++) if (err[j]) best += wd[j];
    uint8_t *cand = (uint8_t *)malloc((size_t)n + 1);
    uint8_t *tc = (uint8_t *)malloc((size_t)m + 1);
    uint8_t *base = (uint8_t *)malloc((size_t)n + 1);
    memcpy(base, err, (size_t)n);
    long ncand;
    int w = osd_order > kk ? kk : osd_order;
    if (osd_method == OQ_OSD_E) ncand = (1L << w) - 1;
    else ncand = (long)kk + (long)w * (w - 1) / 2;
    long best_ic = -1;
    for (long ic = 0; ic < ncand; ic++) {
        int sel[64], nsel = 0;
        if (osd_method == OQ_OSD_E) {
            long pat = ic + 1;
            for (int b = 0; b < w; b++) if ((pat >> b) & 1) sel[nsel++] = b;
        } else if (ic < kk) {
            sel[nsel++] = (int)ic;
        } else {
            long q = ic - kk; int a = 0;
            while (q >= w - 1 - a) { q -= w - 1 - a; a++; }
            sel[0] = a; sel[1] = a + 1 + (int)q; nsel = 2;
        }
        memcpy(cand, base, (size_t)n);
        for (int s = 0; s < nsel; s++) {
            int col = npc[sel[s]];
            memset(tc, 0, (size_t)m);
            int nmask = 0, maskk[OQ_MAX_COL_DEG];
            for (int e = g->cp[col]; e < g->cp[col + 1]; e++) {
                int r = g->ri[e];
                tc[r] ^= 1;
                if (E.rowpiv[r] >= 0) maskk[nmask++] = E.rowpiv[r];
            }
            for (int k = 0; k < E.npiv; k++) {
                int r = E.prow[k];
                const uint64_t *q = E.Q + (size_t)r * E.mw;
                int b = tc[r];
                for (int x = 0; x < nmask; x++) b ^= (int)((q[maskk[x] >> 6] >> (maskk[x] & 63)) & 1);
                if (b) cand[E.pcol[k]] ^= 1;
            }
            cand[col] = 1;
        }
        double wgt = 0;
        for (int j = 0; j < n; j++) if (cand[j]) wgt += wd[j];
        if (wgt < best) { best = wgt; best_ic = ic; memcpy(err, cand, (size_t)n); }
    }
    if (stats) { stats[2] = (int32_t)(best_ic + 1); stats[3] = (int32_t)ncand; }
    free(cand); free(tc); free(base); free(npc); free(ispiv); free(wd); elim_free(&E); free(order);
    return 0;
}

int oq_osd_w(oq_graph *g, const uint8_t *synd, const double *llr, int osd_method, int osd_order, uint8_t *err)
{
    return osd_w_impl(g, synd, llr, osd_method, osd_order, 0, err, NULL);
}

/* stats: pivots, columns examined, 1 + index of the winning candidate (0 = OSD-0 kept), number of candidates */
int oq_osd_w_fixed(oq_graph *g, const uint8_t *synd, const double *llr, int osd_method, int osd_order, uint8_t *err,
                   int32_t *stats)
{
    return osd_w_impl(g, synd, llr, osd_method, osd_order, 1, err, stats);
}

/* BpOsdDecoder.decode (bposd_decoder.pyx): BP; if converged return the BP decision, else OSD on the posteriors.
 * flags_out (optional, int[4]): converged, iterations, osd pivots, osd inconsistent. */
static int g_last_grid = -1, g_last_uncertified = 0;
/* grid the last oq_bposd_decode call ended on (-1: exact LLRs), and whether even that grid's bound tripped */
void oq_last_grid(int32_t *out2) { out2[0] = g_last_grid; out2[1] = g_last_uncertified; }

int oq_bposd_decode(oq_graph *g, const oq_params *prm, const uint8_t *synd, uint8_t *err, int32_t *flags_out)
{
    double *llr = (double *)malloc(sizeof(double) * (size_t)(g->n > 0 ? g->n : 1));
    int iters = 0;
    int32_t st[4] = {0, 0, 0, 0};
    g_max_s = 0.0;
    int conv = oq_bp_decode(g, prm, synd, err, llr, &iters);
    if (conv < 0) { free(llr); return -1; }
    g_last_grid = g->llr_frac_bits; g_last_uncertified = 0;
    if (g->llr_frac_bits >= 0 && g->llr_coarse_bits >= 0 && g_max_s >= ldexp(1.0, 23 - g->llr_frac_bits)) {
        /* the device re-decodes such a shot on the coarse grid (bp_kernels.hip, redo pass) */
        const int fine = g->llr_frac_bits;
        oq_graph_quantize_llr(g, g->llr_coarse_bits);
        g_max_s = 0.0;
        conv = oq_bp_decode(g, prm, synd, err, llr, &iters);
        g_last_grid = g->llr_coarse_bits;
        g_last_uncertified = g_max_s >= ldexp(1.0, 23 - g->llr_coarse_bits);
        oq_graph_quantize_llr(g, fine);
    }
    if (!conv && prm->osd_method != OQ_OSD_OFF) {
        const int dev_costs = prm->form == OQ_FORM_COMPRESSED_F32 || prm->form == OQ_FORM_LDPC_F32 || g->llr_frac_bits >= 0;
        if (prm->osd_method == OQ_LSD_0 || prm->osd_method == OQ_LSD_E || prm->osd_method == OQ_LSD_CS) {
            int32_t st7[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            oq_lsd(g, synd, llr, prm->osd_method, prm->osd_order, dev_costs, err, st7);
            st[0] = st7[0]; st[2] = st7[2];
        }
        else if (prm->osd_method == OQ_OSD_0 || prm->osd_order == 0) oq_osd0(g, synd, llr, 1, err, st);
        else osd_w_impl(g, synd, llr, prm->osd_method, prm->osd_order,
                        prm->form == OQ_FORM_COMPRESSED_F32 || prm->form == OQ_FORM_LDPC_F32 || g->llr_frac_bits >= 0
                        /* the device's arithmetic (float forms, or any form on the LLR grid): integer candidate costs */, err, NULL);
    }
    if (flags_out) { flags_out[0] = conv; flags_out[1] = iters; flags_out[2] = st[0]; flags_out[3] = st[2]; }
    free(llr);
    return 0;
}

int oq_bposd_decode_batch2(oq_graph *g, const oq_params *prm, const uint8_t *synd, int64_t B, uint8_t *err,
                           int32_t *flags /* B x 4 or NULL */, int32_t *grid /* B x 2 or NULL: grid used, bound tripped */)
{
    for (int64_t b = 0; b < B; b++) {
        int rc = oq_bposd_decode(g, prm, synd + b * g->m, err + b * g->n, flags ? flags + 4 * b : NULL);
        if (rc) return rc;
        if (grid) oq_last_grid(grid + 2 * b);
    }
    return 0;
}

int oq_bposd_decode_batch(oq_graph *g, const oq_params *prm, const uint8_t *synd, int64_t B, uint8_t *err,
                          int32_t *flags /* B x 4 or NULL */)
{
    return oq_bposd_decode_batch2(g, prm, synd, B, err, flags, NULL);
}

/* ---------------------------------------------------------------------------------------------------------- */
/* Sliding-window loop, circuit-level (quits/decoder/sliding_window.py:162-186), for a batch of shots.
 * windows[k] decodes rows [row0[k], row0[k] + g_k->m) of the detector record; L[k] / U[k] are CSR matrices over
 * the first ncommit[k] columns of window k (window_observable_set / window_update, base.py:170,177).
 */
typedef struct {
    int nrows, ncols;
    int *rp, *ci;
} oq_csr;

oq_csr *oq_csr_create(int nrows, int ncols, const int32_t *rp, const int32_t *ci)
{
    oq_csr *c = (oq_csr *)malloc(sizeof(oq_csr));
    c->nrows = nrows; c->ncols = ncols;
    c->rp = (int *)malloc(sizeof(int) * (size_t)(nrows + 1));
    c->ci = (int *)malloc(sizeof(int) * (size_t)(rp[nrows] > 0 ? rp[nrows] : 1));
    memcpy(c->rp, rp, sizeof(int) * (size_t)(nrows + 1));
    memcpy(c->ci, ci, sizeof(int) * (size_t)rp[nrows]);
    return c;
}
void oq_csr_destroy(oq_csr *c) { if (c) { free(c->rp); free(c->ci); free(c); } }

int oq_sliding_window_decode(int nwin, oq_graph **gs, oq_csr **Ls, oq_csr **Us, const int32_t *row0,
                             int nz, int ndet, int nobs, const oq_params *prm,
                             const uint8_t *samples /* B x ndet */, int64_t B,
                             uint8_t *pred /* B x nobs */, int64_t *counters /* [4]: bp-converged windows, osd windows, total iters, inconsistent */)
{
    int maxm = 0, maxn = 0;
    for (int k = 0; k < nwin; k++) { if (gs[k]->m > maxm) maxm = gs[k]->m; if (gs[k]->n > maxn) maxn = gs[k]->n; }
    uint8_t *s = (uint8_t *)malloc((size_t)maxm + 1), *e = (uint8_t *)malloc((size_t)maxn + 1);
    uint8_t *upd = (uint8_t *)malloc((size_t)nz + 1);
    int32_t fl[4];
    if (counters) memset(counters, 0, sizeof(int64_t) * 4);
    for (int64_t b = 0; b < B; b++) {
        const uint8_t *row = samples + b * ndet;
        uint8_t *acc = pred + b * nobs;
        memset(acc, 0, (size_t)nobs);
        memset(upd, 0, (size_t)nz);
        for (int k = 0; k < nwin; k++) {
            oq_graph *g = gs[k];
            for (int i = 0; i < g->m; i++) s[i] = row[row0[k] + i] & 1;
            for (int i = 0; i < nz && i < g->m; i++) s[i] ^= upd[i];
            if (oq_bposd_decode(g, prm, s, e, fl)) { free(s); free(e); free(upd); return -1; }
            if (counters) { counters[0] += fl[0]; counters[1] += !fl[0]; counters[2] += fl[1]; counters[3] += fl[3]; }
            const oq_csr *L = Ls[k];
            for (int o = 0; o < L->nrows; o++) {
                int p = 0;
                for (int x = L->rp[o]; x < L->rp[o + 1]; x++) p ^= e[L->ci[x]];
                acc[o] ^= (uint8_t)p;
            }
            if (k + 1 < nwin) {
                const oq_csr *U = Us[k];
                for (int o = 0; o < U->nrows; o++) {
                    int p = 0;
                    for (int x = U->rp[o]; x < U->rp[o + 1]; x++) p ^= e[U->ci[x]];
                    upd[o] = (uint8_t)p;
                }
            }
        }
    }
    free(s); free(e); free(upd);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* DEM sampler: stands in for stim's detector sampler (quits/simulation.py:23-27), which is absent here.
 * e_j ~ Bernoulli(p_j) independently, s = H e, o = L e  (mod 2).  Randomness: Philox4x32-10, key = (seed lo, hi),
 * counter = (shot lo, shot hi, j / 4, 0); word (j & 3) of the output decides fault j:  fires iff word < thr_j,
 * thr_j = floor(p_j * 2^32).  Pure integer work, so the HIP sampler (quits_amd/csrc/sampler.hip) is bit-exact. */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oq_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
    philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

uint32_t oq_prob_threshold(double p)
{
    double t = floor(p * 4294967296.0);
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

/* H given in CSC (cp, ri) over n faults; Lobs in CSC too (lcp, lri).  Outputs are byte arrays. */
int oq_sample_dem(int m, int n, int nobs, const int32_t *cp, const int32_t *ri, const int32_t *lcp,
                  const int32_t *lri, const double *priors, uint64_t seed, int64_t shot0, int64_t B,
                  uint8_t *synd /* B x m */, uint8_t *obs /* B x nobs */, int32_t *nfaults /* B or NULL */)
{
    uint32_t *thr = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; j++) thr[j] = oq_prob_threshold(priors[j]);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t b = 0; b < B; b++) {
        uint64_t shot = (uint64_t)(shot0 + b);
        uint8_t *s = synd + b * m, *o = obs + b * nobs;
        memset(s, 0, (size_t)m); memset(o, 0, (size_t)nobs);
        int cnt = 0;
        for (int j4 = 0; j4 < n; j4 += 4) {
            uint32_t r[4];
            philox4x32_10((uint32_t)shot, (uint32_t)(shot >> 32), (uint32_t)(j4 >> 2), 0u, k0, k1, r);
            for (int x = 0; x < 4 && j4 + x < n; x++) {
                int j = j4 + x;
                if (r[x] < thr[j]) {
                    cnt++;
                    for (int e = cp[j]; e < cp[j + 1]; e++) s[ri[e]] ^= 1;
                    for (int e = lcp[j]; e < lcp[j + 1]; e++) o[lri[e]] ^= 1;
                }
            }
        }
        if (nfaults) nfaults[b] = cnt;
    }
    free(thr);
    return 0;
}

int oq_version(void) { return 1; }
