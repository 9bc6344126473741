// qd_api.hip -- host side of libquits_amd.so: the extern "C" entry points declared in include/quits_amd.h.
#include "../../include/quits_amd.h"
#include "qd_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#define QD_GEN_PREFIX_LDS (32 * 1024)   // LDS a workgroup of the serial BP kernel may spend on row prefixes (128 slots x 64 shots): five workgroups per CU stay resident

hipError_t qd_launch_bp(const BpGraphDev &g, const DecodeArgs &a, int64_t B, hipStream_t s);
hipError_t qd_launch_bp_scatter(const BpGraphDev &g, const ScatGraphDev &sg, const DecodeArgs &a, const ScatArgs &x, int64_t B, hipStream_t s);
hipError_t qd_launch_bp_scatter_wide(const BpGraphDev &g, const ScatGraphDev &sg, const DecodeArgs &a, const ScatArgs &x, int64_t B, hipStream_t s);
hipError_t qd_launch_bp_general(const GenGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, const GenWs &w, int bp_method,
                                int schedule, int64_t shot0, int nshots, hipStream_t s, GenStagePlan *plan);
hipError_t qd_launch_hold(const int32_t *count, int threshold, int microseconds, hipStream_t s);
int qd_bp_ps_lds_bytes(const GenGraphDev &g, int max_rdeg);
hipError_t qd_launch_bp_ps_lds(const GenGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int64_t B, hipStream_t s);
hipError_t qd_launch_osd0(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks_fast,
                          int blocks_full, hipStream_t s, bool handed_over = false);
int qd_osd_sr_layout(int m, int m_pad, int n, int out_words, int *off13, int *threads, int *rpt);
hipError_t qd_launch_osd0_sr(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int blocks, hipStream_t s);
// osd_cs.hip: OSD-CS / OSD-E, the rebuilt column-form kernel
int qd_osdcs_layout(int m, int n, int out_words, uint32_t max_wfix, int *off, int *variant, int *per_cu);
size_t qd_osdcs_ws_words(int variant);
hipError_t qd_launch_osdcs(const OsdGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, const int *off, int variant, int lds,
                           uint64_t *ws, int blocks, hipStream_t s);
size_t qd_osd_sr_ws_words(int m_pad, int mw, int threads, int rpt);
hipError_t qd_launch_lsd0(const GenGraphDev &gg, const BpGraphDev &bg, const DecodeArgs &d, uint64_t *q_ws, int blocks_alloc, int blocks,
                          int lsd_w, int lsd_order, const uint32_t *wfix, hipStream_t s);
size_t qd_lsd_ws_bytes(int m, int n, int blocks, int lsd_w);
int qd_lsd_lds_bytes(int m, int n, int out_words);
int qd_lsd_plane_rows(int m);
hipError_t qd_launch_stage_llr(const float *llr_in, int n, int n_pad, const uint32_t *bit_orig, int64_t B, float *llr_ws,
                               int32_t *fail_list, int32_t *fail_count, int32_t *status, hipStream_t s);
hipError_t qd_launch_spmv(const SpmatDev &A, const uint32_t *err, int64_t err_stride, int64_t B, uint8_t *out,
                          int64_t out_stride, int accumulate, hipStream_t s);
hipError_t qd_launch_unpack(const uint32_t *bits, int64_t stride_words, int nbits, int64_t B, uint8_t *out,
                            int64_t out_stride, hipStream_t s);
hipError_t qd_launch_count(const uint8_t *pred, const uint8_t *obs, int k, int64_t B, int64_t *count, hipStream_t s);
hipError_t qd_launch_sample(const SpmatDev &Ht, const SpmatDev &Lt, const uint32_t *thr, uint64_t seed, int64_t shot0,
                            int64_t B, int m, int nobs, uint8_t *det, int64_t det_stride, uint8_t *obs,
                            int64_t obs_stride, hipStream_t s);

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(QD_EHIP, "%s: %s", #expr, hipGetErrorString(e_));              \
    } while (0)

struct DevAllocs {
    std::vector<void *> ptrs;
    template <class Tp> int upload(const std::vector<Tp> &h, const Tp **out)
    {
        void *d = nullptr;
        size_t bytes = std::max<size_t>(h.size() * sizeof(Tp), 16);
        if (hipMalloc(&d, bytes) != hipSuccess) return -1;
        ptrs.push_back(d);
        if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice) != hipSuccess) return -1;
        *out = reinterpret_cast<const Tp *>(d);
        return 0;
    }
    void release()
    {
        for (void *p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
};

struct qd_graph {
    int device = 0;
    int m = 0, n = 0, nnz = 0, max_rdeg = 0, max_cdeg = 0, rank = -1;
    BpGraphDev bp{};
    GenGraphDev gen{};
    OsdGraphDev osd{};
    DevAllocs mem;
    std::vector<int32_t> h_cp, h_ri;   // host CSC, for the rank
    std::vector<double> h_llr0;        // log((1-p)/p) in double, fault order
    ScatGraphDev sc{};                 // scatter form of the flooding min-sum kernel (bp_scatter.hip); sc.ok = 0: not for this window
    std::vector<uint32_t> h_bit_rec;   // host copy of bp.bit_rec: a decoder on an LLR grid uploads its own with word 0 replaced
    std::vector<uint32_t> h_bit_orig;  // bit slot -> fault
    std::vector<int32_t> h_sc_slot;    // fault -> accumulator slot of the scatter kernels (empty: they are not used)
    long long sc_walk_cycles = 0, sc_walk_ideal = 0;   // modelled LDS cycles of one pass of the scatter kernels' walk, and without any bank conflict
};

struct qd_decoder {
    const qd_graph *g = nullptr;
    qd_params prm{};
    int64_t cap = 0;
    int osd_blocks = 0;
    float *llr_ws = nullptr;
    int32_t *fail_list = nullptr, *fail_count = nullptr;     // fail_count: the counter set of the call in flight = ctr_base + 64 * cset
    // Two sets of counters (fail / hard / redo / recheck counts), used by alternate calls.  A call's BP stage needs its set at zero; the set is zeroed by
    // the post-processing stage of the call BEFORE, on ITS stream (that set was last used two calls ago) -- not by fills at the head of the BP stage: in the
    // pipelined driver those fills sat between two BP kernels and each chunk lost ~2 ms there (a 4-byte fill launched the moment the other stream's OSD
    // kernel starts takes 1.9-2.1 ms, profiles/r06_bp_stream_bubble.txt).  set_clean: the set has been zeroed by an operation already queued.
    int32_t *ctr_base = nullptr;
    int cset = 0;
    bool set_clean[2] = {false, false};
    int64_t last_B = 0;         // batch size of the last BP stage (qd_decoder_post_head_start prices its failure count against it)
    // the failure count of an earlier call, read back without waiting (pinned copy + event queued behind a post-processing stage): OSD-0 decoders launch the
    // hold of qd_decoder_post_head_start only when that says "heavy" -- even an empty launch between two BP kernels costs the headline 0.8 %
    int32_t *host_fail = nullptr;
    hipEvent_t fail_ready = nullptr;
    bool fail_pending = false;
    int64_t fail_pending_B = 0;
    double fail_frac_hint = -1.0;
    uint16_t *order_ws = nullptr;
    uint64_t *q_spill = nullptr, *q_spill_fast = nullptr, *mt_ws = nullptr;
    uint64_t *q_spill_sr = nullptr;
    int32_t *hard_list = nullptr, *hard_list2 = nullptr;
    int osd_blocks_fast = 0;
    int osd_blocks_sr = 0;      // > 0: OSD-0 runs in qd_osd0_sr_kernel (osd_sr.hip), the mirrored kernel only takes the shots it hands over
    int osd_blocks_cs = 0;      // > 0: OSD-CS / OSD-E run in qd_osdcs_kernel (osd_cs.hip)
    int cs_off[16] = {0}, cs_variant = 0, cs_lds = 0;
    uint64_t *cs_ws = nullptr;  // [osd_blocks_cs][qd_osdcs_ws_words] Q columns for the candidate sweep
    int osd_w = 0;
    int general = 0;            // 1: the one-message-per-edge kernel (bp_general.hip) runs BP
    int lds_edge = 0;           // ... its LDS-resident form (flooding product-sum on a window whose messages fit LDS): no HBM message workspace
    int lsd = 0;                // 1: BP-LSD post-processing (lsd_kernels.hip) instead of OSD
    int lsd_blocks = 0;
    int lsd_w = 0;              // higher-order LSD: 0 = LSD-0, 1 = combination sweep, 2 = exhaustive (order = prm.osd_order)
    uint64_t *lsd_ws = nullptr; // [lsd_blocks][mw][m_pad] Q planes, then the work counter
    int64_t gen_ws_limit = 0;   // bytes; 0 = default
    GenWs gws{};
    GenStagePlan gsp{};         // serial schedule: the launches' iteration bounds, the second workspace, the survivor lists (nbounds = 0: one launch)
    // ---- LLR grid (flooding min-sum, ms_scaling 1): decoder-owned prior arrays on the fine and the coarse grid
    int grid_k = -1, grid_kc = -1, grid_floor = 0;   // grid_floor: the fine grid is the 2^-10 floor, not the rule's: any number of shots may need the redo pass
    BpGraphDev bp_fine{}, bp_coarse{};         // copies of g->bp with their own bit_rec
    const float *llr0_q = nullptr;             // fault-order LLRs for the one-message-per-edge kernel (fine grid)
    int32_t *redo_list = nullptr;
    int redo_cap = 0;
    int scatter = 0;                           // 1: the fine-grid pass runs in qd_bp_scatter_kernel
    const int32_t *prior_g = nullptr;          // [n_pad] fine-grid channel LLRs of the bit slots in grid units
    float m2_limit = 0.f;
    int32_t *recheck_list = nullptr;           // shots the scatter kernel's bound could not certify
    int recheck_cap = 0;
    DevAllocs mem;
    int profiling = 0;
    struct Span { int kind; hipEvent_t t0, t1; };   // kind 0 = BP kernel, 1 = OSD kernel(s)
    std::vector<Span> ev;
    double acc_ms[4] = {0, 0, 0, 0};
};

struct qd_spmat {
    int device = 0;
    SpmatDev d{};
    DevAllocs mem;
};

#define QD_GRID_MIN_BITS 10
static inline int align16(int x) { return (x + 15) & ~15; }
static inline int pad64(int x) { return (x + 63) & ~63; }

// ---- LDS banks for the scatter kernels (bp_scatter.hip, bp_scatter_wide.hip) --------------------------------------------------
// Step k of a wavefront's walk is one ds_read_b32 (gather pass) and one ds_add_u32 (scatter pass) over the accumulators of the faults
// its 64 checks meet at that step, serviced in two groups of 32 lanes with bank = (address / 4) mod 32 and one extra LDS cycle per
// additional lane on a busy bank (MI355X_MICROARCH.md, LDS; an atomic does not broadcast).  Two things are free: which bank a fault's
// accumulator lives in (the slot order is the kernels' own) and the order in which a check walks its edges (min, second min and sign
// parity are symmetric).  A group of 32 checks x 32 banks is a bipartite multigraph; if no bank receives more edges from the group
// than the group has steps, it splits into that many matchings (Koenig), i.e. a walk without a single conflict.  So:
//   scatter_banks: choose the banks so that every group's 32 bank loads are as equal as possible (descent on the sum of squared loads,
//                  one fault at a time, deterministic; starts from the degree-sorted bit slots mod 32);
//   scatter_walk:  step by step, a maximum matching of the group's unfinished checks into the banks (banks with the largest remaining
//                  load first: a bank whose remaining load equals the remaining steps must be served now); a check the matching leaves out
//                  takes its edge on the least busy bank; steps beyond a check's degree point at a trash slot on a bank nobody uses.
// Headline window (1008 x 9504): 1932 modelled cycles per pass with the round-3 greedy walk over the degree-sorted slots, 1109 here,
// 1052 without any conflict (WindowGraph.info(): scatter_walk_cycles / _ideal).
static void scatter_banks(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const std::vector<int> &chk_orig,
                          const std::vector<int> &bit_slot_of, std::vector<int> &bank)
{
    const int nh = (m + 31) / 32;
    bank.resize(n);
    for (int j = 0; j < n; ++j) bank[j] = bit_slot_of[j] & 31;
    if (std::getenv("QD_SCATTER_BANKS_BY_SLOT")) return;            // A/B: the round-3 assignment
    // fault -> (group, edges from that group)
    std::vector<std::vector<std::pair<int, int>>> fh(n);
    std::vector<int> load((size_t)nh * 32, 0), cnt(32, 0);
    for (int h = 0; h < nh; ++h)
        for (int s = 32 * h; s < std::min(m, 32 * h + 32); ++s) {
            const int i = chk_orig[s];
            for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
                const int j = col_idx[e];
                if (!fh[j].empty() && fh[j].back().first == h) fh[j].back().second++;
                else fh[j].push_back({h, 1});
                load[(size_t)h * 32 + bank[j]]++;
            }
        }
    for (int j = 0; j < n; ++j) cnt[bank[j]]++;
    const int cap = (n + 31) / 32 + 1;                               // slots per bank: the LDS footprint stays within 32 slots of pad64(n)
    for (int pass = 0; pass < 40; ++pass) {
        int moved = 0;
        for (int j = 0; j < n; ++j) {
            if (fh[j].empty()) continue;
            const int b0 = bank[j];
            long long stay = 0;                                      // what the sum of squares gains back when j leaves b0 ...
            for (auto &hm : fh[j]) stay += 2ll * load[(size_t)hm.first * 32 + b0] * hm.second - (long long)hm.second * hm.second;
            long long best = stay;
            int bb = b0;
            for (int b = 0; b < 32; ++b) {
                if (b == b0 || cnt[b] >= cap) continue;
                long long add = 0;                                   // ... and what it pays to enter b
                for (auto &hm : fh[j]) add += 2ll * load[(size_t)hm.first * 32 + b] * hm.second + (long long)hm.second * hm.second;
                if (add < best) { best = add; bb = b; }
            }
            if (bb != b0) {
                for (auto &hm : fh[j]) { load[(size_t)hm.first * 32 + b0] -= hm.second; load[(size_t)hm.first * 32 + bb] += hm.second; }
                cnt[b0]--; cnt[bb]++; bank[j] = bb; ++moved;
            }
        }
        if (!moved) break;
    }
}

// put(s, k, e, trash_bank): check slot s takes CSR edge e at step k (e < 0: a step beyond its degree, reading / adding 0 to the trash slot of trash_bank)
template <class Put>
static void scatter_walk(int m, const int32_t *row_ptr, const int32_t *col_idx, const std::vector<int> &chk_orig, const std::vector<int> &bank,
                         Put put, long long *cycles, long long *ideal)
{
    const bool greedy_only = std::getenv("QD_SCATTER_WALK_GREEDY") != nullptr;      // A/B: no matching, every check takes its least busy bank
    std::vector<std::vector<int>> rem(32);
    for (int w0 = 0; w0 < m; w0 += 64) {
        int trip = 0;
        for (int s = w0; s < std::min(m, w0 + 64); ++s) trip = std::max(trip, (int)(row_ptr[chk_orig[s] + 1] - row_ptr[chk_orig[s]]));
        trip = (trip + 3) & ~3;                                      // what the wavefront of these 64 slots walks (ScatGraphDev::deg_w)
        for (int g0 = w0; g0 < std::min(m, w0 + 64); g0 += 32) {
            const int gn = std::min(32, m - g0);
            for (int l = 0; l < gn; ++l) {
                const int i = chk_orig[g0 + l];
                rem[l].assign(col_idx + row_ptr[i], col_idx + row_ptr[i + 1]);     // faults; the CSR edge is found again by its column
            }
            auto take = [&](int l, int b, int k) {                   // lane l takes one of its remaining edges on bank b at step k
                const int i = chk_orig[g0 + l];
                for (size_t y = 0; y < rem[l].size(); ++y)
                    if (bank[rem[l][y]] == b) {
                        const int j = rem[l][y];
                        rem[l].erase(rem[l].begin() + (long)y);
                        const int e = (int)(std::lower_bound(col_idx + row_ptr[i], col_idx + row_ptr[i + 1], j) - col_idx);
                        put(g0 + l, k, e, 0);
                        return;
                    }
            };
            for (int k = 0; k < trip; ++k) {
                int rl[32] = {0}, used[32] = {0}, match_b[32], nact = 0, act[32];
                for (int b = 0; b < 32; ++b) match_b[b] = -1;
                for (int l = 0; l < gn; ++l)
                    if (!rem[l].empty()) { act[nact++] = l; for (int j : rem[l]) rl[bank[j]]++; }
                if (nact) {
                    // options of a lane: the banks of its remaining edges, busiest (remaining load) first
                    std::vector<int> opts[32];
                    for (int x = 0; x < nact; ++x) {
                        const int l = act[x];
                        bool has[32] = {false};
                        for (int j : rem[l]) has[bank[j]] = true;
                        for (int b = 0; b < 32; ++b) if (has[b]) opts[l].push_back(b);
                        std::stable_sort(opts[l].begin(), opts[l].end(), [&](int a, int b) { return rl[a] > rl[b]; });
                    }
                    std::stable_sort(act, act + nact, [&](int a, int b) { return opts[a].size() < opts[b].size(); });
                    bool seen[32];
                    // augmenting paths (Kuhn); 32 x 32, recursion depth <= 32
                    struct Aug {
                        std::vector<int> *opts; int *match_b; bool *seen;
                        bool run(int l) {
                            for (int b : opts[l]) {
                                if (seen[b]) continue;
                                seen[b] = true;
                                if (match_b[b] < 0 || run(match_b[b])) { match_b[b] = l; return true; }
                            }
                            return false;
                        }
                    } aug{opts, match_b, seen};
                    std::vector<int> left;
                    for (int x = 0; x < nact; ++x) {
                        if (greedy_only) { left.push_back(act[x]); continue; }
                        for (int b = 0; b < 32; ++b) seen[b] = false;
                        if (!aug.run(act[x])) left.push_back(act[x]);
                    }
                    for (int b = 0; b < 32; ++b)
                        if (match_b[b] >= 0) { take(match_b[b], b, k); used[b]++; }
                    for (int l : left) {                              // no free bank among its edges: the least busy one
                        int bb = -1;
                        for (int j : rem[l]) if (bb < 0 || used[bank[j]] < used[bb]) bb = bank[j];
                        take(l, bb, k); used[bb]++;
                    }
                    int mx = 0;
                    for (int b = 0; b < 32; ++b) mx = std::max(mx, used[b]);
                    *cycles += mx; *ideal += 1;
                }
                // lanes past their degree: the trash slot of a bank nobody uses at this step (there are 32 banks for 32 lanes)
                int fb = 0;
                for (int l = 0; l < 32; ++l) {
                    const bool real_lane = l < gn;
                    const int deg = real_lane ? (int)(row_ptr[chk_orig[g0 + l] + 1] - row_ptr[chk_orig[g0 + l]]) : 0;
                    if (k < deg) continue;
                    while (fb < 32 && used[fb]) ++fb;
                    const int b = fb < 32 ? fb : (l & 31);
                    if (fb < 32) used[fb]++;
                    if (g0 + l < ((m + 63) & ~63)) put(g0 + l, k, -1, b);
                }
            }
        }
    }
}

extern "C" int qd_version(void) { return 103; }      // 103: qd_decoder_post_head_start; 102: qd_graph_info fills 10 entries again, qd_graph_info_ex(g, info, n) the rest; 101: qd_decoder_postproc_kernel
extern "C" const char *qd_last_error(void) { return g_err; }
extern "C" int qd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int qd_graph_create(int32_t m, int32_t n, const int32_t *row_ptr, const int32_t *col_idx,
                               const double *priors, int32_t device, qd_graph **out)
{
    if (!out) return fail(QD_EINVAL, "out is null");
    *out = nullptr;
    if (m <= 0 || n <= 0 || !row_ptr || !col_idx || !priors) return fail(QD_EINVAL, "empty or null graph");
    if (m > 32767) return fail(QD_ECAPACITY, "m = %d detectors per window exceeds 32767", m);
    if (n > 65000) return fail(QD_ECAPACITY, "n = %d faults per window exceeds 65000", n);
    const int nnz = row_ptr[m];
    if (row_ptr[0] != 0 || nnz < 0) return fail(QD_EINVAL, "bad row_ptr");
    std::vector<int> rdeg(m), cdeg(n, 0);
    for (int i = 0; i < m; ++i) {
        rdeg[i] = row_ptr[i + 1] - row_ptr[i];
        if (rdeg[i] < 0) return fail(QD_EINVAL, "row_ptr not monotone");
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            if (col_idx[e] < 0 || col_idx[e] >= n) return fail(QD_EINVAL, "column index out of range");
            if (e > row_ptr[i] && col_idx[e] <= col_idx[e - 1]) return fail(QD_EINVAL, "columns must ascend in a row");
            cdeg[col_idx[e]]++;
        }
    }
    for (int j = 0; j < n; ++j)
        if (!(priors[j] > 0.0 && priors[j] < 1.0)) return fail(QD_EINVAL, "prior %d = %g outside (0,1)", j, priors[j]);
    const int max_rdeg = *std::max_element(rdeg.begin(), rdeg.end());
    const int max_cdeg = *std::max_element(cdeg.begin(), cdeg.end());
    if (max_rdeg > QD_MAX_ROW_DEG || max_rdeg < 1)
        return fail(QD_ECAPACITY, "row weight %d outside 1..%d", max_rdeg, QD_MAX_ROW_DEG);
    if (max_cdeg > QD_MAX_COL_DEG || max_cdeg < 1)
        return fail(QD_ECAPACITY, "column weight %d outside 1..%d", max_cdeg, QD_MAX_COL_DEG);

    qd_graph *g = new qd_graph();
    g->device = device; g->m = m; g->n = n; g->nnz = nnz; g->max_rdeg = max_rdeg; g->max_cdeg = max_cdeg;
    if (hipSetDevice(device) != hipSuccess) { delete g; return fail(QD_EHIP, "hipSetDevice(%d) failed", device); }

    // slots: degree-descending, stable
    std::vector<int> chk_slot_of(m), chk_orig(m), bit_slot_of(n), bit_orig(n);
    std::iota(chk_orig.begin(), chk_orig.end(), 0);
    std::stable_sort(chk_orig.begin(), chk_orig.end(), [&](int a, int b) { return rdeg[a] > rdeg[b]; });
    for (int s = 0; s < m; ++s) chk_slot_of[chk_orig[s]] = s;
    std::iota(bit_orig.begin(), bit_orig.end(), 0);
    std::stable_sort(bit_orig.begin(), bit_orig.end(), [&](int a, int b) { return cdeg[a] > cdeg[b]; });
    for (int s = 0; s < n; ++s) bit_slot_of[bit_orig[s]] = s;

    const int m_pad = pad64(m), n_pad = pad64(n);
    const int max_rdeg_pad = (max_rdeg + 3) & ~3;
    const int bp_threads_ = std::max(m, (n + 9) / 10) <= 256 ? 256 : (std::max(m, (n + 9) / 10) <= 512 ? 512 : 1024);
    // the shape conditions of the scatter kernel (its LDS fit is checked where the layouts are known)
    // (windows small enough for 256-thread workgroups: the one-check-per-lane kernel measured 4.0 vs 4.7 ms per launch on the W = 3
    // windows of the [[72,12,6]] code, 9.0 vs 9.8 ms on those of the [[144,12,12]] code, profiles/r03z_scatter_other_configs.txt; on
    // 128 lanes x 2 checks it is 3.98 -> 3.75 ms on the former (108 checks: taken) and 8.95 -> 9.45 ms on the latter (216 checks:
    // stay with the gather kernel unless QD_SCATTER_SMALL is set), profiles/r03x_scatter_shapes_ab.txt)
    const int min_rdeg_ = *std::min_element(rdeg.begin(), rdeg.end());
    // (round 5: with the accumulators' banks balanced per group of 32 checks -- scatter_banks / scatter_walk -- the 128-lane shape is 8.92 -> 7.73 ms per
    // launch on those 216-check windows and 3.78 -> 3.00 ms on the 108-check ones, so every window of <= 256 checks goes there; QD_NO_SCATTER_SMALL=1:
    // the 129..256-check windows stay with the gather kernel, profiles/r05_scatter_small_ab.txt)
    const bool scatter_narrow = m <= bp_threads_ && (bp_threads_ >= 512 || m <= 128 || !std::getenv("QD_NO_SCATTER_SMALL")) && max_rdeg_pad <= 64 && min_rdeg_ >= 2;
    // ... and of its two-checks-per-lane form (bp_scatter_wide.hip): more checks than a workgroup has lanes, or rows of 65..96 faults --
    // the QLP windows of BASELINE configs[4] (1326 checks of up to 78 faults on 704 lanes)
    const int wide_threads_ = pad64((m + 1) / 2) <= 704 ? 704 : 1024;
    const bool scatter_wide = !scatter_narrow && bp_threads_ == 1024 && m <= 2 * wide_threads_ && max_rdeg_pad <= 96 && min_rdeg_ >= 2 &&
                              (m > 1024 || max_rdeg_pad > 64) && !std::getenv("QD_NO_SCATTER_WIDE");
    const bool scatter_shape = scatter_narrow || scatter_wide;
    const int dummy_bit = n_pad, dummy_chk = m_pad;        // one extra LDS slot each
    if ((m_pad + 1) * 16 > 65535) {
        delete g;
        return fail(QD_ECAPACITY, "m = %d detectors per window: check-state offsets exceed 16 bits", m);
    }
    if (n_pad + 1 > 65535) {
        delete g;
        return fail(QD_ECAPACITY, "n = %d faults per window: fault slots exceed 16 bits", n);
    }
    const int sign_mode = max_rdeg_pad <= 32 ? 0 : (max_rdeg_pad <= 44 ? 1 : 2);    // mode 1: 15 spare bits of the state word hold signs 32..46
    const int neg_words_ = (max_rdeg_pad + 31) / 32;
    const int off_chk_ = 0, off_cneg_ = (m_pad + 4) * 16;
    const int off_llr_ = off_cneg_ + (sign_mode == 2 ? align16((neg_words_ - 1) * m_pad * 4) : 0);
    // 32-bit offsets always (round 6: the 16-bit packing -- half of 66 KB of L2-resident adjacency for one more unpacking instruction per
    // edge, 60.4 -> 62.0 ms when K1 was the headline kernel -- was only reachable through an environment knob and doubled K1's instantiations: removed)
    const int adj32 = 1;
    const int rec_words = ((1 + max_cdeg) + 3) & ~3;
    std::vector<uint8_t> chk_deg(m_pad, 0), bit_deg(n_pad, 0);
    std::vector<int32_t> chk_degp_w(m_pad / 64, 0);
    std::vector<uint32_t> chk_orig_u(m_pad, 0), bit_orig_u(n_pad, 0);
    std::vector<uint32_t> bit_rec((size_t)n_pad * rec_words, (uint32_t)(off_chk_ + dummy_chk * 16) << 16);
    // CSC with the edge's position inside its row
    std::vector<int32_t> cp(n + 1, 0), ri(nnz), pos(nnz);
    for (int j = 0; j < n; ++j) cp[j + 1] = cp[j] + cdeg[j];
    {
        std::vector<int> fill(n, 0);
        for (int i = 0; i < m; ++i)
            for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
                const int j = col_idx[e], dst = cp[j] + fill[j]++;
                ri[dst] = i; pos[dst] = e - row_ptr[i];
            }
    }
    for (int s = 0; s < m; ++s) { chk_deg[s] = (uint8_t)rdeg[chk_orig[s]]; chk_orig_u[s] = (uint32_t)chk_orig[s]; }
    // wave-uniform trip counts (slots are degree-sorted, so a wavefront's lanes nearly agree anyway)
    for (int w0 = 0; w0 < m_pad; w0 += 64) {
        int mx = 0;
        for (int s = w0; s < w0 + 64; ++s) mx = std::max(mx, (int)chk_deg[s]);
        chk_degp_w[w0 / 64] = ((mx + 3) & ~3) | (mx << 16);      // trip count | exact maximum (edges beyond it are padding for every lane)
    }
    // check -> fault adjacency, ELL-transposed, as LDS byte offsets of the posteriors
    std::vector<uint32_t> chk_adj32((size_t)max_rdeg_pad * m_pad, (uint32_t)(off_llr_ + dummy_bit * 4));
    // Order in which a check walks its edges.  Nothing the check pass computes depends on it (minimum, second minimum and the
    // sign parity are symmetric; a tie for the minimum leaves min1 = min2, so it does not matter which edge carries the argmin
    // label), so it is chosen for the LDS: step k of a wavefront is one ds_read_b32 gather, serviced in two groups of 32 lanes
    // with bank = (address / 4) mod 32 (MI355X_MICROARCH.md, LDS), one extra cycle per additional distinct address on a bank.
    // Greedy per group of 32 check slots: at every step each check takes, among its remaining edges, the one whose bank has
    // the fewest distinct addresses so far in that step.  Headline window: 2918 -> 1463 LDS cycles per iteration for these
    // gathers (1062 without any conflict; tools/lds_model.py).
    // The scatter kernel (bp_scatter.hip) walks the same order with ds_add_u32, and lanes of one instruction that add to the SAME
    // address serialise like any other bank conflict (profiles/r03z_lds_atomic_rates.txt: two lanes per address 2.2 -> 4.7 ns per
    // instruction), where a gather broadcasts for free: for the windows that kernel takes, a lane on a busy bank costs one cycle
    // whatever its address.
    const bool walk_for_atomics = scatter_shape && !std::getenv("QD_WALK_BROADCAST");
    std::vector<int32_t> step_of(nnz);              // CSR edge -> position in its check's walk
    {
        std::vector<std::vector<int>> rem(32);
        std::vector<int> used_cnt(32), order32(32);
        std::vector<std::vector<uint32_t>> used_addr(32);
        for (int g0 = 0; g0 < m; g0 += 32) {
            const int gn = std::min(32, m - g0);
            int kmax = 0;
            for (int l = 0; l < gn; ++l) {
                const int i = chk_orig[g0 + l];
                rem[l].clear();
                for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) rem[l].push_back(e);
                kmax = std::max(kmax, rdeg[i]);
            }
            for (int k = 0; k < kmax; ++k) {
                for (int b = 0; b < 32; ++b) { used_cnt[b] = 0; used_addr[b].clear(); }
                int na = 0;
                for (int l = 0; l < gn; ++l) if (!rem[l].empty()) order32[na++] = l;
                std::stable_sort(order32.begin(), order32.begin() + na, [&](int a, int b) { return rem[a].size() > rem[b].size(); });
                for (int x = 0; x < na; ++x) {
                    std::vector<int> &r = rem[order32[x]];
                    int best = 0, bestc = 1 << 30;
                    for (int y = 0; y < (int)r.size(); ++y) {
                        const uint32_t slot = (uint32_t)bit_slot_of[col_idx[r[y]]];
                        const int b = (int)(slot & 31u);
                        int c = used_cnt[b];
                        if (!walk_for_atomics)
                            for (uint32_t a : used_addr[b]) if (a == slot) { c = 0; break; }  // same address: a broadcast, free
                        if (c < bestc) { bestc = c; best = y; if (c == 0) break; }
                    }
                    const uint32_t slot = (uint32_t)bit_slot_of[col_idx[r[best]]];
                    bool dup = false;
                    for (uint32_t a : used_addr[slot & 31u]) dup |= (a == slot);
                    if (!dup) used_addr[slot & 31u].push_back(slot);
                    if (!dup || walk_for_atomics) used_cnt[slot & 31u]++;
                    step_of[r[best]] = k;
                    r.erase(r.begin() + best);
                }
            }
        }
    }
    std::vector<int32_t> wpos(nnz);                 // CSC edge -> position in its check's walk (pos[] stays the natural position)
    {
        std::vector<int> fill(n, 0);
        for (int i = 0; i < m; ++i)
            for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) wpos[cp[col_idx[e]] + fill[col_idx[e]]++] = step_of[e];
    }
    for (int s = 0; s < m; ++s) {
        const int i = chk_orig[s];
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            const int k = step_of[e];
            const uint32_t off = (uint32_t)off_llr_ + (uint32_t)bit_slot_of[col_idx[e]] * 4u;
            const size_t at = ((size_t)(k >> 2) * m_pad + s) * 4 + (k & 3);      // [group of 4 edges][slot][4]: one vector load per group
            chk_adj32[at] = off;
        }
    }
    // fault records: prior + (check state offset, where that check keeps this edge's sign)
    auto rec_at = [&](int slot, int word) { return ((size_t)(word >> 2) * n_pad + slot) * 4 + (word & 3); };   // [chunk][slot][4]
    for (int s = 0; s < n_pad; ++s) {
        const float one = 1.0f;
        std::memcpy(&bit_rec[rec_at(s, 0)], &one, 4);
    }
    for (int s = 0; s < n; ++s) {
        const int j = bit_orig[s];
        bit_deg[s] = (uint8_t)cdeg[j];
        bit_orig_u[s] = (uint32_t)j;
        const float l0 = (float)std::log((1.0 - priors[j]) / priors[j]);
        std::memcpy(&bit_rec[rec_at(s, 0)], &l0, 4);
        for (int q = 0; q < cdeg[j]; ++q) {
            const int cs = chk_slot_of[ri[cp[j] + q]];
            const int k = wpos[cp[j] + q];
            const int degp = chk_degp_w[cs / 64] & 0xFFFF;
            const int w = k >> 5, kend = std::min(degp - 32 * w, 32);
            const int sbit = kend - 1 - (k & 31);              // the check pass shifts signs in from bit 0 (v_alignbit)
            // mode 0/1: bit index into the 64-bit value {w : z} of the state (signs 32..46 sit in w's bits 16..30)
            const uint32_t where = sign_mode == 2 ? (((uint32_t)w << 5) | (uint32_t)sbit)
                                                  : (uint32_t)(w == 0 ? sbit : 32 + 16 + sbit);
            bit_rec[rec_at(s, 1 + q)] = ((uint32_t)(off_chk_ + cs * 16) << 16) | where;
        }
    }
    g->h_cp = cp; g->h_ri = ri;
    g->h_bit_rec = bit_rec; g->h_bit_orig = bit_orig_u;
    g->h_llr0.resize(n);
    for (int j = 0; j < n; ++j) g->h_llr0[j] = std::log((1.0 - priors[j]) / priors[j]);
    {
        GenGraphDev &gg = g->gen;
        gg.m = m; gg.n = n; gg.nnz = nnz; gg.out_words = (n + 31) / 32;
        std::vector<int32_t> rp_v(row_ptr, row_ptr + m + 1), ci_v(col_idx, col_idx + nnz), c2r(nnz);
        for (int e = 0; e < nnz; ++e) c2r[e] = row_ptr[ri[e]] + pos[e];
        std::vector<float> l0(n);
        for (int j = 0; j < n; ++j) l0[j] = (float)std::log((1.0 - priors[j]) / priors[j]);
        int rcg = 0;
        rcg |= g->mem.upload(rp_v, &gg.rp); rcg |= g->mem.upload(ci_v, &gg.ci);
        rcg |= g->mem.upload(cp, &gg.cp); rcg |= g->mem.upload(ri, &gg.ri);
        rcg |= g->mem.upload(c2r, &gg.c2r); rcg |= g->mem.upload(l0, &gg.llr0);
        gg.frec = nullptr; gg.erow = nullptr;
        if (max_cdeg <= 8 && nnz <= 65534) {
            std::vector<uint16_t> erow(nnz);
            for (int i = 0; i < m; ++i)
                for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) erow[e] = (uint16_t)i;
            rcg |= g->mem.upload(erow, &gg.erow);
            std::vector<uint16_t> frec((size_t)n * 16, (uint16_t)0xFFFF);
            for (int j = 0; j < n; ++j)
                for (int e = cp[j]; e < cp[j + 1]; ++e) {
                    frec[(size_t)j * 16 + (e - cp[j])] = (uint16_t)c2r[e];
                    frec[(size_t)j * 16 + 8 + (e - cp[j])] = (uint16_t)ri[e];
                }
            rcg |= g->mem.upload(frec, &gg.frec);
        }
        {
            gg.ell_w = (max_rdeg + 63) / 64 * 64;
            std::vector<int32_t> ell((size_t)m * gg.ell_w * 2, 0);
            for (int i = 0; i < m; ++i)
                for (int x = 0; x < gg.ell_w; ++x) {
                    const int e = row_ptr[i] + x;
                    const bool in = e < row_ptr[i + 1];
                    ell[((size_t)i * gg.ell_w + x) * 2] = in ? col_idx[e] : -1;
                    ell[((size_t)i * gg.ell_w + x) * 2 + 1] = in ? bit_slot_of[col_idx[e]] : 0;
                }
            rcg |= g->mem.upload(ell, &gg.ell);
        }
        {
            // the records below pack the fault index into 24 bits of their first word: refuse before building anything
            if (n >= (1 << 24)) { g->mem.release(); delete g; return fail(QD_EINVAL, "more than 2^24 faults"); }
            std::vector<int32_t> last(m, 0), lev(n, 0);
            int nlev = 0;
            for (int j = 0; j < n; ++j) {
                int l = 0;
                for (int e = cp[j]; e < cp[j + 1]; ++e) l = std::max(l, last[ri[e]]);
                lev[j] = l;                                     // 0-based
                for (int e = cp[j]; e < cp[j + 1]; ++e) last[ri[e]] = l + 1;
                nlev = std::max(nlev, l + 1);
            }
            std::vector<int32_t> lp(nlev + 1, 0), lb(n);
            for (int j = 0; j < n; ++j) lp[lev[j] + 1]++;
            for (int l = 0; l < nlev; ++l) lp[l + 1] += lp[l];
            std::vector<int32_t> fillp(lp.begin(), lp.end() - 1);
            for (int j = 0; j < n; ++j) lb[fillp[lev[j]]++] = j;
            // LDS slots of the rows' running prefixes (see GenGraphDev): a slot is handed on only to a row that starts strictly
            // after its previous owner's last level (two faults of one level touch their rows in any order)
            std::vector<int32_t> first(m, nlev), lastl(m, -1), slot(m, 0), order(m);
            for (int i = 0; i < m; ++i)
                for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
                    first[i] = std::min(first[i], lev[col_idx[e]]);
                    lastl[i] = std::max(lastl[i], lev[col_idx[e]]);
                }
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
            std::vector<int32_t> busy_until;                // per slot: last level of its current owner
            for (int i : order) {
                int sidx = -1;
                for (size_t q = 0; q < busy_until.size(); ++q)
                    if (busy_until[q] < first[i]) { sidx = (int)q; break; }
                if (sidx < 0) { sidx = (int)busy_until.size(); busy_until.push_back(-1); }
                busy_until[sidx] = lastl[i];
                slot[i] = sidx;
            }
            gg.nslots = ((int)busy_until.size() * 256 <= QD_GEN_PREFIX_LDS && m < (1 << 23) && !std::getenv("QD_NO_LDS_PREFIX")) ? (int)busy_until.size() : 0;
            // one record per (step, wavefront): see GenGraphDev
            const int G = QD_GEN_GS, D = qd_gen_unroll(max_cdeg);
            const int RW = (2 + 2 * D + 3) & ~3;
            size_t nstep = 0;
            for (int l = 0; l < nlev; ++l) nstep += (size_t)(lp[l + 1] - lp[l] + G - 1) / G;
            std::vector<uint32_t> srec(nstep * G * RW, 0u);
            size_t st = 0;
            for (int l = 0; l < nlev; ++l) {
                const int cnt = lp[l + 1] - lp[l], steps = (cnt + G - 1) / G;
                for (int q = 0; q < steps; ++q, ++st)
                    for (int wv = 0; wv < G; ++wv) {
                        uint32_t *r = &srec[(st * G + wv) * RW];
                        const uint32_t bar = (q == steps - 1) ? 0x80000000u : 0u;
                        const int x = q * G + wv;
                        if (x >= cnt) { r[0] = bar; continue; }
                        const int j = lb[lp[l] + x], c0 = cp[j], deg = cp[j + 1] - c0;
                        r[0] = (uint32_t)j | ((uint32_t)deg << 24) | bar;
                        std::memcpy(&r[1], &l0[j], 4);
                        for (int k = 0; k < deg; ++k) {
                            const int i = ri[c0 + k];
                            r[2 + k] = (uint32_t)i;
                            if (gg.nslots > 0) r[2 + k] |= ((uint32_t)slot[i] << 24) | (col_idx[row_ptr[i]] == j ? 0x800000u : 0u);
                            r[2 + D + k] = (uint32_t)c2r[c0 + k];
                        }
                    }
            }
            gg.nlev = nlev; gg.nstep = (int)nstep; gg.srec_w = RW;
            rcg |= g->mem.upload(srec, &gg.srec);
        }
        if (rcg) { g->mem.release(); delete g; return fail(QD_EHIP, "device allocation failed while building the graph"); }
    }

    BpGraphDev &bp = g->bp;
    bp.m = m; bp.n = n; bp.m_pad = m_pad; bp.n_pad = n_pad; bp.max_rdeg = max_rdeg; bp.max_cdeg = max_cdeg;
    bp.neg_words = (max_rdeg_pad + 31) / 32; bp.out_words = (n + 31) / 32;
    bp.max_rdeg_pad = max_rdeg_pad; bp.dummy_bit = dummy_bit; bp.dummy_chk = dummy_chk;
    bp.rec_words = rec_words; bp.adj32 = adj32; bp.sign_mode = sign_mode;
    for (int q = 0; q < QD_MAX_COL_DEG; ++q) {
        int cnt = 0;
        for (int w0 = 0; w0 < n_pad; w0 += 64) {
            int mx = 0;
            for (int s = w0; s < w0 + 64; ++s) mx = std::max(mx, (int)bit_deg[s]);
            if (mx > q) cnt = w0 + 64;
        }
        bp.bit_thr[q] = cnt;
    }
    int rc = 0;
    { const uint32_t *p32 = nullptr; rc |= g->mem.upload(chk_adj32, &p32); bp.chk_adj = p32; }
    rc |= g->mem.upload(chk_degp_w, &bp.chk_degp_w);
    rc |= g->mem.upload(chk_orig_u, &bp.chk_orig);
    rc |= g->mem.upload(bit_rec, &bp.bit_rec);
    rc |= g->mem.upload(bit_orig_u, &bp.bit_orig);
    {
        std::vector<uint32_t> slot_of(bit_slot_of.begin(), bit_slot_of.end()), wfix((size_t)n);
        for (int j = 0; j < n; ++j) {
            double w = std::log(1.0 / priors[j]) * 262144.0;
            wfix[j] = (uint32_t)std::llround(std::min(std::max(w, 0.0), 4294967295.0));
            g->osd.max_wfix = std::max(g->osd.max_wfix, wfix[j]);
        }
        rc |= g->mem.upload(slot_of, &bp.bit_slot_of);
        rc |= g->mem.upload(wfix, &g->osd.wfix);
    }
    // LDS carve-up for BP
    int off = 0;
    bp.off_chk = off; off += (m_pad + 4) * 16;                 // + the dummy check
    bp.off_cneg = off; off += (sign_mode == 2 ? align16((bp.neg_words - 1) * m_pad * 4) : 0);
    bp.off_llr = off; off += align16((n_pad + 4) * 4);         // + the dummy bit
    if (bp.off_chk != off_chk_ || bp.off_cneg != off_cneg_ || bp.off_llr != off_llr_) {
        g->mem.release(); delete g;
        return fail(QD_EINVAL, "internal: LDS layout mismatch");
    }
    bp.off_out = off; off += align16(bp.out_words * 4);
    bp.off_misc = off; off += 256;
    bp.lds_bytes = off;
    const int need = std::max(m, (n + 9) / 10);
    bp.threads = need <= 256 ? 256 : (need <= 512 ? 512 : 1024);

    // Scatter form of the flooding min-sum kernel (bp_scatter.hip): one check per lane with its state in registers, two integer
    // posterior buffers in LDS.  Admitted when every check has a lane, rows hold 2..64 faults (two sign words; a row of one fault
    // has no second minimum) and the two buffers leave as many workgroups per CU as the gather kernel's layout does.
    {
        ScatGraphDev &sc = g->sc;
        sc = ScatGraphDev{};
        const int min_rdeg = *std::min_element(rdeg.begin(), rdeg.end());
        // ---- the accumulators' own slot order: a bank for every fault (scatter_banks), then bank + 32 * (rank inside the bank)
        std::vector<int> sc_bank, sc_slot;
        int sc_rows = 0;
        if (scatter_shape) {
            scatter_banks(m, n, row_ptr, col_idx, chk_orig, bit_slot_of, sc_bank);
            std::vector<int> fillb(32, 0);
            sc_slot.resize(n);
            for (int j = 0; j < n; ++j) sc_slot[j] = sc_bank[j] + 32 * fillb[sc_bank[j]]++;
            sc_rows = *std::max_element(fillb.begin(), fillb.end());
        }
        const int sc_trash = 32 * sc_rows;                     // 32 trash slots, one per bank, behind the accumulators
        sc.nslots = scatter_shape ? sc_trash + 32 : n_pad + 4;
        const int buf = align16(sc.nslots * 4);
        sc.offA = 0; sc.offB = 0; sc.off_out = buf;            // one buffer: the scatter pass works in place
        sc.off_bmap = sc.off_out + align16(bp.out_words * 4);
        sc.off_misc = sc.off_bmap;
        sc.lds_bytes = sc.off_misc + 256;
        const int by_threads = 2048 / bp.threads;
        const int res_old = std::min(by_threads, QD_LDS_BYTES / std::max(1, bp.lds_bytes));
        // the shape that will really be launched (the residency test used to look at a shape that never ran): several
        // checks per lane -- see below for the measurements behind each case
        int shape_threads = bp.threads, shape_cpl = 1;
        if (scatter_wide) {
            shape_threads = wide_threads_; shape_cpl = 2;
            if (std::getenv("QD_SCATTER_WIDE_T1024")) { shape_threads = 1024; shape_cpl = 2; }                // A/B: 16 wavefronts x 2 checks
            else if (m <= 1536 && !std::getenv("QD_SCATTER_WIDE_T704")) { shape_threads = 512; shape_cpl = 3; }    // 19.8 -> 19.3 ms per QLP launch
        } else if (!std::getenv("QD_SCATTER_CPL1")) {
            if (bp.threads == 1024 && 4 * sc.lds_bytes <= QD_LDS_BYTES) { shape_threads = 512; shape_cpl = 2; }
            else if (bp.threads == 512 && 8 * sc.lds_bytes <= QD_LDS_BYTES && !std::getenv("QD_SCATTER_NO_CPL2_256")) { shape_threads = 256; shape_cpl = 2; }
            else if (bp.threads == 256 && 16 * sc.lds_bytes <= QD_LDS_BYTES) { shape_threads = 128; shape_cpl = 2; }    // (<= 256 checks)
        }
        const int res_new = std::min(2048 / shape_threads, QD_LDS_BYTES / sc.lds_bytes);
        if (scatter_shape && min_rdeg >= 2 && bp.lds_bytes <= QD_LDS_BYTES && res_new >= 1 && res_new >= res_old) {
            const int rows = max_rdeg_pad / 4 + 2;            // two spare group rows: the kernel loads up to two groups ahead unconditionally
            // every entry starts at the trash slot of bank (lane mod 32): the steps nobody reaches and the two spare rows
            std::vector<uint32_t> adjA((size_t)rows * m_pad * 4);
            for (int r = 0; r < rows; ++r)
                for (int s = 0; s < m_pad; ++s)
                    for (int q = 0; q < 4; ++q) adjA[((size_t)r * m_pad + s) * 4 + q] = (uint32_t)sc.offA + (uint32_t)(sc_trash + (s & 31)) * 4u;
            long long walk_cyc = 0, walk_ideal = 0;
            std::vector<uint8_t> edge_step((size_t)nnz, 0xFF);
            bool walk_ok = true;
            scatter_walk(m, row_ptr, col_idx, chk_orig, sc_bank, [&](int s, int k, int e, int trash_bank) {
                const uint32_t slot = e >= 0 ? (uint32_t)sc_slot[col_idx[e]] : (uint32_t)(sc_trash + trash_bank);
                if (k < 0 || k >= rows * 4 || s < 0 || s >= m_pad) { walk_ok = false; return; }
                if (e >= 0) {                                      // a real edge: once, at a step below its check's degree
                    const int i = s < m ? chk_orig[s] : -1;
                    if (i < 0 || e < row_ptr[i] || e >= row_ptr[i + 1] || k >= rdeg[i] || edge_step[e] != 0xFF) { walk_ok = false; return; }
                    edge_step[e] = (uint8_t)k;
                }
                adjA[((size_t)(k >> 2) * m_pad + s) * 4 + (k & 3)] = (uint32_t)sc.offA + slot * 4u;
            }, &walk_cyc, &walk_ideal);
            for (int i = 0; i < m && walk_ok; ++i) {               // every step below the degree taken exactly once
                uint64_t seen_lo = 0, seen_hi = 0;
                for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
                    const int k = edge_step[e];
                    if (k == 0xFF) { walk_ok = false; break; }
                    uint64_t &w = k < 64 ? seen_lo : seen_hi;
                    if ((w >> (k & 63)) & 1u) { walk_ok = false; break; }
                    w |= 1ull << (k & 63);
                }
            }
            if (!walk_ok) { g->mem.release(); delete g; return fail(QD_ECAPACITY, "scatter walk: an edge was not placed exactly once below its check's degree"); }
            g->sc_walk_cycles = walk_cyc; g->sc_walk_ideal = walk_ideal;
            std::vector<uint32_t> slot_fault((size_t)sc.nslots, 0xFFFFFFFFu), k1_slot((size_t)n, 0u);
            for (int j = 0; j < n; ++j) { slot_fault[sc_slot[j]] = (uint32_t)j; k1_slot[bit_slot_of[j]] = (uint32_t)sc_slot[j]; }
            g->h_sc_slot.assign(sc_slot.begin(), sc_slot.end());
            std::vector<uint32_t> deg_w(m_pad / 64, 0u);
            for (int w0 = 0; w0 < m; w0 += 64) {
                int mx = 0, mn = 255;
                for (int s = w0; s < std::min(m, w0 + 64); ++s) { mx = std::max(mx, (int)chk_deg[s]); mn = std::min(mn, (int)chk_deg[s]); }
                deg_w[w0 / 64] = (uint32_t)((mx + 3) & ~3) | ((uint32_t)mx << 8) | ((uint32_t)mn << 16);
            }
            int rcs = 0;
            rcs |= g->mem.upload(adjA, &sc.adjA); sc.adjB = sc.adjA;
            rcs |= g->mem.upload(slot_fault, &sc.slot_fault); rcs |= g->mem.upload(k1_slot, &sc.k1_slot);
            rcs |= g->mem.upload(deg_w, &sc.deg_w); rcs |= g->mem.upload(chk_deg, &sc.chk_deg);
            if (rcs) { g->mem.release(); delete g; return fail(QD_EHIP, "device allocation failed while uploading the scatter adjacency"); }
            sc.ok = 1;
            // Several checks per lane (bp_scatter_wide.hip).  (a) windows the one-check-per-lane kernel cannot take: three checks per
            // lane on 512 lanes (or two on 704 / 1024).  (b) windows it can take, on HALF the lanes with two checks each: the same number
            // of wavefronts per CU in twice as many workgroups, i.e. half as many wavefronts per barrier -- headline BP 50.3 -> 44.2 ms per
            // 65 536 shots, same bits (profiles/r03x_scatter_cpl2_ab.txt); taken when the LDS holds twice the workgroups.
            // QD_SCATTER_CPL1=1 keeps one check per lane.  Measured and not kept (profiles/r03x_scatter_shapes2_ab.txt): four checks per
            // lane on a quarter of the lanes (headline 46.2 -> 50.9 ms), 384 lanes x 4 checks for the QLP windows (19.2 -> 24.4 ms).
            sc.wide_threads = shape_cpl > 1 ? shape_threads : 0; sc.wide_cpl = shape_cpl > 1 ? shape_cpl : 0;
            if (sc.wide_threads) {
                // deal the slot-waves (64 consecutive check slots, heaviest first) to the workgroup's wavefronts so that the largest
                // number of edges a wavefront walks between two barriers is small: each goes to the wavefront with the fewest edges so
                // far that still has a free round (longest-processing-time rule), then single moves / swaps out of the heaviest
                // wavefront while they lower the maximum (QLP windows on 8 wavefronts x 3 rounds: 222 -> 202 edge steps)
                const int nw = sc.wide_threads / 64, nsw = m_pad / 64, cpl = sc.wide_cpl;
                const bool natural = std::getenv("QD_SCATTER_NATURAL_ROUNDS") != nullptr;      // A/B: round j of wavefront w = slot-wave j * nw + w
                auto cost = [&](int sw) { return (int)(deg_w[sw] & 0xFF) + 6; };               // trip count + the per-check work outside the edge loops
                std::vector<std::vector<int>> bins(nw);
                std::vector<int> load(nw, 0);
                for (int sw = 0; sw < nsw; ++sw) {
                    int best = -1;
                    if (natural) best = sw % nw;
                    else
                        for (int w = 0; w < nw; ++w)
                            if ((int)bins[w].size() < cpl && (best < 0 || load[w] < load[best])) best = w;
                    if (best < 0 || (int)bins[best].size() >= cpl) { g->mem.release(); delete g; return fail(QD_ECAPACITY, "scatter kernel shape does not cover the window"); }
                    bins[best].push_back(sw); load[best] += cost(sw);
                }
                for (int pass = 0; pass < 1000 && !natural; ++pass) {
                    const int hi = (int)(std::max_element(load.begin(), load.end()) - load.begin()), Lh = load[hi];
                    int gain = 0, bi = -1, bw = -1, bk = -1;
                    for (int i = 0; i < (int)bins[hi].size(); ++i) {
                        const int x = cost(bins[hi][i]);
                        for (int w = 0; w < nw; ++w) {
                            if (w == hi) continue;
                            if ((int)bins[w].size() < cpl) {
                                const int nl = std::max(Lh - x, load[w] + x);
                                if (Lh - nl > gain) { gain = Lh - nl; bi = i; bw = w; bk = -1; }
                            }
                            for (int k = 0; k < (int)bins[w].size(); ++k) {
                                const int y = cost(bins[w][k]);
                                if (y >= x) continue;
                                const int nl = std::max(Lh - x + y, load[w] - y + x);
                                if (Lh - nl > gain) { gain = Lh - nl; bi = i; bw = w; bk = k; }
                            }
                        }
                    }
                    if (bi < 0) break;
                    const int x = cost(bins[hi][bi]);
                    if (bk < 0) { bins[bw].push_back(bins[hi][bi]); bins[hi].erase(bins[hi].begin() + bi); load[hi] -= x; load[bw] += x; }
                    else { const int y = cost(bins[bw][bk]); std::swap(bins[hi][bi], bins[bw][bk]); load[hi] += y - x; load[bw] += x - y; }
                }
                std::vector<int32_t> wmap((size_t)cpl * nw, -1);
                for (int w = 0; w < nw; ++w)
                    for (int j = 0; j < (int)bins[w].size(); ++j) wmap[(size_t)j * nw + w] = bins[w][j];
                if (g->mem.upload(wmap, &sc.wave_map)) { g->mem.release(); delete g; return fail(QD_EHIP, "device allocation failed while uploading the scatter wave map"); }
            }
        }
    }

    // OSD view
    OsdGraphDev &od = g->osd;
    od.m = m; od.n = n; od.m_pad = m_pad; od.max_cdeg = max_cdeg; od.mw = (m + 63) / 64;
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    od.npow2 = np2;
    std::vector<uint32_t> csc_ptr(cp.begin(), cp.end());
    std::vector<uint16_t> csc_row(ri.begin(), ri.end());
    rc |= g->mem.upload(csc_ptr, &od.csc_ptr);
    rc |= g->mem.upload(csc_row, &od.csc_row);
    if (rc) { g->mem.release(); delete g; return fail(QD_EHIP, "device allocation/upload failed"); }
    auto carve = [&](int *offs, int qbytes, int base) {
        int o = base;
        offs[0] = o; o += align16(qbytes);
        offs[1] = o; o += align16(m_pad * 8);          // tb
        offs[2] = o; o += align16(m_pad);              // sp
        offs[3] = o; o += align16(m_pad * 2);          // rowpiv
        offs[4] = o; o += align16(m_pad * 2);          // prow
        offs[5] = o; o += align16(m_pad * 4);          // pcol
        offs[6] = o; o += align16(64 * max_cdeg * 4);  // pairs
        offs[7] = o; o += 256;                         // cols
        offs[8] = o; o += 1024;                        // red: 96 words of pivot/flag/counter scratch + 2 x 64 words of block sums
        offs[9] = o; o += align16(bp.out_words * 4);   // out
        return o;
    };
    int tmp[10];
    const int small = carve(tmp, 0, 0);
    const int q_budget = QD_LDS_BYTES - small - 64;
    const int sort_bytes = np2 * 8;
    od.threads = m <= 256 ? 256 : (m <= 512 ? 512 : 1024);
    od.f_threads = m <= 256 ? 256 : 512;       // (256 threads with 4 rows each measured slower at m = 1008: 20.4 vs 16.3 ms)
    if (const char *ev = std::getenv("QD_OSD_THREADS")) {          // tuning knob
        const int v = std::atoi(ev);
        if ((v == 256 || v == 512) && (m + v - 1) / v <= 4) od.f_threads = v;
    }
    od.lds_bytes = 0; od.kw_lds = 0; od.f_lds_bytes = 0; od.f_kw = 0;
    if (sort_bytes <= q_budget) {
        od.kw_lds = std::min(od.mw, q_budget / (m_pad * 8));
        od.lds_bytes = carve(od.off, std::max(sort_bytes, od.kw_lds * m_pad * 8), 0);
    }
    {
        // register kernel: aim at two workgroups per CU, settle for one; the tier sort buffer and the order live beside the
        // Q mirror, which must hold the kernel's 6 register planes (or all mw planes of a small window)
        const int tier = 1024;
        const int sort_b = tier * 8, order_b = align16(tier * 2);
        od.f_lds_bytes = 0; od.w_lds_bytes = 0;
        // `sweep`: higher-order OSD also keeps a pivot-column bit mask and the list of the first non-pivot columns
        auto lay = [&](int per_cu, bool sweep, int *offs, int &o_sort, int &o_order, int &o_piv, int &o_npl, int &kw) -> int {
            const int extra = sweep ? align16(bp.out_words * 4) + 256 : 0;
            const int f_budget = QD_LDS_BYTES / per_cu - 256 - small - sort_b - order_b - extra;
            kw = std::min(od.mw, std::max(0, f_budget / (m_pad * 8)));
            const int min_planes = sweep ? 6 : 2;          // the Q mirror holds at least the planes the kernel keeps in registers (QD_OSD_KWR / QD_OSD_KWR0)
            if (kw < std::min(min_planes, od.mw) || (m + od.f_threads - 1) / od.f_threads > 4) return 0;
            int o = carve(offs, kw * m_pad * 8, 0);
            o_sort = o; o += sort_b;
            o_order = o; o += order_b;
            o_piv = o; o_npl = o;                       // unused by OSD-0
            if (sweep) { o_piv = o; o += align16(bp.out_words * 4); o_npl = o; o += 256; }
            return o <= QD_LDS_BYTES / per_cu ? o : 0;
        };
        od.f_off_hist = 0;
        // three workgroups per CU where the window allows it (two Q planes in LDS, the rest in the HBM spill): most of a shot is spent
        // in single-wavefront phases, so the third workgroup is worth more than the planes (headline 10.8 -> 10.1 ms, p = 6e-3 164 -> 154 ms)
        const int per_cu0 = std::getenv("QD_OSD_PER_CU") ? std::atoi(std::getenv("QD_OSD_PER_CU")) : 2;   // (the register budget of qd_osd0_reg_kernel<512, ., false>: two per CU)
        for (int per_cu = per_cu0; per_cu >= 1 && od.f_lds_bytes == 0; --per_cu)
            od.f_lds_bytes = lay(per_cu, false, od.f_off, od.f_off_sort, od.f_off_order, od.f_off_pivmask, od.f_off_npl, od.f_kw);
        od.w_lds_bytes = lay(1, true, od.w_off, od.w_off_sort, od.w_off_order, od.w_off_pivmask, od.w_off_npl, od.w_kw);
    }
    // OSD-0 with simultaneous singleton pivots (osd_sr.hip): columns in ELL form (one load per entry, no pointer chase), its own
    // LDS layout; taken whenever the mirrored register kernel exists too (it decodes the shots the new kernel hands over)
    od.s_lds_bytes = 0; od.s_per_cu = 0; od.csc_ell = nullptr; od.ell_log2 = 0;
    if (od.f_lds_bytes > 0 && m < 65535) {
        int dl = 1;
        while ((1 << dl) < max_cdeg) ++dl;
        std::vector<uint16_t> ell((size_t)n << dl, (uint16_t)0xFFFFu);
        for (int j = 0; j < n; ++j)
            for (int e = cp[j]; e < cp[j + 1]; ++e) ell[((size_t)j << dl) + (e - cp[j])] = (uint16_t)ri[e];
        if (g->mem.upload(ell, &od.csc_ell)) { g->mem.release(); delete g; return fail(QD_EHIP, "device allocation/upload failed"); }
        od.ell_log2 = dl;
        const int lds = qd_osd_sr_layout(m, m_pad, n, bp.out_words, od.s_off, &od.s_threads, &od.s_rpt);
        if (lds > 0 && lds <= QD_LDS_BYTES) {
            od.s_lds_bytes = lds;
            od.s_per_cu = std::max(1, std::min(QD_LDS_BYTES / lds, QD_SR_WPS_OF(od.s_rpt) * 256 / od.s_threads));   // the kernel's register budget (wavefronts per SIMD)
        }
    }
    // the full kernel sorts all n columns in LDS; windows too large for that rely on the register kernel alone
    if (od.lds_bytes == 0 && od.f_lds_bytes == 0) od.threads = 0;
    if (bp.lds_bytes > QD_LDS_BYTES) {
        g->mem.release();
        const int need_lds = bp.lds_bytes;
        delete g;
        return fail(QD_ECAPACITY, "window needs %d bytes of LDS for BP state; the CU has %d", need_lds, QD_LDS_BYTES);
    }
    *out = g;
    return QD_OK;
}

extern "C" void qd_graph_destroy(qd_graph *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    g->mem.release();
    delete g;
}

static int host_rank(qd_graph *g)
{
    if (g->rank >= 0) return g->rank;
    const int m = g->m, n = g->n, mw = (m + 63) / 64;
    // column-incremental elimination on m-bit columns
    std::vector<std::vector<uint64_t>> basis;   // reduced columns with distinct leading rows
    std::vector<int> lead_of(m, -1);
    int rank = 0;
    std::vector<uint64_t> v(mw);
    for (int j = 0; j < n && rank < m; ++j) {
        std::fill(v.begin(), v.end(), 0ull);
        for (int e = g->h_cp[j]; e < g->h_cp[j + 1]; ++e) v[g->h_ri[e] >> 6] ^= 1ull << (g->h_ri[e] & 63);
        for (;;) {
            int lead = -1;
            for (int w = 0; w < mw; ++w)
                if (v[w]) { lead = w * 64 + __builtin_ctzll(v[w]); break; }
            if (lead < 0) break;
            if (lead_of[lead] < 0) { lead_of[lead] = (int)basis.size(); basis.push_back(v); ++rank; break; }
            const std::vector<uint64_t> &b = basis[lead_of[lead]];
            for (int w = 0; w < mw; ++w) v[w] ^= b[w];
        }
    }
    g->rank = rank;
    return rank;
}

// qd_graph_info fills exactly 10 entries, as it did in library version 100 (version 101 wrote 12 into the caller's buffer, an
// 8-byte overwrite for a C caller compiled against the older header); whoever wants more says how many it has room for.
static void graph_info_fill(const qd_graph *g, int32_t *info, int n_entries)
{
    const int32_t v[12] = {g->m, g->n, g->nnz, g->max_rdeg, g->max_cdeg, g->bp.threads, g->bp.lds_bytes, g->osd.threads, g->osd.lds_bytes,
                           n_entries > 9 ? host_rank(const_cast<qd_graph *>(g)) : 0, (int32_t)g->sc_walk_cycles, (int32_t)g->sc_walk_ideal};
    for (int i = 0; i < n_entries && i < 12; ++i) info[i] = v[i];
    for (int i = 12; i < n_entries; ++i) info[i] = 0;
}

extern "C" int qd_graph_info(const qd_graph *g, int32_t *info)
{
    if (!g || !info) return fail(QD_EINVAL, "null argument");
    graph_info_fill(g, info, 10);
    return QD_OK;
}

extern "C" int qd_graph_info_ex(const qd_graph *g, int32_t *info, int32_t n_entries)
{
    if (!g || !info || n_entries < 0) return fail(QD_EINVAL, "null argument or negative n_entries");
    graph_info_fill(g, info, n_entries);
    return QD_OK;
}

extern "C" int qd_decoder_create(const qd_graph *g, const qd_params *p, qd_decoder **out)
{
    if (!g || !p || !out) return fail(QD_EINVAL, "null argument");
    *out = nullptr;
    if (p->bp_method != QD_BP_MINIMUM_SUM && p->bp_method != QD_BP_PRODUCT_SUM) return fail(QD_EINVAL, "unknown bp_method %d", p->bp_method);
    if (p->schedule != QD_SCHEDULE_PARALLEL && p->schedule != QD_SCHEDULE_SERIAL) return fail(QD_EINVAL, "unknown schedule %d", p->schedule);
    if (p->reserved & ~(QD_FLAG_EDGE_MESSAGES | QD_FLAG_RAW_LLR)) return fail(QD_EINVAL, "unknown flag bits 0x%x", p->reserved);
    const bool lsd = p->osd_method == QD_LSD_0 || p->osd_method == QD_LSD_E || p->osd_method == QD_LSD_CS;
    if (lsd && p->osd_order < 0) return fail(QD_EINVAL, "negative lsd_order");
    if (p->osd_method == QD_LSD_CS && p->osd_order > 64)
        return fail(QD_EUNSUPPORTED, "lsd_cs: lsd_order %d > 64 is not implemented on the device path", p->osd_order);
    if (p->osd_method == QD_LSD_E && p->osd_order > 15)
        return fail(QD_EUNSUPPORTED, "lsd_e: lsd_order %d > 15 is not implemented on the device path", p->osd_order);
    if (lsd && qd_lsd_lds_bytes(g->m, g->n, g->bp.out_words) > QD_LDS_BYTES)
        return fail(QD_ECAPACITY, "window %d x %d does not fit the LSD kernel's LDS layout", g->m, g->n);
    const bool osd0 = lsd || p->osd_method == QD_OSD_0 || ((p->osd_method == QD_OSD_CS || p->osd_method == QD_OSD_E) && p->osd_order == 0);
    if (p->osd_method != QD_OSD_OFF && !osd0) {
        if (p->osd_method != QD_OSD_CS && p->osd_method != QD_OSD_E) return fail(QD_EINVAL, "unknown osd_method %d", p->osd_method);
        if (p->osd_order < 0) return fail(QD_EINVAL, "negative osd_order");
        if (g->osd.w_lds_bytes == 0)
            return fail(QD_EUNSUPPORTED, "osd_cs / osd_e need the register OSD kernel, which this window (%d detectors) does not fit", g->m);
        if (p->osd_method == QD_OSD_CS && p->osd_order > 64)
            return fail(QD_EUNSUPPORTED, "osd_cs: osd_order %d > 64 is not implemented on the device path", p->osd_order);
        if (p->osd_method == QD_OSD_E && p->osd_order > 15)
            return fail(QD_EUNSUPPORTED, "osd_e: osd_order %d > 15 is not implemented on the device path", p->osd_order);
    }
    if (p->osd_method != QD_OSD_OFF && !lsd && g->osd.lds_bytes == 0 && g->osd.f_lds_bytes == 0)
        return fail(QD_ECAPACITY, "window %d x %d does not fit either OSD kernel's LDS layout", g->m, g->n);
    if (p->max_iter < 0 || p->ms_scaling_factor < 0) return fail(QD_EINVAL, "negative max_iter / ms_scaling_factor");
    qd_decoder *d = new qd_decoder();
    d->g = g; d->prm = *p; d->lsd = lsd ? 1 : 0;
    d->lsd_w = (lsd && p->osd_order > 0) ? (p->osd_method == QD_LSD_CS ? 1 : (p->osd_method == QD_LSD_E ? 2 : 0)) : 0;
    d->general = (p->bp_method != QD_BP_MINIMUM_SUM || p->schedule != QD_SCHEDULE_PARALLEL || (p->reserved & QD_FLAG_EDGE_MESSAGES)) ? 1 : 0;
    d->lds_edge = (d->general && p->bp_method == QD_BP_PRODUCT_SUM && p->schedule == QD_SCHEDULE_PARALLEL && !std::getenv("QD_NO_LDS_EDGE") &&
                   qd_bp_ps_lds_bytes(g->gen, g->max_rdeg) > 0) ? 1 : 0;
    d->osd_w = osd0 || p->osd_method == QD_OSD_OFF ? 0 : (p->osd_method == QD_OSD_CS ? 1 : 2);
    if (d->osd_w) host_rank(const_cast<qd_graph *>(g));     // the sweep needs the complete factorisation: rank pivots
    if (d->prm.max_iter == 0) d->prm.max_iter = g->n;       // ldpc: max_iter = 0 -> number of bits
    if (d->prm.max_iter > QD_STATUS_ITER_MASK) d->prm.max_iter = QD_STATUS_ITER_MASK;
    d->bp_fine = g->bp; d->bp_coarse = g->bp; d->llr0_q = g->gen.llr0;
    if (p->bp_method == QD_BP_MINIMUM_SUM && p->schedule == QD_SCHEDULE_PARALLEL && p->ms_scaling_factor == 1.0 &&
        !(p->reserved & QD_FLAG_RAW_LLR)) {
        // Channel LLRs on a binary grid: see qd_decoder_info in quits_amd.h.  (oracle/qd_oracle.c restates this rule.)
        double mx = 0.0;
        for (double l : g->h_llr0) mx = std::max(mx, std::fabs(l));
        const double need = 8.0 * mx * (double)std::max(1, d->prm.max_iter);
        int e = 0;
        while (std::ldexp(1.0, e) < need && e < 40) ++e;
        // fine grid: never coarser than 2^-10, whatever max_iter (a large max_iter -- ldpc's max_iter = 0 means n -- would otherwise
        // put every shot on a grid of 1/4 or 1/8 although most shots converge long before their magnitudes get anywhere near the
        // bound); the shots that do outgrow it are certified on the grid the rule gives, by the redo pass
        const int kr = std::min(20, std::max(2, 23 - e));
        d->grid_k = std::max(kr, QD_GRID_MIN_BITS);
        d->grid_kc = kr >= QD_GRID_MIN_BITS ? std::max(0, kr - 4) : kr;
        d->grid_floor = kr < QD_GRID_MIN_BITS ? 1 : 0;
        if (hipSetDevice(g->device) != hipSuccess) { delete d; return fail(QD_EHIP, "hipSetDevice(%d) failed", g->device); }
        // (+ 0.0f: a prior that rounds to zero from below must be +0, not -0 -- the kernel's sign test reads the bit pattern)
        auto on_grid = [&](double l, int k) { return (float)std::ldexp(std::nearbyint(std::ldexp(l, k)), -k) + 0.0f; };
        int rc = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const int k = pass == 0 ? d->grid_k : d->grid_kc;
            std::vector<uint32_t> rec = g->h_bit_rec;
            for (int s = 0; s < g->n; ++s) {
                const float l0 = on_grid(g->h_llr0[g->h_bit_orig[s]], k);
                std::memcpy(&rec[(size_t)s * 4], &l0, 4);                  // word 0 of chunk 0: [chunk][slot][4]
            }
            rc |= d->mem.upload(rec, pass == 0 ? &d->bp_fine.bit_rec : &d->bp_coarse.bit_rec);
        }
        std::vector<float> lq(g->n);
        for (int j = 0; j < g->n; ++j) lq[j] = on_grid(g->h_llr0[j], d->grid_k);
        rc |= d->mem.upload(lq, &d->llr0_q);
        if (g->sc.ok && !std::getenv("QD_NO_SCATTER")) {
            // scatter kernel: fine-grid priors of the bit slots as integers (grid units) and the second-minimum bound that
            // certifies a run (bp_scatter.hip): max|prior| + max_cdeg * max min2 < 2^23
            std::vector<int32_t> pg((size_t)g->sc.nslots, 0);      // (unused and trash slots: 0)
            long long mxp = 0;
            for (int j = 0; j < g->n; ++j) {
                const long long v = std::llround(std::ldexp((double)on_grid(g->h_llr0[j], d->grid_k), d->grid_k));
                pg[g->h_sc_slot[j]] = (int32_t)v - 1;              // an accumulator holds L - 1: (L <= 0) is its sign bit
                mxp = std::max(mxp, std::llabs(v));
            }
            const long long lim = ((1ll << 23) - mxp) / std::max(1, g->max_cdeg) - 1;
            if (mxp < (1ll << 22) && lim > 0) {
                rc |= d->mem.upload(pg, &d->prior_g);
                d->m2_limit = (float)lim;
                if (const char *ev = std::getenv("QD_SCATTER_M2_LIMIT")) d->m2_limit = std::min(d->m2_limit, (float)std::atof(ev));   // test knob: force the recheck pass
                d->scatter = 1;
            }
        }
        if (rc) { d->mem.release(); delete d; return fail(QD_EHIP, "device allocation failed while building the LLR grid"); }
    }
    *out = d;
    return QD_OK;
}

extern "C" int qd_decoder_info(const qd_decoder *d, int32_t *info)
{
    if (!d || !info) return fail(QD_EINVAL, "null argument");
    info[0] = d->grid_k; info[1] = d->grid_kc; info[2] = d->general; info[3] = d->scatter ? (d->g->sc.wide_threads ? 2 : 1) : 0;
    return QD_OK;
}

extern "C" int qd_decoder_postproc_kernel(const qd_decoder *d)
{
    if (!d) return -1;
    if (d->prm.osd_method == QD_OSD_OFF) return QD_POST_NONE;
    if (d->lsd) return QD_POST_LSD;
    if (d->osd_w) {
        // (the same decision qd_decoder_reserve takes when it sizes the workspace)
        const char *ev = std::getenv("QD_OSDCS_OLD");
        int off[16], var = 0, per = 0;
        if (d->g->osd.csc_ell && !(ev && std::atoi(ev) == 1) &&
            qd_osdcs_layout(d->g->m, d->g->n, d->g->bp.out_words, d->g->osd.max_wfix, off, &var, &per) > 0) return QD_POST_OSD_CS_PANEL;
        return QD_POST_OSD_W_OLD;
    }
    const char *ev = std::getenv("QD_NO_OSD_SR");
    if (d->g->osd.s_lds_bytes > 0 && !(ev && std::atoi(ev) == 1)) return QD_POST_OSD0_SR;
    return QD_POST_OSD0_REG;
}

static void free_ws(qd_decoder *d)
{
    if (d->llr_ws) (void)hipFree(d->llr_ws);
    if (d->fail_list) (void)hipFree(d->fail_list);
    if (d->ctr_base) (void)hipFree(d->ctr_base);
    d->ctr_base = nullptr;
    if (d->host_fail) (void)hipHostFree(d->host_fail);
    if (d->fail_ready) (void)hipEventDestroy(d->fail_ready);
    d->host_fail = nullptr; d->fail_ready = nullptr; d->fail_pending = false;
    if (d->order_ws) (void)hipFree(d->order_ws);
    if (d->q_spill) (void)hipFree(d->q_spill);
    if (d->q_spill_fast) (void)hipFree(d->q_spill_fast);
    d->q_spill_fast = nullptr;
    if (d->q_spill_sr) (void)hipFree(d->q_spill_sr);
    d->q_spill_sr = nullptr;
    if (d->mt_ws) (void)hipFree(d->mt_ws);
    d->mt_ws = nullptr;
    if (d->cs_ws) (void)hipFree(d->cs_ws);
    d->cs_ws = nullptr; d->osd_blocks_cs = 0;
    if (d->gws.b2c) (void)hipFree(d->gws.b2c);
    if (d->gws.c2b) (void)hipFree(d->gws.c2b);
    if (d->gws.th) (void)hipFree(d->gws.th);
    if (d->gws.pre) (void)hipFree(d->gws.pre);
    if (d->gws.llr) (void)hipFree(d->gws.llr);
    if (d->gws.syn) (void)hipFree(d->gws.syn);
    if (d->gws.slot) (void)hipFree(d->gws.slot);
    d->gws = GenWs{};
    if (d->gsp.msg2) (void)hipFree(d->gsp.msg2);
    if (d->gsp.syn2) (void)hipFree(d->gsp.syn2);
    if (d->gsp.host_counts) (void)hipHostFree(d->gsp.host_counts);
    if (d->gsp.counts_ready) (void)hipEventDestroy(d->gsp.counts_ready);
    if (d->gsp.lists[0]) (void)hipFree(d->gsp.lists[0]);
    if (d->gsp.lists[1]) (void)hipFree(d->gsp.lists[1]);
    if (d->gsp.counts) (void)hipFree(d->gsp.counts);
    d->gsp = GenStagePlan{};
    if (d->hard_list) (void)hipFree(d->hard_list);
    if (d->hard_list2) (void)hipFree(d->hard_list2);
    if (d->redo_list) (void)hipFree(d->redo_list);
    if (d->recheck_list) (void)hipFree(d->recheck_list);
    d->recheck_list = nullptr; d->recheck_cap = 0;
    if (d->lsd_ws) (void)hipFree(d->lsd_ws);
    d->lsd_ws = nullptr;
    d->redo_list = nullptr; d->redo_cap = 0;
    d->hard_list = nullptr; d->hard_list2 = nullptr;
    d->llr_ws = nullptr; d->fail_list = nullptr; d->fail_count = nullptr; d->order_ws = nullptr; d->q_spill = nullptr;
    d->cap = 0;
}

extern "C" void qd_decoder_destroy(qd_decoder *d)
{
    if (!d) return;
    (void)hipSetDevice(d->g->device);
    free_ws(d);
    d->mem.release();
    for (auto &sp : d->ev) { (void)hipEventDestroy(sp.t0); (void)hipEventDestroy(sp.t1); }
    delete d;
}

extern "C" int qd_decoder_reserve(qd_decoder *d, int64_t max_batch)
{
    if (!d || max_batch <= 0) return fail(QD_EINVAL, "bad reserve request");
    if (max_batch <= d->cap) return QD_OK;
    HIP_TRY(hipSetDevice(d->g->device));
    HIP_TRY(hipDeviceSynchronize());
    free_ws(d);
    const qd_graph *g = d->g;
    const bool osd = d->prm.osd_method != QD_OSD_OFF;
    HIP_TRY(hipMalloc((void **)&d->ctr_base, 512));
    HIP_TRY(hipMemset(d->ctr_base, 0, 512));
    d->fail_count = d->ctr_base; d->cset = 0; d->set_clean[0] = d->set_clean[1] = true;
    HIP_TRY(hipHostMalloc((void **)&d->host_fail, 2 * sizeof(int32_t)));
    HIP_TRY(hipEventCreateWithFlags(&d->fail_ready, hipEventDisableTiming));
    if (d->grid_k >= 0 && !d->general) {
        d->redo_cap = (int)(d->grid_floor ? max_batch : std::min<int64_t>(max_batch, 4096));
        HIP_TRY(hipMalloc((void **)&d->redo_list, sizeof(int32_t) * (size_t)d->redo_cap));
        if (d->scatter) {
            d->recheck_cap = (int)max_batch;
            HIP_TRY(hipMalloc((void **)&d->recheck_list, sizeof(int32_t) * (size_t)d->recheck_cap));
        }
    }
    if (osd) {
        int ncu = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        const int per_cu = std::max(1, QD_LDS_BYTES / std::max(1, g->osd.lds_bytes));
        d->osd_blocks = g->osd.lds_bytes > 0 ? ncu * std::min(per_cu, 2048 / std::max(1, g->osd.threads)) : 0;
        const int per_cu_fast = std::max(1, QD_LDS_BYTES / std::max(1, g->osd.f_lds_bytes));
        d->osd_blocks_fast = ncu * std::min(per_cu_fast, 2048 / std::max(1, g->osd.f_threads));
        if (const char *ev = std::getenv("QD_OSD_BLOCKS_PER_CU")) {       // tuning knob
            const int v = std::atoi(ev);
            if (v > 0) d->osd_blocks_fast = ncu * v;
        }
        if (d->osd_w) d->osd_blocks_fast = ncu;                                   // higher-order OSD by row: w_* layout, one workgroup per CU
        if (d->lsd) {
            const int lds = qd_lsd_lds_bytes(g->m, g->n, g->bp.out_words);
            d->lsd_blocks = ncu * std::max(1, std::min(8, QD_LDS_BYTES / std::max(1, lds)));     // one wavefront per shot, several shots per CU
            d->lsd_ws = nullptr;
            HIP_TRY(hipMalloc((void **)&d->lsd_ws, qd_lsd_ws_bytes(g->m, g->n, d->lsd_blocks, d->lsd_w)));   // Q planes + work counter (+ debug timers) + pivot columns (+ sweep scratch)
        }
        d->osd_blocks_sr = 0;
        {
            const char *ev = std::getenv("QD_NO_OSD_SR");
            if (!d->osd_w && !d->lsd && g->osd.s_lds_bytes > 0 && !(ev && std::atoi(ev) == 1)) {
                int per = g->osd.s_per_cu;
                if (const char *e2 = std::getenv("QD_OSD_SR_PER_CU")) { const int v = std::atoi(e2); if (v > 0) per = std::min(v, QD_LDS_BYTES / g->osd.s_lds_bytes); }
                d->osd_blocks_sr = ncu * per;
                HIP_TRY(hipMalloc((void **)&d->q_spill_sr, sizeof(uint64_t) * (size_t)d->osd_blocks_sr *
                                                           qd_osd_sr_ws_words(g->osd.m_pad, g->osd.mw, g->osd.s_threads, g->osd.s_rpt)));
            }
        }
        d->osd_blocks_cs = 0;
        {
            // higher-order OSD: the rebuilt column-form kernel (osd_cs.hip) wherever its layout takes the window; QD_OSDCS_OLD=1 keeps
            // the row form (osd_kernels.hip) for A/B runs
            const char *ev = std::getenv("QD_OSDCS_OLD");
            if (d->osd_w && !d->lsd && g->osd.csc_ell && !(ev && std::atoi(ev) == 1)) {
                int per = 0;
                d->cs_lds = qd_osdcs_layout(g->m, g->n, g->bp.out_words, g->osd.max_wfix, d->cs_off, &d->cs_variant, &per);
                if (d->cs_lds > 0) {
                    if (const char *e2 = std::getenv("QD_OSDCS_PER_CU")) { const int v = std::atoi(e2); if (v > 0) per = std::min(v, QD_LDS_BYTES / d->cs_lds); }
                    d->osd_blocks_cs = ncu * per;
                    HIP_TRY(hipMalloc((void **)&d->cs_ws, sizeof(uint64_t) * (size_t)d->osd_blocks_cs * qd_osdcs_ws_words(d->cs_variant)));
                }
            }
        }
        const int spill_fast = g->osd.mw - (d->osd_w ? g->osd.w_kw : g->osd.f_kw);
        if (g->osd.f_lds_bytes > 0 && spill_fast > 0)
            HIP_TRY(hipMalloc((void **)&d->q_spill_fast, sizeof(uint64_t) * (size_t)d->osd_blocks_fast * spill_fast * g->osd.m_pad));
        if (d->osd_w && g->osd.w_lds_bytes > 0)
            HIP_TRY(hipMalloc((void **)&d->mt_ws, sizeof(uint64_t) * (size_t)d->osd_blocks_fast * ((size_t)g->osd.mw * g->osd.m_pad + 64 * 32)));   // + 64 candidate vectors of QD_SWEEP_W words
        HIP_TRY(hipMalloc((void **)&d->hard_list, sizeof(int32_t) * (size_t)max_batch));
        HIP_TRY(hipMalloc((void **)&d->hard_list2, sizeof(int32_t) * (size_t)max_batch));
        HIP_TRY(hipMalloc((void **)&d->llr_ws, sizeof(float) * (size_t)max_batch * g->bp.n_pad));
        HIP_TRY(hipMalloc((void **)&d->fail_list, sizeof(int32_t) * (size_t)max_batch));
        if (d->osd_blocks > 0)
            HIP_TRY(hipMalloc((void **)&d->order_ws, sizeof(uint16_t) * (size_t)d->osd_blocks * g->n));
        const int spill_planes = g->osd.mw - g->osd.kw_lds;
        if (d->osd_blocks > 0 && spill_planes > 0)
            HIP_TRY(hipMalloc((void **)&d->q_spill, sizeof(uint64_t) * (size_t)d->osd_blocks * spill_planes * g->osd.m_pad));
    }
    if (d->general && !d->lds_edge) {
        // [index][shot] message planes for a chunk of S shots; a batch larger than S is decoded chunk by chunk
        const bool ps = d->prm.bp_method == QD_BP_PRODUCT_SUM, serial = d->prm.schedule == QD_SCHEDULE_SERIAL;
        // edge planes: flooding b2c + c2b (+ th for product-sum); serial: messages (th or b2c) + suffixes (the c2b plane), and a row plane
        const int planes = serial ? 2 : (ps ? 3 : 2);
        const bool pre_plane = serial && g->gen.nslots == 0;       // the rows' running prefixes: LDS slots when the graph allows
        // Serial schedule in several launches with the survivors packed in between (GenStage): bounds after iterations 3, 6, 10, 14, 20, ...
        // (x ~1.4) while at least two iterations remain; QD_GEN_STAGES="3,6" sets them, QD_GEN_STAGES=0 = one launch.  Costs a second workspace.
        GenStagePlan &sp = d->gsp;
        sp = GenStagePlan{};
        if (serial) {
            static const int dflt[] = {3, 6, 10, 14, 20, 28, 40, 56, 80, 112};     // (max_iter 10: "3,6" 50.5 ms, "3,5,7" 51.4, "4" 51.4, one launch 56.0: profiles/r06_k1g_staged_ab.txt)
            std::vector<int> bnd(dflt, dflt + sizeof(dflt) / sizeof(dflt[0]));
            if (const char *ev = std::getenv("QD_GEN_STAGES")) {
                bnd.clear();
                for (const char *q = ev; *q;) {
                    char *end = nullptr;
                    const long v = std::strtol(q, &end, 10);
                    if (end == q) break;
                    if (v > 0) bnd.push_back((int)v);
                    q = *end ? end + 1 : end;
                }
            }
            int prev = 0;
            for (int b : bnd)
                if (b > prev && b + 2 <= d->prm.max_iter && sp.nbounds < QD_GEN_MAX_STAGES - 1) { sp.bounds[sp.nbounds++] = b; prev = b; }
        }
        // (the launches alternate between two message planes and two syndrome planes; suffixes, prefixes, posteriors and fail slots are scratch of
        //  ONE launch and shared.  If the second message plane would push the batch into more workspace chunks, the schedule stays in one launch:
        //  a chunk more costs a whole dependency chain, more than packing returns -- W = 3 windows, ten decoders in a 96 GB budget: 98 -> 128 ms)
        const size_t per_shot1 = ((size_t)g->nnz * planes + g->n + (pre_plane ? g->m : 0)) * sizeof(float) + g->m + sizeof(int32_t);
        size_t per_shot = per_shot1 + (sp.nbounds > 0 ? (size_t)g->nnz * sizeof(float) + g->m + 2 * sizeof(int32_t) : 0);
        // Default budget 48 GB of the 288: the kernel is latency-bound (one wavefront per 64 shots), so a launch costs about
        // the same for 8 K or 64 K shots and chunks should be as large as memory allows -- and of equal size.
        double budget_gb = 48.0;
        if (const char *ev = std::getenv("QD_GENERAL_WS_GB")) budget_gb = std::max(0.001, std::atof(ev));
        if (d->gen_ws_limit > 0) budget_gb = (double)d->gen_ws_limit / 1073741824.0;
        const auto chunks_for = [&](size_t ps) { const int64_t s_ = std::max<int64_t>(256, (int64_t)(budget_gb * 1073741824.0 / (double)ps) & ~(int64_t)255); return (max_batch + s_ - 1) / s_; };
        if (sp.nbounds > 0 && chunks_for(per_shot) > chunks_for(per_shot1)) { sp.nbounds = 0; per_shot = per_shot1; }
        int64_t S = (int64_t)(budget_gb * 1073741824.0 / (double)per_shot) & ~(int64_t)255;
        S = std::max<int64_t>(256, S);
        const int64_t nchunks = (max_batch + S - 1) / S;
        S = std::max<int64_t>(256, (((max_batch + nchunks - 1) / nchunks) + 255) & ~(int64_t)255);
        GenWs &w = d->gws;
        w.S = S;
        if (!(ps && serial)) HIP_TRY(hipMalloc((void **)&w.b2c, sizeof(float) * (size_t)g->nnz * S));
        HIP_TRY(hipMalloc((void **)&w.c2b, sizeof(float) * (size_t)g->nnz * S));
        if (ps) HIP_TRY(hipMalloc((void **)&w.th, sizeof(float) * (size_t)g->nnz * S));
        if (pre_plane) HIP_TRY(hipMalloc((void **)&w.pre, sizeof(float) * (size_t)g->m * S));
        HIP_TRY(hipMalloc((void **)&w.llr, sizeof(float) * (size_t)g->n * S));
        HIP_TRY(hipMalloc((void **)&w.syn, (size_t)g->m * S));
        HIP_TRY(hipMalloc((void **)&w.slot, sizeof(int32_t) * (size_t)S));
        if (sp.nbounds > 0) {
            HIP_TRY(hipMalloc((void **)&sp.msg2, sizeof(float) * (size_t)g->nnz * S));
            HIP_TRY(hipMalloc((void **)&sp.syn2, (size_t)g->m * S));
            sp.w2 = w;
            (ps ? sp.w2.th : sp.w2.b2c) = sp.msg2;
            sp.w2.syn = sp.syn2;
            HIP_TRY(hipMalloc((void **)&sp.lists[0], sizeof(int32_t) * (size_t)S));
            HIP_TRY(hipMalloc((void **)&sp.lists[1], sizeof(int32_t) * (size_t)S));
            HIP_TRY(hipMalloc((void **)&sp.counts, sizeof(int32_t) * (QD_GEN_MAX_STAGES + 1)));
            HIP_TRY(hipHostMalloc((void **)&sp.host_counts, sizeof(int32_t) * (QD_GEN_MAX_STAGES + 1)));
            HIP_TRY(hipEventCreateWithFlags(&sp.counts_ready, hipEventDisableTiming));
        }
    }
    d->cap = max_batch;
    return QD_OK;
}

extern "C" int qd_decoder_set_workspace_limit(qd_decoder *d, int64_t bytes)
{
    if (!d || bytes <= 0) return fail(QD_EINVAL, "bad workspace limit");
    d->gen_ws_limit = bytes;
    if (d->general && d->cap > 0) {          // takes effect at the next reserve: drop what is there
        HIP_TRY(hipSetDevice(d->g->device));
        HIP_TRY(hipDeviceSynchronize());
        free_ws(d);
    }
    return QD_OK;
}

extern "C" int qd_decoder_release_workspace(qd_decoder *d)
{
    if (!d) return fail(QD_EINVAL, "null decoder");
    HIP_TRY(hipSetDevice(d->g->device));
    HIP_TRY(hipDeviceSynchronize());
    free_ws(d);
    return QD_OK;
}

extern "C" int qd_decoder_set_profiling(qd_decoder *d, int32_t enable)
{
    if (!d) return fail(QD_EINVAL, "null decoder");
    d->profiling = enable ? 1 : 0;
    return QD_OK;
}

extern "C" int qd_decoder_profile(qd_decoder *d, double *out, int32_t reset)
{
    if (!d || !out) return fail(QD_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(d->g->device));
    for (auto &sp : d->ev) {
        HIP_TRY(hipEventSynchronize(sp.t1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, sp.t0, sp.t1));
        d->acc_ms[sp.kind] += ms; d->acc_ms[2 + sp.kind] += 1;
        (void)hipEventDestroy(sp.t0); (void)hipEventDestroy(sp.t1);
    }
    d->ev.clear();
    for (int i = 0; i < 4; ++i) out[i] = d->acc_ms[i];
    if (reset) for (int i = 0; i < 4; ++i) d->acc_ms[i] = 0;
    return QD_OK;
}

static int decode_impl(qd_decoder *d, const uint8_t *d_det, int64_t det_stride, int64_t det_offset, const uint8_t *d_upd,
                       int64_t upd_stride, int32_t upd_rows, int64_t B, uint32_t *d_err_bits, int32_t *d_status, int stage,
                       void *stream)
{
    if (!d || !d_det || !d_err_bits || !d_status) return fail(QD_EINVAL, "null argument");
    if (stage < 1 || stage > 3) return fail(QD_EINVAL, "stage must be 1 (BP), 2 (OSD) or 3 (both)");
    if (B < 0 || B > 0x7FFFFFFF) return fail(QD_EINVAL, "batch size out of range");
    if (B == 0) return QD_OK;
    if (det_offset < 0 || det_stride < det_offset + d->g->m) return fail(QD_EINVAL, "detector slice [%lld, %lld) exceeds the row stride %lld", (long long)det_offset, (long long)(det_offset + d->g->m), (long long)det_stride);
    if (d_upd && (upd_rows < 0 || upd_rows > d->g->m || upd_stride < upd_rows)) return fail(QD_EINVAL, "bad syndrome-update shape");
    HIP_TRY(hipSetDevice(d->g->device));
    if (B > d->cap) {
        if (stage == 2) return fail(QD_EINVAL, "OSD stage without a preceding BP stage of this batch size");
        int rc = qd_decoder_reserve(d, B);
        if (rc) return rc;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool osd = d->prm.osd_method != QD_OSD_OFF;
    if (stage & 1) {                                     // a new call: the other counter set (see qd_decoder::ctr_base)
        d->cset ^= 1;
        d->fail_count = d->ctr_base + 64 * d->cset;
        if (!d->set_clean[d->cset] || std::getenv("QD_COUNTER_FILLS")) {
            HIP_TRY(hipMemsetAsync(d->fail_count, 0, 3 * sizeof(int32_t), s));
            HIP_TRY(hipMemsetAsync(d->fail_count + 40, 0, 2 * sizeof(int32_t), s));
        }
        d->set_clean[d->cset] = false;
        d->last_B = B;
    }
    DecodeArgs a{};
    a.det = d_det; a.det_stride = det_stride; a.det_offset = det_offset;
    a.upd = d_upd; a.upd_stride = upd_stride; a.upd_rows = d_upd ? upd_rows : 0;
    a.max_iter = d->prm.max_iter; a.ms_scale = (float)d->prm.ms_scaling_factor; a.want_llr = osd ? 1 : 0;
    a.err_bits = d_err_bits; a.status = d_status;
    a.llr_ws = d->llr_ws; a.fail_list = d->fail_list; a.fail_count = d->fail_count;
    a.order_ws = d->order_ws; a.q_spill = d->q_spill; a.q_spill_fast = d->q_spill_fast; a.q_spill_sr = d->q_spill_sr; a.mt_ws = d->mt_ws;
    a.hard_list = d->hard_list; a.hard_list2 = d->hard_list2; a.hard_count = d->fail_count + 1;
    a.dbg = reinterpret_cast<unsigned long long *>(d->ctr_base) + 2;   // bytes 16..143 of the counter block (first set, never zeroed by a call)
    a.osd_w = d->osd_w; a.osd_order = d->prm.osd_order; a.rank = d->g->rank;
    auto span = [&](int kind, hipEvent_t &t0) -> int {
        if (!d->profiling) return QD_OK;
        hipEvent_t t1;
        HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1));
        d->ev.push_back({kind, t0, t1});
        HIP_TRY(hipEventRecord(t0, s));
        return QD_OK;
    };
    if (stage & 1) {
        int32_t *redo_count = d->fail_count + 40;       // bytes 160..163 of the counter set (16..143 of the first set are the debug counters)
        hipEvent_t t0 = nullptr;
        if (int rc = span(0, t0)) return rc;
        if (d->lds_edge) {
            GenGraphDev gg = d->g->gen;
            gg.llr0 = d->llr0_q;
            HIP_TRY(qd_launch_bp_ps_lds(gg, d->g->bp, a, B, s));
        } else if (d->general) {
            GenGraphDev gg = d->g->gen;
            gg.llr0 = d->llr0_q;
            for (int64_t b0 = 0; b0 < B; b0 += d->gws.S)
                HIP_TRY(qd_launch_bp_general(gg, d->g->bp, a, d->gws, d->prm.bp_method, d->prm.schedule, b0,
                                             (int)std::min<int64_t>(d->gws.S, B - b0), s, d->gsp.nbounds > 0 ? &d->gsp : nullptr));
        } else if (d->grid_k >= 0) {
            // grid arithmetic: first pass on the fine grid parks the shots whose exactness bound tripped; they are decoded
            // again on the coarse grid by a second launch (one workgroup per parked shot; the others exit at once)
            DecodeArgs a1 = a;
            a1.s_limit = std::ldexp(1.0f, 23 - d->grid_k);
            a1.redo_list = d->redo_list; a1.redo_count = redo_count; a1.redo_cap = d->redo_cap;
            if (d->scatter) {
                // scatter kernel first; the shots its (looser) bound cannot certify are decoded again by the gather kernel,
                // which carries the per-fault bound the coarse-grid rule is stated on
                int32_t *recheck_count = d->fail_count + 41;
                ScatArgs x{};
                x.prior_g = d->prior_g; x.grid_inv = std::ldexp(1.0f, -d->grid_k); x.m2_limit = d->m2_limit;
                x.recheck_list = d->recheck_list; x.recheck_count = recheck_count; x.recheck_cap = d->recheck_cap;
                if (d->g->sc.wide_threads) HIP_TRY(qd_launch_bp_scatter_wide(d->bp_fine, d->g->sc, a1, x, B, s));
                else HIP_TRY(qd_launch_bp_scatter(d->bp_fine, d->g->sc, a1, x, B, s));
                a1.shot_list = d->recheck_list; a1.shot_count = recheck_count;
                HIP_TRY(qd_launch_bp(d->bp_fine, a1, std::min<int64_t>(B, d->recheck_cap), s));
            } else
                HIP_TRY(qd_launch_bp(d->bp_fine, a1, B, s));
            DecodeArgs a2 = a;
            a2.s_limit = std::ldexp(1.0f, 23 - d->grid_kc);
            a2.shot_list = d->redo_list; a2.shot_count = redo_count; a2.status_or = QD_STATUS_COARSE_GRID;
            HIP_TRY(qd_launch_bp(d->bp_coarse, a2, std::min<int64_t>(B, d->redo_cap), s));
        } else
            HIP_TRY(qd_launch_bp(d->g->bp, a, B, s));
        if (d->profiling) HIP_TRY(hipEventRecord(d->ev.back().t1, s));
        if (std::getenv("QD_DEBUG_SYNC")) { std::fprintf(stderr, "[qd] BP stage queued (B = %lld)\n", (long long)B); HIP_TRY(hipStreamSynchronize(s)); std::fprintf(stderr, "[qd] BP stage done\n"); }
    }
    if ((stage & 2) && osd) {
        hipEvent_t t0 = nullptr;
        if (int rc = span(1, t0)) return rc;
        if (d->lsd)
            HIP_TRY(qd_launch_lsd0(d->g->gen, d->g->bp, a, d->lsd_ws, d->lsd_blocks, (int)std::min<int64_t>(B, d->lsd_blocks), d->lsd_w,
                                   d->prm.osd_order, d->g->osd.wfix, s));
        else if (d->osd_blocks_cs > 0)
            HIP_TRY(qd_launch_osdcs(d->g->osd, d->g->bp, a, d->cs_off, d->cs_variant, d->cs_lds, d->cs_ws,
                                    (int)std::min<int64_t>(B, d->osd_blocks_cs), s));
        else if (d->osd_blocks_sr > 0) {
            // OSD-0: many pivots per round (osd_sr.hip); shots whose syndrome is outside the column space come back on the hard list
            // and are decoded by the one-pivot-per-round kernel, whose lowest-row rule defines their answer
            HIP_TRY(qd_launch_osd0_sr(d->g->osd, d->g->bp, a, (int)std::min<int64_t>(B, d->osd_blocks_sr), s));
            HIP_TRY(qd_launch_osd0(d->g->osd, d->g->bp, a, (int)std::min<int64_t>(B, d->osd_blocks_fast),
                                   (int)std::min<int64_t>(B, d->osd_blocks), s, true));
        } else
            HIP_TRY(qd_launch_osd0(d->g->osd, d->g->bp, a, (int)std::min<int64_t>(B, d->osd_blocks_fast),
                                   (int)std::min<int64_t>(B, d->osd_blocks), s));
        if (d->profiling) HIP_TRY(hipEventRecord(d->ev.back().t1, s));
        if (std::getenv("QD_DEBUG_SYNC")) { std::fprintf(stderr, "[qd] post-processing stage queued (B = %lld)\n", (long long)B); HIP_TRY(hipStreamSynchronize(s)); std::fprintf(stderr, "[qd] post-processing stage done\n"); }
    }
    if ((stage & 2) && osd && !d->fail_pending && d->host_fail) {      // (see qd_decoder::host_fail)
        HIP_TRY(hipMemcpyAsync(d->host_fail, d->fail_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipEventRecord(d->fail_ready, s));
        d->fail_pending = true; d->fail_pending_B = d->last_B;
    }
    if ((stage & 2) && !d->set_clean[d->cset ^ 1]) {     // the next call's counters, zeroed behind this call's last stage (nothing waits for these fills)
        int32_t *other = d->ctr_base + 64 * (d->cset ^ 1);
        HIP_TRY(hipMemsetAsync(other, 0, 3 * sizeof(int32_t), s));
        HIP_TRY(hipMemsetAsync(other + 40, 0, 2 * sizeof(int32_t), s));
        d->set_clean[d->cset ^ 1] = true;
    }
    return QD_OK;
}

// See include/quits_amd.h.  Heavy = the post-processor will want whole CUs for about as long as a BP stage or longer: OSD-CS / OSD-E (two workgroups of
// 78 KB of LDS per CU), BP-LSD, and OSD-0 when at least QD_POST_HEAD_FRAC (0.75) of the batch failed (headline p = 3e-3: 48 %, p = 5e-3: 95 %).  Measured, same box, 30 us
// against none: OSD-CS(1) 361 -> 482 k shots/s, lsd_cs(1) 888 -> 940 k, p = 6e-3 488 -> 569 k, W = 5 / F = 3 1.125 -> 1.153 M; the headline (OSD-0 over
// 48 % of the shots, 4.4 ms beside 40 ms of BP) is the one that loses, 1.552 -> 1.539 M -- even to an EMPTY launch at this point of the BP stream, so an
// OSD-0 decoder launches nothing unless an earlier call's failure count (qd_decoder::host_fail) says "heavy" (profiles/r06_post_head_start.txt).
extern "C" int qd_decoder_post_head_start(qd_decoder *d, int32_t microseconds, void *stream)
{
    if (!d) return fail(QD_EINVAL, "null decoder");
    if (microseconds < 0) microseconds = 50;
    if (const char *ev = std::getenv("QD_POST_HEAD_START_US")) microseconds = std::atoi(ev);
    if (microseconds <= 0 || d->prm.osd_method == QD_OSD_OFF || !d->ctr_base || d->last_B <= 0) return QD_OK;
    microseconds = std::min(microseconds, 5000);
    HIP_TRY(hipSetDevice(d->g->device));
    const bool always = d->lsd || d->osd_blocks_cs > 0;
    double frac = 0.75;
    if (const char *ev = std::getenv("QD_POST_HEAD_FRAC")) frac = std::atof(ev);
    if (d->fail_pending && qd_event_done(d->fail_ready)) {
        d->fail_pending = false;
        if (d->fail_pending_B > 0) d->fail_frac_hint = (double)d->host_fail[0] / (double)d->fail_pending_B;
    }
    if (std::getenv("QD_DEBUG_HEAD")) std::fprintf(stderr, "[qd] head start: decoder %p always %d hint %.3f host_fail %d pending_B %lld last_B %lld\n", (void *)d, (int)always, d->fail_frac_hint, d->host_fail ? d->host_fail[0] : -1, (long long)d->fail_pending_B, (long long)d->last_B);
    if (!always && d->fail_frac_hint < 0.8 * frac) return QD_OK;        // OSD-0 over a minority of the shots (or nothing known yet): no launch at all
    const int threshold = (int)std::min<double>(2147483647.0, std::max(1.0, frac * (double)d->last_B));
    HIP_TRY(qd_launch_hold(always ? nullptr : d->fail_count, threshold, microseconds, reinterpret_cast<hipStream_t>(stream)));
    return QD_OK;
}

extern "C" int qd_decode_batch(qd_decoder *d, const uint8_t *d_det, int64_t det_stride, int64_t det_offset,
                               const uint8_t *d_upd, int64_t upd_stride, int32_t upd_rows, int64_t B,
                               uint32_t *d_err_bits, int32_t *d_status, void *stream)
{
    return decode_impl(d, d_det, det_stride, det_offset, d_upd, upd_stride, upd_rows, B, d_err_bits, d_status, 3, stream);
}

extern "C" int qd_decode_stage(qd_decoder *d, const uint8_t *d_det, int64_t det_stride, int64_t det_offset,
                               const uint8_t *d_upd, int64_t upd_stride, int32_t upd_rows, int64_t B,
                               uint32_t *d_err_bits, int32_t *d_status, int32_t stage, void *stream)
{
    return decode_impl(d, d_det, det_stride, det_offset, d_upd, upd_stride, upd_rows, B, d_err_bits, d_status, stage, stream);
}

extern "C" int qd_osd0_batch(qd_decoder *d, const uint8_t *d_det, int64_t det_stride, int64_t det_offset,
                             const uint8_t *d_upd, int64_t upd_stride, int32_t upd_rows, int64_t B, const float *d_llr,
                             uint32_t *d_err_bits, int32_t *d_status, void *stream)
{
    if (!d || !d_det || !d_llr || !d_err_bits || !d_status) return fail(QD_EINVAL, "null argument");
    if (d->prm.osd_method == QD_OSD_OFF) return fail(QD_EINVAL, "decoder was created with osd_method = off");
    const bool lsd_only = d->lsd != 0;
    if (B < 0 || B > 0x7FFFFFFF) return fail(QD_EINVAL, "batch size out of range");
    if (B == 0) return QD_OK;
    if (det_offset < 0 || det_stride < det_offset + d->g->m) return fail(QD_EINVAL, "detector slice exceeds the row stride");
    if (d_upd && (upd_rows < 0 || upd_rows > d->g->m || upd_stride < upd_rows)) return fail(QD_EINVAL, "bad syndrome-update shape");
    HIP_TRY(hipSetDevice(d->g->device));
    if (B > d->cap) {
        int rc = qd_decoder_reserve(d, B);
        if (rc) return rc;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DecodeArgs a{};
    a.det = d_det; a.det_stride = det_stride; a.det_offset = det_offset;
    a.upd = d_upd; a.upd_stride = upd_stride; a.upd_rows = d_upd ? upd_rows : 0;
    a.max_iter = d->prm.max_iter; a.ms_scale = (float)d->prm.ms_scaling_factor; a.want_llr = 1;
    a.err_bits = d_err_bits; a.status = d_status;
    a.llr_ws = d->llr_ws; a.fail_list = d->fail_list; a.fail_count = d->fail_count;
    a.order_ws = d->order_ws; a.q_spill = d->q_spill; a.q_spill_fast = d->q_spill_fast; a.q_spill_sr = d->q_spill_sr; a.mt_ws = d->mt_ws;
    a.hard_list = d->hard_list; a.hard_list2 = d->hard_list2; a.hard_count = d->fail_count + 1;
    a.dbg = reinterpret_cast<unsigned long long *>(d->ctr_base) + 2;
    a.osd_w = d->osd_w; a.osd_order = d->prm.osd_order; a.rank = d->g->rank;
    HIP_TRY(hipMemsetAsync(d->fail_count, 0, 3 * sizeof(int32_t), s));
    d->set_clean[d->cset] = false;
    HIP_TRY(qd_launch_stage_llr(d_llr, d->g->n, d->g->bp.n_pad, d->g->bp.bit_orig, B, d->llr_ws, d->fail_list, d->fail_count,
                                d_status, s));
    if (lsd_only)
        HIP_TRY(qd_launch_lsd0(d->g->gen, d->g->bp, a, d->lsd_ws, d->lsd_blocks, (int)std::min<int64_t>(B, d->lsd_blocks), d->lsd_w,
                                   d->prm.osd_order, d->g->osd.wfix, s));
    else if (d->osd_blocks_cs > 0)
        HIP_TRY(qd_launch_osdcs(d->g->osd, d->g->bp, a, d->cs_off, d->cs_variant, d->cs_lds, d->cs_ws,
                                (int)std::min<int64_t>(B, d->osd_blocks_cs), s));
    else if (d->osd_blocks_sr > 0) {
        HIP_TRY(qd_launch_osd0_sr(d->g->osd, d->g->bp, a, (int)std::min<int64_t>(B, d->osd_blocks_sr), s));
        HIP_TRY(qd_launch_osd0(d->g->osd, d->g->bp, a, (int)std::min<int64_t>(B, d->osd_blocks_fast),
                               (int)std::min<int64_t>(B, d->osd_blocks), s, true));
    } else
        HIP_TRY(qd_launch_osd0(d->g->osd, d->g->bp, a, (int)std::min<int64_t>(B, d->osd_blocks_fast),
                               (int)std::min<int64_t>(B, d->osd_blocks), s));
    return QD_OK;
}

extern "C" int qd_decoder_debug_counters(qd_decoder *d, uint64_t *out16)
{
    if (!d || !out16 || !d->fail_count) return fail(QD_EINVAL, "no workspace yet");
    HIP_TRY(hipSetDevice(d->g->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, reinterpret_cast<char *>(d->ctr_base) + 16, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(reinterpret_cast<char *>(d->ctr_base) + 16, 0, 16 * sizeof(uint64_t)));
    return QD_OK;
}

extern "C" int qd_decoder_failed_llr(qd_decoder *d, int64_t b, float *d_out, void *stream)
{
    if (!d || !d_out || !d->llr_ws) return fail(QD_EINVAL, "no posterior workspace (OSD off or nothing decoded yet)");
    HIP_TRY(hipSetDevice(d->g->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipStreamSynchronize(s));
    int32_t nfail = 0;
    HIP_TRY(hipMemcpy(&nfail, d->fail_count, sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> list((size_t)std::max(nfail, 1));
    if (nfail > 0) HIP_TRY(hipMemcpy(list.data(), d->fail_list, sizeof(int32_t) * (size_t)nfail, hipMemcpyDeviceToHost));
    for (int i = 0; i < nfail; ++i)
        if (list[i] == (int32_t)b) {
            // workspace rows are in bit-slot order; hand back fault order
            const qd_graph *g = d->g;
            std::vector<float> slot(g->bp.n_pad), outv(g->n);
            std::vector<uint32_t> orig(g->bp.n_pad);
            HIP_TRY(hipMemcpy(slot.data(), d->llr_ws + (size_t)i * g->bp.n_pad, sizeof(float) * g->bp.n_pad, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(orig.data(), g->bp.bit_orig, sizeof(uint32_t) * g->bp.n_pad, hipMemcpyDeviceToHost));
            for (int sidx = 0; sidx < g->n; ++sidx) outv[orig[sidx]] = slot[sidx];
            HIP_TRY(hipMemcpy(d_out, outv.data(), sizeof(float) * g->n, hipMemcpyHostToDevice));
            return QD_OK;
        }
    return fail(QD_EINVAL, "shot %lld converged (no stored posterior)", (long long)b);
}

extern "C" int qd_spmat_create(int32_t nrows, int32_t ncols, const int32_t *row_ptr, const int32_t *col_idx,
                               int32_t device, qd_spmat **out)
{
    if (!out) return fail(QD_EINVAL, "out is null");
    *out = nullptr;
    if (nrows < 0 || ncols < 0 || !row_ptr) return fail(QD_EINVAL, "bad shape");
    const int nnz = row_ptr[nrows];
    if (nnz > 0 && !col_idx) return fail(QD_EINVAL, "null col_idx");
    for (int e = 0; e < nnz; ++e)
        if (col_idx[e] < 0 || col_idx[e] >= ncols) return fail(QD_EINVAL, "column index out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(QD_EHIP, "hipSetDevice(%d) failed", device);
    qd_spmat *s = new qd_spmat();
    s->device = device;
    std::vector<uint32_t> rp(row_ptr, row_ptr + nrows + 1), ci(col_idx, col_idx + nnz);
    int rc = s->mem.upload(rp, &s->d.row_ptr) | s->mem.upload(ci, &s->d.col_idx);
    if (rc) { s->mem.release(); delete s; return fail(QD_EHIP, "device allocation/upload failed"); }
    s->d.nrows = nrows; s->d.ncols = ncols; s->d.nnz = nnz;
    s->d.colmask = nullptr; s->d.mask_words = 0;
    if (nrows > 0 && nrows <= 512 && ncols > 0) {
        const int mw = (nrows + 31) / 32;
        std::vector<uint32_t> cm((size_t)ncols * mw, 0u);
        for (int r = 0; r < nrows; ++r)
            for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) cm[(size_t)col_idx[e] * mw + (r >> 5)] ^= 1u << (r & 31);
        if (s->mem.upload(cm, &s->d.colmask)) { s->mem.release(); delete s; return fail(QD_EHIP, "device allocation/upload failed"); }
        s->d.mask_words = mw;
    }
    *out = s;
    return QD_OK;
}

extern "C" void qd_spmat_destroy(qd_spmat *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    s->mem.release();
    delete s;
}

extern "C" int qd_gf2_spmv_batch(const qd_spmat *A, const uint32_t *d_err_bits, int64_t err_stride_words, int64_t B,
                                 uint8_t *d_out, int64_t out_stride, int32_t accumulate, void *stream)
{
    if (!A || !d_err_bits || !d_out) return fail(QD_EINVAL, "null argument");
    if (err_stride_words * 32 < A->d.ncols) return fail(QD_EINVAL, "error rows hold %lld bits, matrix has %d columns", (long long)err_stride_words * 32, A->d.ncols);
    if (out_stride < A->d.nrows) return fail(QD_EINVAL, "out_stride smaller than the row count");
    HIP_TRY(hipSetDevice(A->device));
    HIP_TRY(qd_launch_spmv(A->d, d_err_bits, err_stride_words, B, d_out, out_stride, accumulate, reinterpret_cast<hipStream_t>(stream)));
    return QD_OK;
}

extern "C" int qd_unpack_bits(const uint32_t *d_bits, int64_t stride_words, int32_t nbits, int64_t B, uint8_t *d_out,
                              int64_t out_stride, void *stream)
{
    if (!d_bits || !d_out || nbits < 0 || stride_words * 32 < nbits || out_stride < nbits) return fail(QD_EINVAL, "bad unpack arguments");
    HIP_TRY(qd_launch_unpack(d_bits, stride_words, nbits, B, d_out, out_stride, reinterpret_cast<hipStream_t>(stream)));
    return QD_OK;
}

extern "C" int qd_count_mismatch(const uint8_t *d_pred, const uint8_t *d_obs, int32_t k, int64_t B, int64_t *d_count,
                                 void *stream)
{
    if (!d_pred || !d_obs || !d_count || k <= 0) return fail(QD_EINVAL, "bad count arguments");
    HIP_TRY(qd_launch_count(d_pred, d_obs, k, B, d_count, reinterpret_cast<hipStream_t>(stream)));
    return QD_OK;
}

extern "C" int qd_sample_dem(const qd_spmat *Ht, const qd_spmat *Lt, const double *priors, uint64_t seed, int64_t shot0,
                             int64_t B, uint8_t *d_det, int64_t det_stride, uint8_t *d_obs, int64_t obs_stride,
                             void *stream)
{
    if (!Ht || !Lt || !priors || !d_det || !d_obs) return fail(QD_EINVAL, "null argument");
    if (Ht->d.nrows != Lt->d.nrows) return fail(QD_EINVAL, "Ht and Lt must both have one row per fault");
    const int n = Ht->d.nrows, m = Ht->d.ncols, nobs = Lt->d.ncols;
    if (det_stride < m || obs_stride < nobs) return fail(QD_EINVAL, "output strides too small");
    if (((m + 31) / 32 + (nobs + 31) / 32) * 4 > 64 * 1024) return fail(QD_ECAPACITY, "too many detectors for the sampler's LDS bit array");
    HIP_TRY(hipSetDevice(Ht->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    std::vector<uint32_t> thr((size_t)n);
    for (int j = 0; j < n; ++j) {
        double t = std::floor(priors[j] * 4294967296.0);
        thr[j] = (uint32_t)std::min(std::max(t, 0.0), 4294967295.0);
    }
    uint32_t *d_thr = nullptr;
    HIP_TRY(hipMalloc((void **)&d_thr, sizeof(uint32_t) * (size_t)std::max(n, 4)));
    hipError_t e = hipMemcpy(d_thr, thr.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = qd_launch_sample(Ht->d, Lt->d, d_thr, seed, shot0, B, m, nobs, d_det, det_stride, d_obs, obs_stride, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_thr);
    if (e != hipSuccess) return fail(QD_EHIP, "sampler: %s", hipGetErrorString(e));
    return QD_OK;
}
