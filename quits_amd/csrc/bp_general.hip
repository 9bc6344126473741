// bp_general.hip -- belief propagation with one message per edge, for every option pair the reference's wrapper can ask
// for: bp_method in {product_sum, minimum_sum} x schedule in {parallel, serial} (quits/decoder/bposd.py:27-28; the
// wrapper's defaults are product_sum + serial, bposd.py:54).  The flooding min-sum pair normally runs in the compressed
// LDS kernel (bp_kernels.hip); this kernel serves the other three pairs, and flooding min-sum on request.
//
// Restates ldpc 2.x src_cpp/bp.hpp (bp_decode_parallel / bp_decode_serial) in float; CPU mirror, bit for bit:
// bp_parallel_edge_f32 / bp_serial_edge_f32 (oracle/bp_core.inc, form OQ_FORM_LDPC_F32).
//
// Mapping: one LANE per shot.  All 64 shots of a wavefront walk the same Tanner graph, so
//   - the adjacency (row_ptr, col_idx, ...) is read with scalar loads and every loop has a uniform trip count;
//   - message arrays are laid out [edge][shot]: a wavefront's access to one edge is one contiguous 256-byte line;
//   - the serial schedule, sequential over the faults of ONE shot, still runs 64 shots per wavefront in lockstep.
// Messages live in HBM (no LDS, no barriers): this is the bandwidth-bound formulation of SURVEY.md 8(d).  A shot that has
// converged leaves the loop; its lanes idle until the wavefront's slowest shot is done.
//
// tanh(x/2) and log((1+c)/(1-c)) are the fixed-operation-order float functions of qd_math.h (shared with the CPU mirror).
// The serial product-sum schedule keeps tanh(b2c/2) of every edge beside the message (bp.hpp re-evaluates it once per
// use, row weight times per sweep; a pure function of an unchanged argument, so the cached value is the same number).
#include "qd_internal.h"
#include "qd_math.h"
#include "../../include/quits_amd.h"

#define QD_GEN_THREADS 256

template <int METHOD, int SCHED>
__global__ void __launch_bounds__(QD_GEN_THREADS) qd_bp_edge_kernel(GenGraphDev g, DecodeArgs a, GenWs w, int64_t shot0, int nshots)
{
    const int ls = blockIdx.x * QD_GEN_THREADS + threadIdx.x;       // shot inside this chunk = column of the workspace
    if (ls >= nshots) return;
    const int64_t shot = shot0 + ls;
    const size_t S = (size_t)w.S;
    float *__restrict__ b2c = w.b2c + ls;                            // [nnz][S], CSR edge order
    float *__restrict__ c2b = w.c2b + ls;
    float *__restrict__ th = w.th ? w.th + ls : nullptr;             // product-sum only
    float *__restrict__ llr = w.llr + ls;                            // [n][S]
    uint8_t *__restrict__ syn = w.syn + ls;                          // [m][S]
    const float BIG = 3.402823466e+38f;

    // ---- window syndrome (sliding_window.py:168-169)
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    uint32_t any = 0;
    for (int i = 0; i < g.m; ++i) {
        uint32_t s = det[i] & 1u;
        if (upd && i < a.upd_rows) s ^= upd[i] & 1u;
        syn[(size_t)i * S] = (uint8_t)s;
        any |= s;
    }
    uint32_t *out = a.err_bits + shot * g.out_words;
    w.slot[ls] = -1;
    if (!any) {   // bposd_decoder.pyx: an all-zero syndrome returns the zero vector without running BP
        for (int x = 0; x < g.out_words; ++x) out[x] = 0u;
        a.status[shot] = (1 << 16) | (1 << 19);
        return;
    }

    // ---- every bit->check message starts as the prior LLR
    for (int j = 0; j < g.n; ++j) {
        const float l0 = g.llr0[j];
        for (int e = g.cp[j]; e < g.cp[j + 1]; ++e) {
            const size_t ce = (size_t)g.c2r[e] * S;
            b2c[ce] = l0;
            if (METHOD == QD_BP_PRODUCT_SUM && SCHED == QD_SCHEDULE_SERIAL) th[ce] = qd_tanh_half(l0);
        }
    }

    int iters = 0, converged = 0;
    for (int it = 1; it <= a.max_iter; ++it) {
        const float alpha = (a.ms_scale == 0.f) ? (1.0f - ldexpf(1.0f, -it)) : a.ms_scale;
        if (SCHED == QD_SCHEDULE_PARALLEL) {
            // ---- check pass: forward / backward exclusive sweeps over each row (bp.hpp)
            for (int i = 0; i < g.m; ++i) {
                const int r0 = g.rp[i], r1 = g.rp[i + 1];
                const uint32_t si = syn[(size_t)i * S];
                if (METHOD == QD_BP_PRODUCT_SUM) {
                    const float sgn = si ? -1.0f : 1.0f;
                    float temp = 1.0f;
                    for (int e = r0; e < r1; ++e) {
                        const float t = qd_tanh_half(b2c[(size_t)e * S]);
                        th[(size_t)e * S] = t;
                        c2b[(size_t)e * S] = temp;
                        temp = temp * t;
                    }
                    temp = 1.0f;
                    for (int e = r1 - 1; e >= r0; --e) {
                        const float c = c2b[(size_t)e * S] * temp;
                        c2b[(size_t)e * S] = sgn * qd_log_ratio(c);
                        temp = temp * th[(size_t)e * S];
                    }
                } else {
                    int total_sgn = (int)si;
                    float temp = BIG;
                    for (int e = r0; e < r1; ++e) {
                        const float v = b2c[(size_t)e * S];
                        if (v <= 0.f) total_sgn += 1;
                        c2b[(size_t)e * S] = temp;
                        const float av = fabsf(v);
                        if (av < temp) temp = av;
                    }
                    temp = BIG;
                    for (int e = r1 - 1; e >= r0; --e) {
                        const float v = b2c[(size_t)e * S];
                        int sgn = total_sgn;
                        if (v <= 0.f) sgn += 1;
                        float c = c2b[(size_t)e * S];
                        if (temp < c) c = temp;
                        const float msign = (sgn % 2 == 0) ? 1.0f : -1.0f;
                        c2b[(size_t)e * S] = c * (msign * alpha);
                        const float av = fabsf(v);
                        if (av < temp) temp = av;
                    }
                }
            }
            // ---- bit pass: posterior, then the prefix / suffix sums that make the outgoing messages
            for (int j = 0; j < g.n; ++j) {
                const int c0 = g.cp[j], c1 = g.cp[j + 1];
                float temp = g.llr0[j];
                for (int e = c0; e < c1; ++e) {
                    const size_t ce = (size_t)g.c2r[e] * S;
                    b2c[ce] = temp;
                    temp += c2b[ce];
                }
                llr[(size_t)j * S] = temp;
                temp = 0.f;
                for (int e = c1 - 1; e >= c0; --e) {
                    const size_t ce = (size_t)g.c2r[e] * S;
                    b2c[ce] += temp;
                    temp += c2b[ce];
                }
            }
        } else {
            // ---- serial schedule: faults in natural order, each one refreshing its incoming messages first
            for (int j = 0; j < g.n; ++j) {
                const int c0 = g.cp[j], c1 = g.cp[j + 1];
                float lj = g.llr0[j];
                for (int e = c0; e < c1; ++e) {
                    const int i = g.ri[e], own = g.c2r[e];
                    const int r0 = g.rp[i], r1 = g.rp[i + 1];
                    const uint32_t si = syn[(size_t)i * S];
                    float c;
                    if (METHOD == QD_BP_PRODUCT_SUM) {
                        float t = 1.0f;
                        for (int f = r0; f < r1; ++f)
                            if (f != own) t = t * th[(size_t)f * S];
                        c = (si ? -1.0f : 1.0f) * qd_log_ratio(t);
                    } else {
                        int sgn = (int)si;
                        float t = BIG;
                        for (int f = r0; f < r1; ++f)
                            if (f != own) {
                                const float v = b2c[(size_t)f * S];
                                const float av = fabsf(v);
                                if (av < t) t = av;
                                if (v <= 0.f) sgn += 1;
                            }
                        c = alpha * ((sgn % 2 == 0) ? 1.0f : -1.0f) * t;
                    }
                    c2b[(size_t)own * S] = c;
                    b2c[(size_t)own * S] = lj;
                    lj += c;
                }
                llr[(size_t)j * S] = lj;
                float temp = 0.f;
                for (int e = c1 - 1; e >= c0; --e) {
                    const size_t ce = (size_t)g.c2r[e] * S;
                    const float v = b2c[ce] + temp;
                    b2c[ce] = v;
                    if (METHOD == QD_BP_PRODUCT_SUM) th[ce] = qd_tanh_half(v);
                    temp += c2b[ce];
                }
            }
        }
        // ---- stop when the hard decision reproduces the syndrome
        uint32_t bad = 0;
        for (int i = 0; i < g.m; ++i) {
            uint32_t p = syn[(size_t)i * S];
            for (int e = g.rp[i]; e < g.rp[i + 1]; ++e) p ^= (llr[(size_t)g.ci[e] * S] <= 0.f) ? 1u : 0u;
            bad |= p;
        }
        iters = it;
        if (!bad) { converged = 1; break; }
    }

    // ---- hard decision, packed by fault index
    for (int x = 0; x < g.out_words; ++x) {
        uint32_t word = 0u;
        const int j1 = min(g.n, 32 * x + 32);
        for (int j = 32 * x; j < j1; ++j) word |= ((llr[(size_t)j * S] <= 0.f) ? 1u : 0u) << (j & 31);
        out[x] = word;
    }
    a.status[shot] = iters | (converged << 16);
    if (!converged && a.want_llr) {
        const int slot = atomicAdd(a.fail_count, 1);
        a.fail_list[slot] = (int32_t)shot;
        w.slot[ls] = slot;
    }
}

// Posteriors of the shots BP could not finish, from [fault][shot] to the OSD workspace's [fail slot][bit slot] rows.
// 64 x 64 tiles through LDS so that both sides move whole lines.
__global__ void __launch_bounds__(256) qd_publish_llr_kernel(const float *__restrict__ llr, const int32_t *__restrict__ slot,
                                                             int64_t S, int nshots, int n, int n_pad,
                                                             const uint32_t *__restrict__ bit_orig, float *__restrict__ llr_ws)
{
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int sb0 = blockIdx.x * 64, sh0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {
        const int sb = sb0 + r;                                       // bit slot (uniform across the wavefront)
        float v = 0.f;
        if (sb < n && sh0 + tx < nshots) v = llr[(size_t)bit_orig[sb] * S + sh0 + tx];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int sh = sh0 + r;
        if (sh >= nshots) continue;
        const int sl = slot[sh];
        if (sl >= 0 && sb0 + tx < n) llr_ws[(size_t)sl * n_pad + sb0 + tx] = tile[tx][r];
    }
}

template <int METHOD, int SCHED>
static hipError_t launch_k(const GenGraphDev &g, const DecodeArgs &a, const GenWs &w, int64_t shot0, int nshots, hipStream_t s)
{
    hipLaunchKernelGGL((qd_bp_edge_kernel<METHOD, SCHED>), dim3((unsigned)((nshots + QD_GEN_THREADS - 1) / QD_GEN_THREADS)),
                       dim3(QD_GEN_THREADS), 0, s, g, a, w, shot0, nshots);
    return hipGetLastError();
}

hipError_t qd_launch_bp_general(const GenGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, const GenWs &w, int bp_method,
                                int schedule, int64_t shot0, int nshots, hipStream_t s)
{
    hipError_t e;
    if (bp_method == QD_BP_PRODUCT_SUM)
        e = schedule == QD_SCHEDULE_PARALLEL ? launch_k<QD_BP_PRODUCT_SUM, QD_SCHEDULE_PARALLEL>(g, a, w, shot0, nshots, s)
                                             : launch_k<QD_BP_PRODUCT_SUM, QD_SCHEDULE_SERIAL>(g, a, w, shot0, nshots, s);
    else
        e = schedule == QD_SCHEDULE_PARALLEL ? launch_k<QD_BP_MINIMUM_SUM, QD_SCHEDULE_PARALLEL>(g, a, w, shot0, nshots, s)
                                             : launch_k<QD_BP_MINIMUM_SUM, QD_SCHEDULE_SERIAL>(g, a, w, shot0, nshots, s);
    if (e != hipSuccess || !a.want_llr) return e;
    hipLaunchKernelGGL(qd_publish_llr_kernel, dim3((unsigned)((g.n + 63) / 64), (unsigned)((nshots + 63) / 64)), dim3(256), 0, s,
                       w.llr, w.slot, w.S, nshots, g.n, bg.n_pad, bg.bit_orig, a.llr_ws);
    return hipGetLastError();
}
