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
// Messages live in HBM: this is the bandwidth-bound formulation of SURVEY.md 8(d).  A shot that has converged goes idle;
// its lanes wait until the slowest of the 64 shots is done.
// Flooding schedule: rows (then columns) are independent, so G wavefronts of one workgroup share the same 64 shots and
// take every G-th row / column, with a barrier between the passes -- G times the loads in flight for the same memory.
// Serial schedule: faults that share no check commute; they are grouped into dependency levels on the host and the G = 4
// wavefronts of a workgroup take the faults of a level in parallel, one barrier per level (3.4 faults per level at the
// headline: 2795 levels for 9504 faults).  bp.hpp recomputes, for every edge (i, j) it visits, the product (minimum) over the
// OTHER edges of row i: row weight loads per edge, 14 dependent memory round trips per fault.  Here a row keeps a running PREFIX
// over the entries the sweep has already refreshed (natural fault order = ascending position in every row) and, per edge, the
// SUFFIX over the entries still to come, computed once at the start of the sweep: one load of each per edge, one round trip per
// fault.  Minimum and sign parity are associative, so min-sum returns bp.hpp's numbers exactly; the product-sum product becomes
// (prefix, left to right) x (suffix, right to left) -- the order bp.hpp itself uses in its flooding schedule -- instead of one
// left-to-right chain: the float mirror (oracle/bp_core.inc, bp_serial_ps_presuf) follows, the double-precision reference
// form keeps bp.hpp's chain.
//
// The adjacency arrays are separate __restrict__ kernel arguments on purpose: only then can the compiler prove that the
// kernel's stores do not clobber them and read them with scalar loads (as members of the by-value graph struct they came
// in through vector loads + readfirstlane, one dependent VMEM round trip each).
//
// tanh(x/2) and log((1+c)/(1-c)) are the fixed-operation-order float functions of qd_math.h (shared with the CPU mirror).
// The serial product-sum schedule keeps tanh(b2c/2) of every edge INSTEAD of the message (bp.hpp re-evaluates it once per
// use, row weight times per sweep; a pure function of an unchanged argument, so the cached value is the same number).
#include "qd_internal.h"
#include "qd_math.h"
#include "../../include/quits_amd.h"

#define QD_GEN_MLP 8          // loads of one row issued together (the arithmetic that follows keeps bp.hpp's order)

// Product-sum values travel as signed u (qd_math.h): magnitude e^-|b2c| = (1 - t) / (1 + t) of t = tanh(|b2c| / 2), sign bit = sign of the tanh.
// u of a product of two tanh values:
__device__ __forceinline__ float qd_ucomb_s(float a, float b)
{
    const uint32_t sg = (__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u;
    return __uint_as_float(__float_as_uint(qd_ucomb(fabsf(a), fabsf(b))) | sg);
}
// the check->bit message of a row product z: +-(-log |z|), negated once more by the syndrome bit
__device__ __forceinline__ float qd_u_llr(float z, uint32_t syndrome_bit)
{
    const uint32_t sg = (__float_as_uint(z) & 0x80000000u) ^ (syndrome_bit << 31);
    return __uint_as_float(__float_as_uint(qd_neg_log(fabsf(z))) ^ sg);
}

// per-lane OR across the G wavefronts of the workgroup (every wavefront gets the result)
template <int G>
__device__ __forceinline__ uint32_t qd_lanes_or(uint32_t v, uint32_t (*red)[64], int wv, int lane)
{
    if (G == 1) return v;
    red[wv][lane] = v;
    __syncthreads();
    uint32_t r = 0u;
#pragma unroll
    for (int k = 0; k < G; ++k) r |= red[k][lane];
    __syncthreads();
    return r;
}

// D: bound on the column weight the instantiation unrolls for (4, 8 or QD_MAX_COL_DEG; registers: five arrays of D in the serial schedule)
//    LP: serial schedule with the rows' running prefixes in LDS slots (GenGraphDev::nslots > 0)
#ifndef QD_GEN_WPE
#define QD_GEN_WPE 1          // wavefronts per SIMD the serial instantiations for column weight <= 8 are budgeted for (1: no bound)
#endif
template <int METHOD, int SCHED, int G, int D, bool LP>
__global__ void __launch_bounds__(64 * G, (SCHED == QD_SCHEDULE_SERIAL && D <= 8) ? QD_GEN_WPE : 1) qd_bp_edge_kernel(GenGraphDev g, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                            const int32_t *__restrict__ cp, const int32_t *__restrict__ ri,
                                                            const int32_t *__restrict__ c2r, const float *__restrict__ llr0,
                                                            const uint32_t *__restrict__ srec,
                                                            DecodeArgs a, GenWs w, int64_t shot0, int nshots, GenStage st)
{
    __shared__ uint32_t red[G][64];
    if (st.in_count) nshots = *st.in_count;                          // a later launch of the staged serial schedule: the survivors of the one before
    if ((int)blockIdx.x * 64 >= nshots) return;
    extern __shared__ float pls[];                                   // [nslots][64]  (LP)
    const int lane = threadIdx.x & 63;
    const int wv = G > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;    // which rows / columns this wavefront takes
    const int ls = blockIdx.x * 64 + lane;                           // shot inside this chunk = column of the workspace
    bool active = ls < nshots;
    // (the shot index is worked out where it is used, at the two ends: held across the sweeps it costs the registers that decide whether five
    //  workgroups fit a CU or four -- 93 -> 97 registers measured 55 -> 76 ms per launch)
#define QD_GEN_SHOT (st.in_shot ? (int64_t)st.in_shot[lsc] : shot0 + lsc)
    // workspace: [index][S shots] -- wavefronts advance through the graph at nearly the same pace, so at any moment they
    // touch the same few index planes: the pages in use are shared by all of them.  (A per-wavefront tiling
    // [tile][index][64] keeps each wavefront's data contiguous but multiplies the pages in flight by the number of
    // wavefronts; it measured 30 % slower in the flooding schedule.)
    const size_t S = (size_t)w.S;
    const int lsc = active ? ls : 0;
    float *__restrict__ b2c = w.b2c ? w.b2c + lsc : nullptr;         // [nnz][S], CSR edge order (not for serial product-sum)
    float *__restrict__ c2b = w.c2b + lsc;
    float *__restrict__ th = w.th ? w.th + lsc : nullptr;            // product-sum only
    float *__restrict__ llr = w.llr + lsc;                           // [n][S]
    uint8_t *__restrict__ syn = w.syn + lsc;                         // [m][S]
    float *__restrict__ rpre = w.pre ? w.pre + lsc : nullptr;        // [m][S] serial schedule: running prefix of each row
    const float BIG = 3.402823466e+38f;

    // ---- window syndrome (sliding_window.py:168-169)
    if (active && wv == 0) w.slot[ls] = -1;
    if (st.it0 == 0) {               // (a later launch finds syndrome and messages in its planes)
    const int64_t shot = shot0 + lsc;
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    uint32_t any = 0;
    if (active)
        for (int i = wv; i < g.m; i += G) {
            uint32_t s = det[i] & 1u;
            if (upd && i < a.upd_rows) s ^= upd[i] & 1u;
            syn[(size_t)i * S] = (uint8_t)s;
            any |= s;
        }
    any = qd_lanes_or<G>(any, red, wv, lane);
    if (active && !any) {   // bposd_decoder.pyx: an all-zero syndrome returns the zero vector without running BP
        if (wv == 0) {
            uint32_t *out = a.err_bits + shot * g.out_words;
            for (int x = 0; x < g.out_words; ++x) out[x] = 0u;
            a.status[shot] = (1 << 16) | (1 << 19);
        }
        active = false;
    }

    // ---- every bit->check message starts as the prior LLR
    if (active)
        for (int j = wv; j < g.n; j += G) {
            const float l0 = llr0[j];
            for (int e = cp[j]; e < cp[j + 1]; ++e) {
                const size_t ce = (size_t)c2r[e] * S;
                if (METHOD == QD_BP_PRODUCT_SUM && SCHED == QD_SCHEDULE_SERIAL) th[ce] = qd_exp_neg(l0);
                else b2c[ce] = l0;
            }
        }
    if (G > 1) __syncthreads();
    }
    const bool ran = active;

    int iters = 0, converged = 0;
    for (int it = st.it0 + 1; it <= st.it_end; ++it) {
        if (G > 1) { if (!__syncthreads_or(active ? 1 : 0)) break; }
        else if (!active) break;
        const float alpha = (a.ms_scale == 0.f) ? (1.0f - ldexpf(1.0f, -it)) : a.ms_scale;
        if (SCHED == QD_SCHEDULE_PARALLEL) {
            // ---- check pass: forward / backward exclusive sweeps over each row (bp.hpp)
            for (int i = wv; active && i < g.m; i += G) {
                const int r0 = rp[i], r1 = rp[i + 1];
                const uint32_t si = syn[(size_t)i * S];
                if (METHOD == QD_BP_PRODUCT_SUM) {
                    float temp = 0.0f;                               // u of the empty product
                    int e = r0;
                    for (; e + QD_GEN_MLP <= r1; e += QD_GEN_MLP) {
                        float v[QD_GEN_MLP];
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) v[k] = b2c[(size_t)(e + k) * S];
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) {
                            const float t = qd_exp_neg(v[k]);
                            th[(size_t)(e + k) * S] = t;
                            c2b[(size_t)(e + k) * S] = temp;
                            temp = qd_ucomb_s(temp, t);
                        }
                    }
                    for (; e < r1; ++e) {
                        const float t = qd_exp_neg(b2c[(size_t)e * S]);
                        th[(size_t)e * S] = t;
                        c2b[(size_t)e * S] = temp;
                        temp = qd_ucomb_s(temp, t);
                    }
                    temp = 0.0f;
                    e = r1 - 1;
                    for (; e - (QD_GEN_MLP - 1) >= r0; e -= QD_GEN_MLP) {
                        float vc[QD_GEN_MLP], vt[QD_GEN_MLP];
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) { vc[k] = c2b[(size_t)(e - k) * S]; vt[k] = th[(size_t)(e - k) * S]; }
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) {
                            c2b[(size_t)(e - k) * S] = qd_u_llr(qd_ucomb_s(vc[k], temp), si);
                            temp = qd_ucomb_s(temp, vt[k]);
                        }
                    }
                    for (; e >= r0; --e) {
                        c2b[(size_t)e * S] = qd_u_llr(qd_ucomb_s(c2b[(size_t)e * S], temp), si);
                        temp = qd_ucomb_s(temp, th[(size_t)e * S]);
                    }
                } else {
                    int total_sgn = (int)si;
                    float temp = BIG;
                    int e = r0;
                    for (; e + QD_GEN_MLP <= r1; e += QD_GEN_MLP) {
                        float v[QD_GEN_MLP];
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) v[k] = b2c[(size_t)(e + k) * S];
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) {
                            if (v[k] <= 0.f) total_sgn += 1;
                            c2b[(size_t)(e + k) * S] = temp;
                            const float av = fabsf(v[k]);
                            if (av < temp) temp = av;
                        }
                    }
                    for (; e < r1; ++e) {
                        const float v = b2c[(size_t)e * S];
                        if (v <= 0.f) total_sgn += 1;
                        c2b[(size_t)e * S] = temp;
                        const float av = fabsf(v);
                        if (av < temp) temp = av;
                    }
                    temp = BIG;
                    e = r1 - 1;
                    for (; e - (QD_GEN_MLP - 1) >= r0; e -= QD_GEN_MLP) {
                        float vb[QD_GEN_MLP], vc[QD_GEN_MLP];
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) { vb[k] = b2c[(size_t)(e - k) * S]; vc[k] = c2b[(size_t)(e - k) * S]; }
#pragma unroll
                        for (int k = 0; k < QD_GEN_MLP; ++k) {
                            int sgn = total_sgn;
                            if (vb[k] <= 0.f) sgn += 1;
                            float c = vc[k];
                            if (temp < c) c = temp;
                            const float msign = (sgn % 2 == 0) ? 1.0f : -1.0f;
                            c2b[(size_t)(e - k) * S] = c * (msign * alpha);
                            const float av = fabsf(vb[k]);
                            if (av < temp) temp = av;
                        }
                    }
                    for (; e >= r0; --e) {
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
            // (a column's messages are gathered together, the two sweeps run in registers in bp.hpp's order, and each
            //  outgoing message is written once: prefix + suffix is the value bp.hpp reaches with its `+=`)
            if (G > 1) __syncthreads();
            for (int j = wv; active && j < g.n; j += G) {
                const int c0 = cp[j], deg = cp[j + 1] - c0;
                float cv[D], pre[D];
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < deg) cv[k] = c2b[(size_t)c2r[c0 + k] * S];
                float temp = llr0[j];
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < deg) { pre[k] = temp; temp += cv[k]; }
                llr[(size_t)j * S] = temp;
                temp = 0.f;
#pragma unroll
                for (int k = D - 1; k >= 0; --k)
                    if (k < deg) { b2c[(size_t)c2r[c0 + k] * S] = pre[k] + temp; temp += cv[k]; }
            }
            if (G > 1) __syncthreads();
        } else {
            // ---- serial schedule (see the header).  msg = tanh(b2c / 2) (product-sum) or b2c (min-sum), per CSR edge; suf = the
            // suffix over the row entries behind an edge; pre = the row's running prefix.  Min-sum packs the parity of the
            // "<= 0" signs into the sign bit of the (non-negative) running minimum.
            float *__restrict__ msg = METHOD == QD_BP_PRODUCT_SUM ? th : b2c;
            float *__restrict__ suf = c2b;
            for (int i = wv; active && i < g.m; i += G) {
                const int r0 = rp[i], r1 = rp[i + 1];
                float sm = METHOD == QD_BP_PRODUCT_SUM ? 0.0f : BIG;
                uint32_t par = 0u;
                int e = r1 - 1;
                for (; e - (QD_GEN_MLP - 1) >= r0; e -= QD_GEN_MLP) {
                    float v[QD_GEN_MLP];
#pragma unroll
                    for (int k = 0; k < QD_GEN_MLP; ++k) v[k] = msg[(size_t)(e - k) * S];
#pragma unroll
                    for (int k = 0; k < QD_GEN_MLP; ++k) {
                        if (METHOD == QD_BP_PRODUCT_SUM) { suf[(size_t)(e - k) * S] = sm; sm = qd_ucomb_s(v[k], sm); }
                        else {
                            suf[(size_t)(e - k) * S] = __uint_as_float(__float_as_uint(sm) | (par << 31));
                            const float av = fabsf(v[k]);
                            if (av < sm) sm = av;
                            par ^= (v[k] <= 0.f) ? 1u : 0u;
                        }
                    }
                }
                for (; e >= r0; --e) {
                    const float v = msg[(size_t)e * S];
                    if (METHOD == QD_BP_PRODUCT_SUM) { suf[(size_t)e * S] = sm; sm = qd_ucomb_s(v, sm); }
                    else {
                        suf[(size_t)e * S] = __uint_as_float(__float_as_uint(sm) | (par << 31));
                        const float av = fabsf(v);
                        if (av < sm) sm = av;
                        par ^= (v <= 0.f) ? 1u : 0u;
                    }
                }
                // the row's syndrome bit rides in the sign of its prefix: log_ratio is odd and a product's sign is the parity of its
                // factors' signs, so the check->bit sign needs no separate load per edge
                if (!LP) {
                    const uint32_t sbit = (uint32_t)syn[(size_t)i * S] << 31;
                    rpre[(size_t)i * S] = __uint_as_float(__float_as_uint(METHOD == QD_BP_PRODUCT_SUM ? 0.0f : BIG) | sbit);
                }
            }
            if (G > 1) __syncthreads();
            // Faults are taken level by level (see GenGraphDev): inside a level they share no check, so the G wavefronts take one
            // each; across levels every pair of faults with a common check keeps its natural order -- the result is that of the
            // natural-order sweep of bp.hpp.  A wavefront's work in a step is one record, loaded (scalar) one step ahead.
            {
            constexpr int RW = (2 + 2 * D + 3) & ~3;
            // lane q < RW fetches dword q of the wavefront's record for the NEXT step with one vector load (a scalar load of two
            // records' worth of registers does not fit beside the kernel's arguments: the compiler parks it in vector lanes at once,
            // which waits for it); the step that uses it broadcasts the dwords it needs with v_readlane
            const uint32_t *__restrict__ rb = srec + (size_t)wv * RW + (lane < RW ? lane : 0);
            uint32_t pf = rb[0];
            for (int st = 0; st < g.nstep; ++st) {
                uint32_t cur[RW];
#pragma unroll
                for (int q = 0; q < RW; ++q) cur[q] = __builtin_amdgcn_readlane(pf, q);
                const uint32_t head = cur[0];
                static_assert(QD_MAX_COL_DEG <= 31, "the record's weight field is decoded with & 31");
                const int deg = (int)((head >> 24) & 31u);
                float P[D], X[D], cv[D], pr[D];
                pf = rb[(size_t)(st + 1 < g.nstep ? st + 1 : st) * (G * RW)];
                if (active && deg) {
                const int j = (int)(head & 0xFFFFFFu);
                float lj = __uint_as_float(cur[1]);
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < deg) {
                        if (LP) {
                            const uint32_t rw = cur[2 + k];
                            if (rw & 0x800000u)      // first entry of its row: nothing refreshed yet
                                P[k] = __uint_as_float(__float_as_uint(METHOD == QD_BP_PRODUCT_SUM ? 0.0f : BIG) |
                                                       ((uint32_t)syn[(size_t)(rw & 0x7FFFFFu) * S] << 31));
                            else P[k] = pls[(rw >> 24) * 64 + lane];
                        } else P[k] = rpre[(size_t)cur[2 + k] * S];
#if defined(QD_GEN_ABL_NOX)      /* timing experiment only (wrong results): the suffix does not come from memory -- what a perfect prefetch of it could return */
                        X[k] = __uint_as_float(0x3f000000u | (cur[2 + D + k] & 0xFFu));
#else
                        X[k] = suf[(size_t)cur[2 + D + k] * S];
#endif
                    }
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k < deg) {
                        float c;
                        if (METHOD == QD_BP_PRODUCT_SUM) c = qd_u_llr(qd_ucomb_s(P[k], X[k]), 0u);
                        else {
                            const float a1 = fabsf(P[k]), a2 = fabsf(X[k]);
                            const uint32_t sg = ((__float_as_uint(P[k]) >> 31) ^ (__float_as_uint(X[k]) >> 31)) & 1u;
                            c = alpha * (sg ? -1.0f : 1.0f) * (a2 < a1 ? a2 : a1);
                        }
                        cv[k] = c;
                        pr[k] = lj;
                        lj += c;
                    }
                llr[(size_t)j * S] = lj;
                float temp = 0.f;
#pragma unroll
                for (int k = D - 1; k >= 0; --k)
                    if (k < deg) {               // b2c = prefix (kept in registers) + suffix
                        const size_t ce = (size_t)cur[2 + D + k] * S;
                        float *__restrict__ pdst = LP ? &pls[(cur[2 + k] >> 24) * 64 + lane] : &rpre[(size_t)cur[2 + k] * S];
                        const float v = pr[k] + temp;
                        temp += cv[k];
                        if (METHOD == QD_BP_PRODUCT_SUM) {
                            const float nt = qd_exp_neg(v);
                            msg[ce] = nt;
                            *pdst = qd_ucomb_s(P[k], nt);
                        } else {
                            msg[ce] = v;
                            const float a1 = fabsf(P[k]), av = fabsf(v);
                            const uint32_t np = (__float_as_uint(P[k]) >> 31) ^ ((v <= 0.f) ? 1u : 0u);
                            *pdst = __uint_as_float(__float_as_uint(av < a1 ? av : a1) | (np << 31));
                        }
                    }
                }
                if (G > 1 && (head >> 31)) __syncthreads();        // (a barrier that orders LDS only -- no wait for the level's global stores -- measured no different: r06_k1g_staged_ab.txt (6))
            }
            }
        }
        // ---- stop when the hard decision reproduces the syndrome
        uint32_t bad = 0;
        for (int i = wv; active && i < g.m; i += G) {
            uint32_t p = syn[(size_t)i * S];
            const int r1 = rp[i + 1];
            int e = rp[i];
            for (; e + QD_GEN_MLP <= r1; e += QD_GEN_MLP) {
                float v[QD_GEN_MLP];
#pragma unroll
                for (int k = 0; k < QD_GEN_MLP; ++k) v[k] = llr[(size_t)ci[e + k] * S];
#pragma unroll
                for (int k = 0; k < QD_GEN_MLP; ++k) p ^= (v[k] <= 0.f) ? 1u : 0u;
            }
            for (; e < r1; ++e) p ^= (llr[(size_t)ci[e] * S] <= 0.f) ? 1u : 0u;
            bad |= p;
        }
        bad = qd_lanes_or<G>(bad, red, wv, lane);
        if (active) {
            iters = it;
            if (!bad) { converged = 1; active = false; }
        }
    }

    // ---- not the last launch: the shots still running move on, packed (GenStage)
    if (SCHED == QD_SCHEDULE_SERIAL && !st.last) {
        const unsigned long long sb = __ballot(active);              // (the same in every wavefront of the workgroup: they share the 64 shots)
        uint32_t base = 0u;
        if (G > 1) {
            if (threadIdx.x == 0) red[0][0] = sb ? (uint32_t)atomicAdd(st.out_count, (int)__popcll(sb)) : 0u;
            __syncthreads();
            base = red[0][0];
        } else {
            if (lane == 0 && sb) base = (uint32_t)atomicAdd(st.out_count, (int)__popcll(sb));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        }
        if (sb) {
            const size_t dcol = (size_t)base + (size_t)__popcll(sb & ((1ull << lane) - 1ull));
            const size_t S2 = (size_t)st.next.S;
            const float *__restrict__ src = (METHOD == QD_BP_PRODUCT_SUM ? w.th : w.b2c) + lsc;
            float *__restrict__ dst = (METHOD == QD_BP_PRODUCT_SUM ? st.next.th : st.next.b2c) + dcol;
            int e = wv;
            for (; e + 7 * G < g.nnz; e += 8 * G) {                  // whole lines in, the survivors' part of them out: eight loads in flight
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(e + k * G) * S];
                if (active) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) dst[(size_t)(e + k * G) * S2] = v[k];
                }
            }
            for (; e < g.nnz; e += G) {
                const float v = src[(size_t)e * S];
                if (active) dst[(size_t)e * S2] = v;
            }
            for (int i = wv; i < g.m; i += G) {
                const uint8_t v = syn[(size_t)i * S];
                if (active) st.next.syn[(size_t)i * S2 + dcol] = v;
            }
            if (active && wv == 0) st.out_shot[dcol] = (int32_t)QD_GEN_SHOT;
        }
    }
    // ---- hard decision, packed by fault index
    if (!ran || (!st.last && active)) return;
    const int64_t shot = QD_GEN_SHOT;
    uint32_t *out = a.err_bits + shot * g.out_words;
    for (int x = wv; x < g.out_words; x += G) {
        uint32_t word = 0u;
        const int j1 = min(g.n, 32 * x + 32);
        for (int j = 32 * x; j < j1; ++j) word |= ((llr[(size_t)j * S] <= 0.f) ? 1u : 0u) << (j & 31);
        out[x] = word;
    }
    if (wv != 0) return;
    a.status[shot] = iters | (converged << 16);
    if (!converged && a.want_llr) {
        const int slot = atomicAdd(a.fail_count, 1);
        a.fail_list[slot] = (int32_t)shot;
        w.slot[ls] = slot;
    }
}

// ---- flooding product-sum with the messages in LDS: one workgroup per shot -------------------------------------------------
// The kernel above gives a lane to a shot because the serial schedule leaves a single shot almost no parallelism; the FLOODING
// schedule does not have that problem -- every row, then every column, is independent -- and for the windows the reference
// really decodes (W = 3..5 rounds: E <= ~25 000 edges) one float per edge fits LDS.  So here a workgroup owns a shot: a lane
// takes an edge (tanh, log), a row (the part of the check pass that is a chain: bp.hpp's forward / backward exclusive products,
// the row's tanh values in registers) or a fault (bit pass: posterior, then the prefix / suffix sums); check->bit messages are
// written over the bit->check messages they were made from, message traffic never leaves the CU, and HBM sees the detector
// bytes in and the decision out.
// Same operations in the same order as qd_bp_edge_kernel<PRODUCT_SUM, PARALLEL> and as the float mirror bp_parallel_edge_f32:
// the two kernels return the same bits.  The parity of the hard decisions is kept per check by LDS atomics from the (rare)
// faults decided 1, as in bp_kernels.hip, so the convergence test costs nothing per edge.
//   DEG = bound on the row weight the instantiation unrolls for (one register array of that length)
//   NREG = 0: posteriors in LDS.  NREG > 0 (a window whose messages alone nearly fill the LDS -- the headline's 33 192 edges = 133 KB):
//          a lane keeps the posteriors of ITS faults (fault j belongs to lane j mod T in every bit pass) in NREG registers, n <= NREG * T
#define QD_PSL_TR 4           // bit-pass trips whose records are in flight together
template <int DEG, int T, int NREG>
__global__ void __launch_bounds__(T) qd_bp_ps_lds_kernel(GenGraphDev g, const int32_t *__restrict__ rp, const int32_t *__restrict__ cp,
                                                         const int32_t *__restrict__ ri, const int32_t *__restrict__ c2r,
                                                         const float *__restrict__ llr0, const uint32_t *__restrict__ slot_of,
                                                         const uint16_t *__restrict__ erow, const uint4 *__restrict__ frec, int n_pad, DecodeArgs a)
{
    extern __shared__ __align__(16) float sm[];
    float *msg = sm;                                                 // [nnz] CSR edge order: b2c before a check pass, c2b after it
    float *llr = msg + ((g.nnz + 3) & ~3);                           // [n]   (NREG = 0)
    uint32_t *par = reinterpret_cast<uint32_t *>(llr + (NREG ? 0 : ((g.n + 3) & ~3)));   // [m] bit 0 syndrome, bit 1 parity of the hard decisions on the check
    float lreg[NREG ? NREG : 1];
    uint32_t *red = par + ((g.m + 3) & ~3);                          // [2][16] block-OR flags, [32] fail slot
    const int tid = threadIdx.x;
    constexpr int NW = T / 64;
    constexpr int TR = NREG ? 2 : QD_PSL_TR;                         // (registers: the NREG instantiation has 128 to live in)
    const int64_t shot = blockIdx.x;
    const uint8_t *det = a.det + shot * a.det_stride + a.det_offset;
    const uint8_t *upd = a.upd ? a.upd + shot * a.upd_stride : nullptr;
    auto block_or = [&](uint32_t v, int phase) -> uint32_t {
        const unsigned long long bal = __ballot(v != 0u);
        if ((tid & 63) == 0) red[phase * 16 + (tid >> 6)] = bal != 0ull ? 1u : 0u;
        __syncthreads();
        uint32_t r = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) r |= red[phase * 16 + w];
        return r;
    };
    uint32_t any = 0u;
    for (int i = tid; i < g.m; i += T) {
        uint32_t sb = det[i] & 1u;
        if (upd && i < a.upd_rows) sb ^= upd[i] & 1u;
        par[i] = sb;
        any |= sb;
    }
    uint32_t *out = a.err_bits + shot * g.out_words;
    if (!block_or(any, 0)) {   // bposd_decoder.pyx: an all-zero syndrome returns the zero vector without running BP
        for (int x = tid; x < g.out_words; x += T) out[x] = 0u;
        if (tid == 0) a.status[shot] = (1 << 16) | (1 << 19);
        return;
    }
    for (int j = tid; j < g.n; j += T) {
        const float l0 = llr0[j];
        for (int e = cp[j]; e < cp[j + 1]; ++e) msg[c2r[e]] = l0;
    }
    __syncthreads();
    int iters = 0, converged = 0, phase = 1;
    const int my_r0 = tid < g.m ? rp[tid] : 0, my_deg = tid < g.m ? rp[tid + 1] - my_r0 : 0;      // this lane's first (usually only) row
    for (int it = 1; it <= a.max_iter; ++it) {
        // ---- check pass, in three steps so that the expensive functions run one edge per lane (every lane busy) and only the two
        // multiply chains of a row (bp.hpp's order) run one row per lane:
        //   tanh(b2c / 2) per edge | forward / backward exclusive products per row | sign * log((1 + c) / (1 - c)) per edge
        // (fusing the two functions into the bit pass -- three barriers per iteration instead of five -- measured 15 % slower: a
        //  fault has 3.2 edges of the 8 a lane unrolls for)
        for (int e = tid; e < g.nnz; e += T) msg[e] = qd_exp_neg(msg[e]);
        __syncthreads();
        for (int i = tid; i < g.m; i += T) {
            const int r0 = i == tid ? my_r0 : rp[i], deg = i == tid ? my_deg : rp[i + 1] - rp[i];
            float th[DEG];
            float temp = 0.0f;                                       // u of the empty product
#pragma unroll
            for (int k = 0; k < DEG; ++k)
                if (k < deg) {
                    const float t = msg[r0 + k];
                    th[k] = t;
                    msg[r0 + k] = temp;
                    temp = qd_ucomb_s(temp, t);
                }
            temp = 0.0f;
#pragma unroll
            for (int k = DEG - 1; k >= 0; --k)
                if (k < deg) {
                    msg[r0 + k] = qd_ucomb_s(msg[r0 + k], temp);
                    temp = qd_ucomb_s(temp, th[k]);
                }
        }
        __syncthreads();
        for (int e = tid; e < g.nnz; e += T) {
            msg[e] = qd_u_llr(msg[e], par[erow[e]] & 1u);
        }
        __syncthreads();
        // ---- bit pass: posterior, then the prefix / suffix sums that make the outgoing messages.  A fault comes as one 32-byte
        // record (eight 16-bit CSR edge indices and their rows, 0xFFFF beyond the column weight); the records and priors of
        // TR trips are requested before any is used, so a pass pays one L2 latency, not one per trip.
#pragma unroll
        for (int ob = 0; ob < (NREG ? NREG / TR : 1 << 20); ++ob) {
            const int base = ob * (T * TR);
            if (base >= g.n) break;
            uint4 rec[TR];
            float l0v[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const int j = base + t * T + tid;
                rec[t] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                l0v[t] = 0.f;
                if (j < g.n) { rec[t] = frec[2 * j]; l0v[t] = llr0[j]; }
            }
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const int j = base + t * T + tid;
                if (j >= g.n) continue;
                const uint32_t w4[4] = {rec[t].x, rec[t].y, rec[t].z, rec[t].w};
                uint32_t idx[8];
                float cv[8], pr[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    idx[k] = (k & 1) ? (w4[k >> 1] >> 16) : (w4[k >> 1] & 0xFFFFu);
                    cv[k] = idx[k] != 0xFFFFu ? msg[idx[k]] : 0.f;
                }
                float temp = l0v[t];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (idx[k] != 0xFFFFu) { pr[k] = temp; temp += cv[k]; }
                if (NREG) lreg[NREG ? ob * TR + t : 0] = temp;
                else llr[j] = temp;
                if (temp <= 0.f) {                                      // hard decision 1 (rare): tell the fault's checks
                    const uint4 rr = frec[2 * j + 1];
                    const uint32_t r4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (idx[k] != 0xFFFFu) atomicXor(&par[(k & 1) ? (r4[k >> 1] >> 16) : (r4[k >> 1] & 0xFFFFu)], 2u);
                }
                temp = 0.f;
#pragma unroll
                for (int k = 7; k >= 0; --k)
                    if (idx[k] != 0xFFFFu) { msg[idx[k]] = pr[k] + temp; temp += cv[k]; }
            }
        }
        __syncthreads();
        // ---- stop when the hard decision reproduces the syndrome
        uint32_t bad = 0u;
        for (int i = tid; i < g.m; i += T) {
            const uint32_t v = par[i];
            bad |= (v ^ (v >> 1)) & 1u;
            par[i] = v & 1u;                                         // (only this lane touches the word until the next bit pass)
        }
        bad = block_or(bad, phase);
        phase ^= 1;
        iters = it;
        if (!bad) { converged = 1; break; }
    }
    // ---- hard decision, packed by fault index
    if (NREG) {
        uint32_t *ow = reinterpret_cast<uint32_t *>(msg);            // the messages are done with
        __syncthreads();
        for (int x = tid; x < g.out_words; x += T) ow[x] = 0u;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (NREG ? NREG : 1); ++q) {
            const int j = q * T + tid;
            if (j < g.n && lreg[q] <= 0.f) atomicOr(&ow[j >> 5], 1u << (j & 31));
        }
        __syncthreads();
        for (int x = tid; x < g.out_words; x += T) out[x] = ow[x];
    } else
        for (int x = tid; x < g.out_words; x += T) {
            uint32_t word = 0u;
            const int j1 = min(g.n, 32 * x + 32);
            for (int j = 32 * x; j < j1; ++j) word |= ((llr[j] <= 0.f) ? 1u : 0u) << (j & 31);
            out[x] = word;
        }
    if (tid == 0) a.status[shot] = iters | (converged << 16);
    if (!converged && a.want_llr) {
        if (tid == 0) { const int slot = atomicAdd(a.fail_count, 1); a.fail_list[slot] = (int32_t)shot; red[32] = (uint32_t)slot; }
        __syncthreads();
        float *dst = a.llr_ws + (int64_t)red[32] * n_pad;
        if (NREG) {
#pragma unroll
            for (int q = 0; q < (NREG ? NREG : 1); ++q) {
                const int j = q * T + tid;
                if (j < g.n) dst[slot_of[j]] = lreg[q];
            }
        } else
            for (int j = tid; j < g.n; j += T) dst[slot_of[j]] = llr[j];
    }
}

#define QD_PSL_BIG_T 1024
#define QD_PSL_BIG_NREG 12
// LDS bytes of that kernel for a window, or 0 if the window does not fit it (row weight beyond the instantiations, or too many edges)
int qd_bp_ps_lds_bytes(const GenGraphDev &g, int max_rdeg)
{
    if (max_rdeg > 56 || g.n > 65535 || g.nnz > 65534 || !g.frec) return 0;          // (frec: only built for column weight <= 8)
    const size_t fixed = ((size_t)((g.nnz + 3) & ~3) + ((g.m + 3) & ~3) + 64) * 4, b = fixed + (size_t)((g.n + 3) & ~3) * 4;
    if (b <= (size_t)QD_LDS_BYTES / 2) return (int)b;                                // two workgroups per CU
    if (fixed <= (size_t)QD_LDS_BYTES - 1024 && g.n <= QD_PSL_BIG_T * QD_PSL_BIG_NREG && max_rdeg <= 36) return (int)fixed;   // posteriors in registers
    return b <= (size_t)QD_LDS_BYTES - 1024 ? (int)b : 0;
}

hipError_t qd_launch_bp_ps_lds(const GenGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, int64_t B, hipStream_t s)
{
    const int lds = qd_bp_ps_lds_bytes(g, bg.max_rdeg);
    if (lds == 0) return hipErrorInvalidValue;
    const bool wide = bg.max_rdeg > 36;
    const bool small = g.nnz <= 9000;           // threads per shot: 256 for the W = 3 windows (6624 edges: 14.1 vs 18.0 ms per launch), 512 beyond (W = 5: 28.6 vs 44.5 ms)
    const bool regs = (size_t)lds == ((size_t)((g.nnz + 3) & ~3) + ((g.m + 3) & ~3) + 64) * 4;
    const dim3 grid((unsigned)B);
#define QD_PSL(D_, T_, N_)                                                                                                  \
    {                                                                                                                       \
        hipError_t e = hipFuncSetAttribute((const void *)qd_bp_ps_lds_kernel<D_, T_, N_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        if (e != hipSuccess) return e;                                                                                      \
        hipLaunchKernelGGL((qd_bp_ps_lds_kernel<D_, T_, N_>), grid, dim3(T_), (size_t)lds, s, g, g.rp, g.cp, g.ri, g.c2r, g.llr0, bg.bit_slot_of, g.erow, \
                           reinterpret_cast<const uint4 *>(g.frec), bg.n_pad, a);                                           \
    }
    if (regs) QD_PSL(36, QD_PSL_BIG_T, QD_PSL_BIG_NREG)
    else if (wide) { if (small) QD_PSL(56, 256, 0) else QD_PSL(56, 512, 0) }
    else { if (small) QD_PSL(36, 256, 0) else QD_PSL(36, 512, 0) }
#undef QD_PSL
    return hipGetLastError();
}

// Posteriors of the shots BP could not finish, from [fault][shot] to the OSD workspace's [fail slot][bit slot] rows.
// 64 x 64 tiles through LDS so that both sides move whole lines.
__global__ void __launch_bounds__(256) qd_publish_llr_kernel(const float *__restrict__ llr, const int32_t *__restrict__ slot,
                                                             int64_t S, int nshots, const int32_t *__restrict__ count, int n, int n_pad,
                                                             const uint32_t *__restrict__ bit_orig, float *__restrict__ llr_ws)
{
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int sb0 = blockIdx.x * 64, sh0 = blockIdx.y * 64;
    if (count) nshots = *count;                                       // (staged serial schedule: the columns of the last launch)
    if (sh0 >= nshots) return;
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

#define QD_GEN_G 8            // wavefronts per 64 shots in the flooding schedule

template <int METHOD, int SCHED, int G, int D>
static hipError_t launch_kd(const GenGraphDev &g, const DecodeArgs &a, const GenWs &w, int64_t shot0, int nshots, const GenStage &st, hipStream_t s)
{
    const dim3 grid((unsigned)((nshots + 63) / 64)), block(64 * G);
    if (SCHED == QD_SCHEDULE_SERIAL && g.nslots > 0)
        hipLaunchKernelGGL((qd_bp_edge_kernel<METHOD, SCHED, G, D, SCHED == QD_SCHEDULE_SERIAL>), grid, block, (size_t)g.nslots * 256, s, g, g.rp, g.ci,
                           g.cp, g.ri, g.c2r, g.llr0, g.srec, a, w, shot0, nshots, st);
    else
        hipLaunchKernelGGL((qd_bp_edge_kernel<METHOD, SCHED, G, D, false>), grid, block, 0, s, g, g.rp, g.ci,
                           g.cp, g.ri, g.c2r, g.llr0, g.srec, a, w, shot0, nshots, st);
    return hipGetLastError();
}

template <int METHOD, int SCHED, int G>
static hipError_t launch_k(const GenGraphDev &g, int max_cdeg, const DecodeArgs &a, const GenWs &w, int64_t shot0, int nshots, const GenStage &st, hipStream_t s)
{
    switch (qd_gen_unroll(max_cdeg)) {
    case 4: return launch_kd<METHOD, SCHED, G, 4>(g, a, w, shot0, nshots, st, s);
#if QD_GEN_D6
    case 6: return launch_kd<METHOD, SCHED, G, 6>(g, a, w, shot0, nshots, st, s);
#endif
    case 8: return launch_kd<METHOD, SCHED, G, 8>(g, a, w, shot0, nshots, st, s);
    default: return launch_kd<METHOD, SCHED, G, QD_MAX_COL_DEG>(g, a, w, shot0, nshots, st, s);
    }
}

hipError_t qd_launch_bp_general(const GenGraphDev &g, const BpGraphDev &bg, const DecodeArgs &a, const GenWs &w, int bp_method,
                                int schedule, int64_t shot0, int nshots, hipStream_t s, GenStagePlan *plan)
{
    // the staged serial schedule (GenStage): launches over iterations (0, b0], (b0, b1], ... (b_last, max_iter], workspaces w / plan->w2 in turn
    int nb = (plan && schedule == QD_SCHEDULE_SERIAL) ? plan->nbounds : 0;
    if (nb > 0) {
        if (plan->pending && qd_event_done(plan->counts_ready)) {      // what an earlier staged call packed (never waited for)
            plan->pending = 0;
            const int left = plan->host_counts[plan->pending_nb - 1];                  // shots that went into its last launch
            if (plan->pending_shots >= 4096 && (double)left > 0.85 * (double)plan->pending_shots) plan->one_launch_calls = QD_GEN_PROBE;
        }
        if (plan->one_launch_calls > 0) { --plan->one_launch_calls; nb = 0; }
    }
    if (nb > 0) {
        hipError_t e0 = hipMemsetAsync(plan->counts, 0, sizeof(int32_t) * (size_t)(nb + 1), s);
        if (e0 != hipSuccess) return e0;
    }
    const GenWs *wsv[2] = {&w, plan ? &plan->w2 : &w};
    for (int k = 0; k <= nb; ++k) {
        GenStage st{};
        st.it0 = k ? plan->bounds[k - 1] : 0;
        st.it_end = k < nb ? plan->bounds[k] : a.max_iter;
        st.last = k == nb;
        if (k) { st.in_shot = plan->lists[(k - 1) & 1]; st.in_count = plan->counts + (k - 1); }
        if (k < nb) { st.out_shot = plan->lists[k & 1]; st.out_count = plan->counts + k; st.next = *wsv[(k + 1) & 1]; }
        const GenWs &wk = *wsv[k & 1];
        hipError_t e;
        const int cd = bg.max_cdeg;
        if (bp_method == QD_BP_PRODUCT_SUM)
            e = schedule == QD_SCHEDULE_PARALLEL ? launch_k<QD_BP_PRODUCT_SUM, QD_SCHEDULE_PARALLEL, QD_GEN_G>(g, cd, a, wk, shot0, nshots, st, s)
                                                 : launch_k<QD_BP_PRODUCT_SUM, QD_SCHEDULE_SERIAL, QD_GEN_GS>(g, cd, a, wk, shot0, nshots, st, s);
        else
            e = schedule == QD_SCHEDULE_PARALLEL ? launch_k<QD_BP_MINIMUM_SUM, QD_SCHEDULE_PARALLEL, QD_GEN_G>(g, cd, a, wk, shot0, nshots, st, s)
                                                 : launch_k<QD_BP_MINIMUM_SUM, QD_SCHEDULE_SERIAL, QD_GEN_GS>(g, cd, a, wk, shot0, nshots, st, s);
        if (e != hipSuccess) return e;
    }
    if (nb > 0 && !plan->pending) {
        hipError_t e1 = hipMemcpyAsync(plan->host_counts, plan->counts, sizeof(int32_t) * (size_t)(nb + 1), hipMemcpyDeviceToHost, s);
        if (e1 == hipSuccess) e1 = hipEventRecord(plan->counts_ready, s);
        if (e1 != hipSuccess) return e1;
        plan->pending = 1; plan->pending_shots = nshots; plan->pending_nb = nb;
    }
    if (!a.want_llr) return hipSuccess;
    // BP failures exist in the last launch only (a shot fails by reaching max_iter): their posteriors are in ITS workspace, by ITS columns
    const GenWs &wl = *wsv[nb & 1];
    hipLaunchKernelGGL(qd_publish_llr_kernel, dim3((unsigned)((g.n + 63) / 64), (unsigned)((nshots + 63) / 64)), dim3(256), 0, s,
                       wl.llr, wl.slot, wl.S, nshots, nb ? plan->counts + (nb - 1) : nullptr, g.n, bg.n_pad, bg.bit_orig, a.llr_ws);
    return hipGetLastError();
}
