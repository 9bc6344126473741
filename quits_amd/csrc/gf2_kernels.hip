// gf2_kernels.hip -- the byte/bit plumbing around the decoder: window hand-off, unpacking, counting, sampling.
#include "qd_internal.h"

// out[b][r] (^)= parity(row r of A AND e_b).  Replaces `window_observable_set[k] @ e % 2` and
// `window_update[k] @ e % 2` (quits/decoder/sliding_window.py:172,174,183).  CPU restatement: the L/U loops of
// oq_sliding_window_decode (oracle/qd_oracle.c).  lane = (shot, row); the <= 1.2 KB of packed error bits of a shot
// stay in L1 across its rows.
__global__ void qd_gf2_spmv_kernel(SpmatDev A, const uint32_t *__restrict__ err, int64_t err_stride, int64_t B,
                                   uint8_t *out, int64_t out_stride, int accumulate)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * A.nrows) return;
    const int64_t b = idx / A.nrows;
    const int r = (int)(idx - b * A.nrows);
    const uint32_t *e = err + b * err_stride;
    uint32_t p = 0;
    for (uint32_t x = A.row_ptr[r]; x < A.row_ptr[r + 1]; ++x) {
        const uint32_t j = A.col_idx[x];
        p ^= (e[j >> 5] >> (j & 31u)) & 1u;
    }
    uint8_t *o = out + b * out_stride + r;
    *o = (uint8_t)((accumulate ? (*o & 1u) : 0u) ^ p);
}

// The same product driven by the set bits of e: a correction has a few dozen faults, a row of L_k or U_k hundreds of
// columns.  One wavefront per shot scans the packed error bits and XORs the row mask of every set column (columns beyond
// A.ncols -- faults of the window that are not committed -- are ignored, as the row form ignores them).
__global__ void __launch_bounds__(256) qd_gf2_colxor_kernel(SpmatDev A, const uint32_t *__restrict__ err, int64_t err_stride,
                                                            int64_t B, uint8_t *out, int64_t out_stride, int accumulate)
{
    __shared__ uint32_t acc[4][16];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wv;
    if (lane < 16) acc[wv][lane] = 0u;
    __syncthreads();
    if (b < B) {
        const uint32_t *e = err + b * err_stride;
        const int nwords = (A.ncols + 31) >> 5;
        for (int w = lane; w < nwords; w += 64) {
            uint32_t bits = e[w];
            if (w == nwords - 1 && (A.ncols & 31)) bits &= (1u << (A.ncols & 31)) - 1u;
            while (bits) {
                const int k = __ffs(bits) - 1;
                bits &= bits - 1u;
                const uint32_t *cm = A.colmask + (size_t)(32 * w + k) * A.mask_words;
                for (int q = 0; q < A.mask_words; ++q) {
                    const uint32_t mk = cm[q];
                    if (mk) atomicXor(&acc[wv][q], mk);
                }
            }
        }
    }
    __syncthreads();
    if (b < B)
        for (int r = lane; r < A.nrows; r += 64) {
            uint8_t *o = out + b * out_stride + r;
            const uint32_t p = (acc[wv][r >> 5] >> (r & 31)) & 1u;
            *o = (uint8_t)((accumulate ? (*o & 1u) : 0u) ^ p);
        }
}

__global__ void qd_unpack_bits_kernel(const uint32_t *__restrict__ bits, int64_t stride_words, int nbits, int64_t B,
                                      uint8_t *out, int64_t out_stride)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * nbits) return;
    const int64_t b = idx / nbits;
    const int j = (int)(idx - b * nbits);
    out[b * out_stride + j] = (uint8_t)((bits[b * stride_words + (j >> 5)] >> (j & 31)) & 1u);
}

// pL numerator (tests/test_sliding_window.py:83): shots where prediction != observable flips on any bit.
__global__ void qd_count_mismatch_kernel(const uint8_t *__restrict__ pred, const uint8_t *__restrict__ obs, int k,
                                         int64_t B, unsigned long long *count)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    if (b < B)
        for (int i = 0; i < k; ++i) bad |= ((pred[b * k + i] ^ obs[b * k + i]) & 1u);
    const unsigned long long bal = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(count, (unsigned long long)__popcll(bal));
}

// ---- DEM sampler (stands in for stim's detector sampler, quits/simulation.py:23-27).  Same integer recipe as
// oq_sample_dem (oracle/qd_oracle.c): Philox4x32-10, key = seed, counter = (shot lo, shot hi, j / 4, 0); word j & 3
// fires fault j iff it is < floor(p_j * 2^32).  One workgroup per shot, detector/observable bits accumulated in LDS.
__device__ __forceinline__ void qd_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void qd_sample_dem_kernel(SpmatDev Ht, SpmatDev Lt, const uint32_t *__restrict__ thr, uint32_t k0,
                                     uint32_t k1, int64_t shot0, int m, int nobs, uint8_t *det, int64_t det_stride,
                                     uint8_t *obs, int64_t obs_stride)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *dbits = reinterpret_cast<uint32_t *>(smem);
    const int dwords = (m + 31) >> 5, owords = (nobs + 31) >> 5;
    uint32_t *obits = dbits + dwords;
    const int tid = threadIdx.x, T = blockDim.x;
    const int64_t b = blockIdx.x;
    const uint64_t shot = (uint64_t)(shot0 + b);
    for (int w = tid; w < dwords + owords; w += T) dbits[w] = 0u;
    __syncthreads();
    const int n = Ht.nrows;
    for (int c = tid; 4 * c < n; c += T) {
        uint32_t r[4];
        qd_philox4x32_10((uint32_t)shot, (uint32_t)(shot >> 32), (uint32_t)c, 0u, k0, k1, r);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int j = 4 * c + x;
            if (j < n && r[x] < thr[j]) {
                for (uint32_t e = Ht.row_ptr[j]; e < Ht.row_ptr[j + 1]; ++e) {
                    const uint32_t d = Ht.col_idx[e];
                    atomicXor(&dbits[d >> 5], 1u << (d & 31u));
                }
                for (uint32_t e = Lt.row_ptr[j]; e < Lt.row_ptr[j + 1]; ++e) {
                    const uint32_t o = Lt.col_idx[e];
                    atomicXor(&obits[o >> 5], 1u << (o & 31u));
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < m; i += T) det[b * det_stride + i] = (uint8_t)((dbits[i >> 5] >> (i & 31)) & 1u);
    for (int i = tid; i < nobs; i += T) obs[b * obs_stride + i] = (uint8_t)((obits[i >> 5] >> (i & 31)) & 1u);
}

// Stage caller-supplied LLRs (fault order) as if the BP kernel had published them: slot order rows, identity fail list.
__global__ void qd_stage_llr_kernel(const float *__restrict__ llr_in, int n, int n_pad, const uint32_t *__restrict__ bit_orig,
                                    int64_t B, float *llr_ws, int32_t *fail_list, int32_t *fail_count, int32_t *status)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) *fail_count = (int32_t)B;
    if (idx < B) { fail_list[idx] = (int32_t)idx; status[idx] = 0; }
    if (idx >= B * n) return;
    const int64_t b = idx / n;
    const int slot = (int)(idx - b * n);
    llr_ws[b * n_pad + slot] = llr_in[b * n + bit_orig[slot]];
}

hipError_t qd_launch_stage_llr(const float *llr_in, int n, int n_pad, const uint32_t *bit_orig, int64_t B, float *llr_ws,
                               int32_t *fail_list, int32_t *fail_count, int32_t *status, hipStream_t s)
{
    const int64_t total = B * n;
    hipLaunchKernelGGL(qd_stage_llr_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, llr_in, n, n_pad, bit_orig,
                       B, llr_ws, fail_list, fail_count, status);
    return hipGetLastError();
}

// ---- launch wrappers
hipError_t qd_launch_spmv(const SpmatDev &A, const uint32_t *err, int64_t err_stride, int64_t B, uint8_t *out,
                          int64_t out_stride, int accumulate, hipStream_t s)
{
    const int64_t total = B * A.nrows;
    if (total == 0) return hipSuccess;
    if (A.colmask) {
        hipLaunchKernelGGL(qd_gf2_colxor_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, A, err, err_stride, B, out,
                           out_stride, accumulate);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(qd_gf2_spmv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, err, err_stride, B,
                       out, out_stride, accumulate);
    return hipGetLastError();
}

hipError_t qd_launch_unpack(const uint32_t *bits, int64_t stride_words, int nbits, int64_t B, uint8_t *out,
                            int64_t out_stride, hipStream_t s)
{
    const int64_t total = B * nbits;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(qd_unpack_bits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, bits, stride_words,
                       nbits, B, out, out_stride);
    return hipGetLastError();
}

// One wavefront that holds its stream for `ticks` of the 100 MHz wall clock if *count >= threshold (count == nullptr: always).  Bounded: it waits for
// nobody.  The pipelined driver puts it behind a BP stage whose post-processing is heavy, so that the post-processor -- starting on the other stream at
// that moment -- gets onto the CUs before the next BP kernel fills every wavefront slot (qd_decoder_post_head_start).
__global__ void qd_hold_kernel(const int32_t *__restrict__ count, int threshold, unsigned long long ticks)
{
    if (count && *count < threshold) return;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

hipError_t qd_launch_hold(const int32_t *count, int threshold, int microseconds, hipStream_t s)
{
    hipLaunchKernelGGL(qd_hold_kernel, dim3(1), dim3(64), 0, s, count, threshold, (unsigned long long)microseconds * 100ull);
    return hipGetLastError();
}

hipError_t qd_launch_count(const uint8_t *pred, const uint8_t *obs, int k, int64_t B, int64_t *count, hipStream_t s)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(qd_count_mismatch_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, pred, obs, k, B,
                       reinterpret_cast<unsigned long long *>(count));
    return hipGetLastError();
}

hipError_t qd_launch_sample(const SpmatDev &Ht, const SpmatDev &Lt, const uint32_t *thr, uint64_t seed, int64_t shot0,
                            int64_t B, int m, int nobs, uint8_t *det, int64_t det_stride, uint8_t *obs,
                            int64_t obs_stride, hipStream_t s)
{
    if (B == 0) return hipSuccess;
    const size_t lds = sizeof(uint32_t) * (size_t)(((m + 31) >> 5) + ((nobs + 31) >> 5));
    hipLaunchKernelGGL(qd_sample_dem_kernel, dim3((unsigned)B), dim3(256), lds, s, Ht, Lt, thr, (uint32_t)seed,
                       (uint32_t)(seed >> 32), shot0, m, nobs, det, det_stride, obs, obs_stride);
    return hipGetLastError();
}
