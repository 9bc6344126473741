// Issue rate of the vector instructions the BP kernels are made of, relative to v_add_u32: 16 independent accumulators per lane, 8 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/microbench/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(512) k_##NAME(uint32_t *out, int iters, uint32_t sv)                  \
    {                                                                                                       \
        uint32_t a[16], b = threadIdx.x * 2654435761u, c = blockIdx.x + 12345u;                             \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;                                                \
        const uint32_t s = __builtin_amdgcn_readfirstlane(sv);                                              \
        for (int it = 0; it < iters; ++it) {                                                                \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                 \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(s)); \
            }                                                                                               \
        }                                                                                                   \
        uint32_t x = 0; for (int i = 0; i < 16; ++i) x ^= a[i];                                             \
        if (x == 0x12345678u) out[0] = x;                                                                   \
    }
KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL(sub_f32, "v_sub_f32 %0, %0, %1")
KERNEL(add3_u32, "v_add3_u32 %0, %0, %1, 2")
KERNEL(add3_u32_vvv, "v_add3_u32 %0, %0, %1, %2")
KERNEL(xad_u32, "v_xad_u32 %0, %0, %1, %2")
KERNEL(xad_u32_c, "v_xad_u32 %0, %0, %1, 2")
KERNEL(bfe_i32_s, "v_bfe_i32 %0, %0, %3, 1")
KERNEL(bfe_i32_i, "v_bfe_i32 %0, %0, 7, 1")
KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 31, %1")
KERNEL(alignbit, "v_alignbit_b32 %0, %0, %1, 31")
KERNEL(med3_f32, "v_med3_f32 %0, %0, %1, |%2|")
KERNEL(min_f32_abs, "v_min_f32 %0, %0, |%1|")
KERNEL(bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(cmp_addc, "v_cmp_lt_f32 vcc, |%1|, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc")
KERNEL(cmp_eq_cnd, "v_cmp_eq_u32 vcc, 5, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL(lshrrev_s, "v_lshrrev_b32 %0, %3, %0")
KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(mul_i24, "v_mul_i32_i24 %0, %0, %1")
KERNEL(sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(add_f32, "v_add_f32 %0, 1.0, %0")
KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(cmp_e32_addc, "v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc")
KERNEL(cmp_e32_only, "v_cmp_lt_f32 vcc, %1, %0\n\tv_add_u32 %0, %0, %1")
KERNEL(cmp_abs_only, "v_cmp_lt_f32 vcc, |%1|, %0\n\tv_add_u32 %0, %0, %1")
KERNEL(addc_only, "v_addc_co_u32 %0, vcc, %0, %0, vcc")
KERNEL(min_f32_e32, "v_min_f32 %0, %0, %1")
KERNEL(max_f32_e32, "v_max_f32 %0, %0, %1")
KERNEL(and_b32, "v_and_b32 %0, 0x7fffffff, %0")
KERNEL(lshlrev_s, "v_lshlrev_b32 %0, %3, %0")
KERNEL(lshlrev_i, "v_lshlrev_b32 %0, 3, %0")
KERNEL(bfe_u32_i, "v_bfe_u32 %0, %0, 7, 1")
KERNEL(mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(mov_b32, "v_mov_b32 %0, %1")
KERNEL(cndmask_real, "v_cmp_eq_u32 vcc, 5, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc")
KERNEL(bitop3_ea, "v_bitop3_b32 %0, %0, %3, %1 bitop3:0xea")
KERNEL(or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(min3_f32, "v_min3_f32 %0, %0, %1, %2")
KERNEL(add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(sub_f32_abs, "v_sub_f32 %0, |%0|, %1")
KERNEL(fma_mix, "v_fma_f32 %0, %0, 1.0, %1")
KERNEL(dpp_mov, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(sdwa_or, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0")
template <class K> static float run(K k, uint32_t *d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d, 16, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    uint32_t *d; hipMalloc(&d, 64);
    const int iters = 2048;
    const double base = run(k_add_u32, d, iters);
#define R(NAME, N) { const double ms = run(k_##NAME, d, iters); std::printf("%-16s %8.3f ms   %.2f x v_add_u32  (%d instruction(s) per slot)\n", #NAME, ms, ms / base, N); }
    R(add_u32, 1) R(xor_b32, 1) R(sub_u32, 1) R(sub_f32, 1) R(add_f32, 1) R(fma_f32, 1) R(add3_u32, 1) R(add3_u32_vvv, 1) R(xad_u32, 1) R(xad_u32_c, 1)
    R(bfe_i32_s, 1) R(bfe_i32_i, 1) R(lshrrev_s, 1) R(lshl_or, 1) R(lshl_add, 1) R(and_or, 1) R(alignbit, 1) R(perm, 1) R(bitop3, 1) R(med3_f32, 1) R(min_f32_abs, 1)
    R(cvt_f32_i32, 1) R(cndmask, 1) R(mad_i24, 1) R(mul_i24, 1) R(cmp_addc, 2) R(cmp_eq_cnd, 2)
    R(cmp_e32_addc, 2) R(cmp_e32_only, 2) R(cmp_abs_only, 2) R(addc_only, 1) R(min_f32_e32, 1) R(max_f32_e32, 1) R(and_b32, 1) R(lshlrev_s, 1) R(lshlrev_i, 1) R(bfe_u32_i, 1)
    R(mul_f32, 1) R(mov_b32, 1) R(cndmask_real, 3) R(bitop3_ea, 1) R(or3, 1) R(min3_f32, 1) R(add_co, 1) R(cvt_i32_f32, 1) R(sub_f32_abs, 1) R(fma_mix, 1) R(dpp_mov, 1) R(sdwa_or, 1)
    return 0;
}
