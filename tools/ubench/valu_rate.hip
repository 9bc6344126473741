// Issue rate of individual gfx950 VALU / SALU instructions at full occupancy (8 waves per SIMD), to price kernel inner loops.
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int K>
__global__ void __launch_bounds__(512) k(uint32_t *out, int iters, unsigned long long *cyc)
{
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 9, e = 11, f = 13, g2 = 5, h = 3;
    uint64_t p = threadIdx.x, q = 12345;
    float fa = a, fb = b, fc = c, fd = d;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (K == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 1) { REP16(asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(p) : "v"(b));) }
        if (K == 2) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 31, %1" : "+v"(a) : "v"(b));) }
        if (K == 3) { REP16(asm volatile("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:WORD_0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
        if (K == 4) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));) }
        if (K == 5) { REP16(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));) }
        if (K == 6) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a) : "v"(b));) }
        if (K == 7) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(p) : "v"(q));) }
        if (K == 8) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a) : "v"(b));) }
        if (K == 9) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(fa) : "v"(fb) : "vcc");) }
        if (K == 10) { REP16(asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, 1" : "+v"(a), "+s"(c) : "v"(b));) }          // VALU + SALU pair
        if (K == 11) { REP16(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(d) : "v"(b));) }           // two independent
        if (K == 12) { REP16(asm volatile("v_min_f32 %0, %0, |%1|" : "+v"(fa) : "v"(fb));) }
        if (K == 13) { REP16(asm volatile("v_cmp_ge_f32 %0, 0, %1" : "=s"(q) : "v"(fb));) }
        if (K == 14) { REP16(asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, %0" : "+s"(c), "+s"(d));) }                    // SALU only
        if (K == 15) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
        if (K == 16) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));) }
        if (K == 17) { REP16(asm volatile("v_lshlrev_b32 %0, 17, %0\n v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");) }
        if (K == 18) { REP16(asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(c) : "v"(a));) }
        if (K == 19) { REP16(asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(a));) }
        if (K == 20) { REP16(asm volatile("v_bfe_u32 %0, %0, 16, 15" : "+v"(a));) }
        if (K == 21) { REP16(asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(e));) }
        if (K == 22) { REP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 24) { REP16(asm volatile("v_lshlrev_b32 %0, 31, %0\n v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 25) { REP16(asm volatile("v_min_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
        if (K == 26) { REP16(asm volatile("v_and_b32 %0, 0x7fffffff, %0\n v_min_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
        if (K == 27) { REP16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
        if (K == 28) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
        if (K == 29) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));) }
        if (K == 30) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(q));) }
        if (K == 31) { REP16(asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a) : "v"(b));) }
        if (K == 32) { REP16(asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(a));) }
        if (K == 33) { REP16(asm volatile("v_max_f32 %0, %0, %1" : "+v"(fa) : "v"(fb));) }
        if (K == 34) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));) }
        if (K == 35) { REP16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 36) { REP16(asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");) }
        if (K == 37) { REP16(asm volatile("v_cmp_eq_u16 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");) }
        if (K == 38) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_lshl_or_b32 %2, %2, 31, %1" : "+v"(a), "+v"(d) : "v"(b));) }
        if (K == 39) { REP16(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(e));) }
        if (K == 40) { REP16(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (K == 41) { REP16(asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(a));) }
        if (K == 23) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1" : "+v"(d) : "v"(a));) }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g2 + h + (uint32_t)p + (uint32_t)q + (uint32_t)(fa + fb + fc + fd);
    if (threadIdx.x == 0) atomicAdd(cyc, t1 - t0);
}
template <int K> static void run(const char *name, int per_rep, uint32_t *out, unsigned long long *cyc)
{
    const int iters = 2000, blocks = 256 * 4;   // 4 x 512 threads per CU = 8 waves per SIMD
    hipMemset(cyc, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K><<<blocks, 512>>>(out, 10, cyc); hipDeviceSynchronize(); hipMemset(cyc, 0, 8);
    hipEventRecord(e0); k<K><<<blocks, 512>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double cycles = (double)c / blocks;                        // block lifetime in clock64 ticks
    const double instr_per_simd = (double)iters * 16 * per_rep * 8;  // 8 waves per SIMD
    printf("%-44s %7.3f ms   %6.3f ns per instruction per SIMD   %6.2f ticks (%d instr / rep)\n", name, ms, (ms - 0.02) * 1e6 / instr_per_simd, cycles / instr_per_simd, per_rep);
}
int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    uint32_t *out; unsigned long long *cyc; hipMalloc(&out, 4 * 512 * 1024); hipMalloc(&cyc, 8);
    run<0>("v_add_u32", 1, out, cyc);
    run<1>("v_lshrrev_b64", 1, out, cyc);
    run<2>("v_lshl_or_b32", 1, out, cyc);
    run<3>("v_cmp_eq_u32_sdwa + v_cndmask", 2, out, cyc);
    run<4>("v_pk_add_f32", 1, out, cyc);
    run<5>("v_med3_f32", 1, out, cyc);
    run<6>("v_alignbit_b32", 1, out, cyc);
    run<7>("v_lshl_add_u64", 1, out, cyc);
    run<8>("v_add_u32_sdwa", 1, out, cyc);
    run<9>("v_cmp_lt_f32 + v_cndmask", 2, out, cyc);
    run<11>("v_add_u32 x2 independent", 2, out, cyc);
    run<12>("v_min_f32 |abs|", 1, out, cyc);
    run<13>("v_cmp_ge_f32 -> sgpr", 1, out, cyc);
    run<15>("v_mul_f32", 1, out, cyc);
    run<16>("v_pk_mul_f32", 1, out, cyc);
    run<17>("v_lshlrev + v_cmp_eq + v_cndmask", 3, out, cyc);
    run<18>("v_readfirstlane_b32", 1, out, cyc);
    run<19>("v_mov_b32", 1, out, cyc);
    run<20>("v_bfe_u32", 1, out, cyc);
    run<21>("v_and_or_b32", 1, out, cyc);
    run<22>("v_xor_b32", 1, out, cyc);
    run<23>("v_mov_b32_dpp row_shr", 1, out, cyc);
    run<24>("v_lshlrev_b32 + v_or_b32", 2, out, cyc);
    run<25>("v_min_f32 (e32)", 1, out, cyc);
    run<26>("v_and_b32 literal + v_min_f32", 2, out, cyc);
    run<27>("v_sub_f32", 1, out, cyc);
    run<28>("v_add_f32", 1, out, cyc);
    run<29>("v_cndmask_b32 vcc (e32)", 1, out, cyc);
    run<30>("v_cndmask_b32_e64 sgpr mask", 1, out, cyc);
    run<31>("v_lshrrev_b32 vgpr shift", 1, out, cyc);
    run<32>("v_lshrrev_b32 const shift", 1, out, cyc);
    run<33>("v_max_f32", 1, out, cyc);
    run<34>("v_fma_f32", 1, out, cyc);
    run<35>("v_and_b32", 1, out, cyc);
    run<36>("v_cmp_eq_u32 vcc (e32)", 1, out, cyc);
    run<37>("v_cmp_eq_u16 vcc (e32)", 1, out, cyc);
    run<38>("v_add_u32 + v_lshl_or_b32 independent", 2, out, cyc);
    run<39>("v_bfi_b32", 1, out, cyc);
    run<40>("v_sub_u32", 1, out, cyc);
    run<41>("v_xor_b32 literal", 1, out, cyc);
    return 0;
}
