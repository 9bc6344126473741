// Does gfx950 accumulate the IEEE "inexact" status bit (TRAPSTS.EXCP[5]) for plain VALU float adds when traps are
// disabled?  If so a wavefront can certify, at no cost per operation, that every float operation it executed was exact.
//   hipcc --offload-arch=gfx950 -O2 -o build_ablate/inexact_flag tools/ubench/inexact_flag.hip && build_ablate/inexact_flag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t trapsts_excp()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(v));
    return v;
}
__device__ __forceinline__ void trapsts_clear()
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory");
}

__global__ void probe(const float *in, float *outf, uint32_t *out)
{
    const int t = threadIdx.x;
    float a = in[0], b = in[1], c = in[2], big = in[3];
    trapsts_clear();
    const uint32_t e0 = trapsts_excp();
    float r1 = a + b;                                   // 1.0 + 2.0: exact
    asm volatile("s_nop 7" : "+v"(r1));
    const uint32_t e1 = trapsts_excp();
    float r2 = fminf(fabsf(r1 - a), b);                 // sub, min, abs: exact
    asm volatile("s_nop 7" : "+v"(r2));
    const uint32_t e2 = trapsts_excp();
    float r3 = big - c;                                 // inf - x: exact (no inexact, no overflow)
    asm volatile("s_nop 7" : "+v"(r3));
    const uint32_t e3 = trapsts_excp();
    float r4 = (t == 5) ? a + c : a + b;                // lane 5 only: 1.0 + 2^-30 -> inexact
    asm volatile("s_nop 7" : "+v"(r4));
    const uint32_t e4 = trapsts_excp();
    trapsts_clear();
    const uint32_t e5 = trapsts_excp();
    float r6 = 16777216.0f + a;                         // 2^24 + 1: inexact
    asm volatile("s_nop 7" : "+v"(r6));
    const uint32_t e6 = trapsts_excp();
    if (t == 0) { out[0] = e0; out[1] = e1; out[2] = e2; out[3] = e3; out[4] = e4; out[5] = e5; out[6] = e6; }
    outf[t] = r1 + r2 + r3 + r4 + r6;
}

int main()
{
    float h[4] = {1.0f, 2.0f, 9.313225746154785e-10f, __builtin_inff()};
    float *din, *df; uint32_t *dout, ho[8] = {0};
    hipMalloc(&din, sizeof(h)); hipMalloc(&df, 64 * 4); hipMalloc(&dout, 32);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, df, dout);
    hipMemcpy(ho, dout, 28, hipMemcpyDeviceToHost);
    const char *what[7] = {"after clear", "after exact add", "after exact sub/min/abs", "after inf - x", "after one lane's inexact add",
                           "after clear", "after 2^24 + 1"};
    for (int i = 0; i < 7; ++i) printf("TRAPSTS.EXCP %-32s 0x%03x  inexact=%u\n", what[i], ho[i], (ho[i] >> 5) & 1u);
    return 0;
}
