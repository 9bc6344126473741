// Throughput of scattered LDS accesses on gfx950 at the occupancy of the flooding min-sum kernel (two 1024-thread workgroups per
// CU, 76 KB of LDS each): ds_read_b32 gathers, ds_read_b128 gathers, ds_add_f32 (no return) scatters, and gather + scatter mixed.
// Addresses are pseudo-random 4-byte (16-byte for b128) slots of a 38 KB array: the access pattern of a check walking its faults.
// build: hipcc --offload-arch=gfx950 -O2 -o lds_rate lds_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int K>
__global__ void __launch_bounds__(1024, 8) k(float *out, int iters, int same_bank, int dup)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *arr = reinterpret_cast<float *>(smem);
    const int N = 9504;
    for (int i = threadIdx.x; i < 2 * N; i += 1024) arr[i] = 1.0f;
    __syncthreads();
    uint32_t ad[8];
    uint32_t s = (threadIdx.x / dup) * 2654435761u + 12345u;          // dup consecutive lanes share every address
    for (int j = 0; j < 8; ++j) {
        s = s * 1664525u + 1013904223u;
        uint32_t slot = (s >> 8) % N;
        if (same_bank == 1) slot = (((threadIdx.x / dup) * 37 + j * 101) % N);            // stride pattern: distinct banks in a group of 32
        ad[j] = (K == 1) ? ((slot & ~3u) * 4u) : slot * 4u;
    }
    float acc = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (K == 0) { float v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad[j]) : "memory"); acc += v; }
            if (K == 1) { f4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad[j]) : "memory"); acc += v.x; }
            if (K == 2) { asm volatile("ds_add_f32 %0, %1" : : "v"(ad[j]), "v"(1.0f) : "memory"); }
            if (K == 3) { float v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad[j]) : "memory"); acc += v;
                          asm volatile("ds_add_f32 %0, %1 offset:38016" : : "v"(ad[(j + 3) & 7]), "v"(1.0f) : "memory"); }
            if (K == 4) { asm volatile("ds_add_u32 %0, %1" : : "v"(ad[j]), "v"(1u) : "memory"); }
            if (K == 6 || K == 7 || K == 8) {       // the scatter loop's mix: 8 VALU instructions per ds_add_u32 (6: both, 7: VALU only, 8: add only)
                uint32_t w = ad[j] ^ (uint32_t)i, v;
                if (K != 8) asm volatile("v_bfe_i32 %0, %1, 30, 1\n v_xor_b32 %0, %0, %1\n v_sub_u32 %0, %0, %1\n v_bfe_i32 %1, %1, 29, 1\n"
                                         "v_xor_b32 %1, %1, %0\n v_sub_u32 %0, %1, %0\n v_lshlrev_b32 %1, 1, %1\n v_sub_u32 %0, %0, %1" : "=&v"(v), "+v"(w));
                else v = w;
                if (K != 7) asm volatile("ds_add_u32 %0, %1" : : "v"(ad[j]), "v"(v) : "memory");
                else acc += __uint_as_float(v & 1u);
            }
            if (K == 5) { float v; asm volatile("ds_add_rtn_f32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad[j]), "v"(1.0f) : "memory"); acc += v; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 1024 + threadIdx.x] = acc + arr[threadIdx.x];
}
template <int K> static void run(const char *name, int per_rep, float *out, int same_bank, int dup = 1)
{
    const int iters = 2000, blocks = 256 * 2 * 4;
    auto kk = k<K>;
    hipFuncSetAttribute((const void *)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 77824);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kk, dim3(blocks), dim3(1024), 77824, 0, out, 10, same_bank, dup); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kk, dim3(blocks), dim3(1024), 77824, 0, out, iters, same_bank, dup); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU: blocks / 256 workgroups in sequence-pairs, each 16 waves x iters x 8 x per_rep LDS instructions
    const double wave_instr_per_cu = (double)blocks / 256 * 16 * iters * 8 * per_rep;
    printf("%-52s %8.3f ms   %6.2f ns per wave-instruction per CU (%s)\n", name, ms, ms * 1e6 / wave_instr_per_cu, hipGetErrorString(hipGetLastError()));
}
int main()
{
    setvbuf(stdout, NULL, _IONBF, 0);
    float *out; hipMalloc(&out, 4 * 1024 * 2048);
    for (int sb = 0; sb < 2; ++sb) {
        printf("# addresses: %s\n", sb ? "strided (conflict-free within 32 lanes)" : "pseudo-random");
        run<0>("ds_read_b32 gather + wait", 1, out, sb);
        run<1>("ds_read_b128 gather + wait", 1, out, sb);
        run<2>("ds_add_f32 scatter (no return, no wait)", 1, out, sb);
        run<4>("ds_add_u32 scatter (no return, no wait)", 1, out, sb);
        run<5>("ds_add_rtn_f32 + wait", 1, out, sb);
        run<3>("ds_read_b32 gather + wait, ds_add_f32 scatter", 2, out, sb);
        run<4>("ds_add_u32 scatter, lane pairs share an address", 1, out, sb, 2);
        run<4>("ds_add_u32 scatter, 4 lanes share an address", 1, out, sb, 4);
        run<0>("ds_read_b32 gather, lane pairs share an address", 1, out, sb, 2);
        run<7>("8 VALU instructions", 1, out, sb);
        run<8>("ds_add_u32 (value from a register)", 1, out, sb);
        run<6>("8 VALU instructions + ds_add_u32", 1, out, sb);
    }
    return 0;
}
