// Diagnostic micro-benchmark (not part of the product): issue cost of the integer VALU operations the scan kernel is made of,
// per wave-instruction, with 1, 2 and 4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define OPS8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(1024) void bench(unsigned *out, unsigned long long *cyc, unsigned seed) {
    unsigned a[8], b = seed * 3u + threadIdx.x, c = seed * 7u + 1u;
    for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#define ONE(i)                                                                                                      \
    if (OP == 0) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf6" : "+v"(a[i]) : "v"(b), "v"(c));            \
    if (OP == 1) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                           \
    if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                       \
    if (OP == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                        \
    if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                      \
    if (OP == 5) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(*(unsigned long long *)&a[i & 6]));                  \
    if (OP == 6) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                           \
    if (OP == 7) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[0]) : "v"(b), "v"(c));            \
    if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                        \
    if (OP == 9) asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            OPS8(ONE)
#undef ONE
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(const char *name) {
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 16 * 8);
    for (int threads : {256, 512, 1024}) {
        bench<OP><<<256, threads>>>(out, cyc, 1);
        bench<OP><<<256, threads>>>(out, cyc, 2);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double per = sum / h.size() / (REP * 32.0);
        printf("%-34s waves/SIMD %d: %.2f cycles per instruction per wave -> %.2f per SIMD\n", name, threads / 256, per, per / (threads / 256));
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_bitop3_b32 (8 chains)");
    run<7>("v_bitop3_b32 (1 dependent chain)");
    run<1>("v_or3_b32");
    run<2>("v_xor_b32");
    run<3>("v_and_or_b32");
    run<4>("v_alignbit_b32");
    run<5>("v_lshrrev_b64");
    run<6>("v_xad_u32");
    run<8>("v_add_u32");
    run<9>("v_bfe_u32");
    return 0;
}
