// Microbenchmark (tuning aid): issue cost of further VALU candidates for the fill kernel's max step (gfx950) --
// integer min / max (non-positive floats order like unsigned integers, reversed), and the adds beside them.
// Same method as tools/valu_rate.hip: 8 independent instructions per asm statement, s_memtime around the loop.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate2 tools/valu_rate2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define OPERANDS : "+v"(u[0]),"+v"(u[1]),"+v"(u[2]),"+v"(u[3]),"+v"(u[4]),"+v"(u[5]),"+v"(u[6]),"+v"(u[7]) : "v"(seed), "s"(sc), "v"(seed2) : "vcc"

template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(unsigned long long* cyc, unsigned* out, int iters, unsigned seed, unsigned sc, unsigned seed2) {
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = i * 77u + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define I0(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define I1(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I2(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n"
#define I3(n) "v_max3_i32 %" #n ", %" #n ", %8, %10\n"
#define I4(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I5(n) "v_add_f32 %" #n ", %" #n ", %9\n"
#define I6(n) "v_max_f32 %" #n ", %" #n ", %9\n"
#define I7(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define I8(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I9(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I10(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I11(n) "v_add_f32_dpp %" #n ", %10, %" #n " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I12(n) "v_add_f32_dpp %" #n ", %10, %" #n " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I13(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define I14(n) "v_med3_f32 %" #n ", %" #n ", %8, %10\n"
#define I15(n) "v_mov_b32 %" #n ", %8\n"
        if (OP == 0) asm volatile(REP8(I0) OPERANDS);
        if (OP == 1) asm volatile(REP8(I1) OPERANDS);
        if (OP == 2) asm volatile(REP8(I2) OPERANDS);
        if (OP == 3) asm volatile(REP8(I3) OPERANDS);
        if (OP == 4) asm volatile(REP8(I4) OPERANDS);
        if (OP == 5) asm volatile(REP8(I5) OPERANDS);
        if (OP == 6) asm volatile(REP8(I6) OPERANDS);
        if (OP == 7) asm volatile(REP8(I7) OPERANDS);
        if (OP == 8) asm volatile(REP8(I8) OPERANDS);
        if (OP == 9) asm volatile(REP8(I9) OPERANDS);
        if (OP == 10) asm volatile(REP8(I10) OPERANDS);
        if (OP == 11) asm volatile(REP8(I11) OPERANDS);
        if (OP == 12) asm volatile(REP8(I12) OPERANDS);
        if (OP == 13) asm volatile(REP8(I13) OPERANDS);
        if (OP == 14) asm volatile(REP8(I14) OPERANDS);
        if (OP == 15) asm volatile(REP8(I15) OPERANDS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v ^= u[i];
    out[blockIdx.x * 64 + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name) {
    unsigned* d; unsigned long long* c;
    (void)hipMalloc(&d, 1024 * 8 * 64 * 4); (void)hipMalloc(&c, 1024 * 8 * 8);
    const int iters = 50000;
    printf("%-42s", name);
    for (int wps : {1, 2, 3, 4, 8}) {
        int grid = 1024 * wps;
        rate_kernel<OP><<<grid, 64>>>(c, d, iters, 0xbf800000u, 0xc0000000u, 0xbf000000u);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), c, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  w%d: %5.2f", wps, (double)h[grid / 2] / ((double)iters * 8 * wps));
    }
    printf("   cyc/inst/SIMD\n");
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    run<0>("v_min_u32 (vop2)");
    run<1>("v_max_i32 (vop2)");
    run<2>("v_min3_u32 v,v,s");
    run<3>("v_max3_i32 v,v,v");
    run<4>("v_add_f32 v,v,v (vop2)");
    run<5>("v_add_f32 v,v,s");
    run<6>("v_max_f32 v,v,s");
    run<7>("v_pk_max_i16");
    run<8>("v_sub_u32");
    run<9>("v_xor_b32");
    run<10>("v_mul_f32");
    run<11>("v_add_f32_dpp wave_shr:1");
    run<12>("v_add_f32_dpp row_shr:1");
    run<13>("v_min_f32");
    run<14>("v_med3_f32");
    run<15>("v_mov_b32");
    return 0;
}
