// Microbenchmark (tuning aid): issue cost of v_pk_add_f32 vs v_add_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(64) k(unsigned long long* cyc, float* out, int iters, float seed) {
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{seed + i + threadIdx.x, seed * i};
    f2 c = {seed, seed * 0.5f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define R8(S) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) : "+v"(p[0]),"+v"(p[1]),"+v"(p[2]),"+v"(p[3]),"+v"(p[4]),"+v"(p[5]),"+v"(p[6]),"+v"(p[7]) : "v"(c));
#define PK(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define PKM(i) "v_pk_add_f32 %" #i ", %" #i ", %8 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
#define PKF(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define ADD2(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define MOV64(i) "v_mov_b64 %" #i ", %8\n"
        if (OP == 0) R8(PK)
        if (OP == 1) R8(PKM)
        if (OP == 2) R8(PKF)
        if (OP == 3) R8(PKMUL)
        if (OP == 4) R8(MOV64)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name) {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 1024 * 8 * 64 * 4); (void)hipMalloc(&c, 1024 * 8 * 8);
    const int iters = 100000;
    printf("%-42s", name);
    for (int wps : {1, 2, 4, 8}) {
        int grid = 1024 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        k<OP><<<grid, 64>>>(c, d, iters, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), c, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  w%d: %5.2f (%.0f ticks/us)", wps, (double)h[grid / 2] / ((double)iters * 8 * wps), (double)h[grid / 2] / (ms * 1e3));
    }
    printf("   ticks/inst/SIMD\n");
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    run<0>("v_pk_add_f32");
    run<1>("v_pk_add_f32 op_sel_hi/neg");
    run<2>("v_pk_fma_f32");
    run<3>("v_pk_mul_f32");
    run<4>("v_mov_b64");
    return 0;
}
