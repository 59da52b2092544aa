// Microbenchmark (tuning aid, not product code): LDS instruction cost per CU on gfx950 for
// the access shapes the CTC fill kernel uses.  Every wave issues ITERS x 16 LDS instructions
// (lgkmcnt drained every 16); s_memtime brackets the loop.  Output: LDS-pipe cycles per
// wave-instruction per CU with 4/8/16 waves resident on the CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(1024) lds_kernel(unsigned long long* cyc, float* out, int iters, const int* labels) {
    __shared__ __align__(16) unsigned char smem[32768];
    const int lane = threadIdx.x & 63;
    float4* s4 = reinterpret_cast<float4*>(smem);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s4[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const unsigned lab = (unsigned)labels[threadIdx.x & 63];
    unsigned a_gather8 = lab * 8u;          // 32 entries x 8 B
    unsigned a_gather16 = lab * 16u;        // 32 entries x 16 B
    unsigned a_bcast = 4096u;               // same address for every lane
    unsigned a_lane4 = 8192u + lane * 4u;   // distinct dword per lane
    unsigned a_lane16 = 8192u + lane * 16u; // distinct 16 B per lane
    float acc = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f4 v4 = {(float)lane, 1.f, 2.f, 3.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { f2 r; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(a_gather8), "i"(u * 264)); acc += 0; }
            if (OP == 1) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a_bcast), "i"(u * 4)); }
            if (OP == 2) { f4 r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a_bcast), "i"(u * 16)); }
            if (OP == 3) { asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(a_lane4), "v"(acc), "i"(u * 4)); }
            if (OP == 4) { asm volatile("s_mov_b64 exec, 1\n\tds_write_b32 %0, %1 offset:%2\n\ts_mov_b64 exec, -1" :: "v"(a_lane4), "v"(acc), "i"(u * 4)); }
            if (OP == 5) { asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a_lane16), "v"(v4), "i"(u * 16)); }
            if (OP == 6) { asm volatile("s_mov_b64 exec, 1\n\tds_write_b128 %0, %1 offset:%2\n\ts_mov_b64 exec, -1" :: "v"(a_lane16), "v"(v4), "i"(u * 16)); }
            if (OP == 7) { f4 r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a_gather16), "i"(u * 528)); }
            if (OP == 8) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a_gather8), "i"(u * 264)); }
            if (OP == 10) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(lab * 4u), "i"(u * 136)); }
            if (OP == 11) { f2 r; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(lab * 4u), "i"(u), "i"(u + 34)); }
            if (OP == 9) { asm volatile("s_mov_b64 exec, 1\n\tds_read_b32 %0, %1 offset:%2\n\ts_mov_b64 exec, -1" : "=v"(acc) : "v"(a_bcast), "i"(u * 4)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v4.x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, const int* d_lab) {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 1024 * 4); (void)hipMalloc(&c, 256 * 8);
    const int iters = 20000;
    printf("%-34s", name);
    for (int waves : {4, 8, 16}) {
        lds_kernel<OP><<<256, 64 * waves>>>(c, d, iters, d_lab);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        (void)hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  %2d waves: %5.2f", waves, (double)h[128] / ((double)iters * 16 * waves));
    }
    printf("   cyc/inst/CU\n");
    (void)hipFree(d); (void)hipFree(c);
}

int main() {
    int lab[64];
    unsigned x = 12345;
    for (int i = 0; i < 64; ++i) { x = x * 1664525u + 1013904223u; lab[i] = (x >> 16) % 32; }
    int* d_lab; (void)hipMalloc(&d_lab, sizeof(lab)); (void)hipMemcpy(d_lab, lab, sizeof(lab), hipMemcpyHostToDevice);
    run<0>("ds_read_b64 gather(32x8B)", d_lab);
    run<8>("ds_read_b32 gather(32x8B)", d_lab);
    run<10>("ds_read_b32 gather(32x4B)", d_lab);
    run<11>("ds_read2_b32 gather(32x4B, 2 rows)", d_lab);
    run<7>("ds_read_b128 gather(32x16B)", d_lab);
    run<1>("ds_read_b32 broadcast", d_lab);
    run<9>("ds_read_b32 broadcast exec=1", d_lab);
    run<2>("ds_read_b128 broadcast", d_lab);
    run<3>("ds_write_b32 all lanes", d_lab);
    run<4>("ds_write_b32 exec=1 lane", d_lab);
    run<5>("ds_write_b128 all lanes", d_lab);
    run<6>("ds_write_b128 exec=1 lane", d_lab);
    return 0;
}
