// Microbenchmark (tuning aid, not product code): cycles per trellis row of the fill kernel's inner
// loop for ONE wave, as a function of columns per lane K, waves per SIMD and what is left out:
//   mode 0  full row: K ds_read_b64 gathers (prefetch 2 rows), dpp-add, add, max3
//   mode 1  no gathers (operands from registers)
//   mode 2  gathers, but no cross-lane DPP (plain add)
//   mode 3  gathers, row-to-row dependency broken (reads a value that never changes)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/row_rate tools/row_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float dpp_shr1(float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <int K, int MODE>
__global__ void __launch_bounds__(1024) row_kernel(unsigned long long* cyc, float* out, int blocks, const int* labels) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int PITCH = 34;
    const int lane = threadIdx.x & 63;
    using lds_vint = volatile __attribute__((address_space(3))) int;
    lds_vint* flag = (lds_vint*)(smem + 32 * 34 * 8);
    if (threadIdx.x == 0) *flag = 0;
    float2* s2 = reinterpret_cast<float2*>(smem);
    for (int i = threadIdx.x; i < 32 * PITCH; i += blockDim.x) s2[i] = make_float2(-1.0f - 0.01f * (i % 7), -0.5f);
    __syncthreads();
    if (MODE >= 4 && threadIdx.x >= 64 && threadIdx.x < 128) {   // a "producer" that has nothing to do: polls a counter in LDS, as wait_space does
        if (MODE == 4) {
            while (__builtin_amdgcn_ballot_w64(*flag < 1) != 0ull) __builtin_amdgcn_s_sleep(2);
        } else {
            while (__builtin_amdgcn_ballot_w64(*flag < 1) != 0ull) __builtin_amdgcn_s_sleep(100);
        }
        return;
    }
    if (MODE >= 7 && threadIdx.x >= 128) return;
    if (MODE == 6 || MODE == 7) __builtin_amdgcn_s_setprio(1);
    float prev[K];
    uint32_t gaddr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        prev[k] = -0.001f * (lane * K + k);
        gaddr[k] = (unsigned)labels[(lane * K + k) & 1023] * 8u;
    }
    const float frozen = -3.0f + lane;
    const float kProbMax = -1e9f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < blocks; ++b) {
        float2 emq[2][K];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 1) emq[d][k] = make_float2(-1.0f - 0.001f * (float)gaddr[k], -1.0f);
                else emq[d][k] = *reinterpret_cast<const float2*>(smem + gaddr[k] + d * (PITCH * 8));
            }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            float2 em[K];
#pragma unroll
            for (int k = 0; k < K; ++k) em[k] = emq[i % 2][k];
            if (i + 2 < 32 && MODE != 1) {
#pragma unroll
                for (int k = 0; k < K; ++k)
                    emq[i % 2][k] = *reinterpret_cast<const float2*>(smem + gaddr[k] + (i + 2) * (PITCH * 8));
            }
            const float leftv = (MODE == 2) ? prev[K - 1] : dpp_shr1(MODE == 3 ? frozen : prev[K - 1]);
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const float pl = (k == 0) ? leftv : (MODE == 3 ? frozen : prev[k > 0 ? k - 1 : 0]);
                const float pk = MODE == 3 ? frozen : prev[k];
                const float a = pl + em[k].x;
                const float bb = pk + em[k].y;
                const float nw = max3f(a, bb, kProbMax);
                if (MODE == 3) prev[k] = __builtin_fmaxf(prev[k], nw);
                else prev[k] = nw;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" : "+v"(prev[k]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE >= 4 && threadIdx.x == 0) *flag = 1;
    float s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += prev[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int K, int MODE>
void run_pair(const int* d_lab) {   // one computing wave + one polling wave per workgroup, one workgroup per CU
    float* d;
    unsigned long long* c;
    (void)hipMalloc(&d, 512 * 1024 * 4);
    (void)hipMalloc(&c, 512 * 16 * 8);
    const int blocks = 94;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((row_kernel<K, MODE>), dim3(MODE == 8 ? 1 : 256), dim3(MODE >= 7 ? 256 : 128), MODE >= 7 ? 40960 : 32 * 34 * 8 + 64, 0, c, d, blocks, d_lab);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 2);
    (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> t;
    const int nb = MODE == 8 ? 1 : 256, wpb = MODE >= 7 ? 4 : 2;
    h.resize(nb * wpb);
    (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < nb; ++i) t.push_back(h[wpb * i]);
    std::sort(t.begin(), t.end());
    printf("K=%d mode=%d (1 computing + 1 polling wave): %6.1f cyc/row\n", K, MODE, (double)t[nb / 2] / (blocks * 32.0));
    (void)hipFree(d);
    (void)hipFree(c);
}

template <int K, int MODE>
void run(const int* d_lab) {
    float* d;
    unsigned long long* c;
    (void)hipMalloc(&d, 512 * 1024 * 4);
    (void)hipMalloc(&c, 512 * 16 * 8);
    const int blocks = 94;
    printf("K=%d mode=%d:", K, MODE);
    for (int wps : {1, 2, 3, 4}) {   // waves per SIMD: one workgroup of 4*wps waves per CU
        const int nw = 4 * wps;
        hipLaunchKernelGGL((row_kernel<K, MODE>), dim3(256), dim3(64 * nw), 32 * 34 * 8 + 64, 0, c, d, blocks, d_lab);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;   // wall time of the same launch: ns per row (bench.py's issue_floor_ms takes this, not the cycles)
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        // ~0.3 s of the same load first: the card's clocks follow the load (ramp from idle; under a saturating vector load they
        // settle LOWER than under a light one), and the number wanted is the sustained one
        for (int warm = 0; warm < 400; ++warm)
            hipLaunchKernelGGL((row_kernel<K, MODE>), dim3(256), dim3(64 * nw), 32 * 34 * 8 + 64, 0, c, d, 8 * blocks, d_lab);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep)
            hipLaunchKernelGGL((row_kernel<K, MODE>), dim3(256), dim3(64 * nw), 32 * 34 * 8 + 64, 0, c, d, 8 * blocks, d_lab);   // (8 x as long: the launch itself is < 1 % of it)
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 10.f;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        hipLaunchKernelGGL((row_kernel<K, MODE>), dim3(256), dim3(64 * nw), 32 * 34 * 8 + 64, 0, c, d, blocks, d_lab);   // (the cycle counts of a launch of the usual length)
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * nw);
        (void)hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc_row = (double)h[h.size() / 2] / (blocks * 32.0), ns_row = ms * 1e6 / (8 * blocks * 32.0);
        printf("  w%d: %6.1f cyc/row %5.1f ns/row (%.2f GHz)", wps, cyc_row, ns_row, cyc_row / ns_row);
    }
    printf("\n");
    (void)hipFree(d);
    (void)hipFree(c);
}

int main() {
    std::vector<int> lab(1024);
    unsigned x = 12345;
    for (auto& l : lab) { x = x * 1664525u + 1013904223u; l = 1 + (x >> 8) % 31; }
    int* d_lab;
    (void)hipMalloc(&d_lab, 4096);
    (void)hipMemcpy(d_lab, lab.data(), 4096, hipMemcpyHostToDevice);
    run<1, 0>(d_lab); run<2, 0>(d_lab); run<3, 0>(d_lab); run<4, 0>(d_lab);
    run<2, 1>(d_lab); run<3, 1>(d_lab);
    run<2, 2>(d_lab); run<3, 2>(d_lab);
    run<2, 3>(d_lab); run<3, 3>(d_lab);
    run_pair<1, 4>(d_lab); run_pair<2, 4>(d_lab); run_pair<2, 5>(d_lab);
    run_pair<2, 6>(d_lab); run_pair<2, 7>(d_lab); run_pair<2, 8>(d_lab); run_pair<1, 8>(d_lab);
    return 0;
}
