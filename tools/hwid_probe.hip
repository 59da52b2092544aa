// Probe (tuning aid): where do the 6 waves of each fill-kernel workgroup land?  Same launch
// shape as config 3 (512 workgroups x 384 threads, 57 KB dynamic LDS): prints, per CU, the
// number of waves per SIMD and which wave indices they are.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(1024) probe(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep every workgroup resident for a while so that all 512 are placed together
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { smem[threadIdx.x] = 1; }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main(int argc, char** argv) {
    const int B = 512, waves = argc > 1 ? atoi(argv[1]) : 6, lds = argc > 2 ? atoi(argv[2]) : 58000;
    unsigned* d; (void)hipMalloc(&d, B * 16 * 2 * 4); (void)hipMemset(d, 0xff, B * 16 * 2 * 4);
    (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<<<B, 64 * waves, lds>>>(d, 200000);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(B * 16 * 2);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // key: (xcc, se, sh, cu) -> simd -> list of (block, wave)
    std::map<unsigned, std::map<int, std::vector<std::pair<int,int>>>> m;
    for (int b = 0; b < B; ++b) for (int w = 0; w < waves; ++w) {
        unsigned hw = h[(b * 16 + w) * 2], xcc = h[(b * 16 + w) * 2 + 1] & 0xf;
        int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        m[(xcc << 16) | (se << 8) | (sh << 4) | cu][simd].push_back({b, w});
    }
    printf("CUs seen: %zu\n", m.size());
    int shown = 0; std::map<std::string, int> hist;
    for (auto& cu : m) {
        char key[256]; int n = 0;
        for (int s = 0; s < 4; ++s) {
            n += snprintf(key + n, sizeof(key) - n, "[");
            for (auto& bw : cu.second[s]) n += snprintf(key + n, sizeof(key) - n, "%c%d", 'A' + (bw.first != cu.second[0].empty() ? 0 : 0), bw.second);
            n += snprintf(key + n, sizeof(key) - n, "] ");
        }
        hist[key]++;
        if (shown++ < 4) {
            printf("cu %06x:", cu.first);
            for (int s = 0; s < 4; ++s) { printf("  simd%d:", s); for (auto& bw : cu.second[s]) printf(" b%d.w%d", bw.first, bw.second); }
            printf("\n");
        }
    }
    for (auto& kv : hist) printf("%4d CUs  wave indices per SIMD: %s\n", kv.second, kv.first.c_str());
    return 0;
}
