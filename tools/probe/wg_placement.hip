// Scratch probe (not part of the library): where do the workgroups of a grid that exactly fills the chip land?
//   hipcc --offload-arch=gfx950 -O2 wg_placement.hip -o wg_placement && ./wg_placement <threads> <lds_kb> <nblocks>
// Prints, per workgroup in blockIdx order: XCC id, SE / CU id, and the order in which it arrived on its CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
__global__ void probe(unsigned* out, unsigned long long* clk)
{
    extern __shared__ double sm[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        clk[blockIdx.x] = wall_clock64();
        sm[0] = 1.0;
    }
    // stay resident for a while so that the whole grid must be co-resident
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 200000ull) __builtin_amdgcn_s_sleep(10);     // 2 ms
}
int main(int argc, char** argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 192, ldskb = argc > 2 ? atoi(argv[2]) : 36, nb = argc > 3 ? atoi(argv[3]) : 1024;
    unsigned* d; unsigned long long* c;
    hipMalloc(&d, nb * 8); hipMalloc(&c, nb * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, ldskb * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), ldskb * 1024, 0, d, c);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * nb); std::vector<unsigned long long> hc(nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, nb * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0; for (auto v : hc) { if (v < t0) t0 = v; if (v > t1) t1 = v; }
    printf("threads %d lds %d KB blocks %d: start spread %.1f us (<< 2000 us means all co-resident)\n", threads, ldskb, nb, (t1 - t0) / 100.0);
    std::map<unsigned, std::vector<int>> percu;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        percu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("distinct CUs %zu\n", percu.size());
    int shown = 0;
    std::map<size_t, int> hist;
    for (auto& kv : percu) {
        hist[kv.second.size()]++;
        if (shown < 12) { printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); ++shown; }
    }
    for (auto& kv : hist) printf("CUs with %zu workgroups: %d\n", kv.first, kv.second);
    // is the slot (arrival order on the CU) a simple function of blockIdx?
    int ok256 = 0, tot = 0;
    for (auto& kv : percu) for (size_t s = 0; s < kv.second.size(); ++s) { ++tot; if ((size_t)(kv.second[s] / 256) == s) ++ok256; }
    printf("blocks whose slot == blockIdx / 256: %d of %d\n", ok256, tot);
    return 0;
}
