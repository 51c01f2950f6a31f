// tools/probe/lds_atomic_probe.hip -- what does ONE LDS atomic wave-instruction cost on this part?  Every wave issues `n` independent returnless
// LDS adds (no dependent read between them) in one of four address patterns, with 1-8 waves per SIMD resident; the clock per instruction and CU
// follows from the kernel time.  Motivation: K3 (histogram binning) and the K8 radix histogram both run at about one LDS atomic instruction per
// 35-60 cycles and CU, whatever the bank conflicts (profiles/r05_notes.md): is that the LDS atomic unit?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_atomic_probe tools/probe/lds_atomic_probe.hip && /tmp/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// KIND 0: ds_add_u32, 1: ds_add_u64, 2: ds_add_f64, 3: plain ds_write_b32 (for comparison), 4: ds_add_rtn_u32 (value used)
// PAT 0: lane-private address (conflict-free), 1: all lanes one address, 2: pseudo-random over 256 words, 3: 8 lanes per address
template <int KIND, int PAT>
__global__ __launch_bounds__(1024) void k_probe(int n, unsigned* sink)
{
    __shared__ unsigned long long s[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) s[i] = 0;
    __syncthreads();
    unsigned h = tid * 2654435761u;
    unsigned acc = 0;
    for (int r = 0; r < n; ++r) {
        h = h * 1664525u + 1013904223u;
        int a;
        if (PAT == 0) a = lane;
        else if (PAT == 1) a = 0;
        else if (PAT == 2) a = (h >> 16) & 255;
        else a = lane >> 3;
        a += wave * 256;                                   // wave-private region
        if (KIND == 0) atomicAdd((unsigned*)&s[a], 1u);
        else if (KIND == 1) atomicAdd(&s[a], 1ull);
        else if (KIND == 2) atomicAdd((double*)&s[a], 1.0);
        else if (KIND == 3) ((volatile unsigned*)&s[a])[0] = h;
        else acc += atomicAdd((unsigned*)&s[a], 1u);
    }
    __syncthreads();
    if (tid == 0) sink[blockIdx.x] = (unsigned)s[lane] + (unsigned)s[256] + acc;
}

template <int KIND, int PAT>
static int run(const char* name, int threads, hipStream_t st, unsigned* sink, int cus)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 4096, blocks = cus;                      // one block per CU (16 KB + 32 KB LDS: the launch bounds leave room for one 1024-thread block)
    hipLaunchKernelGGL((k_probe<KIND, PAT>), dim3(blocks), dim3(threads), 0, st, n, sink);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_probe<KIND, PAT>), dim3(blocks), dim3(threads), 0, st, n, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int clk_khz = 0; CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    const double us = ms * 1e3 / 5, instr = (double)n * (threads / 64);
    printf("{\"op\": \"%s\", \"waves_per_cu\": %d, \"us\": %.1f, \"ns_per_wave_instr_per_cu\": %.2f, \"clk_per_wave_instr_at_%d_MHz\": %.1f}\n",
           name, threads / 64, us, us * 1e3 / instr, clk_khz / 1000, us * 1e3 / instr * clk_khz / 1e6);
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    unsigned* sink; CK(hipMalloc(&sink, 4096 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int cus = p.multiProcessorCount;
    // several workgroups per CU (the radix histogram has eight 256-thread workgroups per CU): 4 x 256 threads against 1 x 1024
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int bpc : {1, 2, 4}) for (int n : {4096, 16}) {
            const int reps = n == 16 ? 200 : 5, blocks = cus * bpc * (n == 16 ? 8 : 1);      // n = 16: many short workgroups, like one tile each
            hipLaunchKernelGGL((k_probe<0, 2>), dim3(blocks), dim3(256), 0, st, n, sink);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_probe<0, 2>), dim3(blocks), dim3(256), 0, st, n, sink);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps, instr = (double)n * 4 * blocks / cus;
            printf("{\"op\": \"ds_add_u32 random, 256-thread workgroups\", \"workgroups_per_cu_resident\": %d, \"adds_per_wave\": %d, \"workgroups\": %d, \"us\": %.1f, \"ns_per_wave_instr_per_cu\": %.2f}\n",
                   bpc, n, blocks, us, us * 1e3 / instr);
        }
    }
    for (int threads : {256, 1024}) {
        if (run<0, 0>("ds_add_u32 lane-private", threads, st, sink, cus)) return 1;
        if (run<0, 2>("ds_add_u32 random over 256 words", threads, st, sink, cus)) return 1;
        if (run<0, 3>("ds_add_u32 8 lanes per address", threads, st, sink, cus)) return 1;
        if (run<0, 1>("ds_add_u32 one address", threads, st, sink, cus)) return 1;
        if (run<1, 0>("ds_add_u64 lane-private", threads, st, sink, cus)) return 1;
        if (run<2, 0>("ds_add_f64 lane-private", threads, st, sink, cus)) return 1;
        if (run<2, 2>("ds_add_f64 random over 256 words", threads, st, sink, cus)) return 1;
        if (run<3, 0>("ds_write_b32 lane-private", threads, st, sink, cus)) return 1;
        if (run<4, 0>("ds_add_rtn_u32 lane-private", threads, st, sink, cus)) return 1;
    }
    return 0;
}
