// tools/probe/seam_probe.hip -- what does a SEAM between two dependent phases cost on this part: a kernel boundary (N small kernels back to back on
// one stream) against a grid barrier inside ONE resident launch (monotonic counter, relaxed polling, one release / acquire fence pair per barrier)?
// The round-4 review proposed to run the ~18 phases of the small-plane sort (K8, cfg5) as one persistent launch; this probe prices the trade.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/seam_probe tools/probe/seam_probe.hip && /tmp/seam_probe [blocks] [phases]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ void phase_work(double* buf, int n, int phase)
{
    // a little dependent work per phase: every thread touches 4 elements another block wrote in the phase before
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int k = 0; k < 4; ++k) {
        const int i = (t * 4 + k + phase * 977) % n;
        buf[i] = buf[(i + nt) % n] * 0.5 + 1.0;
    }
}

__global__ __launch_bounds__(256) void k_phase(double* buf, int n, int phase) { phase_work(buf, n, phase); }

__global__ __launch_bounds__(256) void k_persistent(double* buf, int n, int phases, unsigned* counter, unsigned* fail)
{
    unsigned epoch = 0;
    for (int p = 0; p < phases; ++p) {
        phase_work(buf, n, p);
        __syncthreads();
        if (threadIdx.x == 0) {
            ++epoch;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * gridDim.x;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 4000000u) { *fail = 1u; break; }           // bounded: a block that is not resident must not hang the chip
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (*fail) return;
    }
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 330, phases = argc > 2 ? atoi(argv[2]) : 16, n = 1 << 20, reps = 50;
    double* buf; unsigned* ctr;
    CK(hipMalloc(&buf, n * sizeof(double))); CK(hipMemset(buf, 0, n * sizeof(double)));
    CK(hipMalloc(&ctr, 256)); 
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // (a) kernel boundaries
    for (int w = 0; w < 3; ++w) for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(256), 0, st, buf, n, p);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(256), 0, st, buf, n, p);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_launch = ms * 1e3 / reps / phases;
    // (b) one resident launch with grid barriers
    double us_barrier = -1.0; unsigned failed = 0;
    int maxb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_persistent, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    if (blocks <= maxb * prop.multiProcessorCount) {
        for (int r = 0; r < reps + 3; ++r) {
            CK(hipMemsetAsync(ctr, 0, 256, st));
            if (r == 3) CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_persistent, dim3(blocks), dim3(256), 0, st, buf, n, phases, ctr, ctr + 16);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&failed, ctr + 16, 4, hipMemcpyDeviceToHost));
        us_barrier = ms * 1e3 / reps / phases;            // includes the memset + launch of the one kernel, spread over its phases
    }
    printf("{\"blocks\": %d, \"phases\": %d, \"us_per_phase_as_kernels\": %.3f, \"us_per_phase_with_grid_barriers\": %.3f, \"barrier_spin_gave_up\": %u}\n",
           blocks, phases, us_launch, us_barrier, failed);
    return 0;
}
