// tools/probe/bw_probe.hip -- what a pure read stream reaches on this MI355X, by access pattern.  Not part of the library:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bw_probe tools/probe/bw_probe.hip && /tmp/bw_probe
// Every kernel sums a 3.3 GB float64 buffer (16-byte loads) and writes one value per thread group, so nothing is optimised away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// grid-stride: consecutive workgroups read consecutive 16 B x blockDim chunks; U loads in flight per lane
template <int U, bool NT>
__global__ void k_stride(const double2* __restrict__ p, size_t n16, double* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) { v[u].x = __builtin_nontemporal_load(&p[i + u * stride].x); v[u].y = __builtin_nontemporal_load(&p[i + u * stride].y); }
            else v[u] = p[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n16; i += stride) s += p[i].x + p[i].y;
    if (s == 12345.678) out[blockIdx.x] = s;
}

// block-contiguous: workgroup b owns one contiguous slice of the buffer (what a slab / strip decomposition does)
template <int U>
__global__ void k_chunk(const double2* __restrict__ p, size_t n16, double* __restrict__ out)
{
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
    double s = 0.0;
    size_t i = b0 + threadIdx.x;
    for (; i + (size_t)(U - 1) * blockDim.x < b1; i += (size_t)U * blockDim.x) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
    for (; i < b1; i += blockDim.x) s += p[i].x + p[i].y;
    if (s == 12345.678) out[blockIdx.x] = s;
}

// copy: 16-byte loads and stores, U in flight (read + write traffic counted)
template <int U, bool NT>
__global__ void k_copy(const double2* __restrict__ p, double2* __restrict__ o, size_t n16)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        d2v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p + i + u * stride)) : *reinterpret_cast<const d2v*>(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], reinterpret_cast<d2v*>(o + i + u * stride));
            else *reinterpret_cast<d2v*>(o + i + u * stride) = v[u];
        }
    }
    for (; i < n16; i += stride) o[i] = p[i];
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t bytes = (size_t)64 * 1801 * 3600 * 8;               // one cfg2 batch: 3.32 GB
    const size_t n16 = bytes / 16;
    double2* p; double* out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(p, 0, bytes));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"bytes\": %zu, \"rows\": [\n", pr.gcnArchName, cus, bytes);
    bool first = true;
    auto row = [&](const char* name, int blocks, int threads, int unroll, double ms) {
        printf("%s {\"kernel\": \"%s\", \"blocks\": %d, \"threads\": %d, \"loads_in_flight_per_lane\": %d, \"ms\": %.4f, \"TBps\": %.3f}",
               first ? "" : ",\n", name, blocks, threads, unroll, ms, bytes / ms / 1e9);
        first = false;
    };
    for (int threads : {256, 512, 1024})
        for (int mult : {1, 2, 4, 8, 16, 32}) {
            const int blocks = cus * mult;
            if ((size_t)blocks * threads > 4u * 1024 * 1024) continue;
            row("stride_u4", blocks, threads, 4, time_ms([&] { hipLaunchKernelGGL((k_stride<4, false>), dim3(blocks), dim3(threads), 0, 0, p, n16, out); }, 5));
            row("stride_u8", blocks, threads, 8, time_ms([&] { hipLaunchKernelGGL((k_stride<8, false>), dim3(blocks), dim3(threads), 0, 0, p, n16, out); }, 5));
            row("stride_u8_nt", blocks, threads, 8, time_ms([&] { hipLaunchKernelGGL((k_stride<8, true>), dim3(blocks), dim3(threads), 0, 0, p, n16, out); }, 5));
            row("chunk_u8", blocks, threads, 8, time_ms([&] { hipLaunchKernelGGL((k_chunk<8>), dim3(blocks), dim3(threads), 0, 0, p, n16, out); }, 5));
        }
    for (int mult : {64, 256, 1024})
        row("stride_u4", cus * mult, 256, 4, time_ms([&] { hipLaunchKernelGGL((k_stride<4, false>), dim3(cus * mult), dim3(256), 0, 0, p, n16, out); }, 5));
    {   // copy of half the buffer into the other half: bytes moved = read + written
        const size_t h16 = n16 / 2;
        for (int mult : {1, 4, 16, 64}) {
            const int blocks = cus * mult;
            const double ms0 = time_ms([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(blocks), dim3(256), 0, 0, p, p + h16, h16); }, 5);
            printf(",\n {\"kernel\": \"copy_u4\", \"blocks\": %d, \"threads\": 256, \"loads_in_flight_per_lane\": 4, \"ms\": %.4f, \"TBps\": %.3f}", blocks, ms0, 2.0 * h16 * 16 / ms0 / 1e9);
            const double ms1 = time_ms([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(blocks), dim3(256), 0, 0, p, p + h16, h16); }, 5);
            printf(",\n {\"kernel\": \"copy_u4_nt\", \"blocks\": %d, \"threads\": 256, \"loads_in_flight_per_lane\": 4, \"ms\": %.4f, \"TBps\": %.3f}", blocks, ms1, 2.0 * h16 * 16 / ms1 / 1e9);
        }
    }
    printf("\n]}\n");
    CK(hipFree(p)); CK(hipFree(out));
    return 0;
}
