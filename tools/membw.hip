// membw.hip -- measured HBM ceilings on this box (read-only, write-only, copy) with 16-byte lane accesses,
// for the "achieved / measured ceiling" column next to the 8 TB/s datasheet peak.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/membw tools/membw.hip && /tmp/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ a, size_t n, float *sink)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
__global__ __launch_bounds__(256) void k_read_nt(const f4 *__restrict__ a, size_t n, float *sink)
{
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += __builtin_nontemporal_load(&a[i]);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
// each block streams a contiguous chunk, 4 loads in flight per lane
__global__ __launch_bounds__(256) void k_read_chunk(const f4 *__restrict__ a, size_t n, float *sink)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    f4 acc = {0, 0, 0, 0};
    size_t i = b + threadIdx.x;
    for (; i + 768 < e; i += 1024) { f4 x0 = a[i], x1 = a[i + 256], x2 = a[i + 512], x3 = a[i + 768]; acc += x0 + x1 + x2 + x3; }
    for (; i < e; i += 256) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
__global__ __launch_bounds__(256) void k_write(f4 *__restrict__ a, size_t n)
{
    f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main()
{
    const size_t bytes = (size_t)2 << 30;   // 2 GiB per buffer, far above the 256 MiB Infinity Cache
    const size_t n = bytes / 16;
    f4 *a, *b; float *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch) {
        std::vector<float> ms;
        for (int r = 0; r < 7; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t); }
        std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
    };
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        float tr = time([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, sink); });
        float tn = time([&] { hipLaunchKernelGGL(k_read_nt, dim3(blocks), dim3(256), 0, 0, a, n, sink); });
        float tc = time([&] { hipLaunchKernelGGL(k_read_chunk, dim3(blocks), dim3(256), 0, 0, a, n, sink); });
        float tw = time([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n); });
        float tp = time([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        printf("blocks %5d: read %.0f GB/s  read_nt %.0f GB/s  read_chunk %.0f GB/s  write %.0f GB/s  copy %.0f GB/s (r+w)\n", blocks,
               bytes / tr / 1e6, bytes / tn / 1e6, bytes / tc / 1e6, bytes / tw / 1e6, 2.0 * bytes / tp / 1e6);
    }
    return 0;
}
