// misalign.hip -- cost of byte-misaligned vector loads on gfx950 (8/16-bit sources start strips at arbitrary
// element offsets).  Reads 1 GiB with 4-, 8- and 16-byte lane loads whose addresses are offset by `off` bytes.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/misalign tools/misalign.hip && /tmp/misalign
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef unsigned u1 __attribute__((aligned(1)));
typedef unsigned u2 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));

template <typename V>
__global__ __launch_bounds__(256) void k_read(const char *__restrict__ a, size_t n, int off, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const V v = __builtin_nontemporal_load(reinterpret_cast<const V *>(a + off + i * sizeof(V)));
        if constexpr (sizeof(V) == 4) acc += v;
        else if constexpr (sizeof(V) == 8) acc += v.x + v.y;
        else acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    char *a; unsigned *sink;
    hipMalloc(&a, bytes + 4096); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes + 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch) {
        std::vector<float> ms;
        for (int r = 0; r < 7; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); ms.push_back(t); }
        std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
    };
    for (int off : {0, 1, 2, 4, 8, 16}) {
        float t4 = time([&] { hipLaunchKernelGGL(k_read<u1>, dim3(8192), dim3(256), 0, 0, a, bytes / 4, off, sink); });
        float t8 = time([&] { hipLaunchKernelGGL(k_read<u2>, dim3(8192), dim3(256), 0, 0, a, bytes / 8, off, sink); });
        float t16 = time([&] { hipLaunchKernelGGL(k_read<u4>, dim3(8192), dim3(256), 0, 0, a, bytes / 16, off, sink); });
        printf("offset %2d B: 4-byte loads %7.1f GB/s   8-byte %7.1f GB/s   16-byte %7.1f GB/s\n", off, bytes / t4 * 1e-6, bytes / t8 * 1e-6, bytes / t16 * 1e-6);
    }
    return 0;
}
