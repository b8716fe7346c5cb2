// aai_util.hip -- small device utilities: the synthetic-image generator and the
// f64 <-> f32 converters behind aai_resample_f64 (the reference's IMG holds doubles, Source.cpp:31).
#include "aai_kernels.hpp"

namespace aai {

namespace {

__global__ __launch_bounds__(256) void aai_synth_kernel(float *__restrict__ dst, int W, int H, int64_t stride, uint64_t seed, int yBase)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = yBase + blockIdx.y;
    if (x >= W || y >= H) return;
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    uint64_t z = seed * G + ((uint64_t)y * (uint64_t)W + (uint64_t)x) + G;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    dst[(int64_t)y * stride + x] = (float)(z >> 40) * 0x1p-24f;
}

__global__ __launch_bounds__(256) void aai_f64_to_f32_kernel(const double *__restrict__ s, float *__restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = (float)s[i];
}

__global__ __launch_bounds__(256) void aai_f32_to_f64_kernel(const float *__restrict__ s, double *__restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = (double)s[i];
}

unsigned stride_grid(size_t n)
{
    size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

// rows [row0, row1) of the W x H pattern; dst addresses row row0
hipError_t launch_synth_rows(float *dst, int W, int H, int row0, int row1, int64_t stride, uint64_t seed, hipStream_t stream)
{
    if (W <= 0 || row1 <= row0) return hipSuccess;
    for (int y0 = row0; y0 < row1; y0 += 32768) {   // grid.y is limited to 65535
        const int rows = (row1 - y0 < 32768) ? row1 - y0 : 32768;
        hipLaunchKernelGGL(aai_synth_kernel, dim3((W + 255) / 256, rows), dim3(256), 0, stream, dst - (int64_t)row0 * stride, W, row1, stride, seed, y0);
    }
    (void)H;
    return hipGetLastError();
}

hipError_t launch_synth(float *dst, int W, int H, int64_t stride, uint64_t seed, hipStream_t stream)
{
    return launch_synth_rows(dst, W, H, 0, H, stride, seed, stream);
}

hipError_t launch_f64_to_f32(const double *src, float *dst, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(aai_f64_to_f32_kernel, dim3(stride_grid(n)), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_f32_to_f64(const float *src, double *dst, size_t n, hipStream_t stream)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(aai_f32_to_f64_kernel, dim3(stride_grid(n)), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

}  // namespace aai
