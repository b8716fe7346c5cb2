// tools/valu_rate.hip -- issue rate of the vector instructions the rotated kernels are built from (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip && ./valu_rate
// Every wave runs ITER x 32 independent instances of one instruction; 8 waves per SIMD keep the issue port busy.
// Output: wave-instructions per SIMD per microsecond and, at the clock the run held (s_memtime / wall), cycles each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, unsigned long long *clk)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3;
    double d0 = a, d1 = a + 1, d2 = b, d3 = c;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a, a + 1}, p1 = {a + 2, a + 3}, pb = {b, b}, pc = {c, c};
    unsigned long long m0 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r0) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r1) : "v"(b), "v"(c));) }
        if (OP == 1) { REP32(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pb), "v"(pc));) }
        if (OP == 2) { REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d2), "v"(d3)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(d2), "v"(d3));) }
        if (OP == 3) { REP32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(r1) : "v"(b));) }
        if (OP == 4) { REP32(asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m0) : "v"(r0), "v"(b)); asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r1) : "v"(b), "v"(c), "s"(m0));) }
        if (OP == 5) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(d2)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d1) : "v"(d2));) }
        if (OP == 6) { REP32(asm volatile("v_min_f64 %0, %0, %1" : "+v"(d0) : "v"(d2)); asm volatile("v_max_f64 %0, %0, %1" : "+v"(d1) : "v"(d2));) }
        if (OP == 7) { REP32(asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r0) : "v"(b), "v"(c)); asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(r1) : "v"(b), "v"(c));) }
        if (OP == 8) { REP32(asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m0) : "v"(d0), "v"(d2)); asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r1) : "v"(b), "v"(c), "s"(m0));) }
        if (OP == 9) { REP32(asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(d0), "=s"(m0) : "v"(r0), "v"(r1)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d1));) }
        if (OP == 10) { REP32(asm volatile("v_rcp_f32 %0, %0" : "+v"(r0)); asm volatile("v_rcp_f32 %0, %0" : "+v"(r1));) }
        if (OP == 11) { REP32(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r0) : "v"(d0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(r1));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + (float)(d0 + d1) + p0.x + p0.y + p1.x + p1.y + (float)m0;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int OP>
void run(const char *name, float *out, unsigned long long *clk)
{
    const int iters = 2000, blocks = 256 * 8;      // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    rate_kernel<OP><<<blocks, 256>>>(out, 10, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    rate_kernel<OP><<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long c = 0;
    hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
    const double instr = (double)iters * 64.0 * 8.0;            // per SIMD: 8 waves x 64 instructions per iteration
    const double ghz = (double)c / (ms * 1e6);                  // s_memtime ticks per ns of the first wave (shader clock)
    printf("%-34s %8.3f ms  %7.1f wave-instr/us/SIMD  clock %.2f GHz  -> %.2f cycles per instruction\n", name, ms, instr / (ms * 1e3), ghz, ms * 1e6 * ghz / instr);
}

int main()
{
    float *out; unsigned long long *clk;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float)); hipMalloc(&clk, 8);
    run<0>("v_fma_f32", out, clk);
    run<3>("v_add_f32", out, clk);
    run<1>("v_pk_fma_f32 (2 lanes of work)", out, clk);
    run<2>("v_fma_f64", out, clk);
    run<5>("v_add_f64", out, clk);
    run<6>("v_min_f64 / v_max_f64", out, clk);
    run<4>("v_cmp_lt_f32 -> sgpr, v_cndmask", out, clk);
    run<8>("v_cmp_lt_f64 -> sgpr, v_cndmask", out, clk);
    run<7>("v_or3_b32", out, clk);
    run<9>("v_mad_u64_u32 / v_lshlrev_b64", out, clk);
    run<10>("v_rcp_f32", out, clk);
    run<11>("v_cvt_f32_f64 / v_cvt_f64_f32", out, clk);
    return 0;
}
