// Diagnostic micro-benchmark: per-CU rate of 16-byte-per-lane stores by address pattern, one workgroup per CU.
//   linear : lane l writes 16 B at l*16 (1 KB contiguous per instruction)
//   pixel  : lane l = (pixel l & 31, half l >> 5) writes piece k of its pixel's 128-byte row, four instructions k = 0..3
//            (the forward cell's factor stores)
//   quad   : same bytes, the four lanes of a quad write the four pieces of one pixel-half (64 B contiguous per quad)
//   halves : lane l = (pixel l & 31, half l >> 5) writes 16 B of its pixel's 32-byte row (the h / c stores)
//   own64  : lane l writes its own 64-byte row, one piece per instruction (a 32-channel dx row per pixel)
//   own32  : lane l writes its own 32-byte row, one piece per instruction (a 16-channel dh row per pixel)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/store_patterns tools/micro/store_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(uint4* out, int iters, unsigned long long* cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // each wave owns runs of 32 pixels x 128 B = 4 KB; successive iterations move on by the grid's footprint
    char* base = reinterpret_cast<char*>(out) + ((size_t)blockIdx.x * nw + wave) * 4096;
    const size_t step = (size_t)gridDim.x * nw * 4096;
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 7u);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        char* p = base + (size_t)it * step;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            size_t off;
            if (PATTERN == 0) off = (size_t)k * 1024 + lane * 16;
            else if (PATTERN == 1) off = (size_t)(lane & 31) * 128 + (lane >> 5) * 64 + k * 16;
            else if (PATTERN == 2) off = (size_t)((lane & 28) + k) * 128 + (lane >> 5) * 64 + (lane & 3) * 16;
            else if (PATTERN == 3) off = (size_t)k * 1024 + (lane & 31) * 32 + (lane >> 5) * 16;
            else if (PATTERN == 4) off = (size_t)lane * 64 + k * 16;
            else off = (size_t)(k >> 1) * 2048 + lane * 32 + (k & 1) * 16;
            *reinterpret_cast<uint4*>(p + off) = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int PATTERN>
void run(const char* name, int threads, uint4* buf, size_t bytes, unsigned long long* cyc) {
    const int blocks = 256, nw = threads / 64;
    const int iters = (int)(bytes / ((size_t)blocks * nw * 4096));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    store_kernel<PATTERN><<<blocks, threads>>>(buf, iters, cyc);
    hipEventRecord(e0);
    store_kernel<PATTERN><<<blocks, threads>>>(buf, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double moved = (double)iters * blocks * nw * 4096;
    printf("%-7s %d waves/CU: %.2f TB/s, %.1f B per s_memtime tick and CU, %.0f ticks per wave instruction\n", name, nw,
           moved / ms * 1e-9, (double)iters * nw * 4096 / h[9], (double)h[9] / (iters * 4.0));
}

int main() {
    const size_t bytes = 1ull << 30;
    uint4* buf;
    unsigned long long* cyc;
    hipMalloc(&buf, bytes);
    hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    run<0>("linear", 256, buf, bytes, cyc), run<1>("pixel", 256, buf, bytes, cyc), run<2>("quad", 256, buf, bytes, cyc);
    run<3>("halves", 256, buf, bytes, cyc), run<4>("own64", 256, buf, bytes, cyc), run<5>("own32", 256, buf, bytes, cyc);
    run<0>("linear", 512, buf, bytes, cyc), run<1>("pixel", 512, buf, bytes, cyc), run<2>("quad", 512, buf, bytes, cyc);
    run<3>("halves", 512, buf, bytes, cyc), run<4>("own64", 512, buf, bytes, cyc), run<5>("own32", 512, buf, bytes, cyc);
    return 0;
}
