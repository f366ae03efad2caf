// Diagnostic micro-benchmark: cycles per vector instruction of ONE wave per SIMD as a function of the number of
// independent dependency chains it interleaves (fma chains, and chains shaped like a sigmoid: mul, exp2, add, rcp).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/valu_latency tools/micro/valu_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, bool TRANS>
__global__ void chain_kernel(float* out, unsigned long long* cycles, int iters) {
    float v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = 0.001f * (threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if constexpr (TRANS) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44f * v[i]));
                else v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int ILP, bool TRANS>
void run(int waves_per_simd) {
    float* out;
    unsigned long long* cyc;
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 2000;
    hipMalloc(&out, blocks * threads * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    for (int k = 0; k < 2; ++k) chain_kernel<ILP, TRANS><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double n_inst = (double)iters * 8 * ILP * (TRANS ? 4 : 1);
    printf("%s ILP=%d waves/SIMD=%d: %.2f cycles per instruction of a wave\n", TRANS ? "sigmoid" : "fma", ILP, waves_per_simd,
           h[17] / n_inst);
    hipFree(out), hipFree(cyc);
}

int main() {
    run<1, false>(1), run<2, false>(1), run<4, false>(1), run<8, false>(1), run<1, false>(2), run<4, false>(2);
    run<1, true>(1), run<2, true>(1), run<4, true>(1), run<8, true>(1), run<1, true>(2), run<4, true>(2);
    return 0;
}
