// The optimiser step of the reference (tf.train.AdamOptimizer after tf.clip_by_global_norm, reference
// polyrnn/Model.py:106-112) over one flat float parameter buffer, in ONE pass: HBM-bound, 4 reads + 3
// writes of 4 bytes per parameter.
#include "isg_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// p -= lr_t m' / (sqrt(v') + eps),  m' = b1 m + (1 - b1) g,  v' = b2 v + (1 - b2) g^2,  g = scale * grad,
// scale = clip / max(norm, clip) when a clip is given (the norm is read from device memory: no host round trip)
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                      const float* __restrict__ g, long long n, float lr_t, float b1,
                                                      float omb1, float b2, float omb2, float eps,
                                                      const float* __restrict__ norm, float clip) {
    float scale = 1.0f;
    if (norm != nullptr && clip > 0.0f) scale = clip / fmaxf(*norm, clip);
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * scale;
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pv = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mv[k] = b1 * mv[k] + omb1 * gv[k];
            vv[k] = b2 * vv[k] + omb2 * gv[k] * gv[k];
            pv[k] -= lr_t * mv[k] / (sqrtf(vv[k]) + eps);
        }
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        reinterpret_cast<f32x4*>(p)[i] = pv;
    }
    // the last n % 4 values
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gv = g[i] * scale;
        const float mv = b1 * m[i] + omb1 * gv;
        const float vv = b2 * v[i] + omb2 * gv * gv;
        m[i] = mv, v[i] = vv;
        p[i] -= lr_t * mv / (sqrtf(vv) + eps);
    }
}

// Everything the kernels read that is a re-arrangement (and narrowing) of parameter values - MFMA operand images of
// the convolution weights, tap tables, compute-type copies of the dense head - from the flat parameter buffer in ONE
// launch: element i of a section comes from src[map[i]] (map[i] < 0: zero), the ranges are plain conversions.
struct RefreshParams {
    const float* src;
    const int* map16; long long n16; __bf16* out16;   // gathered, stored as bf16
    const int* map32; long long n32; float* out32;    // gathered, stored as float
    long long range_src[4], range_n[4]; __bf16* range_out[4];  // contiguous runs converted to bf16
    long long range_first[5];  // work items (8 values each) before each range, the total last
};

__global__ __launch_bounds__(256) void refresh_derived_kernel(RefreshParams p) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = tid; i < p.n16; i += stride) {
        const int k = p.map16[i];
        p.out16[i] = (__bf16)(k >= 0 ? p.src[k] : 0.0f);
    }
    for (long long i = tid; i < p.n32; i += stride) {
        const int k = p.map32[i];
        p.out32[i] = k >= 0 ? p.src[k] : 0.0f;
    }
    for (long long w = tid; w < p.range_first[4]; w += stride) {
        int r = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) r += w >= p.range_first[k];
        const long long e = (w - p.range_first[r]) * 8;
        const float* s = p.src + p.range_src[r] + e;
        __bf16* d = p.range_out[r] + e;
        if (e + 8 <= p.range_n[r] && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
            const f32x4 a = reinterpret_cast<const f32x4*>(s)[0], b = reinterpret_cast<const f32x4*>(s)[1];
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            *reinterpret_cast<bf16x8*>(d) = bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3],
                                                   (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
        } else {
            for (long long j = e; j < p.range_n[r] && j < e + 8; ++j) p.range_out[r][j] = (__bf16)p.src[p.range_src[r] + j];
        }
    }
}

}  // namespace

extern "C" int isg_refresh_derived(const float* src, long long n_src, const int32_t* map_bf16, long long n_bf16,
                                   void* out_bf16, const int32_t* map_f32, long long n_f32, float* out_f32, int n_ranges,
                                   const long long* range_src, const long long* range_n, void* const* range_out,
                                   void* stream) {
    ISG_REQUIRE(src && n_src > 0 && n_bf16 >= 0 && n_f32 >= 0 && n_ranges >= 0 && n_ranges <= 4, ISG_EINVAL,
                "isg_refresh_derived: bad arguments (at most 4 ranges)");
    ISG_REQUIRE((n_bf16 == 0 || (map_bf16 && out_bf16)) && (n_f32 == 0 || (map_f32 && out_f32)), ISG_EINVAL,
                "isg_refresh_derived: a section without its map or its output");
    RefreshParams p{};
    p.src = src;
    p.map16 = map_bf16, p.n16 = n_bf16, p.out16 = static_cast<__bf16*>(out_bf16);
    p.map32 = map_f32, p.n32 = n_f32, p.out32 = out_f32;
    long long items = 0;
    for (int r = 0; r < 4; ++r) {
        p.range_first[r] = items;
        if (r < n_ranges) {
            ISG_REQUIRE(range_src && range_n && range_out && range_out[r] && range_src[r] >= 0 && range_n[r] >= 0 &&
                            range_src[r] + range_n[r] <= n_src,
                        ISG_EINVAL, "isg_refresh_derived: range %d lies outside the source", r);
            p.range_src[r] = range_src[r], p.range_n[r] = range_n[r], p.range_out[r] = static_cast<__bf16*>(range_out[r]);
            items += (range_n[r] + 7) / 8;
        }
    }
    p.range_first[4] = items;
    long long work = n_bf16 > n_f32 ? n_bf16 : n_f32;
    work = work > items ? work : items;
    if (work == 0) return ISG_OK;
    const int blocks = (int)((work + 255) / 256 < 16384 ? (work + 255) / 256 : 16384);
    hipLaunchKernelGGL(refresh_derived_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return isg::check_launch("refresh_derived_kernel");
}

extern "C" int isg_adam_tf(float* param, float* m, float* v, const float* grad, long long n, double lr_t, double beta1,
                           double beta2, double eps, const float* grad_norm, double clip_norm, void* stream) {
    ISG_REQUIRE(n >= 0, ISG_EINVAL, "isg_adam_tf: n=%lld", n);
    if (n == 0) return ISG_OK;
    ISG_REQUIRE(param && m && v && grad, ISG_EINVAL, "isg_adam_tf: null pointer");
    ISG_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                  reinterpret_cast<uintptr_t>(grad)) & 15) == 0,
                ISG_EINVAL, "isg_adam_tf: buffers must be 16-byte aligned");
    const long long work = (n + 3) / 4;
    const int blocks = (int)((work + 255) / 256 < 65536 ? (work + 255) / 256 : 65536);
    hipLaunchKernelGGL(adam_tf_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), param, m, v, grad, n,
                       (float)lr_t, (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       grad_norm, (float)clip_norm);  // 1 - beta in double: 1 - 0.999f is off by 1e-5 of itself
    return isg::check_launch("adam_tf_kernel");
}
