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

}  // namespace

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
