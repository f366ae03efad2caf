// Average pooling of NHWC activations over non-overlapping k x k windows: the step between the
// recurrent output on a large grid and the dense head on a coarser one.  HBM-bound: the input is
// read once with 16-byte loads, k * F * sizeof(T) contiguous bytes per window row.
#include <hip/hip_bf16.h>

#include "isg_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VPL = values per 16-byte load.  The lanes of a window are (kx, channel group): one load
// instruction of a wave reads whole window rows back to back, and the kx lanes of a channel group
// are then summed with shuffles.  Requires LPW = k * F / VPL lanes per window to divide 64.
template <typename T, int VPL>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, long long windows, int H, int W, int F,
                                                      int k, T* __restrict__ out) {
    const int lpw = k * F / VPL;  // lanes per window
    const int cgs = F / VPL;      // channel groups
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long win = gid / lpw;
    const int l = (int)(gid - win * lpw);
    if (win >= windows) return;  // lpw divides 64: a wave's windows are all in or all out together per lane group
    const int ow = W / k, oh = H / k;
    const int ox = (int)(win % ow);
    const int oy = (int)((win / ow) % oh);
    const long long n = win / ((long long)ow * oh);
    const int kx = l / cgs, cg = l - kx * cgs;
    const T* src = x + (((size_t)n * H + (size_t)oy * k) * W + (size_t)ox * k + kx) * F + cg * VPL;
    float acc[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) acc[i] = 0.0f;
    for (int ky = 0; ky < k; ++ky) {
        if constexpr (VPL == 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (size_t)ky * W * F);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += (float)v[i];
        } else {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)ky * W * F);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += v[i];
        }
    }
    // sum over the kx lanes of this channel group: they sit cgs lanes apart
    for (int o = cgs; o < lpw; o <<= 1) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) acc[i] += __shfl_xor(acc[i], o);
    }
    if (kx == 0) {
        const float inv = 1.0f / (float)(k * k);
        T* dst = out + (size_t)win * F + cg * VPL;
        if constexpr (VPL == 8) {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (__bf16)(acc[i] * inv);
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
            *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
        }
    }
}

}  // namespace

extern "C" int isg_avgpool_fwd(const void* x, long long N, int H, int W, int F, int k, int dtype, void* out,
                               void* stream) {
    ISG_REQUIRE(dtype == ISG_F32 || dtype == ISG_BF16, ISG_EDTYPE, "isg_avgpool_fwd: dtype %d", dtype);
    ISG_REQUIRE(N >= 0 && H >= 1 && W >= 1 && F >= 1 && k >= 1 && H % k == 0 && W % k == 0, ISG_EINVAL,
                "isg_avgpool_fwd: bad shape N=%lld H=%d W=%d F=%d k=%d", N, H, W, F, k);
    const int vpl = dtype == ISG_F32 ? 4 : 8;
    ISG_REQUIRE(F % vpl == 0, ISG_EINVAL, "isg_avgpool_fwd: channels must be a multiple of %d (got %d)", vpl, F);
    const int lpw = k * F / vpl;
    // the shuffle reduction needs k a power of two and a window's lanes inside one wave
    ISG_REQUIRE((k & (k - 1)) == 0 && ((F / vpl) & (F / vpl - 1)) == 0 && lpw <= 64 && 64 % lpw == 0, ISG_EINVAL,
                "isg_avgpool_fwd: k must be a power of two with k * F / %d <= 64 lanes (k=%d, F=%d)", vpl, k, F);
    if (N == 0) return ISG_OK;
    ISG_REQUIRE(x && out, ISG_EINVAL, "isg_avgpool_fwd: null pointer");
    const long long windows = N * (H / k) * (W / k);
    const long long threads = windows * lpw;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL((avgpool_kernel<float, 4>), grid, block, 0, s, static_cast<const float*>(x), windows, H, W, F, k,
                           static_cast<float*>(out));
    else
        hipLaunchKernelGGL((avgpool_kernel<__bf16, 8>), grid, block, 0, s, static_cast<const __bf16*>(x), windows, H, W,
                           F, k, static_cast<__bf16*>(out));
    return isg::check_launch("avgpool_kernel");
}
