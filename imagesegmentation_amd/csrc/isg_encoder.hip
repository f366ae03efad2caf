// Pointwise halves of the encoder's conv + bias + ReLU layers (reference
// polyrnn/History2LargeExperiment.py:15-22: tf.layers.conv2d(..., activation=tf.nn.relu)).  The
// convolution itself runs in the vendor library; bias add and ReLU are one in-place pass here instead
// of two library passes, and their backward (ReLU mask + bias gradient) is one pass instead of three.
// HBM-bound: every element is read once and written once with 16-byte accesses.
#include <hip/hip_bf16.h>

#include <cstdint>
#include <mutex>

#include "isg_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kMaxPartialBlocks = 1024;

template <typename T>
struct Vec;
template <>
struct Vec<__bf16> {
    static constexpr int N = 8;
    using type = bf16x8;
};
template <>
struct Vec<float> {
    static constexpr int N = 4;
    using type = f32x4;
};

// y[i][c] = max(y[i][c] + bias[c], 0), NHWC with C a multiple of the vector width
template <typename T>
__global__ __launch_bounds__(kBlock) void bias_relu_kernel(T* __restrict__ y, const float* __restrict__ bias,
                                                         long long nvec, int C) {
    constexpr int N = Vec<T>::N;
    using V = typename Vec<T>::type;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += (long long)gridDim.x * kBlock) {
        const int c0 = (int)((i * N) % C);
        V v = reinterpret_cast<V*>(y)[i];
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = (T)fmaxf((float)v[k] + bias[c0 + k], 0.0f);
        reinterpret_cast<V*>(y)[i] = v;
    }
}

// dz = dy where y > 0 else 0; partial[block][c] = sum over the block's elements of dz[.][c].
// A thread keeps the same channel group in every iteration (the grid stride is a multiple of C / N
// vectors), so its N running sums belong to fixed channels.
template <typename T>
__global__ __launch_bounds__(kBlock) void relu_bwd_bias_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                             T* __restrict__ dz, long long nvec, int C,
                                                             float* __restrict__ partial, int y_vec_stride,
                                                             int dy_vec_stride) {
    constexpr int N = Vec<T>::N;
    using V = typename Vec<T>::type;
    extern __shared__ float red[];  // [kBlock][N]
    float sum[N];
#pragma unroll
    for (int k = 0; k < N; ++k) sum[k] = 0.0f;
    const long long start = (long long)blockIdx.x * kBlock + threadIdx.x;
    for (long long i = start; i < nvec; i += (long long)gridDim.x * kBlock) {
        const V g = reinterpret_cast<const V*>(dy)[dy_vec_stride == C / N ? i : (i / (C / N)) * dy_vec_stride + i % (C / N)];
        // (y may be a channel slice of wider pixels: y_vec_stride vectors from one pixel to the next)
        const V a = reinterpret_cast<const V*>(y)[y_vec_stride == C / N ? i : (i / (C / N)) * y_vec_stride + i % (C / N)];
        V o;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float v = (float)a[k] > 0.0f ? (float)g[k] : 0.0f;
            o[k] = (T)v;
            sum[k] += v;
        }
        reinterpret_cast<V*>(dz)[i] = o;
    }
#pragma unroll
    for (int k = 0; k < N; ++k) red[threadIdx.x * N + k] = sum[k];
    __syncthreads();
    // threads with the same (threadIdx.x mod groups) hold the same channels
    const int groups = C / N;
    if ((int)threadIdx.x < groups) {
        float tot[N];
#pragma unroll
        for (int k = 0; k < N; ++k) tot[k] = 0.0f;
        for (int t = threadIdx.x; t < kBlock; t += groups)
#pragma unroll
            for (int k = 0; k < N; ++k) tot[k] += red[t * N + k];
        const int g0 = (int)((start % groups) * N);  // channel of this thread's first value
#pragma unroll
        for (int k = 0; k < N; ++k) partial[(size_t)blockIdx.x * C + g0 + k] = tot[k];
    }
}

// one workgroup per channel: 256 threads share the per-block partial sums, then a tree in LDS
__global__ __launch_bounds__(kBlock) void bias_partial_reduce_kernel(const float* __restrict__ partial, int nblocks,
                                                                   int C, float* __restrict__ dbias) {
    __shared__ float red[kBlock];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) s += partial[(size_t)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dbias[c] = red[0];
}

int check(const char* what, long long n, int C, int dtype) {
    ISG_REQUIRE(dtype == ISG_F32 || dtype == ISG_BF16, ISG_EDTYPE, "%s: dtype %d", what, dtype);
    const int N = dtype == ISG_F32 ? 4 : 8;
    ISG_REQUIRE(n >= 0 && C >= N && C % N == 0 && kBlock % (C / N) == 0, ISG_EINVAL,
                "%s: channels must be a multiple of %d that divides %d vectors (n=%lld, C=%d)", what, N, kBlock, n, C);
    return ISG_OK;
}

int grid_for(long long nvec) {
    const long long want = (nvec + kBlock - 1) / kBlock;
    return (int)(want < kMaxPartialBlocks ? (want < 1 ? 1 : want) : kMaxPartialBlocks);
}

// Per-block partial sums of isg_relu_bwd_bias_static: a fixed array in the code object instead of a
// caller's workspace, for callers that cannot allocate.  Every GPU the library runs on holds its own
// copy (256 KiB) at its own address until the process ends.  In host code this name is only a handle
// for hipGetSymbolAddress; kernels get the address that call returned.
constexpr int kStaticMaxC = ISG_RELU_BWD_BIAS_STATIC_MAX_C;
}  // namespace
static __device__ __align__(256) float scratch[kMaxPartialBlocks * kStaticMaxC];
namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_lock;
float* g_scratch_ptr[kMaxDevices];  // device address of `scratch` on each device, 0 = not asked yet

// The current device's copy of `scratch`; resolved once per device, with that device current.
int static_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        isg::set_error("isg_relu_bwd_bias_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
        return ISG_ELAUNCH;
    }
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            isg::set_error("isg_relu_bwd_bias_static: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
            return ISG_ELAUNCH;
        }
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_ptr[dev];
    return ISG_OK;
}

}  // namespace

extern "C" int isg_bias_relu(void* y, const float* bias, long long n, int C, int dtype, void* stream) {
    int rc = check("isg_bias_relu", n, C, dtype);
    if (rc != ISG_OK) return rc;
    if (n == 0) return ISG_OK;
    ISG_REQUIRE(y && bias, ISG_EINVAL, "isg_bias_relu: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32) {
        const long long nvec = n * C / 4;
        hipLaunchKernelGGL(bias_relu_kernel<float>, dim3(grid_for(nvec) * 8), dim3(kBlock), 0, s, static_cast<float*>(y),
                           bias, nvec, C);
    } else {
        const long long nvec = n * C / 8;
        hipLaunchKernelGGL(bias_relu_kernel<__bf16>, dim3(grid_for(nvec) * 8), dim3(kBlock), 0, s,
                           static_cast<__bf16*>(y), bias, nvec, C);
    }
    return isg::check_launch("bias_relu_kernel");
}

extern "C" long long isg_relu_bwd_bias_workspace_bytes(int C) {
    return (long long)kMaxPartialBlocks * C * (long long)sizeof(float);
}

namespace {
int relu_bwd_bias_launch(const void* dy, const void* y, long long n, int C, int dtype, void* dz, float* dbias,
                         void* workspace, void* stream, long long y_pixel_stride = 0, long long dy_pixel_stride = 0);
}

extern "C" int isg_relu_bwd_bias(const void* dy, const void* y, long long n, int C, int dtype, void* dz, float* dbias,
                                 void* workspace, void* stream) {
    int rc = check("isg_relu_bwd_bias", n, C, dtype);
    if (rc != ISG_OK) return rc;
    ISG_REQUIRE(dbias && workspace, ISG_EINVAL, "isg_relu_bwd_bias: null pointer");
    return relu_bwd_bias_launch(dy, y, n, C, dtype, dz, dbias, workspace, stream);
}

extern "C" int isg_relu_bwd_bias_strided(const void* dy, long long dy_pixel_stride, const void* y,
                                         long long y_pixel_stride, long long n, int C, int dtype, void* dz,
                                         float* dbias, void* workspace, void* stream) {
    int rc = check("isg_relu_bwd_bias_strided", n, C, dtype);
    if (rc != ISG_OK) return rc;
    ISG_REQUIRE(dbias && workspace, ISG_EINVAL, "isg_relu_bwd_bias_strided: null pointer");
    const int N = dtype == ISG_F32 ? 4 : 8;
    ISG_REQUIRE(y_pixel_stride >= C && y_pixel_stride % N == 0 && y_pixel_stride / N < (1ll << 31) &&
                    dy_pixel_stride >= C && dy_pixel_stride % N == 0 && dy_pixel_stride / N < (1ll << 31),
                ISG_EINVAL, "isg_relu_bwd_bias_strided: pixel strides %lld, %lld (multiples of %d, at least C = %d)",
                dy_pixel_stride, y_pixel_stride, N, C);
    ISG_REQUIRE(dy_pixel_stride == C || dz != dy, ISG_EINVAL,
                "isg_relu_bwd_bias_strided: dz (contiguous) cannot alias a dy of wider pixels");
    return relu_bwd_bias_launch(dy, y, n, C, dtype, dz, dbias, workspace, stream, y_pixel_stride, dy_pixel_stride);
}

// The partial sums go to this GPU's `scratch` array: no allocation, no workspace argument.  Every row the
// reduction reads is written by the first kernel of the same call, so nothing has to be cleared between
// calls; two calls in flight at once on one GPU (different streams) would share those rows and race.
extern "C" int isg_relu_bwd_bias_static(const void* dy, const void* y, long long n, int C, int dtype, void* dz,
                                        float* dbias, void* stream) {
    int rc = check("isg_relu_bwd_bias_static", n, C, dtype);
    if (rc != ISG_OK) return rc;
    ISG_REQUIRE(C <= kStaticMaxC, ISG_EINVAL,
                "isg_relu_bwd_bias_static: %d channels exceed the built-in scratch (at most %d); use "
                "isg_relu_bwd_bias with a workspace", C, kStaticMaxC);
    ISG_REQUIRE(dbias, ISG_EINVAL, "isg_relu_bwd_bias_static: null pointer");
    if (stream) {
        // the scratch handed to the kernels is the CURRENT device's copy: a stream of another GPU would get a
        // pointer that is not its own
        int cur = -1, sdev = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e == hipSuccess) e = hipStreamGetDevice(static_cast<hipStream_t>(stream), &sdev);
        ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "isg_relu_bwd_bias_static: device query: %s", hipGetErrorString(e));
        ISG_REQUIRE(cur == sdev, ISG_EINVAL,
                    "isg_relu_bwd_bias_static: the stream belongs to device %d but device %d is current", sdev, cur);
    }
    float* partial = nullptr;
    if ((rc = static_scratch(&partial)) != ISG_OK) return rc;
    return relu_bwd_bias_launch(dy, y, n, C, dtype, dz, dbias, partial, stream);
}

extern "C" int isg_relu_bwd_bias_static_init(void) {
    float* partial = nullptr;
    return static_scratch(&partial);
}

namespace {
int relu_bwd_bias_launch(const void* dy, const void* y, long long n, int C, int dtype, void* dz, float* dbias,
                         void* workspace, void* stream, long long y_pixel_stride, long long dy_pixel_stride) {
    int rc;
    const int ystride = (int)((y_pixel_stride ? y_pixel_stride : C) / (dtype == ISG_F32 ? 4 : 8));
    const int dystride = (int)((dy_pixel_stride ? dy_pixel_stride : C) / (dtype == ISG_F32 ? 4 : 8));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return hipMemsetAsync(dbias, 0, sizeof(float) * C, s) == hipSuccess ? ISG_OK : ISG_ELAUNCH;
    ISG_REQUIRE(dy && y && dz, ISG_EINVAL, "isg_relu_bwd_bias: null pointer");
    float* partial = static_cast<float*>(workspace);
    int grid;
    if (dtype == ISG_F32) {
        const long long nvec = n * C / 4;
        grid = grid_for(nvec);
        hipLaunchKernelGGL(relu_bwd_bias_kernel<float>, dim3(grid), dim3(kBlock), kBlock * 4 * sizeof(float), s,
                           static_cast<const float*>(dy), static_cast<const float*>(y), static_cast<float*>(dz), nvec, C,
                           partial, ystride, dystride);
    } else {
        const long long nvec = n * C / 8;
        grid = grid_for(nvec);
        hipLaunchKernelGGL(relu_bwd_bias_kernel<__bf16>, dim3(grid), dim3(kBlock), kBlock * 8 * sizeof(float), s,
                           static_cast<const __bf16*>(dy), static_cast<const __bf16*>(y), static_cast<__bf16*>(dz), nvec,
                           C, partial, ystride, dystride);
    }
    if ((rc = isg::check_launch("relu_bwd_bias_kernel")) != ISG_OK) return rc;
    hipLaunchKernelGGL(bias_partial_reduce_kernel, dim3(C), dim3(kBlock), 0, s, partial, grid, C, dbias);
    return isg::check_launch("bias_partial_reduce_kernel");
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Weight images of the encoder convolutions from HWIO float weights: one launch per image (the weights change every
// training step).  One thread per element of the image.
// ---------------------------------------------------------------------------------------------
namespace {

// [KS*KS*C/16][64][8]: element [tap * (C/16) + q][32 kg + m][j] = A[tap][m][16 q + 8 kg + j]; forward
// A[tap][m][k] = w[ty][tx][k][m], data gradient A[tap][m][k] = w[KS-1-ty][KS-1-tx][m][k]; rows m >= C are zero.
// (cin < C, forward image only: weights [KS][KS][cin][C] of a layer whose input is padded to C channels with zeros)
__global__ __launch_bounds__(256) void conv_square_pack_kernel(const float* __restrict__ w, int KS, int C, int data_gradient,
                                                               __bf16* __restrict__ out, int total, int cin) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, s = idx >> 9;
    const int spt = C / 16;
    const int tap = s / spt, q = s - tap * spt;
    const int m = lane & 31, k = 16 * q + 8 * (lane >> 5) + j;
    const int ty = tap / KS, tx = tap - ty * KS;
    float v = 0.0f;
    if (m < C && k < cin)
        v = data_gradient ? w[(((KS - 1 - ty) * KS + (KS - 1 - tx)) * C + m) * C + k] : w[((ty * KS + tx) * cin + k) * C + m];
    out[idx] = (__bf16)v;
}

// [KS * KXP/4][64][8], KXP = KS rounded up to 4: element [ky * KXP/4 + s][32 kg + m][j] =
// w[ky][4 s + 2 kg + j / 4][j % 4][m], 0 where kx >= KS, the channel is 3 or m >= cout.
__global__ __launch_bounds__(256) void conv_in_pack_kernel(const float* __restrict__ w, int KS, int cout,
                                                           __bf16* __restrict__ out, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, r = idx >> 9;
    const int spk = ((KS + 3) / 4 * 4) / 4;
    const int ky = r / spk, s = r - ky * spk;
    const int m = lane & 31, kx = 4 * s + 2 * (lane >> 5) + (j >> 2), c = j & 3;
    float v = 0.0f;
    if (kx < KS && c < 3 && m < cout) v = w[((ky * KS + kx) * 3 + c) * cout + m];
    out[idx] = (__bf16)v;
}

}  // namespace

namespace {
struct SliceCopy {
    const char* src;
    char* dst;
    long long src_stride[3], dst_stride[3];  // bytes: outer block, inner block, pixel
    long long n1, npix;                      // inner blocks per outer block, pixels per inner block
    int chunks;                              // 16-byte chunks per pixel
    long long total;
};

// 16 bytes per thread: chunk c of pixel (i0, i1, q) moves from src + i0 s0 + i1 s1 + q s2 + 16 c to the same place
// by dst's strides
__global__ __launch_bounds__(256) void copy_channel_slice_kernel(SliceCopy p) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (long long)gridDim.x * 256) {
        const long long px = i / p.chunks;
        const int c = (int)(i - px * p.chunks);
        const long long blk = px / p.npix, q = px - blk * p.npix;
        const long long i0 = blk / p.n1, i1 = blk - i0 * p.n1;
        *reinterpret_cast<uint4*>(p.dst + i0 * p.dst_stride[0] + i1 * p.dst_stride[1] + q * p.dst_stride[2] + 16 * c) =
            *reinterpret_cast<const uint4*>(p.src + i0 * p.src_stride[0] + i1 * p.src_stride[1] + q * p.src_stride[2] + 16 * c);
    }
}

// [npix][cin] float or bf16 -> [npix][cout] bf16, channels cin.. zero: thread = one 16-byte run of 8 output channels
template <typename T>
__global__ __launch_bounds__(256) void pad_channels_kernel(const T* __restrict__ x, int cin, __bf16* __restrict__ y,
                                                           int cout, long long total) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const int chunks = cout / 8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long px = i / chunks;
        const int c0 = (int)(i - px * chunks) * 8;
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = c0 + k < cin ? (__bf16)(float)x[px * cin + c0 + k] : (__bf16)0.0f;
        *reinterpret_cast<bf16x8*>(y + px * cout + c0) = v;
    }
}
}  // namespace

extern "C" int isg_copy_channel_slice(const void* src, const long long* src_strides, void* dst,
                                      const long long* dst_strides, int slice_bytes, long long n0, long long n1,
                                      long long npix, void* stream) {
    ISG_REQUIRE(src && dst && src_strides && dst_strides, ISG_EINVAL, "isg_copy_channel_slice: null pointer");
    bool ok = slice_bytes >= 16 && slice_bytes % 16 == 0 && n0 >= 0 && n1 >= 0 && npix >= 0 &&
              (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int k = 0; k < 3 && ok; ++k) ok = src_strides[k] % 16 == 0 && dst_strides[k] % 16 == 0;
    ok = ok && src_strides[2] >= slice_bytes && dst_strides[2] >= slice_bytes;
    ISG_REQUIRE(ok, ISG_EINVAL, "isg_copy_channel_slice: the slice, every stride and both addresses are multiples of 16 "
                "bytes, the pixel strides at least the slice (got slice %d)", slice_bytes);
    SliceCopy p;
    p.src = static_cast<const char*>(src), p.dst = static_cast<char*>(dst);
    for (int k = 0; k < 3; ++k) p.src_stride[k] = src_strides[k], p.dst_stride[k] = dst_strides[k];
    p.n1 = n1, p.npix = npix, p.chunks = slice_bytes / 16;
    p.total = n0 * n1 * npix * p.chunks;
    if (p.total == 0) return ISG_OK;
    long long blocks = (p.total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(copy_channel_slice_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return isg::check_launch("copy_channel_slice_kernel");
}

extern "C" int isg_pad_channels(const void* x, int dtype, int cin, void* y, int cout, long long npix, void* stream) {
    ISG_REQUIRE(dtype == ISG_F32 || dtype == ISG_BF16, ISG_EDTYPE, "isg_pad_channels: dtype %d", dtype);
    ISG_REQUIRE(cin >= 1 && cout >= cin && cout % 8 == 0 && npix >= 0, ISG_EINVAL,
                "isg_pad_channels: 1 <= cin <= cout, cout a multiple of 8 (got %d -> %d)", cin, cout);
    if (npix == 0) return ISG_OK;
    ISG_REQUIRE(x && y, ISG_EINVAL, "isg_pad_channels: null pointer");
    const long long total = npix * (cout / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(pad_channels_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float*>(x),
                           cin, static_cast<__bf16*>(y), cout, total);
    else
        hipLaunchKernelGGL(pad_channels_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s,
                           static_cast<const __bf16*>(x), cin, static_cast<__bf16*>(y), cout, total);
    return isg::check_launch("pad_channels_kernel");
}

extern "C" int isg_conv32_pack(const float* w, int KS, int C, int data_gradient, void* wimg, void* stream) {
    ISG_REQUIRE(w && wimg, ISG_EINVAL, "isg_conv32_pack: null argument");
    ISG_REQUIRE((KS == 3 || KS == 5) && (C == 16 || C == 32), ISG_EINVAL, "isg_conv32_pack: 3 x 3 or 5 x 5 taps, 16 or 32 channels");
    const int total = KS * KS * (C / 16) * 512;
    hipLaunchKernelGGL(conv_square_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, KS,
                       C, data_gradient, static_cast<__bf16*>(wimg), total, C);
    return isg::check_launch("conv_square_pack_kernel");
}

extern "C" int isg_conv32_pack_padded(const float* w, int KS, int C, int cin, void* wimg, void* stream) {
    ISG_REQUIRE(w && wimg, ISG_EINVAL, "isg_conv32_pack_padded: null argument");
    ISG_REQUIRE((KS == 3 || KS == 5) && (C == 16 || C == 32) && cin >= 1 && cin <= C, ISG_EINVAL,
                "isg_conv32_pack_padded: 3 x 3 or 5 x 5 taps, 16 or 32 output channels, 1..C input channels");
    const int total = KS * KS * (C / 16) * 512;
    hipLaunchKernelGGL(conv_square_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, KS,
                       C, 0, static_cast<__bf16*>(wimg), total, cin);
    return isg::check_launch("conv_square_pack_kernel");
}

extern "C" int isg_conv_in_pack(const float* w, int KS, int cout, void* wimg, void* stream) {
    ISG_REQUIRE(w && wimg, ISG_EINVAL, "isg_conv_in_pack: null argument");
    ISG_REQUIRE((KS == 14 || KS == 5) && (cout == 16 || cout == 32), ISG_EINVAL, "isg_conv_in_pack: 14 x 14 or 5 x 5 taps, 16 or 32 output channels");
    const int total = KS * (((KS + 3) / 4 * 4) / 4) * 512;
    hipLaunchKernelGGL(conv_in_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, KS,
                       cout, static_cast<__bf16*>(wimg), total);
    return isg::check_launch("conv_in_pack_kernel");
}
