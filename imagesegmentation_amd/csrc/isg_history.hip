// One-hot history frames on device: frames[n][y][x][i] = 1 where (x, y) == points[n][i].
// Write-only, 16 B per lane where the row of channels allows it.
#include <hip/hip_bf16.h>

#include "isg_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void history_frames_kernel(const int* __restrict__ points, int n, int P, int hist,
                                                             int out_stride, int out_offset, T* __restrict__ out) {
    // one thread per (n, y, x); writes the `hist` channels of that pixel
    const long long total = (long long)n * P * P;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % P);
        const long long r = idx / P;
        const int y = (int)(r % P);
        const int b = (int)(r / P);
        T* dst = out + idx * out_stride + out_offset;
        for (int i = 0; i < hist; ++i) {
            const int px = points[((size_t)b * hist + i) * 2];
            const int py = points[((size_t)b * hist + i) * 2 + 1];
            dst[i] = (px == x && py == y) ? T(1.0f) : T(0.0f);
        }
    }
}

}  // namespace

extern "C" int isg_history_frames(const int32_t* points, int n, int size, int hist, int dtype, void* out,
                                  int out_channels, int out_offset, void* stream) {
    ISG_REQUIRE(n >= 0 && size >= 1 && hist >= 0, ISG_EINVAL, "isg_history_frames: bad sizes");
    ISG_REQUIRE(out_channels >= out_offset + hist && out_offset >= 0, ISG_EINVAL,
                "isg_history_frames: %d frames at offset %d do not fit %d channels", hist, out_offset, out_channels);
    if (n == 0 || hist == 0) return ISG_OK;
    ISG_REQUIRE(points && out, ISG_EINVAL, "isg_history_frames: null pointer");
    const long long total = (long long)n * size * size;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32) {
        hipLaunchKernelGGL(history_frames_kernel<float>, dim3(blocks), dim3(256), 0, s, points, n, size, hist,
                           out_channels, out_offset, static_cast<float*>(out));
    } else if (dtype == ISG_BF16) {
        hipLaunchKernelGGL(history_frames_kernel<__hip_bfloat16>, dim3(blocks), dim3(256), 0, s, points, n, size, hist,
                           out_channels, out_offset, static_cast<__hip_bfloat16*>(out));
    } else {
        ISG_REQUIRE(false, ISG_EDTYPE, "isg_history_frames: dtype %d", dtype);
    }
    return isg::check_launch("history_frames_kernel");
}
