// Vertex decoding over the P*P (+1) class grid: softmax cross-entropy (forward / backward), argmax
// with first-maximum tie-break, and categorical sampling.  One 64-lane wave owns one row; the row
// (<= a few KB) is streamed once with a running (max, sum) pair and reduced across the wave with
// shuffles - no LDS, no second pass over HBM.
#include <hip/hip_bf16.h>

#include "isg_common.h"

namespace {

constexpr int kWavesPerBlock = 4;

template <typename T>
__device__ __forceinline__ float ld(const T* p, long long i) {
    return (float)p[i];
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// (value, index) maximum with the smaller index winning ties: tf.argmax's first-maximum rule
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void xent_fwd_kernel(const T* __restrict__ logits,
                                                                      const int* __restrict__ labels,
                                                                      const float* __restrict__ weights, int rows,
                                                                      int cols, float* __restrict__ loss_rows,
                                                                      float* __restrict__ lse_rows,
                                                                      int* __restrict__ argmax_rows,
                                                                      const int* __restrict__ targets_xy, int grid_side,
                                                                      int* __restrict__ labels_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* z = logits + (size_t)row * cols;
    float m = -INFINITY, s = 0.0f;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float v = ld(z, c);
        // strictly greater: within a lane columns come in increasing order.  The lane's first column always
        // becomes its candidate, so a row of -inf (or NaN) still reports a column of the row, the first one
        if (v > bv || bi == 0x7fffffff) {
            bv = v;
            bi = c;
        }
        if (v > m) {
            s = s * expf(m - v) + 1.0f;
            m = v;
        } else if (v != -INFINITY) {  // a masked logit (-inf) adds nothing; exp(-inf - -inf) would be NaN
            s += expf(v - m);
        }
    }
    const float gm = wave_max(m);
    s = m == -INFINITY ? 0.0f : s * expf(m - gm);
    s = wave_sum(s);
    wave_argmax(bv, bi);
    if (lane == 0) {
        const float lse = gm + logf(s);
        if (lse_rows) lse_rows[row] = lse;
        if (argmax_rows) argmax_rows[row] = bi;
        if (loss_rows) {
            // (targets given as grid positions: class = x * side + y, reference polyrnn/Model.py:39)
            const int y = targets_xy ? targets_xy[2 * row] * grid_side + targets_xy[2 * row + 1] : labels[row];
            if (labels_out) labels_out[row] = y;
            float l = (y >= 0 && y < cols) ? lse - ld(z, y) : NAN;  // TF yields NaN for a bad label
            if (weights) l *= weights[row];
            loss_rows[row] = l;
        }
    }
}

// dlogits[r][c] = (softmax[r][c] - [c == label]) * row_scale[r]
template <typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void xent_bwd_kernel(const T* __restrict__ logits,
                                                                      const int* __restrict__ labels,
                                                                      const float* __restrict__ lse_rows,
                                                                      const float* __restrict__ row_scale,
                                                                      float uniform_scale, int rows, int cols,
                                                                      T* __restrict__ dlogits,
                                                                      const float* __restrict__ upstream) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* z = logits + (size_t)row * cols;
    T* d = dlogits + (size_t)row * cols;
    const float lse = lse_rows[row];
    const float sc = (row_scale ? row_scale[row] : 1.0f) * uniform_scale * (upstream ? *upstream : 1.0f);
    const int y = labels[row];
    for (int c = lane; c < cols; c += 64) {
        const float p = expf(ld(z, c) - lse);
        d[c] = (T)((p - (c == y ? 1.0f : 0.0f)) * sc);
    }
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// Gumbel-max categorical sample of softmax(logits[row]): argmax_c (z_c - log(-log u_c)),
// u_c from a counter hash of (seed, global row, column).
template <typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void sample_kernel(const T* __restrict__ logits, int rows, int cols,
                                                                    unsigned long long seed,
                                                                    unsigned long long row_offset,
                                                                    const unsigned long long* __restrict__ offset_dev,
                                                                    int grid_side, int* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* z = logits + (size_t)row * cols;
    const unsigned long long grow = row_offset + (offset_dev ? *offset_dev : 0ull) + (unsigned long long)row;
    const uint32_t key = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9U)) ^
                         mix32((uint32_t)grow * 0x85ebca6bU + (uint32_t)(grow >> 32));
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const uint32_t r = mix32(key + (uint32_t)c * 0xc2b2ae35U);
        const float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
        const float v = ld(z, c) - logf(-logf(u));
        if (v > bv || bi == 0x7fffffff) {  // a row of -inf still yields one of its columns
            bv = v;
            bi = c;
        }
    }
    wave_argmax(bv, bi);
    if (lane == 0) {
        if (grid_side > 0) {  // as an action (x, y) = (class / P, class % P); the end token P*P is (P, 0)
            const int x = bi / grid_side;
            *reinterpret_cast<int2*>(out_idx + 2 * (size_t)row) = make_int2(x, bi - x * grid_side);
        } else {
            out_idx[row] = bi;
        }
    }
}

// One step of the autoregressive decode loop for every sample, after the head: the argmax class of the row (first
// maximum), its vertex (x, y) = (class / P, class % P), then the stop rule, the predicted polygon and the history
// points of the next step - everything the host loop of the reference does between two session runs.
template <typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void decode_step_kernel(
    const T* __restrict__ logits, int rows, int cols, int P, int end_token, const int* __restrict__ close_xy,
    int* __restrict__ required, int* __restrict__ durations, int* __restrict__ predicted, int Tmax, int step,
    int* __restrict__ history, int hist) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* z = logits + (size_t)row * cols;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float v = ld(z, c);
        if (v > bv || bi == 0x7fffffff) {
            bv = v;
            bi = c;
        }
    }
    wave_argmax(bv, bi);
    if (lane != 0) return;
    const int x = bi / P, y = bi - x * P;
    durations[row] += required[row];
    const int cx = close_xy[2 * row], cy = close_xy[2 * row + 1];
    int px = x, py = y;  // the vertex filed in the predicted polygon
    int fx = x, fy = y;  // the point the next step's newest history frame shows
    if (end_token) {
        if (x == P && y == 0) {  // the end token closes the polygon; the frame shows pixel (0, 0)
            required[row] = 0;
            px = cx, py = cy;
            fx = 0, fy = 0;
        }
    } else if (x == cx && y == cy) {
        required[row] = 0;
    }
    predicted[((size_t)row * Tmax + step) * 2] = px;
    predicted[((size_t)row * Tmax + step) * 2 + 1] = py;
    if (hist > 0) {  // np.roll(+1) over the frames, then the newest frame is overwritten
        int* hp = history + (size_t)row * hist * 2;
        const int lx = hp[2 * (hist - 1)], ly = hp[2 * (hist - 1) + 1];
        for (int i = hist - 1; i > 0; --i) hp[2 * i] = hp[2 * (i - 1)], hp[2 * i + 1] = hp[2 * (i - 1) + 1];
        hp[0] = lx, hp[1] = ly;
        hp[2 * (hist - 1)] = fx, hp[2 * (hist - 1) + 1] = fy;
    }
}

int check_rows(const char* who, int rows, int cols, int dtype) {
    ISG_REQUIRE(rows >= 0 && cols >= 1, ISG_EINVAL, "%s: bad shape rows=%d cols=%d", who, rows, cols);
    ISG_REQUIRE(dtype == ISG_F32 || dtype == ISG_BF16, ISG_EDTYPE, "%s: dtype %d", who, dtype);
    return ISG_OK;
}


// Masked sequence metrics in one launch of one workgroup (rows = B*T is a few hundred to a few
// thousand): out = {#(pred == class(target) and t < duration), sum of masked L2 errors,
// max of masked L2 errors, sum of durations}, all double.
__global__ __launch_bounds__(256) void sequence_metrics_kernel(const int* __restrict__ pred,
                                                              const int* __restrict__ targets,
                                                              const int* __restrict__ durations, int B, int T, int P,
                                                              double* __restrict__ out,
                                                              const float* __restrict__ loss_rows,
                                                              float* __restrict__ loss_mean) {
    __shared__ double red[4][256];
    double correct = 0.0, esum = 0.0, emax = 0.0, lsum = 0.0;
    for (int r = threadIdx.x; r < B * T; r += 256) {
        if (loss_rows) lsum += (double)loss_rows[r];
        const int b = r / T, t = r - b * T;
        const bool live = t < durations[b];
        const int tx = targets[2 * r], ty = targets[2 * r + 1];
        const int a = pred[r];
        const int ax = a / P, ay = a - ax * P;
        if (live && a == tx * P + ty) correct += 1.0;
        const double dx = (double)(tx - ax), dy = (double)(ty - ay);
        const double e = live ? sqrt(dx * dx + dy * dy) : 0.0;
        esum += e;
        emax = fmax(emax, e);
    }
    red[0][threadIdx.x] = correct, red[1][threadIdx.x] = esum, red[2][threadIdx.x] = emax, red[3][threadIdx.x] = lsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
            red[2][threadIdx.x] = fmax(red[2][threadIdx.x], red[2][threadIdx.x + o]);
            red[3][threadIdx.x] += red[3][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        long long num = 0;
        for (int b = 0; b < B; ++b) num += durations[b];
        out[0] = red[0][0], out[1] = red[1][0], out[2] = red[2][0], out[3] = (double)num;
        if (loss_mean) {  // the summary form: the two quotients and the mean loss as well
            out[4] = red[0][0] / (double)num, out[5] = red[1][0] / (double)num;
            *loss_mean = (float)(red[3][0] / (double)(B * T));
        }
    }
}

}  // namespace

extern "C" int isg_softmax_xent_fwd(const void* logits, const int32_t* labels, const float* weights, int rows,
                                    int cols, int dtype, float* loss_rows, float* lse_rows, int32_t* argmax_rows,
                                    void* stream) {
    int rc = check_rows("isg_softmax_xent_fwd", rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && (labels || !loss_rows), ISG_EINVAL, "isg_softmax_xent_fwd: null pointer");
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(xent_fwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), labels,
                           weights, rows, cols, loss_rows, lse_rows, argmax_rows, nullptr, 0, nullptr);
    else
        hipLaunchKernelGGL(xent_fwd_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), labels, weights, rows, cols, loss_rows,
                           lse_rows, argmax_rows, nullptr, 0, nullptr);
    return isg::check_launch("xent_fwd_kernel");
}

extern "C" int isg_sequence_xent_fwd(const void* logits, const int32_t* targets_xy, int rows, int cols, int grid_side,
                                     int dtype, float* loss_rows, float* lse_rows, int32_t* argmax_rows,
                                     int32_t* labels_out, void* stream) {
    int rc = check_rows("isg_sequence_xent_fwd", rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && targets_xy && loss_rows && grid_side >= 1, ISG_EINVAL, "isg_sequence_xent_fwd: bad arguments");
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(xent_fwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), nullptr, nullptr,
                           rows, cols, loss_rows, lse_rows, argmax_rows, targets_xy, grid_side, labels_out);
    else
        hipLaunchKernelGGL(xent_fwd_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), nullptr, nullptr, rows, cols, loss_rows, lse_rows,
                           argmax_rows, targets_xy, grid_side, labels_out);
    return isg::check_launch("xent_fwd_kernel");
}

extern "C" int isg_softmax_xent_bwd(const void* logits, const int32_t* labels, const float* lse_rows,
                                    const float* row_scale, float scale, int rows, int cols, int dtype, void* dlogits,
                                    void* stream) {
    int rc = check_rows("isg_softmax_xent_bwd", rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && labels && lse_rows && dlogits, ISG_EINVAL, "isg_softmax_xent_bwd: null pointer");
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(xent_bwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), labels,
                           lse_rows, row_scale, scale, rows, cols, static_cast<float*>(dlogits), nullptr);
    else
        hipLaunchKernelGGL(xent_bwd_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), labels, lse_rows, row_scale, scale, rows, cols,
                           static_cast<__hip_bfloat16*>(dlogits), nullptr);
    return isg::check_launch("xent_bwd_kernel");
}

extern "C" int isg_softmax_xent_mean_bwd(const void* logits, const int32_t* labels, const float* lse_rows,
                                         const float* upstream, int rows, int cols, int dtype, void* dlogits,
                                         void* stream) {
    int rc = check_rows("isg_softmax_xent_mean_bwd", rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && labels && lse_rows && dlogits && upstream, ISG_EINVAL, "isg_softmax_xent_mean_bwd: null pointer");
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float scale = 1.0f / (float)rows;
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(xent_bwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), labels,
                           lse_rows, nullptr, scale, rows, cols, static_cast<float*>(dlogits), upstream);
    else
        hipLaunchKernelGGL(xent_bwd_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), labels, lse_rows, nullptr, scale, rows, cols,
                           static_cast<__hip_bfloat16*>(dlogits), upstream);
    return isg::check_launch("xent_bwd_kernel");
}

extern "C" int isg_argmax_rows(const void* logits, int rows, int cols, int dtype, int32_t* out_idx, void* stream) {
    ISG_REQUIRE(out_idx || rows == 0, ISG_EINVAL, "isg_argmax_rows: null output");
    return isg_softmax_xent_fwd(logits, nullptr, nullptr, rows, cols, dtype, nullptr, nullptr, out_idx, stream);
}

namespace {
int launch_sample(const char* who, const void* logits, int rows, int cols, int dtype, unsigned long long seed,
                  unsigned long long row_offset, const unsigned long long* offset_dev, int grid_side, int32_t* out,
                  void* stream) {
    int rc = check_rows(who, rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && out, ISG_EINVAL, "%s: null pointer", who);
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(sample_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), rows, cols,
                           seed, row_offset, offset_dev, grid_side, out);
    else
        hipLaunchKernelGGL(sample_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), rows, cols, seed, row_offset, offset_dev,
                           grid_side, out);
    return isg::check_launch("sample_kernel");
}
}  // namespace

extern "C" int isg_sample_rows(const void* logits, int rows, int cols, int dtype, unsigned long long seed,
                               unsigned long long row_offset, int32_t* out_idx, void* stream) {
    return launch_sample("isg_sample_rows", logits, rows, cols, dtype, seed, row_offset, nullptr, 0, out_idx, stream);
}

extern "C" int isg_sample_actions(const void* logits, int rows, int cols, int dtype, unsigned long long seed,
                                  unsigned long long row_offset, const unsigned long long* row_offset_dev, int P,
                                  int32_t* out_xy, void* stream) {
    ISG_REQUIRE(P >= 1 && cols <= P * P + P, ISG_EINVAL, "isg_sample_actions: cols=%d does not fit a %d x %d grid", cols,
                P, P);
    return launch_sample("isg_sample_actions", logits, rows, cols, dtype, seed, row_offset, row_offset_dev, P, out_xy,
                         stream);
}

extern "C" int isg_decode_step(const void* logits, int rows, int cols, int dtype, int P, int end_token,
                               const int32_t* close_xy, int32_t* required, int32_t* durations, int32_t* predicted,
                               int T, int step, int32_t* history_points, int hist, void* stream) {
    int rc = check_rows("isg_decode_step", rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    ISG_REQUIRE(P >= 1 && cols <= P * P + P && hist >= 0 && T >= 1 && step >= 0 && step < T, ISG_EINVAL,
                "isg_decode_step: cols=%d P=%d hist=%d step=%d T=%d", cols, P, hist, step, T);
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && close_xy && required && durations && predicted && (hist == 0 || history_points), ISG_EINVAL,
                "isg_decode_step: null pointer");
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(decode_step_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), rows, cols, P,
                           end_token, close_xy, required, durations, predicted, T, step, history_points, hist);
    else
        hipLaunchKernelGGL(decode_step_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), rows, cols, P, end_token, close_xy, required,
                           durations, predicted, T, step, history_points, hist);
    return isg::check_launch("decode_step_kernel");
}

namespace {
// one thread per sample: what the decode loop starts from
__global__ __launch_bounds__(256) void decode_begin_kernel(const int32_t* __restrict__ vt, const int32_t* __restrict__ nvt,
                                                           int B, int vmax, int hist, int T, int32_t* __restrict__ close_xy,
                                                           int32_t* __restrict__ pts0, int32_t* __restrict__ pts,
                                                           int32_t* __restrict__ required, int32_t* __restrict__ durations,
                                                           int32_t* __restrict__ predicted) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int n = nvt[b];
    const int frames = hist > 0 ? hist : 1;
    for (int i = 0; i < frames; ++i) {
        int v = 0;
        if (n > 0) {
            v = (n - hist + i) % n;
            v = v < 0 ? v + n : v;
        }
        const int x = vt[((size_t)b * vmax + v) * 2], y = vt[((size_t)b * vmax + v) * 2 + 1];
        if (i == 0) close_xy[2 * b] = x, close_xy[2 * b + 1] = y;
        if (i < hist) {
            const size_t o = ((size_t)b * hist + i) * 2;
            pts0[o] = x, pts0[o + 1] = y, pts[o] = x, pts[o + 1] = y;
        }
    }
    required[b] = 1, durations[b] = 0;
    for (int i = 0; i < 2 * T; ++i) predicted[(size_t)b * 2 * T + i] = 0;
}

// one thread per (sample, output vertex): the predicted polygon as the rasteriser takes it
__global__ __launch_bounds__(256) void decode_polygons_kernel(const int32_t* __restrict__ predicted,
                                                              const int32_t* __restrict__ durations,
                                                              const int32_t* __restrict__ pts0, int B, int T, int hist,
                                                              int prepend, int by_duration, int scale, int vmax,
                                                              int32_t* __restrict__ vp, int32_t* __restrict__ count) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * vmax) return;
    const int b = (int)(i / vmax), j = (int)(i - (long long)b * vmax);
    const int lead = prepend ? hist : 0;
    int x = 0, y = 0;
    if (j < lead) {
        x = pts0[((size_t)b * hist + j) * 2], y = pts0[((size_t)b * hist + j) * 2 + 1];
    } else if (j - lead < T) {
        x = predicted[((size_t)b * T + (j - lead)) * 2] * scale, y = predicted[((size_t)b * T + (j - lead)) * 2 + 1] * scale;
    }
    vp[i * 2] = x, vp[i * 2 + 1] = y;
    if (j == 0) count[b] = (by_duration ? durations[b] : T) + lead;
}

// out[r][c] = sum over slices of part[s][r][c] + bias[c], in the order of the slices
template <typename T>
__global__ __launch_bounds__(256) void sum_slices_bias_kernel(const float* __restrict__ part, int slices, long long n,
                                                              int cols, const T* __restrict__ bias, T* __restrict__ out,
                                                              int part_cols, long long slice_stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // (the partial results may have more columns than are wanted: a weight matrix padded with zero columns)
    const long long src = part_cols == cols ? i : (i / cols) * part_cols + i % cols;
    float v = 0.0f;
    for (int s = 0; s < slices; ++s) v += part[(long long)s * slice_stride + src];
    out[i] = (T)(v + (float)bias[i % cols]);
}

// out[row] = x[row][:] . w[:] + bias: a dense layer with ONE output column (a value head).  As a library GEMM it is a
// matrix-vector product that streams x at a fraction of the memory rate; here one workgroup owns a row, every lane
// reads 16 bytes of x and of w (which stays in the L2) per trip and keeps float sums, then wave shuffles and LDS.
__global__ __launch_bounds__(256) void rows_dot_bf16_kernel(const __hip_bfloat16* __restrict__ x,
                                                            const __hip_bfloat16* __restrict__ w,
                                                            const __hip_bfloat16* __restrict__ bias, int cols,
                                                            float* __restrict__ out) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ float red[4];
    const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)blockIdx.x * cols);
    const u32x4* wr = reinterpret_cast<const u32x4*>(w);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = threadIdx.x; i < cols / 8; i += 256) {
        const u32x4 a = __builtin_nontemporal_load(xr + i), b = wr[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (a bf16 is the upper half of its float)
            acc[k] = __builtin_fmaf(__uint_as_float(a[k] << 16), __uint_as_float(b[k] << 16), acc[k]);
            acc[k] = __builtin_fmaf(__uint_as_float(a[k] & 0xffff0000u), __uint_as_float(b[k] & 0xffff0000u), acc[k]);
        }
    }
    const float v = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]) + (bias ? (float)bias[0] : 0.0f);
}
}  // namespace

namespace {
// One wave per row of the actor-critic loss (reference polyrl/a3c/a3c.py:201-216): log-softmax of the row, the
// chosen action's log-probability and the row's entropy in two passes over the logits (online max / sum, then
// sum of p z), and the row's three weighted contributions:
//   terms[row] = w * (-logp[a] * adv,  0.5 * (vf - ret)^2,  H)        stats[row] = (lse, H)
template <typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void a3c_loss_fwd_kernel(
    const T* __restrict__ logits, const float* __restrict__ vf, const int* __restrict__ actions,
    const float* __restrict__ adv, const float* __restrict__ ret, const unsigned char* __restrict__ live, int rows,
    int cols, float* __restrict__ terms, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* z = logits + (size_t)row * cols;
    float m = -INFINITY, s = 0.0f;
    for (int c = lane; c < cols; c += 64) {
        const float v = ld(z, c);
        if (v > m) {
            s = s * expf(m - v) + 1.0f;
            m = v;
        } else if (v != -INFINITY) {
            s += expf(v - m);
        }
    }
    const float gm = wave_max(m);
    s = m == -INFINITY ? 0.0f : s * expf(m - gm);
    s = wave_sum(s);
    const float lse = gm + logf(s);
    float pz = 0.0f;  // sum of p z over the row: H = lse - sum p z
    for (int c = lane; c < cols; c += 64) {
        const float v = ld(z, c);
        if (v != -INFINITY) pz += expf(v - lse) * v;
    }
    pz = wave_sum(pz);
    if (lane == 0) {
        const float w = live == nullptr || live[row] ? 1.0f : 0.0f;
        const int a = actions[row];
        const float logp_a = (a >= 0 && a < cols) ? ld(z, a) - lse : NAN;
        const float H = lse - pz;
        const float dv = vf[row] - ret[row];
        terms[3 * row] = w * (-logp_a * adv[row]);
        terms[3 * row + 1] = w * (0.5f * dv * dv);
        terms[3 * row + 2] = w * H;
        stats[2 * row] = lse, stats[2 * row + 1] = H;
    }
}

// dlogits[row][k] = w (g_pi * adv * (p_k - [k = a]) - g_ent * p_k (logp_k + H)),  dvf[row] = w g_vf (vf - ret)
// with upstream = (g_pi, g_vf, g_ent) read from device memory (no host round trip for the loss's own scale).
template <typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void a3c_loss_bwd_kernel(
    const T* __restrict__ logits, const float* __restrict__ vf, const int* __restrict__ actions,
    const float* __restrict__ adv, const float* __restrict__ ret, const unsigned char* __restrict__ live,
    const float* __restrict__ stats, const float* __restrict__ upstream, int rows, int cols, T* __restrict__ dlogits,
    float* __restrict__ dvf) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* z = logits + (size_t)row * cols;
    T* d = dlogits + (size_t)row * cols;
    const float w = live == nullptr || live[row] ? 1.0f : 0.0f;
    const float g_pi = upstream[0], g_vf = upstream[1], g_ent = upstream[2];
    const float lse = stats[2 * row], H = stats[2 * row + 1];
    const int a = actions[row];
    const float ca = w * g_pi * adv[row], ce = w * g_ent;
    for (int c = lane; c < cols; c += 64) {
        const float v = ld(z, c);
        const float logp = v - lse;
        const float p = v == -INFINITY ? 0.0f : expf(logp);
        float g = ca * (p - (c == a ? 1.0f : 0.0f));
        if (p > 0.0f) g -= ce * p * (logp + H);
        d[c] = (T)g;
    }
    if (lane == 0) dvf[row] = w * g_vf * (vf[row] - ret[row]);
}

constexpr int kColSumMaxBlocks = 256;

// thread = column; a workgroup sums its share of the rows (every row read is 256 adjacent elements)
template <typename T>
__global__ __launch_bounds__(256) void column_sums_partial_kernel(const T* __restrict__ x, long long rows, int cols,
                                                                  float* __restrict__ partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long per = (rows + gridDim.y - 1) / gridDim.y;
    const long long r0 = (long long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // (four loads in flight; summed in a fixed order)
    long long r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += ld(x, r * cols + c), s1 += ld(x, (r + 1) * cols + c);
        s2 += ld(x, (r + 2) * cols + c), s3 += ld(x, (r + 3) * cols + c);
    }
    for (; r < r1; ++r) s0 += ld(x, r * cols + c);
    partial[(size_t)blockIdx.y * cols + c] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void column_sums_finish_kernel(const float* __restrict__ partial, int nblk, int cols,
                                                                 float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.0f;
    for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * cols + c];
    out[c] = s;
}
}  // namespace

extern "C" long long isg_column_sums_workspace_bytes(int cols) {
    return cols >= 1 ? (long long)kColSumMaxBlocks * cols * (long long)sizeof(float) : -1;
}

extern "C" int isg_column_sums(const void* x, long long rows, int cols, int dtype, float* out, void* workspace,
                               void* stream) {
    int rc = check_rows("isg_column_sums", rows > 0x7fffffff ? 0x7fffffff : (int)rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    ISG_REQUIRE(out && workspace && (rows == 0 || x), ISG_EINVAL, "isg_column_sums: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    long long want = (rows + 31) / 32;
    const int nblk = want < 1 ? 1 : (want > kColSumMaxBlocks ? kColSumMaxBlocks : (int)want);
    const dim3 grid((cols + 255) / 256, nblk);
    float* partial = static_cast<float*>(workspace);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(column_sums_partial_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(x), rows,
                           cols, partial);
    else
        hipLaunchKernelGGL(column_sums_partial_kernel<__hip_bfloat16>, grid, dim3(256), 0, s,
                           static_cast<const __hip_bfloat16*>(x), rows, cols, partial);
    rc = isg::check_launch("column_sums_partial_kernel");
    if (rc != ISG_OK) return rc;
    hipLaunchKernelGGL(column_sums_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, partial, nblk, cols, out);
    return isg::check_launch("column_sums_finish_kernel");
}

extern "C" int isg_a3c_loss_fwd(const void* logits, const float* vf, const int32_t* actions, const float* adv,
                                const float* ret, const unsigned char* live, int rows, int cols, int dtype,
                                float* terms, float* stats, void* stream) {
    int rc = check_rows("isg_a3c_loss_fwd", rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && vf && actions && adv && ret && terms && stats, ISG_EINVAL, "isg_a3c_loss_fwd: null pointer");
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(a3c_loss_fwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), vf, actions,
                           adv, ret, live, rows, cols, terms, stats);
    else
        hipLaunchKernelGGL(a3c_loss_fwd_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), vf, actions, adv, ret, live, rows, cols, terms,
                           stats);
    return isg::check_launch("a3c_loss_fwd_kernel");
}

extern "C" int isg_a3c_loss_bwd(const void* logits, const float* vf, const int32_t* actions, const float* adv,
                                const float* ret, const unsigned char* live, const float* stats, const float* upstream3,
                                int rows, int cols, int dtype, void* dlogits, float* dvf, void* stream) {
    int rc = check_rows("isg_a3c_loss_bwd", rows, cols, dtype);
    if (rc != ISG_OK) return rc;
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(logits && vf && actions && adv && ret && stats && upstream3 && dlogits && dvf, ISG_EINVAL,
                "isg_a3c_loss_bwd: null pointer");
    const dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(a3c_loss_bwd_kernel<float>, grid, block, 0, s, static_cast<const float*>(logits), vf, actions,
                           adv, ret, live, stats, upstream3, rows, cols, static_cast<float*>(dlogits), dvf);
    else
        hipLaunchKernelGGL(a3c_loss_bwd_kernel<__hip_bfloat16>, grid, block, 0, s,
                           static_cast<const __hip_bfloat16*>(logits), vf, actions, adv, ret, live, stats, upstream3, rows,
                           cols, static_cast<__hip_bfloat16*>(dlogits), dvf);
    return isg::check_launch("a3c_loss_bwd_kernel");
}

extern "C" int isg_decode_begin(const int32_t* true_vertices, const int32_t* num_true, int B, int vmax, int hist, int T,
                                int32_t* close_xy, int32_t* first_points, int32_t* history_points, int32_t* required,
                                int32_t* durations, int32_t* predicted, void* stream) {
    ISG_REQUIRE(B >= 0 && vmax >= 1 && hist >= 0 && T >= 1, ISG_EINVAL, "isg_decode_begin: B=%d vmax=%d hist=%d T=%d", B,
                vmax, hist, T);
    if (B == 0) return ISG_OK;
    ISG_REQUIRE(true_vertices && num_true && close_xy && required && durations && predicted &&
                    (hist == 0 || (first_points && history_points)), ISG_EINVAL, "isg_decode_begin: null pointer");
    hipLaunchKernelGGL(decode_begin_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       true_vertices, num_true, B, vmax, hist, T, close_xy, first_points, history_points, required,
                       durations, predicted);
    return isg::check_launch("decode_begin_kernel");
}

extern "C" int isg_decode_polygons(const int32_t* predicted, const int32_t* durations, const int32_t* first_points, int B,
                                   int T, int hist, int prepend_history, int slice_by_duration, int scale, int vmax,
                                   int32_t* polygons, int32_t* counts, void* stream) {
    ISG_REQUIRE(B >= 0 && T >= 1 && hist >= 0 && scale >= 1 && vmax >= T + (prepend_history ? hist : 0), ISG_EINVAL,
                "isg_decode_polygons: B=%d T=%d hist=%d scale=%d vmax=%d", B, T, hist, scale, vmax);
    if (B == 0) return ISG_OK;
    ISG_REQUIRE(predicted && durations && polygons && counts && (!prepend_history || hist == 0 || first_points),
                ISG_EINVAL, "isg_decode_polygons: null pointer");
    const long long n = (long long)B * vmax;
    hipLaunchKernelGGL(decode_polygons_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), predicted, durations, first_points, B, T, hist,
                       prepend_history ? 1 : 0, slice_by_duration ? 1 : 0, scale, vmax, polygons, counts);
    return isg::check_launch("decode_polygons_kernel");
}

namespace {
int sum_slices_bias_impl(const char* who, const float* part, int slices, int rows, int cols, int part_cols,
                         const void* bias, int dtype, void* out, void* stream) {
    ISG_REQUIRE(dtype == ISG_F32 || dtype == ISG_BF16, ISG_EDTYPE, "%s: dtype %d", who, dtype);
    ISG_REQUIRE(slices >= 1 && rows >= 0 && cols >= 1 && part_cols >= cols, ISG_EINVAL,
                "%s: slices=%d rows=%d cols=%d part_cols=%d", who, slices, rows, cols, part_cols);
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(part && bias && out, ISG_EINVAL, "%s: null pointer", who);
    const long long n = (long long)rows * cols;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long slice_stride = (long long)rows * part_cols;
    if (dtype == ISG_F32)
        hipLaunchKernelGGL(sum_slices_bias_kernel<float>, grid, block, 0, s, part, slices, n, cols,
                           static_cast<const float*>(bias), static_cast<float*>(out), part_cols, slice_stride);
    else
        hipLaunchKernelGGL(sum_slices_bias_kernel<__hip_bfloat16>, grid, block, 0, s, part, slices, n, cols,
                           static_cast<const __hip_bfloat16*>(bias), static_cast<__hip_bfloat16*>(out), part_cols,
                           slice_stride);
    return isg::check_launch("sum_slices_bias_kernel");
}
}  // namespace

extern "C" int isg_sum_slices_bias(const float* part, int slices, int rows, int cols, const void* bias, int dtype,
                                   void* out, void* stream) {
    return sum_slices_bias_impl("isg_sum_slices_bias", part, slices, rows, cols, cols, bias, dtype, out, stream);
}

extern "C" int isg_sum_slices_bias_padded(const float* part, int slices, int rows, int cols, int part_cols,
                                          const void* bias, int dtype, void* out, void* stream) {
    return sum_slices_bias_impl("isg_sum_slices_bias_padded", part, slices, rows, cols, part_cols, bias, dtype, out, stream);
}

extern "C" int isg_rows_dot(const void* x, const void* w, const void* bias, int rows, int cols, int dtype, float* out,
                            void* stream) {
    ISG_REQUIRE(dtype == ISG_BF16, ISG_EDTYPE, "isg_rows_dot: dtype %d (bf16 only)", dtype);
    ISG_REQUIRE(rows >= 0 && cols >= 8 && cols % 8 == 0, ISG_EINVAL, "isg_rows_dot: rows=%d cols=%d (a multiple of 8)", rows,
                cols);
    if (rows == 0) return ISG_OK;
    ISG_REQUIRE(x && w && out, ISG_EINVAL, "isg_rows_dot: null pointer");
    ISG_REQUIRE(((uintptr_t)x | (uintptr_t)w) % 16 == 0, ISG_EINVAL, "isg_rows_dot: x and w must be 16-byte aligned");
    hipLaunchKernelGGL(rows_dot_bf16_kernel, dim3((unsigned)rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const __hip_bfloat16*>(x), static_cast<const __hip_bfloat16*>(w),
                       static_cast<const __hip_bfloat16*>(bias), cols, out);
    return isg::check_launch("rows_dot_bf16_kernel");
}

extern "C" int isg_sequence_metrics(const int32_t* pred, const int32_t* targets, const int32_t* durations, int B, int T,
                                    int P, double* out4, void* stream) {
    ISG_REQUIRE(B >= 0 && T >= 0 && P >= 1, ISG_EINVAL, "isg_sequence_metrics: bad shape B=%d T=%d P=%d", B, T, P);
    ISG_REQUIRE(out4 && (B * T == 0 || (pred && targets && durations)), ISG_EINVAL,
                "isg_sequence_metrics: null pointer");
    hipLaunchKernelGGL(sequence_metrics_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), pred, targets,
                       durations, B, T, P, out4, nullptr, nullptr);
    return isg::check_launch("sequence_metrics_kernel");
}

extern "C" int isg_sequence_summary(const int32_t* pred, const int32_t* targets, const int32_t* durations,
                                    const float* loss_rows, int B, int T, int P, double* out6, float* loss_mean,
                                    void* stream) {
    ISG_REQUIRE(B >= 1 && T >= 1 && P >= 1, ISG_EINVAL, "isg_sequence_summary: bad shape B=%d T=%d P=%d", B, T, P);
    ISG_REQUIRE(out6 && loss_mean && pred && targets && durations && loss_rows, ISG_EINVAL,
                "isg_sequence_summary: null pointer");
    hipLaunchKernelGGL(sequence_metrics_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), pred, targets,
                       durations, B, T, P, out6, loss_rows, loss_mean);
    return isg::check_launch("sequence_metrics_kernel");
}
