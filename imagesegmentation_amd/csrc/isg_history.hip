// History frames, environment steps, the teacher-forcing batch builder and the decode step: the small integer
// kernels around the recurrence.
// history_frames_kernel: frames[n][y][x][i] = 1 where (x, y) == points[n][i]; write-only, one thread per pixel
// writing that pixel's `hist` channels (the frames usually land beside other channels of a state tensor, so the
// stride between pixels is the caller's and the stores are element-sized).
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

// One environment per thread: the whole of a batched env.step except the reward.
__global__ __launch_bounds__(256) void env_step_kernel(const int* __restrict__ actions, const int* __restrict__ truth,
                                                       int truth_stride, int* __restrict__ predicted, int vcap,
                                                       int* __restrict__ num_predicted, int* __restrict__ history, int hist,
                                                       unsigned char* __restrict__ done, int n, int size, int end_token,
                                                       int* __restrict__ rec_points, int* __restrict__ rec_actions,
                                                       int* __restrict__ rec_lengths, int T, int t) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int ax = actions[2 * e], ay = actions[2 * e + 1];
    const bool active = done[e] == 0;
    int* hp = history + (size_t)e * hist * 2;
    if (rec_points != nullptr)
        for (int i = 0; i < 2 * hist; ++i) rec_points[((size_t)e * T + t) * hist * 2 + i] = hp[i];
    if (rec_actions != nullptr) rec_actions[((size_t)e * T + t) * 2] = ax, rec_actions[((size_t)e * T + t) * 2 + 1] = ay;
    if (rec_lengths != nullptr && active) rec_lengths[e] += 1;
    if (!active) return;
    const int fx = truth[(size_t)e * truth_stride], fy = truth[(size_t)e * truth_stride + 1];
    bool terminal, push;
    int vx = ax, vy = ay;
    if (end_token) {
        terminal = ax == size && ay == 0;  // the end token closes the polygon on the first true vertex
        push = !terminal;
        if (terminal) vx = fx, vy = fy;
    } else {
        terminal = ax == fx && ay == fy;
        push = true;
    }
    const int slot = min(num_predicted[e], vcap - 1);
    predicted[((size_t)e * vcap + slot) * 2] = vx;
    predicted[((size_t)e * vcap + slot) * 2 + 1] = vy;
    num_predicted[e] += 1;
    if (push && hist > 0) {
        // frames roll by one slot (the last wraps to the front), then the action takes the last slot
        const int lx = hp[2 * (hist - 1)], ly = hp[2 * (hist - 1) + 1];
        for (int i = hist - 1; i > 0; --i) hp[2 * i] = hp[2 * (i - 1)], hp[2 * i + 1] = hp[2 * (i - 1) + 1];
        hp[0] = lx, hp[1] = ly;
        hp[2 * (hist - 1)] = ax, hp[2 * (hist - 1) + 1] = ay;
    }
    if (terminal) done[e] = 1;
}

// One thread per (sample, slot): slot t < T writes step t's history points and target, slot k < vcap writes
// rolled vertex k.  np.roll(v, s) puts v[(k - s) mod n] at position k.
__global__ __launch_bounds__(256) void batch_points_kernel(const int* __restrict__ table, const int* __restrict__ counts,
                                                           int vcap, const int* __restrict__ indices,
                                                           const int* __restrict__ starts, int B, int T, int hist, int P,
                                                           int end_token, int* __restrict__ durations,
                                                           int* __restrict__ points, int* __restrict__ targets,
                                                           int* __restrict__ rolled, int* __restrict__ nverts) {
    const int slots = T > vcap ? T : vcap;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * slots) return;
    const int b = gid / slots, slot = gid - b * slots;
    const int sample = indices[b];
    // The counts live on the device, where the host cannot check them: a polygon of no vertices (or a count past
    // the table's width) is treated as one of a single vertex / of vcap vertices instead of dividing by zero or
    // reading past its row.  PolygonDataset never holds either.
    int n = counts[sample];
    n = n < 1 ? 1 : (n > vcap ? vcap : n);
    const int* v = table + (size_t)sample * vcap * 2;
    // floor modulo: start vertices may be given negative or beyond n, as np.roll takes them
    const int s = ((starts[b] % n) + n) % n;
    auto vertex = [&](int k, int& x, int& y) {  // rolled vertex k mod n
        int r = ((k % n) + n) % n - s;
        r = r < 0 ? r + n : r;
        x = v[2 * r], y = v[2 * r + 1];
    };
    const int dur = n - hist + 1;
    const int filled = dur < T ? dur : T;  // steps the reference's loop writes
    if (slot == 0) {
        durations[b] = dur;
        if (nverts) nverts[b] = n;
    }
    if (slot < T) {
        const int t = slot;
        const bool live = t < filled;
        for (int i = 0; i < hist; ++i) {
            int x = -1, y = -1;
            if (live) vertex(t - hist + i + 1, x, y);
            int* d = points + (((size_t)b * T + t) * hist + i) * 2;
            d[0] = x, d[1] = y;
        }
        int tx = 0, ty = 0;
        if (live) vertex(t + 1, tx, ty);
        const int last = filled > 0 ? filled - 1 : 0;
        if (end_token && t == last) tx = P, ty = 0;
        targets[((size_t)b * T + t) * 2] = tx;
        targets[((size_t)b * T + t) * 2 + 1] = ty;
    }
    if (rolled && slot < vcap) {
        int x = 0, y = 0;
        if (slot < n) vertex(slot, x, y);
        rolled[((size_t)b * vcap + slot) * 2] = x;
        rolled[((size_t)b * vcap + slot) * 2 + 1] = y;
    }
}

}  // namespace

extern "C" int isg_batch_points(const int32_t* table, const int32_t* counts, int vcap, const int32_t* indices,
                                const int32_t* starts, int B, int T, int hist, int P, int end_token,
                                int32_t* durations, int32_t* points, int32_t* targets, int32_t* rolled, int32_t* nverts,
                                void* stream) {
    ISG_REQUIRE(B >= 0 && T >= 1 && hist >= 0 && vcap >= 1 && P >= 1, ISG_EINVAL, "isg_batch_points: bad sizes");
    if (B == 0) return ISG_OK;
    ISG_REQUIRE(table && counts && indices && starts && durations && targets && (hist == 0 || points), ISG_EINVAL,
                "isg_batch_points: null pointer");
    const long long threads = (long long)B * (T > vcap ? T : vcap);
    ISG_REQUIRE(threads < (1LL << 31), ISG_EINVAL, "isg_batch_points: batch too large");
    hipLaunchKernelGGL(batch_points_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), table, counts, vcap, indices, starts, B, T, hist, P, end_token,
                       durations, points, targets, rolled, nverts);
    return isg::check_launch("batch_points_kernel");
}

extern "C" int isg_env_step(const int32_t* actions, const int32_t* true_vertices, int true_stride, int32_t* predicted,
                            int vcap, int32_t* num_predicted, int32_t* history_points, int hist, uint8_t* done, int n,
                            int size, int end_token, int32_t* rec_points, int32_t* rec_actions, int32_t* rec_lengths,
                            int T, int t, void* stream) {
    ISG_REQUIRE(n >= 0 && vcap >= 1 && hist >= 0 && size >= 1 && true_stride >= 2, ISG_EINVAL, "isg_env_step: bad sizes");
    ISG_REQUIRE((!rec_points && !rec_actions) || (T >= 1 && t >= 0 && t < T), ISG_EINVAL,
                "isg_env_step: record slot %d outside [0, %d)", t, T);
    if (n == 0) return ISG_OK;
    ISG_REQUIRE(actions && true_vertices && predicted && num_predicted && done && (hist == 0 || history_points),
                ISG_EINVAL, "isg_env_step: null pointer");
    hipLaunchKernelGGL(env_step_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), actions,
                       true_vertices, true_stride, predicted, vcap, num_predicted, history_points, hist, done, n, size,
                       end_token, rec_points, rec_actions, rec_lengths, T, t);
    return isg::check_launch("env_step_kernel");
}

namespace {
// one thread per rollout: the two backward scans of reference polyrl/a3c/a3c.py:15-32 in double, each sum rounded
// as written (no fused multiply-add: the host code this replaces rounds the product first)
__global__ __launch_bounds__(256) void gae_kernel(const double* __restrict__ rewards, const float* __restrict__ values,
                                                  const int32_t* __restrict__ lengths, const double* __restrict__ boot,
                                                  int N, int T, double gamma, double gl, double* __restrict__ adv,
                                                  double* __restrict__ ret) {
#pragma clang fp contract(off)
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    int L = lengths[n];
    L = L < 0 ? 0 : (L > T ? T : L);
    const double b = boot ? boot[n] : 0.0;
    double acc_a = 0.0, acc_r = 0.0;
    for (int t = T - 1; t >= 0; --t) {
        const size_t o = (size_t)n * T + t;
        if (t >= L) {
            adv[o] = 0.0, ret[o] = 0.0;
            continue;
        }
        const bool last = t == L - 1;
        const double r = rewards[o], v = (double)values[o], v_next = last ? b : (double)values[o + 1];
        const double td = r + gamma * v_next - v;
        const double r_in = last ? r + gamma * b : r;
        acc_a = td + gl * acc_a;
        acc_r = r_in + gamma * acc_r;
        adv[o] = acc_a, ret[o] = acc_r;
    }
}
}  // namespace

extern "C" int isg_gae(const double* rewards, const float* values, const int32_t* lengths, const double* bootstrap, int N,
                       int T, double gamma, double lambda, double* advantages, double* returns, void* stream) {
    ISG_REQUIRE(N >= 0 && T >= 1, ISG_EINVAL, "isg_gae: N=%d T=%d", N, T);
    if (N == 0) return ISG_OK;
    ISG_REQUIRE(rewards && values && lengths && advantages && returns, ISG_EINVAL, "isg_gae: null pointer");
    hipLaunchKernelGGL(gae_kernel, dim3((N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), rewards, values,
                       lengths, bootstrap, N, T, gamma, gamma * lambda, advantages, returns);
    return isg::check_launch("gae_kernel");
}

namespace {
// one thread per environment: the step's entries of a rollout's record.  An environment ended at THIS step when it is
// done and its length - the count of steps it took while running - is t + 1.
__global__ __launch_bounds__(256) void rollout_record_kernel(const uint8_t* __restrict__ done,
                                                             const int32_t* __restrict__ lengths,
                                                             const int32_t* __restrict__ inter,
                                                             const int32_t* __restrict__ uni,
                                                             const float* __restrict__ step_values, int n, int T, int t,
                                                             double* __restrict__ rewards, float* __restrict__ values) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const bool terminal = done[e] != 0 && lengths[e] == t + 1;
    rewards[(size_t)e * T + t] = terminal ? (double)inter[e] / (double)uni[e] : 0.0;
    if (values) values[(size_t)e * T + t] = step_values[e];
}
}  // namespace

extern "C" int isg_rollout_record(const uint8_t* done, const int32_t* lengths, const int32_t* inter, const int32_t* uni,
                                  const float* step_values, int n, int T, int t, double* rewards, float* values,
                                  void* stream) {
    ISG_REQUIRE(n >= 0 && T >= 1 && t >= 0 && t < T, ISG_EINVAL, "isg_rollout_record: n=%d, slot %d outside [0, %d)", n, t, T);
    if (n == 0) return ISG_OK;
    ISG_REQUIRE(done && lengths && inter && uni && rewards && ((values == nullptr) == (step_values == nullptr)),
                ISG_EINVAL, "isg_rollout_record: null pointer (values and step_values go together)");
    hipLaunchKernelGGL(rollout_record_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), done,
                       lengths, inter, uni, step_values, n, T, t, rewards, values);
    return isg::check_launch("rollout_record_kernel");
}

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
