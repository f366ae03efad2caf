/* libisg - C ABI of the MI355X (gfx950) hot-path kernels for Shadowen/ImageSegmentation.
 *
 * The reference is pure Python (TensorFlow 1.x graph ops + scikit-image); it has no FFI of its
 * own, so each entry point below names the reference call sequence it replaces.  A reference-side
 * binding is a ctypes stub: load the library after the process's HIP runtime, set argtypes, pass device pointers.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless the comment says "host";
 *  - nothing here allocates, frees or synchronises; each call only enqueues work on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream);
 *  - return value: ISG_OK or a negative ISG_E* code; isg_last_error() gives the text for the
 *    calling thread;
 *  - tensors are dense, row-major, NHWC where images are involved, exactly as the reference
 *    feeds them to TensorFlow.
 */
#ifndef ISG_H
#define ISG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISG_OK 0
#define ISG_EINVAL (-1)   /* bad shape / size / null pointer */
#define ISG_EDTYPE (-2)   /* unsupported dtype code */
#define ISG_ELAUNCH (-3)  /* HIP launch or runtime failure */

/* dtype codes for activation / weight storage */
#define ISG_F32 0
#define ISG_BF16 1

#define ISG_RASTER_MAX_SIZE 512  /* largest mask edge P */
#define ISG_RASTER_MAX_VERTS 64  /* largest Vmax */

/* ABI version (major*100 + minor). */
int isg_version(void);

/* Text of the last error raised on the calling thread ("" if none). Host pointer, do not free. */
const char* isg_last_error(void);

/* ---- polygon rasteriser + mask IoU ------------------------------------------------------
 * Replaces, for n independent rollouts in one launch,
 *   util.create_shape_mask            (reference util.py:60-65, skimage.draw.polygon fill rule)
 *   PolygonEnv._calculate_iou         (reference polyrl/polygon_env.py:48-53,
 *                                      polyrl/polygon_env_end_token.py:49-54; inline copies at
 *                                      polyrnn/Model.py:252-260,
 *                                      polyrnn/EndTokenDeepExperiment_2.py:173-182).
 *
 * verts_pred / verts_true : int32 [n][vmax][2]  (x, y) pairs; entries past nverts[i] are ignored
 * nverts_pred / nverts_true: int32 [n]          vertex counts, 0..vmax (0 -> empty mask)
 * size                    : mask edge P (1..ISG_RASTER_MAX_SIZE); masks are [P][P], index [y][x]
 * out_inter / out_union   : int32 [n]           count(pred AND true), count(pred OR true)
 * mask_pred / mask_true   : uint8 [n][P][P] with values 0/1, or NULL to skip materialising
 *
 * Pixels beyond the image are clipped (the reference raises IndexError there, util.py:64);
 * vertex coordinates are clamped to [-16383, 16383] before use.
 */
int isg_raster_iou(const int32_t* verts_pred, const int32_t* nverts_pred, const int32_t* verts_true,
                   const int32_t* nverts_true, int n, int vmax, int size, int32_t* out_inter,
                   int32_t* out_union, uint8_t* mask_pred, uint8_t* mask_true, void* stream);

/* Single-polygon form of the above: masks only (util.create_shape_mask for a batch).
 * verts int32 [n][vmax][2], nverts int32 [n], mask uint8 [n][P][P]; area int32 [n] or NULL. */
int isg_shape_mask(const int32_t* verts, const int32_t* nverts, int n, int vmax, int size, uint8_t* mask,
                   int32_t* area, void* stream);

/* ---- one-hot history frames --------------------------------------------------------------
 * Replaces util.create_history / util.create_point_mask + np.roll (reference util.py:44-57,
 * polyrl/polygon_env.py:37-39, Dataset.py:145-148) for n environments or batch rows at once.
 *
 * points : int32 [n][hist][2]  (x, y) of the vertex shown in frame i; a negative x = empty frame
 * out    : dtype [n][P][P][out_channels]; channels out_offset .. out_offset+hist-1 are written
 *          (1 at [y][x] of the frame's point, 0 elsewhere), other channels are left untouched,
 *          so the frames can land beside the image channels of an environment state.
 */
int isg_history_frames(const int32_t* points, int n, int size, int hist, int dtype, void* out, int out_channels,
                       int out_offset, void* stream);

/* The teacher-forcing batch builder (reference Dataset.py:104-150 get_batch_for_rnn / _create_sample_sequence,
 * EndTokenDataset.py:132-152) for B samples in one launch, from a dataset kept on the device as a padded
 * vertex table.  The reference's random draws (which samples, which start vertex) are the caller's: `indices`
 * and `starts`.  Sample b = polygon indices[b] rolled by starts[b] as np.roll does (rolled vertex k is vertex
 * (k - start) mod n).  With n its vertex count:
 *   durations[b] = n - hist + 1;  steps t < min(durations[b], T) are filled:
 *   points [B][T][hist][2]: frame i of step t = rolled vertex t - hist + i + 1 (the reference's create_history
 *                           with end_idx = t), as (x, y); (-1, -1) for steps that are not filled;
 *   targets [B][T][2]:      rolled vertex t + 1; (0, 0) for steps that are not filled; with end_token the
 *                           target of the last filled step is (P, 0) (EndTokenDataset.py:150);
 *   rolled [B][vcap][2] (may be NULL): the rolled polygon, zeros past n;  nverts [B] (may be NULL): n.
 * table int32 [N][vcap][2], counts int32 [N] (each 1..vcap), indices / starts int32 [B]. */
int isg_batch_points(const int32_t* table, const int32_t* counts, int vcap, const int32_t* indices,
                     const int32_t* starts, int B, int T, int hist, int P, int end_token, int32_t* durations,
                     int32_t* points, int32_t* targets, int32_t* rolled, int32_t* nverts, void* stream);

/* One step of n lock-stepped polygon environments, everything but the reward (reference
 * polyrl/polygon_env.py:25-46 and polyrl/polygon_env_end_token.py:25-47 for one environment):
 * an environment that is not done appends the action to its predicted polygon (the end token
 * (size, 0) appends the first TRUE vertex instead and ends the episode; without end token the episode
 * ends when the action equals the first true vertex), rolls its history points (last frame wraps to the
 * front, the action takes the last slot; not on the end token) and sets its done flag.  Done
 * environments ignore the action.
 *   actions int32 [n][2] (x, y); true_vertices int32, first vertex of env e at [e * true_stride];
 *   predicted int32 [n][vcap][2]; num_predicted int32 [n]; history_points int32 [n][hist][2];
 *   done uint8 [n].  Optional episode record (any may be NULL): rec_points int32 [n][T][hist][2] gets the
 *   history points BEFORE the step at slot t, rec_actions int32 [n][T][2] the action, rec_lengths [n] is
 *   incremented for environments that were not done. */
int isg_env_step(const int32_t* actions, const int32_t* true_vertices, int true_stride, int32_t* predicted, int vcap,
                 int32_t* num_predicted, int32_t* history_points, int hist, uint8_t* done, int n, int size,
                 int end_token, int32_t* rec_points, int32_t* rec_actions, int32_t* rec_lengths, int T, int t,
                 void* stream);

/* Advantages and returns of N rollouts of at most T steps (reference polyrl/a3c/a3c.py:11-32: td_t = r_t +
 * gamma v_{t+1} - v_t, advantage = discount(td, gamma lambda), return = discount(r ++ bootstrap, gamma)), one
 * launch: rewards double [N][T], values float [N][T], lengths int32 [N], bootstrap double [N] or NULL (0) - the value
 * after a rollout's last step; advantages / returns double [N][T], zero at t >= lengths[n]. */
int isg_gae(const double* rewards, const float* values, const int32_t* lengths, const double* bootstrap, int N, int T,
            double gamma, double lambda, double* advantages, double* returns, void* stream);

/* Slot t of a rollout's record after isg_env_step (with rec_lengths) and the IoU launch of that step (reference
 * polyrl/polygon_env.py:33-46 pays the IoU at the terminal step, polyrl/a3c/a3c.py:122-147 files reward and value):
 * rewards double [n][T] at [e][t] = inter[e] / union[e] (in double) where environment e ended at this step - done[e]
 * and lengths[e] == t + 1, lengths counting from 0 at the rollout's first step - else 0; values float [n][T] at
 * [e][t] = step_values[e] (both NULL: no values kept). */
int isg_rollout_record(const uint8_t* done, const int32_t* lengths, const int32_t* inter, const int32_t* uni,
                       const float* step_values, int n, int T, int t, double* rewards, float* values, void* stream);

/* ---- ConvLSTM cell ------------------------------------------------------------------------
 * One step of the reference cell (reference convLSTM.py:26-45):
 *     y = conv(h_prev, W_h, SAME) + conv(x, W_x, SAME) + b ;  j, i, f, o = split(y, 4)
 *     c = sigmoid(f) * c_prev + sigmoid(i) * tanh(j) ;  h = sigmoid(o) * tanh(c)
 * as it runs inside tf.nn.dynamic_rnn(sequence_length=duration) (reference
 * polyrnn/History2LargeExperiment.py:31-46): where durations != NULL and t >= durations[b], the
 * state of sample b is copied through and its emitted output is zero.
 *
 * Storage: activations (x, h, gate buffers, their gradients) have `dtype` ISG_F32 or ISG_BF16,
 * NHWC; the cell state c and all parameter gradients are always float.  F must be 16.
 * x has Cs allocated channels per pixel, Cs % 16 == 0 (bf16) or Cs % 4 == 0 (f32); channels beyond
 * the C real ones must hold finite values (they meet zero weights).
 * Weights stay in the reference's layout, w_x float [3][3][C][4F], w_h float [3][3][F][4F], bias
 * float [4F] (gate blocks j|i|f|o); isg_convlstm_pack_weights turns them into the MFMA operand
 * images the kernels read (call it again whenever the weights change).
 */

/* One-hot history inputs.  The reference concatenates `hist` one-hot frames to x as extra channels
 * (reference polyrnn/History2LargeExperiment.py:25).  A one-hot channel through a 3x3 convolution is a
 * 9-tap weight scatter, so the kernels can take those channels as vertex points instead of dense
 * frames: w_x then has C + hist input channels, of which the first C go through the GEMM (x holds
 * only those) and the last `hist` are applied from `hist_table` around each frame's point.  Pass
 * hist = 0 to treat every channel densely.
 */

/* Elements (of `dtype`) of a packed image: kind 0 = forward image, 1 = data-gradient image. */
long long isg_convlstm_packed_elems(int Cs, int F, int kind);

/* w_x float [3][3][C + hist][4F].  hist_table: float [9][hist][F][4] out (NULL when hist == 0). */
int isg_convlstm_pack_weights(const float* w_x, const float* w_h, int C, int Cs, int hist, int F, int dtype,
                              void* packed_fwd, void* packed_dgrad /* either may be NULL */, float* hist_table,
                              void* stream);

/* Forward step.  h_prev and c_prev may both be NULL: the zero state (nothing is read for it; the first step of a
 * sequence that starts from dynamic_rnn's default state).  h_out/c_out: new state [B][H][W][F].  out (optional): the emitted output, written
 * at out + b*out_batch_stride + (y*W + x)*F so that it can land in a [B][T][H][W][F] tensor.
 * gates (optional): what the backward pass keeps of this step, in the layout `save_layout` names:
 *   ISG_SAVE_GATES    [B][H][W][F][4] post-activation (tanh j, sig i, sig f, sig o), for
 *                     isg_convlstm_pointwise_bwd (which also reads c_prev and c_out);
 *   ISG_SAVE_FACTORS  bf16 only, same size: [B][H][W][F/2][4][2], per channel pair A0 A1 B0 B1 f0 f1 o0 o1 with
 *                     A = i(1 - j^2), B = j i(1 - i), for isg_convlstm_bwd_fused.  The gate backward is linear
 *                     in (dh, dc): dct = dc + dh E, dy = (dct A, dct B, dct C, dh D), dc_prev = dct f, where
 *                     C = c_prev f(1 - f), D = h(1 - o) and E = o - h^2/o (h = this step's h_out) need neither
 *                     a hyperbolic tangent nor the new cell state: that kernel reads c_prev and h_out instead
 *                     of c_prev and c_out (224 instead of 256 bytes per pixel).
 * hist_points: int32 (x, y) pairs of this step's frames, sample b at hist_points + b*hist_batch_stride,
 * x < 0 = empty frame. */
enum { ISG_SAVE_GATES = 0, ISG_SAVE_FACTORS = 1 };
/* save_layout | ISG_STATE_BF16: c_prev and c_out hold bf16 instead of float (the arithmetic stays float; the new
 * cell state is rounded once, when it is stored): 64 fewer bytes per pixel of a step that runs at the memory
 * system's pace.  Only where isg_convlstm_state_bf16_supported(...) returns 1 (bf16 activations, Cs 16, 32 or 48:
 * the shapes the ping-pong kernels take).  The reference keeps the cell state in float
 * (convLSTM.py:42); what this costs in accuracy is bounded in tests/ (bf16 tolerance, 200-step training curve).
 * isg_convlstm_bwd_fused takes such a c_prev with mode | ISG_CPREV_BF16. */
enum { ISG_STATE_BF16 = 0x10 };
int isg_convlstm_state_bf16_supported(int Cs, int F, int dtype, int B, int H, int W);
int isg_convlstm_cell_fwd(const void* x, const void* h_prev, const void* c_prev, const void* packed_fwd,
                          const float* bias, int B, int H, int W, int Cs, int F, int dtype,
                          const int32_t* durations, int t, const int32_t* hist_points, int hist,
                          long long hist_batch_stride, const float* hist_table, void* h_out, void* c_out, void* out,
                          long long out_batch_stride, void* gates, int save_layout, void* stream);

/* The same step with its two activation outputs as channel slices of wider arrays: pixel p of h_out is written at
 * h_out + p * h_pixel_stride elements and pixel p of a sample of `out` at out + sample * out_batch_stride +
 * p * out_pixel_stride (strides multiples of 8 elements, at least F; F = dense, which is isg_convlstm_cell_fwd).
 * For callers that would otherwise copy the step's h into a joint buffer afterwards: a cell of 16 k filters run as k
 * groups writes each group's h straight into the next step's [x | h] input and into the joint [.., 16 k] output
 * (reference convLSTM.py:33, the concat of x and h, and polyrl/a3c/model.py:31-36, whose cell has 32 filters).
 * The strided destination must not overlap anything the launch reads (x, h_prev, c_prev). */
int isg_convlstm_cell_fwd_strided(const void* x, const void* h_prev, const void* c_prev, const void* packed_fwd,
                                  const float* bias, int B, int H, int W, int Cs, int F, int dtype,
                                  const int32_t* durations, int t, const int32_t* hist_points, int hist,
                                  long long hist_batch_stride, const float* hist_table, void* h_out,
                                  int h_pixel_stride, void* c_out, void* out, long long out_batch_stride,
                                  int out_pixel_stride, void* gates, int save_layout, void* stream);

/* The same step for a caller that only wants the emitted output AVERAGED over pool x pool windows (the head on
 * a coarser grid, see isg_avgpool_fwd): pool_out `dtype` [H/pool][W/pool][F] per sample, samples
 * pool_batch_stride elements apart, zeros for a masked sample; the full-resolution output is never written.
 * No counterpart in the reference (its head reads the full grid, polyrnn/History2LargeExperiment.py:48-50).
 * Only where isg_convlstm_pooled_fwd_supported(...) returns 1 (bf16, Cs 16 or 32, pool 8, H % 8 == 0,
 * W % 32 == 0); otherwise use isg_convlstm_cell_fwd + isg_avgpool_fwd. */
int isg_convlstm_pooled_fwd_supported(int Cs, int F, int dtype, int B, int H, int W, int pool, int hist);
int isg_convlstm_cell_fwd_pooled(const void* x, const void* h_prev, const void* c_prev, const void* packed_fwd,
                                 const float* bias, int B, int H, int W, int Cs, int F, int dtype,
                                 const int32_t* durations, int t, const int32_t* hist_points, int hist,
                                 long long hist_batch_stride, const float* hist_table, void* h_out, void* c_out,
                                 void* gates, int save_layout, int pool, void* pool_out, long long pool_batch_stride,
                                 void* stream);

/* Backward, step 1 (pointwise): from dL/dh (= dh_above as emitted output + dh_state as carried
 * state, either may be NULL) and dL/dc (dc_state, may be NULL) to the pre-activation gradient
 * dy [B][H][W][F][4] and dc_prev (may alias dc_state).  Replaces the gradient of
 * reference convLSTM.py:40-43 that tf.gradients derives (reference polyrnn/Model.py:108).
 * dh_above_pool = 1: dh_above is [H][W][F] per sample.  dh_above_pool = k > 1: the emitted output went
 * through a k x k average pool (isg_avgpool_fwd) and dh_above is the gradient of the POOLED tensor,
 * [H/k][W/k][F] per sample; each pixel takes 1/k^2 of its window's value, never materialised.
 * dc_dtype = storage type of the carried cell-state gradient (dc_state and dc_prev): ISG_F32, or ISG_BF16
 * to carry it like dL/dh already is in a bf16 run (64 fewer bytes per pixel-step; arithmetic stays float). */
int isg_convlstm_pointwise_bwd(const void* gates, const float* c_prev, const float* c_cur, const void* dh_above,
                               long long dh_above_batch_stride, int dh_above_pool, const void* dh_state,
                               const void* dc_state, int dc_dtype, int B, int H, int W, int F, int dtype,
                               const int32_t* durations, int t, void* dy, void* dc_prev, void* stream);

/* Backward, step 2 (data gradient of reference convLSTM.py:36-37): dx (optional; first
 * dx_channels channels, pixel stride dx_channel_stride, += when dx_accumulate) and dh_prev
 * [B][H][W][F].  For a masked sample dh_prev is left as it is (it already holds the carried
 * state gradient) and dx is zero. */
int isg_convlstm_dgrad(const void* dy, const void* packed_dgrad, int B, int H, int W, int Cs, int F, int dtype,
                       const int32_t* durations, int t, void* dx, int dx_channels, int dx_channel_stride,
                       int dx_accumulate, void* dh_prev, void* stream);

/* Backward, step 3 (parameter gradients): dw_x [3][3][C + hist][4F], dw_h [3][3][F][4F], dbias [4F],
 * all float.  workspace: device scratch of isg_convlstm_wgrad_workspace_bytes(Cs, F) bytes holding one
 * partial sum per workgroup; needs Cs + F <= 64.  mode:
 *   ISG_WGRAD_ASSIGN       dw = this step's gradient
 *   ISG_WGRAD_ADD          dw += this step's gradient
 *   ISG_WGRAD_DEFER_FIRST  the partial sums stay in the workspace (overwriting it); dw_h, dbias and the C
 *                          GEMM channels of dw_x are not touched; the hist channels of dw_x are assigned
 *   ISG_WGRAD_DEFER        as above but added to the workspace / to the hist channels of dw_x
 * After a run of deferred calls with one workspace (same B, H, W, Cs), isg_convlstm_wgrad_finish folds
 * the partial sums into dw (accumulate != 0: adds) - one reduction per sequence instead of one per step. */
enum { ISG_WGRAD_ASSIGN = 0, ISG_WGRAD_ADD = 1, ISG_WGRAD_DEFER_FIRST = 2, ISG_WGRAD_DEFER = 3 };
long long isg_convlstm_wgrad_workspace_bytes(int Cs, int F);
int isg_convlstm_wgrad(const void* x, const void* h_prev, const void* dy, int B, int H, int W, int C, int Cs, int F,
                       int dtype, const int32_t* durations, int t, const int32_t* hist_points, int hist,
                       long long hist_batch_stride, void* workspace, int mode, float* dw_x, float* dw_h,
                       float* dbias, void* stream);
int isg_convlstm_wgrad_finish(const void* workspace, int B, int H, int W, int C, int Cs, int hist, int F,
                              int accumulate, float* dw_x, float* dw_h, float* dbias, void* stream);

/* Backward, all three steps in ONE launch (bf16): the pre-activation gradient dy is computed per tile, halo
 * included, from the backward factors the forward step saved (`factors`: the ISG_SAVE_FACTORS layout of
 * isg_convlstm_cell_fwd), the cell state before the step (c_prev), the step's output state (h_cur = its h_out)
 * and the incoming gradients, lives only in on-chip memory, and
 * feeds the data gradient and the weight gradient there - it is never written to or read from device memory
 * (384 fewer bytes per pixel than isg_convlstm_pointwise_bwd + _dgrad + _wgrad).  Same results as those three
 * calls up to the bf16 rounding of the factors and the order of float additions.  Replaces the gradient of reference convLSTM.py:36-43 that
 * tf.gradients derives (reference polyrnn/Model.py:108).
 *   Only where isg_convlstm_bwd_fused_supported(...) returns 1: bf16, F = 16, Cs = 16, 32 or 48, hist <= 2; dx (when
 *   given) in runs of 8 channels: dx_channels and dx_channel_stride multiples of 8.
 *   dc_state / dc_prev are bf16 and must be different buffers, as must dh_state / dh_prev (a tile recomputes its
 *   neighbours' border pixels from the incoming gradients); for a masked sample both pass through as copies
 *   and dx is zero (left alone when accumulating).
 *   workspace / mode: as isg_convlstm_wgrad with ISG_WGRAD_DEFER_FIRST (overwrite) or ISG_WGRAD_DEFER (add);
 *   close the sequence with isg_convlstm_wgrad_finish.  With hist > 0 the gradient of the history taps goes
 *   to the hist channels of dw_x (assigned / added by the same mode) through hist_scratch, device scratch of
 *   isg_convlstm_bwd_fused_hist_bytes(B, hist) bytes.  Over a sequence the zero fill and the fold of that
 *   scratch can be done once instead of per step: scratch of isg_convlstm_bwd_fused_hist_bytes(B * T, hist)
 *   bytes, isg_convlstm_bwd_fused_hist_begin(scratch, B * T, hist) before the first step, every step with
 *   mode | ISG_HIST_DEFER and hist_scratch = scratch + t * B * hist * 9 * 64 * 2 bytes (a slot of its own per
 *   step; dw_x is then not touched by the step), and isg_convlstm_bwd_fused_hist_finish(scratch, B * T, hist, C,
 *   accumulate, dw_x) after the last one (assigns the hist channels of dw_x, or adds to them).
 *   dh_above must be given (zeros where there is no gradient); dh_state and dc_state may both be NULL: no
 *   gradient is carried in (the last step of a sequence whose final state is not differentiated; nothing is
 *   read for them).  dh_above_pool is a power of two.  c_prev and h_prev may both be NULL: the step out of the zero state (nothing is read for
 *   them; the rows of dw_h and of the h part get zero contributions).  dc_prev and dh_prev may both be NULL:
 *   the gradient of the state before this step is not wanted (nothing is written, and where the kernel has a
 *   row tile of its own for dh_prev it is skipped).  dx is written in 16-byte pieces: dx_channels and
 *   dx_channel_stride are multiples of 8 and dx is 16-byte aligned.  Other arguments: as the three calls it
 *   replaces. */
int isg_convlstm_bwd_fused_supported(int Cs, int F, int dtype, int B, int H, int W, int hist);
long long isg_convlstm_bwd_fused_hist_bytes(int B, int hist);
enum { ISG_HIST_DEFER = 0x100, ISG_CPREV_BF16 = 0x200 };  /* mode | ISG_CPREV_BF16: c_prev holds bf16 (see ISG_STATE_BF16) */
int isg_convlstm_bwd_fused_hist_begin(void* hist_scratch, int samples, int hist, void* stream);
int isg_convlstm_bwd_fused_hist_finish(void* hist_scratch, int samples, int hist, int C, int accumulate, float* dw_x,
                                       void* stream);
int isg_convlstm_bwd_fused(const void* factors, const void* c_prev, const void* h_cur, const void* dh_above,
                           long long dh_above_batch_stride, int dh_above_pool, const void* dh_state, const void* dc_state, const void* x,
                           const void* h_prev, const void* packed_dgrad, int B, int H, int W, int C, int Cs, int F,
                           int dtype, const int32_t* durations, int t,
                           const int32_t* hist_points, int hist, long long hist_batch_stride, void* hist_scratch,
                           void* dc_prev, void* dh_prev, void* dx, int dx_channels, int dx_channel_stride,
                           int dx_accumulate, void* workspace, int mode, float* dw_x, void* stream);
/* isg_convlstm_bwd_fused without history taps (hist = 0) for Cs = 48 - one of the two 16-filter groups a 32-filter
 * cell (reference polyrl/a3c/model.py:21-23) is run as, which share the joint input [features | the other group's h] -
 * with h_cur and dh_state read as 16-channel slices of wider pixels, h_cur_pixel_stride / dh_state_pixel_stride
 * elements apart (multiples of 8 in [16, 64]): the group's h where the forward step wrote it, inside the next step's
 * joint input, and its carried gradient where the other group's backward step wrote it, inside dx - no copies into
 * arrays of their own.  Likewise dh_above, dh_above_pixel_stride elements apart: the group's 16 channels of the
 * gradient of the cell's joint output (pooling needs dh_above_pixel_stride = 16). */
int isg_convlstm_bwd_fused_strided(const void* factors, const void* c_prev, const void* h_cur, int h_cur_pixel_stride,
                                   const void* dh_above, long long dh_above_batch_stride, int dh_above_pixel_stride,
                                   int dh_above_pool, const void* dh_state, int dh_state_pixel_stride, const void* dc_state, const void* x,
                                   const void* h_prev, const void* packed_dgrad, int B, int H, int W, int C, int Cs,
                                   int F, int dtype, const int32_t* durations, int t, void* dc_prev, void* dh_prev,
                                   void* dx, int dx_channels, int dx_channel_stride, int dx_accumulate, void* workspace,
                                   int mode, void* stream);

/* ---- vertex decoding over the class grid ----------------------------------------------------
 * logits: `dtype` [rows][cols] (cols = P*P or P*P + 1).  One wave per row.
 *
 * isg_softmax_xent_fwd replaces tf.nn.sparse_softmax_cross_entropy_with_logits (reference
 * polyrnn/Model.py:98-99; weighted by td_target in polyrl/ReinforceOverfitExperiment.py:47-48) and, in
 * the same pass, tf.argmax(logits, axis=2) with its first-maximum rule (reference polyrnn/Model.py:44).
 *   labels int32 [rows]; weights float [rows] or NULL; any of loss_rows / lse_rows / argmax_rows
 *   (float, float, int32 [rows]) may be NULL.  loss_rows[r] = (lse - logit[label]) * weights[r].
 * isg_softmax_xent_bwd: dlogits[r][c] = (softmax[r][c] - [c == label[r]]) * row_scale[r] * scale
 *   (row_scale may be NULL) - the gradient tf.gradients derives for the loss above.
 * isg_sample_rows replaces tf.multinomial(logits - max, 1) (reference
 *   polyrl/ReinforceOverfitExperiment.py:101-103): one categorical draw per row by the Gumbel-max
 *   rule from a counter-based generator keyed by (seed, row_offset + row, column).  Same
 *   distribution as the reference, not the same random stream.
 */
int isg_softmax_xent_fwd(const void* logits, const int32_t* labels, const float* weights, int rows, int cols,
                         int dtype, float* loss_rows, float* lse_rows, int32_t* argmax_rows, void* stream);
int isg_softmax_xent_bwd(const void* logits, const int32_t* labels, const float* lse_rows, const float* row_scale,
                         float scale, int rows, int cols, int dtype, void* dlogits, void* stream);

/* The gradient of the MEAN of the per-row losses (reference polyrnn/Model.py:98-100, tf.reduce_mean over every step
 * of every sequence): dlogits[r][c] = (softmax[r][c] - [c == label[r]]) * *upstream / rows, the gradient that arrives
 * at the mean read from device memory (a float), so that no pass spreads it over the rows first. */
int isg_softmax_xent_mean_bwd(const void* logits, const int32_t* labels, const float* lse_rows, const float* upstream,
                              int rows, int cols, int dtype, void* dlogits, void* stream);

/* isg_softmax_xent_fwd with the labels given as grid positions: label[r] = targets_xy[r][0] * grid_side +
 * targets_xy[r][1] (x-major, reference polyrnn/Model.py:39; the end token (grid_side, 0) is class grid_side^2);
 * labels_out (may be NULL) receives the labels, which the backward pass reads. */
int isg_sequence_xent_fwd(const void* logits, const int32_t* targets_xy, int rows, int cols, int grid_side, int dtype,
                          float* loss_rows, float* lse_rows, int32_t* argmax_rows, int32_t* labels_out, void* stream);
int isg_argmax_rows(const void* logits, int rows, int cols, int dtype, int32_t* out_idx, void* stream);
int isg_sample_rows(const void* logits, int rows, int cols, int dtype, unsigned long long seed,
                    unsigned long long row_offset, int32_t* out_idx, void* stream);
/* The same draw delivered as an action: out_xy int32 [rows][2] = (class / P, class % P), the pair the
 * reference builds from the sampled class (polyrl/ReinforceOverfitExperiment.py:103-105); the end token
 * (class P*P) comes out as (P, 0).  row_offset_dev (device, may be NULL) is added to row_offset when the
 * kernel runs, so a captured HIP graph of a rollout draws fresh numbers on every replay once its owner
 * advances that word between replays. */
int isg_sample_actions(const void* logits, int rows, int cols, int dtype, unsigned long long seed,
                       unsigned long long row_offset, const unsigned long long* row_offset_dev, int P,
                       int32_t* out_xy, void* stream);

/* The actor-critic loss of reference polyrl/a3c/a3c.py:201-216 over rows = all steps of all rollouts, as per-row
 * contributions (sum them with isg_column_sums): with p = softmax(logits[r]), a = actions[r], w = live[r] (NULL: 1),
 *   terms[r] = w * (-log p[a] * adv[r],  0.5 * (vf[r] - ret[r])^2,  -sum_k p_k log p_k)       float [rows][3]
 *   stats[r] = (log-sum-exp, entropy)                                                         float [rows][2]
 * total loss = sum terms[.,0] + 0.5 * sum terms[.,1] - 0.01 * sum terms[.,2].
 * isg_a3c_loss_bwd: from upstream3 = d loss / d (the three sums), DEVICE float [3], to dlogits `dtype` [rows][cols]
 * and dvf float [rows]; logits `dtype` (float or bf16), vf / adv / ret float, actions int32, live uint8. */
int isg_a3c_loss_fwd(const void* logits, const float* vf, const int32_t* actions, const float* adv, const float* ret,
                     const unsigned char* live, int rows, int cols, int dtype, float* terms, float* stats, void* stream);
int isg_a3c_loss_bwd(const void* logits, const float* vf, const int32_t* actions, const float* adv, const float* ret,
                     const unsigned char* live, const float* stats, const float* upstream3, int rows, int cols, int dtype,
                     void* dlogits, float* dvf, void* stream);

/* out[c] = sum over rows of x[r][c] (float, overwritten; a fixed order of additions): the bias gradient of the dense
 * head, tf.gradients of the bias add in reference polyrnn/History2LargeExperiment.py:48-50, whose rows are every
 * (sample, step) of a batch - tens of thousands in a policy update.  x `dtype` [rows][cols];
 * workspace: device scratch of isg_column_sums_workspace_bytes(cols) bytes. */
long long isg_column_sums_workspace_bytes(int cols);
int isg_column_sums(const void* x, long long rows, int cols, int dtype, float* out, void* workspace, void* stream);

/* What the reference's decode loop starts from (polyrnn/Model.py:218-230), for every sample in one launch: with n =
 * num_true[r] and frame i of the first step showing true vertex (n - hist + i) mod n (create_history with end_idx =
 * len(v) - 1), first_points = history_points = those hist points, close_xy = frame 0's point (the closing vertex,
 * Model.py:240; hist = 0: true vertex 0), required = 1, durations = 0, predicted = 0.
 * true_vertices int32 [B][vmax][2], num_true int32 [B]; outputs as isg_decode_step takes them. */
int isg_decode_begin(const int32_t* true_vertices, const int32_t* num_true, int B, int vmax, int hist, int T,
                     int32_t* close_xy, int32_t* first_points, int32_t* history_points, int32_t* required,
                     int32_t* durations, int32_t* predicted, void* stream);

/* The decoded polygons as the rasteriser takes them (polyrnn/Model.py:252-260): polygons int32 [B][vmax][2] = the
 * first_points (when prepend_history) followed by predicted * scale, zero beyond; counts[r] = (durations[r] when
 * slice_by_duration, else T) + the prepended points. */
int isg_decode_polygons(const int32_t* predicted, const int32_t* durations, const int32_t* first_points, int B, int T,
                        int hist, int prepend_history, int slice_by_duration, int scale, int vmax, int32_t* polygons,
                        int32_t* counts, void* stream);

/* out `dtype` [rows][cols] = sum over s < slices of part[s][rows][cols] (float) + bias[cols] (`dtype`): closes a dense
 * layer whose reduction axis was cut into `slices` batched GEMMs with float results - the decode loop's head at a
 * handful of rows (polyrnn/Model.py:231-249 runs it once per vertex), where one GEMM has too few output tiles. */
int isg_sum_slices_bias(const float* part, int slices, int rows, int cols, const void* bias, int dtype, void* out,
                        void* stream);
/* isg_sum_slices_bias over partial results of part_cols >= cols columns each ([slices][rows][part_cols]; the first cols
 * are kept): the batched GEMMs ran on a weight matrix padded with zero columns to a multiple of 8, so that its bf16
 * rows start on 16-byte boundaries (785 = 28 x 28 + 1 classes -> 792). */
int isg_sum_slices_bias_padded(const float* part, int slices, int rows, int cols, int part_cols, const void* bias,
                               int dtype, void* out, void* stream);

/* out float [rows] = x [rows][cols] . w [cols] + bias[0] (bias may be NULL), x, w and bias bf16: a dense layer with one
 * output column - the value head of polyrl/a3c/model.py:39 at an acting step, which as a library GEMM is a
 * matrix-vector product far below the memory rate.  cols a multiple of 8; x and w 16-byte aligned. */
int isg_rows_dot(const void* x, const void* w, const void* bias, int rows, int cols, int dtype, float* out,
                 void* stream);

/* One iteration of the reference's autoregressive decode loop (polyrnn/Model.py:231-249; end-token variant
 * polyrnn/EndTokenExperiment.py:158-170, EndTokenDeepExperiment_2.py:150-170) for every sample, in one launch
 * after the head, with no host round trip.  Per row r of logits [rows][cols]:
 *   class = argmax (first maximum);  (x, y) = (class / P, class % P)  - the end token class P*P is (P, 0);
 *   durations[r] += required[r]                              (steps the sample was still decoding);
 *   without end token: (x, y) == close_xy[r] clears required[r]   (Model.py:239-242: the closing vertex);
 *   with end token:    (x, y) == (P, 0) clears required[r], the predicted vertex becomes close_xy[r] and the
 *                      next history frame shows (0, 0);
 *   predicted[r][step] = the vertex;  history_points[r] rolls by one frame (the last wraps to the front) and
 *   its last frame becomes the vertex's point (Model.py:244-249: np.roll(+1), newest frame overwritten).
 * required int32 [rows] is what the next step's recurrence takes as sequence lengths (1 = still decoding).
 * predicted int32 [rows][T][2], history_points int32 [rows][hist][2], close_xy int32 [rows][2]. */
int isg_decode_step(const void* logits, int rows, int cols, int dtype, int P, int end_token, const int32_t* close_xy,
                    int32_t* required, int32_t* durations, int32_t* predicted, int T, int step,
                    int32_t* history_points, int hist, void* stream);

/* Masked metrics of reference polyrnn/Model.py:44-54 in one launch: pred int32 [B*T] (argmax class),
 * targets int32 [B][T][2] (x, y), durations int32 [B], P = side of the class grid (class = x*P + y;
 * the end token (P, 0) is class P*P).  out4 (device, double): {number of steps with t < duration and
 * pred == class(target); sum over those steps of ||target - (pred / P, pred % P)||_2; the maximum of
 * the same (0 for masked steps); sum of durations}.  accuracy = out4[0] / out4[3],
 * avg_error = out4[1] / out4[3], max_error = out4[2]. */
int isg_sequence_metrics(const int32_t* pred, const int32_t* targets, const int32_t* durations, int B, int T, int P,
                         double* out4, void* stream);

/* The same launch with the step's scalar summaries finished on the device (reference polyrnn/Model.py:44-54 and
 * :98-100: three tf.reduce_mean / divisions): out6 = out4 followed by accuracy = out6[0] / out6[3] and
 * avg_error = out6[1] / out6[3]; *loss_mean (device float) = the mean of loss_rows [B*T] (summed in double, in a
 * fixed order), the un-masked mean the reference takes over every step of every sequence. */
int isg_sequence_summary(const int32_t* pred, const int32_t* targets, const int32_t* durations, const float* loss_rows,
                         int B, int T, int P, double* out6, float* loss_mean, void* stream);

/* ---- optimiser ----------------------------------------------------------------------------------
 * tf.clip_by_global_norm + tf.train.AdamOptimizer as the reference applies them (polyrnn/Model.py:106-112;
 * polyrl/a3c/a3c.py:244-258 with clip 40) over flat float buffers of n values, one pass:
 *   g = grad * (clip_norm / max(*grad_norm, clip_norm))      (grad_norm: device float, the global norm of grad;
 *                                                             NULL or clip_norm <= 0: no clipping)
 *   m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;  param -= lr_t m / (sqrt(v) + eps)
 * lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t) is the caller's (TensorFlow's form: eps is not rescaled).
 * grad is not modified.  All four buffers 16-byte aligned. */
int isg_adam_tf(float* param, float* m, float* v, const float* grad, long long n, double lr_t, double beta1,
                double beta2, double eps, const float* grad_norm, double clip_norm, void* stream);

/* Everything the kernels read that is a re-arrangement (and narrowing) of parameter values - the packed operand
 * images of isg_convlstm_pack_weights / isg_conv32_pack / isg_conv_in_pack, tap tables, compute-type copies of dense
 * weights - refreshed from the flat float parameter buffer in ONE launch after an optimiser step, instead of one
 * packing launch per image (the reference has no counterpart: TensorFlow reads its variables in place,
 * polyrnn/Model.py:106-112 only updates them).  out_bf16[i] = bf16(src[map_bf16[i]]), out_f32[i] = src[map_f32[i]]
 * (a negative map entry gives zero; the caller derives the maps once, e.g. by packing a buffer of index values);
 * range r converts src[range_src[r] .. + range_n[r]) to bf16 at range_out[r] (n_ranges <= 4).  The maps are device
 * int32 arrays, the range descriptions host arrays.  Every map entry must be < n_src (not checked on the device). */
int isg_refresh_derived(const float* src, long long n_src, const int32_t* map_bf16, long long n_bf16, void* out_bf16,
                        const int32_t* map_f32, long long n_f32, float* out_f32, int n_ranges, const long long* range_src,
                        const long long* range_n, void* const* range_out, void* stream);

/* ---- head pooling ----------------------------------------------------------------------------
 * Mean over non-overlapping k x k windows of x [N][H][W][F] -> out [N][H/k][W/k][F], same dtype,
 * float accumulation.  The reference feeds the recurrent output straight into
 * dense(P*P*16 -> P*P) (polyrnn/History2LargeExperiment.py:48-50), which has 4e10 weights at P = 224; this
 * pool puts the head on a coarser grid and has no counterpart in the reference.  k a power of two,
 * F a multiple of 8 (bf16) / 4 (float) with k * F / that <= 64.  Its gradient is folded into
 * isg_convlstm_pointwise_bwd (dh_above_pool). */
int isg_avgpool_fwd(const void* x, long long N, int H, int W, int F, int k, int dtype, void* out, void* stream);

/* ---- encoder pointwise halves ------------------------------------------------------------------
 * The bias + ReLU of tf.layers.conv2d(..., activation=tf.nn.relu) (reference
 * polyrnn/History2LargeExperiment.py:15-22) around a library convolution, NHWC, n pixels of C channels
 * (C a multiple of 8 for bf16 / 4 for float, C / that a divisor of 256).
 * isg_bias_relu:      y = max(y + bias, 0) in place; bias float [C].
 * isg_relu_bwd_bias:  dz = dy where y > 0 else 0 (dz may alias dy); dbias[c] = sum of dz[., c] (float,
 *                     overwritten, deterministic two-stage sum).  workspace: device scratch of
 *                     isg_relu_bwd_bias_workspace_bytes(C) bytes. */
int isg_bias_relu(void* y, const float* bias, long long n, int C, int dtype, void* stream);
long long isg_relu_bwd_bias_workspace_bytes(int C);
int isg_relu_bwd_bias(const void* dy, const void* y, long long n, int C, int dtype, void* dz, float* dbias,
                      void* workspace, void* stream);
/* isg_relu_bwd_bias with dy and y as C-channel slices of wider pixels, dy_pixel_stride / y_pixel_stride elements apart
 * (multiples of the vector width, at least C): the backward of a layer whose output was written by
 * isg_conv32_strided, on a gradient that lies inside the gradient of that wider array.  dz is contiguous. */
int isg_relu_bwd_bias_strided(const void* dy, long long dy_pixel_stride, const void* y, long long y_pixel_stride,
                              long long n, int C, int dtype, void* dz, float* dbias, void* workspace, void* stream);
/* isg_relu_bwd_bias without the workspace argument and without any allocation: the per-workgroup partial
 * sums live in an array built into the library (256 KiB per GPU, held until the process ends), whose
 * address is looked up on the first call on each device.  No counterpart in the reference; it is for
 * callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  At most ISG_RELU_BWD_BIAS_STATIC_MAX_C channels (more: ISG_EINVAL);
 * results are identical to isg_relu_bwd_bias.  All calls on one GPU share the array: calls that overlap
 * on one GPU (different streams, different threads) race; calls queued on one stream do not.
 * The array used is the CURRENT device's copy: `stream` must belong to the current device (checked:
 * ISG_EINVAL otherwise).  The address lookup of the first call on a device (hipGetSymbolAddress under a
 * lock) must not happen inside a stream capture: call isg_relu_bwd_bias_static_init() once per device,
 * with that device current, before capturing (it does only that lookup and launches nothing). */
#define ISG_RELU_BWD_BIAS_STATIC_MAX_C 64
int isg_relu_bwd_bias_static_init(void);
int isg_relu_bwd_bias_static(const void* dy, const void* y, long long n, int C, int dtype, void* dz, float* dbias,
                             void* stream);

/* A channel slice of an NHWC array copied into a channel slice of another: slice_bytes bytes of each pixel
 * (i0, i1, q), i0 < n0, i1 < n1, q < npix, from src + i0 s[0] + i1 s[1] + q s[2] (s = src_strides, HOST array of 3,
 * bytes) to the same place by dst_strides; every stride, the slice and both addresses multiples of 16, the two ranges
 * must not overlap.  What a cell of 16 k filters run as k groups needs between steps (a group's h joined into the
 * next step's [x | h] input: reference convLSTM.py:33, polyrl/a3c/model.py:31-36) and around a sequence (batch-major
 * features into the time-major joint input and back: the two outer strides) as one launch of 16-byte moves. */
int isg_copy_channel_slice(const void* src, const long long* src_strides, void* dst, const long long* dst_strides,
                           int slice_bytes, long long n0, long long n1, long long npix, void* stream);

/* x `dtype` (float or bf16) [npix][cin] -> y bf16 [npix][cout], channels cin.. zero: conversion and padding of an
 * observation (image + history frames, reference polyrl/a3c/model.py:19-20) in one pass, for a first encoder layer
 * that runs on the 32 -> 32 (16 -> 16) kernels with zero weight rows (isg_conv32_pack_padded).  cout a multiple of 8. */
int isg_pad_channels(const void* x, int dtype, int cin, void* y, int cout, long long npix, void* stream);

/* ---- the encoder's first convolution -------------------------------------------------------------
 * KS x KS taps over a 3-channel image -> cout channels, TensorFlow SAME padding ((KS-1)/2 before, the rest
 * after), + bias (+ ReLU): the first tf.layers.conv2d of reference polyrnn/RNNConvExperiment.py (14 x 14 -> 32)
 * and of polyrnn/History2LargeExperiment.py:15-17 (5 x 5 -> 16) as one MFMA kernel whose reduction index is
 * (ky, kx, channel padded to 4).  bf16; KS = 14 or 5, cout = 32 or 16.
 *   isg_conv_in_pad_image:   x3 bf16 [npix][3] -> x4 bf16 [npix][4] (fourth channel 0).
 *   wimg: bf16 [KS * KXP/4][64][8] with KXP = KS rounded up to 4; element [ky * KXP/4 + s][32 kg + m][j] =
 *         w[ky][4 s + 2 kg + j / 4][j % 4][m] (HWIO), 0 where kx >= KS, the channel is 3 or m >= cout;
 *         isg_conv_in_packed_elems(KS, cout) gives its length (-1: no such instantiation).
 *   isg_conv_in_pack:  that image from float HWIO weights w [KS][KS][3][cout], one launch.
 *   isg_conv_in_fwd:   y bf16 [B][H][W][cout] = act(conv(x4, w) + bias), act = ReLU when relu != 0.
 *   isg_conv_in_wgrad: dw float [KS][KS][3][cout] (overwritten) = gradient of the convolution for
 *         dz bf16 [B][H][W][cout] (the gradient after the ReLU mask, see isg_relu_bwd_bias); workspace:
 *         device scratch of isg_conv_in_wgrad_workspace_bytes(KS, cout) bytes. */
long long isg_conv_in_packed_elems(int KS, int cout);
int isg_conv_in_pack(const float* w, int KS, int cout, void* wimg, void* stream);
int isg_conv_in_pad_image(const void* x3, long long npix, void* x4, void* stream);
int isg_conv_in_fwd(const void* x4, const void* wimg, const float* bias, int B, int H, int W, int KS, int cout, int relu,
                    void* y, void* stream);
long long isg_conv_in_wgrad_workspace_bytes(int KS, int cout);
int isg_conv_in_wgrad(const void* x4, const void* dz, int B, int H, int W, int KS, int cout, void* workspace, float* dw,
                      void* stream);

/* ---- the encoder's 32 -> 32 channel convolutions ----------------------------------------------------
 * KS x KS taps (KS = 3 or 5), 32 channels in and out, SAME padding, bf16 with float accumulation: the later
 * tf.layers.conv2d layers of reference polyrnn/RNNConvExperiment.py.
 *   isg_conv32: y [B][H][W][32] = act(sum over taps and k of A * x + bias); bias may be NULL; act = ReLU when
 *       relu != 0.  wimg: bf16 [KS*KS*2][64][8], element [2 tap + q][32 kg + m][j] = A[tap][m][16 q + 8 kg + j].
 *       Forward: A[tap][m][k] = w[ty][tx][k][m] (HWIO).  Data gradient: x = the output gradient,
 *       A[tap][m][k] = w[KS-1-ty][KS-1-tx][m][k], no bias, no ReLU.
 *   isg_conv32_wgrad: dw float [KS][KS][32][32] (HWIO, overwritten) for input x and output gradient dz
 *       (after the ReLU mask); workspace of isg_conv32_wgrad_workspace_bytes(KS) bytes. */
long long isg_conv32_packed_elems(int KS);
/* The image of isg_conv32 / isg_conv16 (C = 32 / 16: [KS*KS*C/16][64][8], element [tap * C/16 + q][32 kg + m][j], rows
 * m >= C zero) from float HWIO weights w [KS][KS][C][C], forward or (data_gradient != 0) data gradient; one launch. */
int isg_conv32_pack(const float* w, int KS, int C, int data_gradient, void* wimg, void* stream);
/* The forward image of a layer with cin < C input channels run on isg_conv32 / isg_conv16 with its input padded to C
 * channels by zeros (reference polyrl/a3c/model.py:19-20: the first of four 3 x 3, 32-filter layers sees the 3 image
 * channels + the history frames): w float [KS][KS][cin][C], the image's k >= cin entries are zero. */
int isg_conv32_pack_padded(const float* w, int KS, int C, int cin, void* wimg, void* stream);
int isg_conv32(const void* x, const void* wimg, const float* bias, int B, int H, int W, int KS, int relu, void* y,
               void* stream);
/* isg_conv32 whose output pixels lie y_pixel_stride elements apart (a multiple of 8, at least 32): y is a
 * 32-channel slice of wider pixels - the encoder's top layer writing its features straight into the joint
 * [features | h] input of the recurrent cell (reference polyrl/a3c/model.py:19-23) instead of being copied there. */
int isg_conv32_strided(const void* x, const void* wimg, const float* bias, int B, int H, int W, int KS, int relu,
                       void* y, long long y_pixel_stride, void* stream);
long long isg_conv32_wgrad_workspace_bytes(int KS);
int isg_conv32_wgrad(const void* x, const void* dz, int B, int H, int W, int KS, void* workspace, float* dw,
                     void* stream);
/* The data gradient of a layer whose input is the ReLU output of the layer below, with that layer's ReLU
 * backward in the epilogue: y = conv(x, wimg) where mask > 0, else 0 (mask = the lower layer's output, same
 * shape as y), and dbias float [32] = the sum of y over all pixels - the lower layer's bias gradient.  Replaces
 * isg_conv32(relu = 0) followed by isg_relu_bwd_bias: one pass over the tensor less.  Part of the gradient
 * tf.gradients derives for the encoder (reference polyrnn/Model.py:108).  workspace: device scratch of
 * isg_conv32_masked_workspace_bytes() bytes. */
long long isg_conv32_masked_workspace_bytes(void);
int isg_conv32_masked(const void* x, const void* wimg, const void* mask, int B, int H, int W, int KS, void* y,
                      float* dbias, void* workspace, void* stream);

/* ---- the same for 16 -> 16 channels ---------------------------------------------------------------------
 * The 16-channel encoder layers of reference polyrnn/History2LargeExperiment.py:15-22 (5 x 5 and 3 x 3) and of
 * polyrnn/EndTokenDeepExperiment_2.py.  Same kernels with one 16-channel k-step per tap:
 * wimg bf16 [KS*KS][64][8], element [tap][32 kg + m][j] = A[tap][m][8 kg + j] with rows m >= 16 zero;
 * x, y, dz [B][H][W][16]; dw float [KS][KS][16][16]. */
long long isg_conv16_packed_elems(int KS);
int isg_conv16(const void* x, const void* wimg, const float* bias, int B, int H, int W, int KS, int relu, void* y,
               void* stream);
int isg_conv16_masked(const void* x, const void* wimg, const void* mask, int B, int H, int W, int KS, void* y,
                      float* dbias /* [16] */, void* workspace, void* stream);
long long isg_conv16_wgrad_workspace_bytes(int KS);
int isg_conv16_wgrad(const void* x, const void* dz, int B, int H, int W, int KS, void* workspace, float* dw,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISG_H */
