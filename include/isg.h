/* libisg - C ABI of the MI355X (gfx950) hot-path kernels for Shadowen/ImageSegmentation.
 *
 * The reference is pure Python (TensorFlow 1.x graph ops + scikit-image); it has no FFI of its
 * own, so each entry point below names the reference call sequence it replaces.  A reference-side
 * binding is a ctypes stub.
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

#ifdef __cplusplus
}
#endif
#endif /* ISG_H */
