/* CPU oracle for the polygon rasteriser + IoU counts, in plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the shipped package links or loads this file.  It is the checker used by
 * tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py.
 *
 * It restates, literally and in double precision as the original does,
 *   - skimage/_shared/geometry.pxd  point_in_polygon   (scikit-image 0.18.3; third-party,
 *     not vendored by the reference; reached through reference util.py:63),
 *   - skimage/draw/_draw.pyx        _polygon           (bounding-box scan, same version),
 *   - reference util.py:60-65       create_shape_mask,
 *   - reference polyrl/polygon_env.py:48-53 _calculate_iou (the two counts).
 *
 * Pinned by tests/test_oracle_raster.py against fixtures made by running the real reference
 * (tests/golden/make_raster_fixtures.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* 0 outside, 1 inside, 2 vertex, 3 edge */
static unsigned char point_in_polygon(int nr_verts, const double *xp, const double *yp, double x, double y) {
    unsigned int l_cross = 0, r_cross = 0;
    const double eps = 1e-12;
    double x1 = xp[nr_verts - 1] - x;
    double y1 = yp[nr_verts - 1] - y;
    for (int i = 0; i < nr_verts; ++i) {
        double x0 = xp[i] - x;
        double y0 = yp[i] - y;
        if ((-eps < x0 && x0 < eps) && (-eps < y0 && y0 < eps)) return 2;
        if ((y0 > 0) != (y1 > 0)) {
            if (((x0 * y1 - x1 * y0) / (y1 - y0)) > 0) r_cross += 1;
        }
        if ((y0 < 0) != (y1 < 0)) {
            if (((x0 * y1 - x1 * y0) / (y1 - y0)) < 0) l_cross += 1;
        }
        x1 = x0;
        y1 = y0;
    }
    if ((r_cross & 1) != (l_cross & 1)) return 3;
    if (r_cross & 1) return 1;
    return 0;
}

/* verts: nv pairs (x, y).  mask: size*size bytes, row-major [y][x], written 0/1.
 * Returns the number of covered pixels that fall outside the image (the reference raises
 * IndexError for those at util.py:64); such pixels are skipped. */
int isg_oracle_shape_mask(const int32_t *verts, int nv, int size, uint8_t *mask) {
    memset(mask, 0, (size_t)size * size);
    if (nv <= 0) return 0;
    double xp[64], yp[64];
    double *xh = xp, *yh = yp;
    if (nv > 64) {
        xh = (double *)malloc(sizeof(double) * nv * 2);
        yh = xh + nv;
    }
    int minx = verts[0], maxx = verts[0], miny = verts[1], maxy = verts[1];
    for (int i = 0; i < nv; ++i) {
        int x = verts[2 * i], y = verts[2 * i + 1];
        xh[i] = x;
        yh[i] = y;
        if (x < minx) minx = x;
        if (x > maxx) maxx = x;
        if (y < miny) miny = y;
        if (y > maxy) maxy = y;
    }
    int minr = miny > 0 ? miny : 0, minc = minx > 0 ? minx : 0;
    int outside = 0;
    for (int r = minr; r <= maxy; ++r)
        for (int c = minc; c <= maxx; ++c)
            if (point_in_polygon(nv, xh, yh, c, r)) {
                if (r < size && c < size) mask[(size_t)r * size + c] = 1;
                else ++outside;
            }
    if (xh != xp) free(xh);
    return outside;
}

/* n rollouts; vertex arrays are [n][vmax][2] int32 with nverts[n] valid entries each.
 * inter/uni: [n] int32.  mask_pred/mask_true: optional [n][size][size] uint8 (may be NULL).
 * threads <= 1 runs serially. Returns the total count of out-of-image pixels skipped. */
int isg_oracle_raster_iou(const int32_t *verts_pred, const int32_t *nverts_pred, const int32_t *verts_true,
                          const int32_t *nverts_true, int n, int vmax, int size, int32_t *inter, int32_t *uni,
                          uint8_t *mask_pred, uint8_t *mask_true, int threads) {
    int outside_total = 0;
    size_t px = (size_t)size * size;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 1 ? threads : 1) reduction(+ : outside_total)
    for (int i = 0; i < n; ++i) {
        uint8_t *mp = (uint8_t *)malloc(px * 2);
        uint8_t *mt = mp + px;
        outside_total += isg_oracle_shape_mask(verts_pred + (size_t)i * vmax * 2, nverts_pred[i], size, mp);
        outside_total += isg_oracle_shape_mask(verts_true + (size_t)i * vmax * 2, nverts_true[i], size, mt);
        int a = 0, o = 0;
        for (size_t k = 0; k < px; ++k) {
            a += (mp[k] & mt[k]);
            o += (mp[k] | mt[k]);
        }
        inter[i] = a;
        uni[i] = o;
        if (mask_pred) memcpy(mask_pred + (size_t)i * px, mp, px);
        if (mask_true) memcpy(mask_true + (size_t)i * px, mt, px);
        free(mp);
    }
    return outside_total;
}
