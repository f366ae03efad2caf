// Scanline polygon rasteriser + mask-IoU reduction for gfx950.
//
// One lane owns one mask row.  For that row it walks the polygon's edge table (held in LDS) and
// builds the row as bit words: every edge that straddles the row contributes one integer
// threshold, and "all pixels left of the threshold" is XOR-ed into a parity word - the even-odd
// fill of the row is the XOR of those prefixes.  The fill rule is the boundary-inclusive one of
// skimage.draw.polygon (point_in_polygon: inside by crossing parity, or on an edge, or on a
// vertex), which for a test point (x, y) and an edge (X1,Y1)->(X0,Y0) reads
//     r_cross counts edges with (y0>0)!=(y1>0) whose intercept lies right of x,
//     l_cross counts edges with (y0<0)!=(y1<0) whose intercept lies left  of x,
//     filled  = vertex | (r_cross odd) | (l_cross odd)            [edge <=> parities differ]
// All arithmetic is int32 and exact; intersection / union counts come from popcounts of the bit
// rows, and the optional byte masks are expanded from the bit rows with 16-byte stores.
#include "isg_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCoordClamp = 16383;

// bits [0, t) of the 32-pixel word that starts at pixel 32*w
__device__ __forceinline__ uint32_t prefix_word(int t, int w) {
    int k = t - 32 * w;
    return k <= 0 ? 0u : (k >= 32 ? 0xFFFFFFFFu : ((1u << k) - 1u));
}

struct PolyLds {
    const int* v;  // [vmax][2] in LDS
    int nv;
    int minr, maxr, minc, maxc;
};

// min(max(floor(N / D), -1), P) for D > 0, exact.  Outside (-1, P) only the clamp matters; inside, 0 <= N < P * D
// < 2^24 is exact in float, the reciprocal estimate is off by at most one, and an integer remainder check fixes it.
__device__ __forceinline__ int floor_div_clamped(int N, int D, int P) {
    if (N < 0) return -1;
    if ((long long)N >= (long long)P * D) return P;
    int q = (int)((float)N * __builtin_amdgcn_rcpf((float)D));
    int r = N - q * D;
    if (r < 0) --q, r += D;
    if (r >= D) ++q;
    return q;
}

// Bit words of mask row y for one polygon.  Two passes: the first walks every vertex with cheap tests only
// (vertex on the row, which edges straddle the row) and leaves a bit mask of the straddling edges; the second
// visits just those.  Lanes are rows, so a wave pays for the largest straddle count among its rows (2 for a convex
// polygon) instead of for every edge.
template <int WPR>
__device__ __forceinline__ void row_bits(const PolyLds& p, int y, int P, uint32_t (&out)[WPR]) {
    uint32_t R[WPR], L[WPR], V[WPR];
#pragma unroll
    for (int w = 0; w < WPR; ++w) R[w] = L[w] = V[w] = 0u;
    uint32_t lodd = 0u;
    const bool live = p.nv > 0 && y >= p.minr && y <= p.maxr;
    unsigned long long straddle = 0ull;
    if (live) {
        int yp = p.v[2 * (p.nv - 1) + 1];
        for (int i = 0; i < p.nv; ++i) {
            const int xi = p.v[2 * i], yi = p.v[2 * i + 1];
            const int y0 = yi - y, y1 = yp - y;
            if (y0 == 0 && xi >= 0 && xi < P) {
#pragma unroll
                for (int w = 0; w < WPR; ++w)
                    if ((xi >> 5) == w) V[w] |= 1u << (xi & 31);
            }
            if (((y0 > 0) != (y1 > 0)) || ((y0 < 0) != (y1 < 0))) straddle |= 1ull << i;
            yp = yi;
        }
    }
    while (straddle != 0ull) {
        const int i = __ffsll((long long)straddle) - 1;
        straddle &= straddle - 1ull;
        const int j = i == 0 ? p.nv - 1 : i - 1;
        const int xi = p.v[2 * i], yi = p.v[2 * i + 1];
        const int xp = p.v[2 * j], yp = p.v[2 * j + 1];
        const int y0 = yi - y, y1 = yp - y;
        const bool rs = (y0 > 0) != (y1 > 0);
        const bool ls = (y0 < 0) != (y1 < 0);
        int D = y1 - y0;
        int N = xi * y1 - xp * y0;
        if (D < 0) {
            D = -D;
            N = -N;
        }
        if (rs) {  // pixels x with x*D < N  <=>  x < ceil(N / D) = floor((N - 1) / D) + 1
            const int t = min(max(floor_div_clamped(N - 1, D, P) + 1, 0), P);
#pragma unroll
            for (int w = 0; w < WPR; ++w) R[w] ^= prefix_word(t, w);
        }
        if (ls) {  // pixels x with x*D > N  <=>  x >= floor(N / D) + 1
            const int s = min(max(floor_div_clamped(N, D, P) + 1, 0), P);
            lodd ^= 0xFFFFFFFFu;
#pragma unroll
            for (int w = 0; w < WPR; ++w) L[w] ^= prefix_word(s, w);
        }
    }
    const int c1 = min(p.maxc, P - 1) + 1;
#pragma unroll
    for (int w = 0; w < WPR; ++w) {
        const uint32_t cols = prefix_word(c1, w) & ~prefix_word(p.minc, w);
        out[w] = live ? (((R[w] | (L[w] ^ lodd)) & cols) | V[w]) : 0u;
    }
}

// G rollouts per block, G*P <= 256 rows unless G == 1 (then rows are strided over the block).
template <int WPR>
__global__ __launch_bounds__(kThreads) void raster_iou_kernel(
    const int* __restrict__ verts_a, const int* __restrict__ nverts_a, const int* __restrict__ verts_b,
    const int* __restrict__ nverts_b, int n, int vmax, int P, int G, int npoly, int* __restrict__ out_inter,
    int* __restrict__ out_union, uint8_t* __restrict__ mask_a, uint8_t* __restrict__ mask_b) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    // LDS carve (ints): verts[2][G][vmax][2] | meta[2][G][8] | cnt[G][2] | bits[2][G*P][WPR]
    int* s_verts = smem;
    int* s_meta = s_verts + 2 * G * vmax * 2;
    int* s_cnt = s_meta + 2 * G * 8;
    uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_cnt + ((2 * G + 3) & ~3));

    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * G;  // first rollout of this block
    const int rows = G * P;

    for (int idx = tid; idx < 2 * G * vmax * 2; idx += kThreads) {
        const int poly = idx / (G * vmax * 2);
        const int rem = idx - poly * (G * vmax * 2);
        const int g = rem / (vmax * 2);
        const int* src = poly ? verts_b : verts_a;
        int val = 0;
        if (poly < npoly && r0 + g < n) val = src[(size_t)(r0 + g) * vmax * 2 + (rem - g * vmax * 2)];
        s_verts[idx] = min(max(val, -kCoordClamp), kCoordClamp);
    }
    for (int k = tid; k < 2 * G; k += kThreads) s_cnt[k] = 0;
    __syncthreads();
    for (int k = tid; k < 2 * G; k += kThreads) {
        const int poly = k / G, g = k - poly * G;
        int nv = 0;
        if (poly < npoly && r0 + g < n) nv = min(max((poly ? nverts_b : nverts_a)[r0 + g], 0), vmax);
        const int* v = s_verts + (poly * G + g) * vmax * 2;
        int minx = 0, maxx = -1, miny = 0, maxy = -1;
        if (nv > 0) {
            minx = maxx = v[0];
            miny = maxy = v[1];
            for (int i = 1; i < nv; ++i) {
                minx = min(minx, v[2 * i]);
                maxx = max(maxx, v[2 * i]);
                miny = min(miny, v[2 * i + 1]);
                maxy = max(maxy, v[2 * i + 1]);
            }
        }
        int* m = s_meta + k * 8;
        m[0] = nv;
        m[1] = max(miny, 0);
        m[2] = maxy;
        m[3] = max(minx, 0);
        m[4] = maxx;
    }
    __syncthreads();

    for (int base = 0; base < rows; base += kThreads) {  // uniform trip count: the wave reduction below needs every lane
        const int row = base + tid;
        const bool active = row < rows;
        const int g = active ? row / P : (rows - 1) / P;
        const int y = row - g * P;
        int ci = 0, cu = 0;
        if (active) {
            uint32_t a[WPR], b[WPR];
            {
                const int* m = s_meta + g * 8;
                PolyLds p{s_verts + g * vmax * 2, m[0], m[1], m[2], m[3], m[4]};
                row_bits<WPR>(p, y, P, a);
            }
            if (npoly > 1) {
                const int* m = s_meta + (G + g) * 8;
                PolyLds p{s_verts + (G + g) * vmax * 2, m[0], m[1], m[2], m[3], m[4]};
                row_bits<WPR>(p, y, P, b);
            } else {
#pragma unroll
                for (int w = 0; w < WPR; ++w) b[w] = a[w];
            }
#pragma unroll
            for (int w = 0; w < WPR; ++w) {
                ci += __popc(a[w] & b[w]);
                cu += __popc(a[w] | b[w]);
                s_bits[(size_t)row * WPR + w] = a[w];
                s_bits[(size_t)(rows + row) * WPR + w] = b[w];
            }
        }
        // rows of one wave mostly belong to one rollout: one LDS atomic per wave instead of one per row
        const int g0 = __builtin_amdgcn_readfirstlane(g);
        if (__all(g == g0)) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ci += __shfl_xor(ci, o);
                cu += __shfl_xor(cu, o);
            }
            if ((tid & 63) == 0) {
                if (ci) atomicAdd(&s_cnt[2 * g0], ci);
                if (cu) atomicAdd(&s_cnt[2 * g0 + 1], cu);
            }
        } else {
            if (ci) atomicAdd(&s_cnt[2 * g], ci);
            if (cu) atomicAdd(&s_cnt[2 * g + 1], cu);
        }
    }
    __syncthreads();
    for (int g = tid; g < G && r0 + g < n; g += kThreads) {
        if (out_inter) out_inter[r0 + g] = s_cnt[2 * g];
        if (out_union) out_union[r0 + g] = s_cnt[2 * g + 1];
    }

    // ---- optional byte masks: expand bit rows, 16 B (P % 16 == 0), 4 B (P % 4 == 0) or 1 B stores
    const long long valid_rows = (long long)n * P - (long long)r0 * P;  // rows of real rollouts in this block
    for (int poly = 0; poly < npoly; ++poly) {
        uint8_t* dst = poly ? mask_b : mask_a;
        if (!dst) continue;
        dst += (size_t)r0 * P * P;
        const uint32_t* bits = s_bits + (size_t)poly * rows * WPR;
        if ((P & 15) == 0) {
            constexpr int SLOTS = 2 * WPR;  // 16-pixel chunks per row, padded to a power of two
            const int chunk = tid & (SLOTS - 1);
            const int rsub = tid / SLOTS;
            if (chunk * 16 < P) {
                for (int row = rsub; row < rows && row < valid_rows; row += kThreads / SLOTS) {
                    const uint32_t half = (bits[(size_t)row * WPR + (chunk >> 1)] >> ((chunk & 1) * 16)) & 0xFFFFu;
                    uint4 o;
                    o.x = ((half & 0xFu) * 0x00204081u) & 0x01010101u;
                    o.y = (((half >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
                    o.z = (((half >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
                    o.w = (((half >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
                    *reinterpret_cast<uint4*>(dst + (size_t)row * P + chunk * 16) = o;
                }
            }
        } else if ((P & 3) == 0) {
            constexpr int SLOTS = 8 * WPR;  // 4-pixel groups per row, padded
            constexpr int RSTEP = kThreads / SLOTS > 0 ? kThreads / SLOTS : 1;
            const int slot = tid % SLOTS;
            if (slot * 4 < P) {
                for (int row = tid / SLOTS; row < rows && row < valid_rows; row += RSTEP) {
                    const uint32_t nib = (bits[(size_t)row * WPR + (slot >> 3)] >> ((slot & 7) * 4)) & 0xFu;
                    *reinterpret_cast<uint32_t*>(dst + (size_t)row * P + slot * 4) = (nib * 0x00204081u) & 0x01010101u;
                }
            }
        } else {
            for (long long px = tid; px < (long long)rows * P; px += kThreads) {
                const int row = (int)(px / P), x = (int)(px - (long long)row * P);
                if (row < valid_rows) dst[px] = (bits[(size_t)row * WPR + (x >> 5)] >> (x & 31)) & 1u;
            }
        }
    }
}

template <int WPR>
int launch(const int* va, const int* na, const int* vb, const int* nb, int n, int vmax, int P, int npoly, int* oi,
           int* ou, uint8_t* ma, uint8_t* mb, hipStream_t stream) {
    const int G = P >= kThreads ? 1 : kThreads / P;
    const int blocks = (n + G - 1) / G;
    const size_t lds_ints = (size_t)2 * G * vmax * 2 + 2 * G * 8 + ((2 * G + 3) & ~3) + (size_t)2 * G * P * WPR;
    const size_t lds = lds_ints * sizeof(int);
    ISG_REQUIRE(lds <= 160 * 1024, ISG_EINVAL, "isg_raster_iou: LDS request %zu exceeds 160 KiB", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&raster_iou_kernel<WPR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(raster_iou_kernel<WPR>, dim3(blocks), dim3(kThreads), lds, stream, va, na, vb, nb, n, vmax, P,
                       G, npoly, oi, ou, ma, mb);
    return isg::check_launch("raster_iou_kernel");
}

int dispatch(const int* va, const int* na, const int* vb, const int* nb, int n, int vmax, int P, int npoly, int* oi,
             int* ou, uint8_t* ma, uint8_t* mb, void* stream) {
    ISG_REQUIRE(n >= 0 && vmax >= 1 && vmax <= ISG_RASTER_MAX_VERTS, ISG_EINVAL,
                "raster: need n >= 0 and 1 <= vmax <= %d (n=%d vmax=%d)", ISG_RASTER_MAX_VERTS, n, vmax);
    ISG_REQUIRE(P >= 1 && P <= ISG_RASTER_MAX_SIZE, ISG_EINVAL, "raster: size %d outside 1..%d", P,
                ISG_RASTER_MAX_SIZE);
    ISG_REQUIRE(va && na && (npoly == 1 || (vb && nb)), ISG_EINVAL, "raster: null vertex pointer");
    const int align = (P & 15) == 0 ? 16 : ((P & 3) == 0 ? 4 : 1);
    ISG_REQUIRE((reinterpret_cast<uintptr_t>(ma) % align) == 0 && (reinterpret_cast<uintptr_t>(mb) % align) == 0,
                ISG_EINVAL, "raster: mask pointers must be %d-byte aligned for size %d", align, P);
    if (n == 0) return ISG_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int wpr = (P + 31) / 32;
    if (wpr <= 1) return launch<1>(va, na, vb, nb, n, vmax, P, npoly, oi, ou, ma, mb, s);
    if (wpr <= 2) return launch<2>(va, na, vb, nb, n, vmax, P, npoly, oi, ou, ma, mb, s);
    if (wpr <= 4) return launch<4>(va, na, vb, nb, n, vmax, P, npoly, oi, ou, ma, mb, s);
    if (wpr <= 8) return launch<8>(va, na, vb, nb, n, vmax, P, npoly, oi, ou, ma, mb, s);
    return launch<16>(va, na, vb, nb, n, vmax, P, npoly, oi, ou, ma, mb, s);
}

}  // namespace

extern "C" int isg_raster_iou(const int32_t* verts_pred, const int32_t* nverts_pred, const int32_t* verts_true,
                              const int32_t* nverts_true, int n, int vmax, int size, int32_t* out_inter,
                              int32_t* out_union, uint8_t* mask_pred, uint8_t* mask_true, void* stream) {
    ISG_REQUIRE(n >= 0, ISG_EINVAL, "isg_raster_iou: n=%d", n);
    if (n == 0) return ISG_OK;  // an empty batch has no buffers to speak of
    ISG_REQUIRE(out_inter && out_union, ISG_EINVAL, "isg_raster_iou: null output pointer");
    return dispatch(verts_pred, nverts_pred, verts_true, nverts_true, n, vmax, size, 2, out_inter, out_union,
                    mask_pred, mask_true, stream);
}

extern "C" int isg_shape_mask(const int32_t* verts, const int32_t* nverts, int n, int vmax, int size, uint8_t* mask,
                              int32_t* area, void* stream) {
    ISG_REQUIRE(mask || area, ISG_EINVAL, "isg_shape_mask: nothing to write");
    return dispatch(verts, nverts, nullptr, nullptr, n, vmax, size, 1, area, nullptr, mask, nullptr, stream);
}
