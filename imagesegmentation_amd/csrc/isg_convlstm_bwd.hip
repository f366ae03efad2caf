// One backward step of the ConvLSTM cell in ONE kernel (bf16): gate backward + data gradient + weight gradient.
//
// The three-kernel backward (pointwise -> dY in HBM -> data gradient and weight gradient each reading dY back)
// moves dY three times: 384 of its ~1,000 bytes per pixel.  Here the pre-activation gradient dY exists only in
// LDS.  Per 8 x 32 (or 4 x 32) pixel tile a 512-thread workgroup runs
//   - the gate backward: dY for the tile AND its one-pixel halo from the backward factors the forward step saved
//     ((A, B, f, o) per channel, see lstm_backward_factors: with c_prev and this step's h they give the whole gate
//     backward without a hyperbolic tangent and without the new cell state) and the incoming gradients (the
//     halo pixels' values are recomputed by the neighbouring tiles, bit for bit the same) into a padded
//     pixel-major halo image in LDS; dc_prev of the tile's own pixels goes to global memory; every thread keeps
//     the sum of its own pixels' dY (the bias gradient) in registers over all tiles;
//   - a short store phase that puts the [x | h_prev] halo tile beside it (64-byte regions of 32 channels);
//   - phase B: the weight gradient (each wave a group of taps x one 32-column half of dY, both operands through
//     transposing LDS reads; the accumulators stay in registers over all tiles and leave as one slab per
//     workgroup), then the data gradient (each wave one tile row x one 32-row tile of input channels, weight
//     image resident in LDS) with its dx / dh_prev stores.
// Pipelined schedule (PIPE): three tiles are in flight - there are TWO dY images, so the gate backward of tile
// n + 1 (vector ALU) runs while phase B of tile n (matrix pipe) reads the other image, in opposite order on the two
// waves of every SIMD, and the operands of tile n + 2 are requested into each staging register as the gate
// backward frees it.  The 48-row kernel of the first layer (32 x channels + 16 h_prev channels, a 54 KB weight image)
// has LDS for ONE dY image on 8-row tiles: it runs the same schedule with the gate backward's dY held in registers
// until the join (HOLD), where no wave reads the image any more.  Its data gradient takes the 48 rows as three
// 16-row tiles on 16 x 16 x 32 MFMAs, so that every dY fragment is read from LDS once for all of them, and the
// operands of the last gate-backward pass are requested after phase B (DEFER_LAST) - without that the kernel does
// not fit its 256 registers.  The 64-row kernel (48 x channels) runs the three parts in order on 4-row tiles.
// What bounds it: LDS delivery in phase B (both k-loops read ~2 KB of fragments per MFMA; the matrix pipe idles
// about half the time) and the vector ALU in the gate backward, with ~600 - 800 bytes per pixel in flight behind them.
#include <utility>

#include "isg_convlstm_shared.h"

using namespace isgc;

namespace {

constexpr int kBwdThreads = 512;

// Diagnostic build only (-DISG_STAMPS, tools/stamp_bwd.py): per-phase cycle sums of waves 0 and 7, written to a
// buffer that nothing else reads.  In the shipped library no stamp executes.
#ifdef ISG_STAMPS
__device__ unsigned long long* g_bwd_stamp_buf = nullptr;
__device__ __forceinline__ unsigned long long bwd_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define BSTAMP(var) const unsigned long long var = bwd_stamp()
#define BSTAMP_ADD(slot, a, b) stamp_sum[slot] += (b) - (a)
#else
#define BSTAMP(var)
#define BSTAMP_ADD(slot, a, b)
#endif

struct BwdParams {
    const void* gates;      // backward factors [B,H,W,F/2,4,2] bf16 (A, B, f, o of lstm_backward_factors)
    const float* c_prev;    // [B,H,W,F] float, or bf16 when cprev_bf16
    int cprev_bf16;
    const void* h_cur;      // [B,H,W,F]: this step's new h
    int h_cur_pstride;      // C0 == 48 only: bytes from pixel to pixel of h_cur and of dh_state (32, or more: 16-channel
    int dh_state_pstride;   // slices of wider pixels - the other group's slot of the joint input and of its gradient)
    int dh_above_pstride;   // C0 == 48 only: likewise dh_above (the group's 16 channels of the cell's joint output gradient)
    const void* dh_above;   // gradient of the emitted output, sample stride dh_above_bstride
    long long dh_above_bstride;
    int pool_shift;         // > 0: dh_above lives on the grid pooled by 2^s x 2^s windows (average pool)
    const void* dh_state;   // [B,H,W,F]
    const void* dc_state;   // [B,H,W,F] bf16
    const void* x;          // [B,H,W,C0]
    const void* h_prev;     // [B,H,W,F]
    int B, H, W, tiles_x, tiles_y;
    const int* durations;
    int t;
    const void* wimg;       // data-gradient weight image, row-tile-major
    void* dc_prev;          // [B,H,W,F] bf16, NOT the dc_state buffer (halo pixels are read by neighbouring tiles)
    void* dh_prev;          // [B,H,W,F], NOT the dh_state buffer
    void* dx;               // [B,H,W,dx_cstride] or null
    int dx_channels, dx_cstride, dx_accumulate;
    float* slab;            // [grid][9][MT*32][64]
    float* bias_slab;       // [grid][64]
    int slab_accumulate;
    const int* hist_points; // [B][hist][2] of this step, sample stride hist_bstride ints
    int hist;
    long long hist_bstride;
    void* hist_dy;          // [B][hist][9][64] bf16, zeroed by the host: dY at the pixel each tap reaches from a point
};

// PIPE: three tiles in flight over two dY images (see the top of the file).  !PIPE: one dY image, every wave in the
// same order (gate backward, store phase, phase B), two tiles in flight - what an 8-row tile of the two-row-tile
// kernel can afford: its weight image alone is 54 - 72 KB.
// HOLD (with PIPE): the pipelined schedule on ONE dY image - the gate backward keeps its results in registers (8 per
// pass) and they are written into the image at the join, between the two barriers, once every wave is done reading
// the previous tile's.  For the kernel whose weight image leaves no room for a second dY image on 8-row tiles.
template <int C0, int MT, int TR, bool PIPE, bool HOLD = false>
struct FB {
    static constexpr int NPXH = (TR + 2) * HALO_W;       // halo tile pixels
    static constexpr int DY_PSTRIDE = NG * 2 + 16;       // padded pixel stride of the halo dY image
    static constexpr int DYH_BYTES = NPXH * DY_PSTRIDE;
    static constexpr int NDY = PIPE && !HOLD ? 2 : 1;    // dY images
    static constexpr int RSTRIDE = 64;                   // 32 channels: one region row of the [x | h_prev] image
    static constexpr int IN_REGION = NPXH * RSTRIDE;
    static constexpr int IN_BYTES = MT * IN_REGION;
    static constexpr int NS = 9 * (NG / 16);             // k-steps of the data gradient
    // Rows C0 + F .. 63 of the second row tile are padding whose results nobody stores: when they are the tile's
    // upper 16 rows (48 real rows) the weight image in LDS is three 16-row tiles in the A layout of the 16 x 16 x 32
    // MFMA, and a wave takes all three for its tile row.
    static constexpr bool WCOMPACT = MT == 2 && TR == 8 && C0 + F_ <= 48;
#ifndef ISG_DG_AHEAD
#define ISG_DG_AHEAD 3
#endif
    // k-steps the fragments of the 32-row data gradient / of the weight gradient are requested ahead of their MFMAs
    // (one row tile: 3 measured 3.5 % faster than 1 and as fast as 5; weight gradient: 2 and 3 no faster than 1)
    static constexpr int DG_AHEAD = MT == 1 ? ISG_DG_AHEAD : 1;
#ifndef ISG_WG_AHEAD
#define ISG_WG_AHEAD 1
#endif
    static constexpr int WG_AHEAD = MT == 1 ? ISG_WG_AHEAD : 1;
    static constexpr bool DEFER_LAST = WCOMPACT && PIPE;      // (see request_last_pass in the kernel)
    static constexpr int W1_STEP = WCOMPACT ? 512 : 1024;     // bytes per k-step of the second row tile
    static constexpr int W_BYTES = NS * 1024 + (MT == 2 ? NS * W1_STEP : 0);
    static constexpr int OFF_DYH = W_BYTES, OFF_IN = OFF_DYH + NDY * DYH_BYTES;
    static constexpr int LDS_BYTES = OFF_IN + IN_BYTES;
    static constexpr int NPW = (NPXH * 4 + kBwdThreads - 1) / kBwdThreads;  // gate-backward passes (4 channels per item)
    static constexpr int NCH0 = C0 / 8, NCHT = NCH0 + 2;                    // 16-byte chunks of x / of [x | h_prev]
    static constexpr int NXP = (NPXH * NCHT + kBwdThreads - 1) / kBwdThreads;
    // Weight-gradient output tiles (32 input-channel rows x 64 dY columns, one per wave slot and column half).  One
    // row tile: 9 (taps).  Two row tiles: 18 (tap, row tile) pairs - or, when the second row tile holds only the 16
    // channels of h_prev (HMERGE), 9 tiles of x rows + 5 tiles that stack the h rows of TWO taps (rows 0-15: tap
    // 2 j, rows 16-31: tap 2 j + 1): 14 tiles instead of 18, no MFMA row spent on padding channels.
    static constexpr bool HMERGE = MT == 2 && C0 + F_ <= 48;
    static constexpr int NWT = MT == 1 ? 9 : HMERGE ? 14 : 18;
    static constexpr int NSLOT = (NWT + 3) / 4;          // tiles per wave (4 wave pairs share them)
    // With stacked h rows the last tile's upper 16 rows have no tap: they are fed ones instead, and every one of
    // them then holds the sum of dY over the pixels - the bias gradient - at the price of four selects per k-step
    // on one wave pair; the other kernels sum it in the gate backward's registers (16 per thread).
    static constexpr bool BIAS_MFMA = HMERGE;
    static constexpr int BIAS_TILE = 13, BIAS_GROUP = 3, BIAS_SLOT = 2;  // where that tile sits (see the dealing below)
    static constexpr int NMT = TR * MT / 8;              // row tiles of ONE tile row a wave takes in the data gradient
    static_assert(TR * MT == 8 || (TR == 8 && MT == 2), "data-gradient units (tile row x row tile) per wave: 1 or 2");
    static_assert(C0 + F_ <= MT * 32 && C0 % 8 == 0, "input channels must fit the row tiles");
    static_assert(DYH_BYTES % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(NDY * DYH_BYTES >= 16 * kBwdThreads * 4, "the closing bias reduction borrows the dY images");
};

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// two floats -> one dword of two bf16 (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// ACC: dx is added to what the array holds (the static features' gradient over the steps of a sequence).
// EDGE: the two ends of a sequence as instantiations of their own (as run-time facts their scalars cost every launch
// 2 - 4 %: a dozen more spilled scalar registers in the tile loop).  Bit 0: no gradient is carried in (dh_state and
// dc_state null: the first step of the reverse sweep when the final state is not differentiated); bit 1: the step
// out of the zero state (c_prev and h_prev null).  Nothing is read for a null operand; zeros take its place.
// CB: c_prev is stored as bf16 (a forward pass run with ISG_STATE_BF16): 8 instead of 16 bytes per item.
template <int C0, int MT, int TR, bool PIPE, bool ACC, int EDGE, bool HOLD = false, bool CB = false>
__global__ __launch_bounds__(kBwdThreads) void convlstm_bwd_fused_kernel(BwdParams p) {
    constexpr bool no_carry = (EDGE & 1) != 0, zero_state = (EDGE & 2) != 0;
    using K = FB<C0, MT, TR, PIPE, HOLD>;
    using T = __bf16;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    (void)sizeof(s16x8);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w = smem;
    char* lds_dyh = smem + K::OFF_DYH;
    char* lds_in = smem + K::OFF_IN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, px_l = lane & 31;

    {  // weight image, once per workgroup; channels past the real ones of the [x | h_prev] image stay zero
        const uint4* src = static_cast<const uint4*>(p.wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds_w);
        if constexpr (!K::WCOMPACT) {
            for (int i = tid; i < K::W_BYTES / 16; i += kBwdThreads) dst[i] = src[i];
        } else {
            // Three 16-row tiles in the A layout of the 16 x 16 x 32 MFMA: fragment (32-deep k-step s32, lane l) =
            // row l % 16, k = 8 (l / 16) .. + 7, a 16-byte slot at (s32 * 64 + l) * 16 of the tile.  Tiles 0 and 1
            // are the x rows, row 4 q + i of tile j = channel 8 q + 4 j + i (a lane's accumulators of the two tiles
            // are then 8 consecutive channels); tile 2 is the h_prev rows in order.  The source fragments are those of
            // the 32-row image: 16-deep k-step 2 s32 + (l / 32), k half (l / 16) % 2, row as stored there.
            for (int i = tid; i < 3 * K::NS * 32; i += kBwdThreads) {
                const int j = i / (K::NS * 32), r = i - j * (K::NS * 32);
                const int sidx = r >> 5, kg = (r >> 4) & 1, m = r & 15;
                const int c = 8 * (m >> 2) + 4 * j + (m & 3);                 // x channel (tiles 0, 1)
                const int prow = 8 * ((c >> 2) & 3) + 4 * (c >> 4) + (c & 3);  // its row of the permuted first tile
                dst[i] = j < 2 ? src[sidx * 64 + kg * 32 + prow] : src[(K::NS + sidx) * 64 + kg * 32 + m];
            }
        }
        if constexpr (C0 + F_ < MT * 32) {
            for (int i = tid; i < K::IN_BYTES / 16; i += kBwdThreads)
                reinterpret_cast<uint4*>(lds_in)[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    }

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8 labels the workgroups that share an L2).  Each
    // label sweeps its own contiguous eighth of the tiles, its workgroups side by side along the image rows: a
    // tile's halo is then re-read by workgroups on the same L2 at about the same time, and the workgroups of a
    // label read neighbouring addresses.  (Measured against runs of vertically adjacent tiles per workgroup, which
    // re-read the shared halo rows from the same CU: 3 - 14 % slower, growing with the run length - 256 separate
    // streams cost more in the memory system than the halo hits save.  Speed only: the result does not depend on
    // which workgroup takes which tile.)
    const bool by_xcd = ((int)gridDim.x & 7) == 0;
    const int wg_per = by_xcd ? (int)gridDim.x >> 3 : (int)gridDim.x;       // workgroups per label
    const int tiles_per = by_xcd ? (ntiles + 7) >> 3 : ntiles;              // tiles per label
    const int label = by_xcd ? (int)blockIdx.x & 7 : 0, rank_in = by_xcd ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int n_iter = (tiles_per + wg_per - 1) / wg_per;

    // ---- where this thread's items of every pass sit in the halo tile (fixed for the whole kernel) ----
    // gate-backward item = (halo pixel, 4 channels 4 q .. 4 q + 3): 32 bytes of factors, 16 of c_prev, 8 of h and of
    // each incoming gradient.  (Two channels per item was measured: the per-item bookkeeping then weighs as much as the
    // gate arithmetic.)  512 % 4 == 0 keeps q per thread.
    const int q = tid & 3;
    int pinfo[K::NPW];      // hy | hx << 8 | exists << 31
#pragma unroll
    for (int j = 0; j < K::NPW; ++j) {
        const int px = j * (kBwdThreads / 4) + (tid >> 2);
        const bool exists = px < K::NPXH;
        const int cpx = exists ? px : K::NPXH - 1;
        const int hy = cpx / HALO_W, hx = cpx - hy * HALO_W;
        pinfo[j] = hy | hx << 8 | (exists ? (int)0x80000000 : 0);
    }
    int xinfo[K::NXP];      // LDS byte offset (bits 0-17) | hy << 18 | hx << 22 | chunk << 28 | exists << 31
#pragma unroll
    for (int j = 0; j < K::NXP; ++j) {
        const int idx = j * kBwdThreads + tid;
        const bool exists = idx < K::NPXH * K::NCHT;
        const int cidx = exists ? idx : K::NPXH * K::NCHT - 1;
        const int px = cidx / K::NCHT, ch = cidx - px * K::NCHT;
        const int hy = px / HALO_W, hx = px - hy * HALO_W;
        const int cbyte = (ch < K::NCH0 ? ch * 8 : C0 + (ch - K::NCH0) * 8) * 2;  // byte in the [x | h_prev] row
        const int ldso = (cbyte / K::RSTRIDE) * K::IN_REGION + px * K::RSTRIDE + (cbyte % K::RSTRIDE);
        xinfo[j] = ldso | hy << 18 | hx << 22 | ch << 28 | (exists ? (int)0x80000000 : 0);
    }
    static_assert(K::IN_BYTES < (1 << 18), "LDS offsets are packed into 18 bits");

    // ---- staged operands of one tile ----
    uint4 sg[K::NPW][2];             // factors of the item's two channel pairs: A A B B f f o o
    std::conditional_t<CB, uint2, uint4> scp[K::NPW];  // c_prev: 4 floats, or 4 bf16
    uint2 shc[K::NPW];               // this step's h: 4 bf16
    uint2 sda[K::NPW], sds[K::NPW], sdc[K::NPW];  // dh_above, dh_state, dc_state: 4 bf16
    uint4 sx[K::NXP];
    unsigned pw_inside = 0u, x_inside = 0u;
    int st_dur = 0x7fffffff;
    int st_hp[4] = {-1, 0, -1, 0};
    // (only the 48-channel kernel - two groups of a wider cell sharing one joint input - meets h and its carried
    // gradient as slices of wider pixels; the others keep the constant)
    const unsigned hc_stride = C0 == 48 ? (unsigned)p.h_cur_pstride : 32u;
    const unsigned ds_stride = C0 == 48 ? (unsigned)p.dh_state_pstride : 32u;
    const unsigned da_stride = C0 == 48 ? (unsigned)p.dh_above_pstride : (unsigned)(F_ * 2);

    struct Tile {
        int b, ty0, tx0;
        int valid;  // (an int: a bool's padding bytes were copied through scratch memory)
    };
    // Every load below is issued unconditionally, from addresses clamped into the tensors (a tile past the end
    // re-reads the last one, a missing optional tensor reads the gate tensor instead) and its value is dropped
    // or zeroed at the use: with loads under branches the compiler can no longer count what is in flight and
    // waits for ALL of it (s_waitcnt vmcnt(0)) at the first use, which serialises the prefetch with the math.
    auto tile_ref = [&](int k) {
        Tile r;
        const int within = k * wg_per + rank_in;
        const int tile = label * tiles_per + within;
        r.valid = k < n_iter && within < tiles_per && tile < ntiles;
        const int ctile = min(tile, ntiles - 1);
        r.b = ctile / tiles_per_img;
        const int rem = ctile - r.b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        r.ty0 = tyi * TR, r.tx0 = (rem - tyi * p.tiles_x) * TW;
        return r;
    };
    const float inv_pool = 1.0f / (float)(1 << (2 * p.pool_shift));

    const int* dur_base = p.durations != nullptr ? p.durations : reinterpret_cast<const int*>(p.gates);
    const int dur_mul = p.durations != nullptr ? 1 : 0;
    const int* hp_base = p.hist > 0 ? p.hist_points : reinterpret_cast<const int*>(p.gates);
    const long long hp_bstride = p.hist > 0 ? p.hist_bstride : 0;
    const int hp_last = p.hist > 0 ? 2 * p.hist - 1 : 0;

    auto load_meta = [&](const Tile& r) {
        st_dur = dur_base[r.b * dur_mul];
#pragma unroll
        for (int i = 0; i < 4; ++i) st_hp[i] = hp_base[(size_t)r.b * hp_bstride + min(i, hp_last)];
    };
    auto load_pw = [&](const Tile& r, auto jc) {
        constexpr int j = decltype(jc)::value;
        // (the packed word is laundered at every use: otherwise its unpacked fields are hoisted out of the tile
        // loop into registers of their own, a dozen of them, and the kernel spills)
        int inf = pinfo[j];
        asm volatile("" : "+v"(inf));
        const int hy = inf & 0xff, hx = (inf >> 8) & 0xff;
        const int gy = r.ty0 - 1 + hy, gx = r.tx0 - 1 + hx;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && inf < 0) pw_inside |= 1u << j;
        const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
        const unsigned pix = (unsigned)(cy * p.W + cx);
        const unsigned pixel = (unsigned)(r.b * p.H) * (unsigned)p.W + pix;
        sg[j][0] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.gates) + (size_t)(pixel * 128u + q * 32u));
        sg[j][1] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.gates) + (size_t)(pixel * 128u + q * 32u + 16u));
        if constexpr (CB) {
            if constexpr (zero_state) scp[j] = make_uint2(0u, 0u);
            else scp[j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(p.c_prev) + (size_t)(pixel * 32u + q * 8u));
        } else {
            if constexpr (zero_state) scp[j] = make_uint4(0u, 0u, 0u, 0u);
            else scp[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.c_prev) + (size_t)(pixel * 64u + q * 16u));
        }
        shc[j] = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.h_cur) + (size_t)(pixel * hc_stride + q * 8u));
        const unsigned apix = (unsigned)((cy >> p.pool_shift) * (p.W >> p.pool_shift) + (cx >> p.pool_shift));
        sda[j] = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.dh_above) + (size_t)r.b * p.dh_above_bstride * 2 +
                                                 (size_t)(apix * da_stride + q * 8u));
        if constexpr (no_carry) {
            sds[j] = sdc[j] = make_uint2(0u, 0u);
        } else {
            sds[j] = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.dh_state) + (size_t)(pixel * ds_stride + q * 8u));
            sdc[j] = *reinterpret_cast<const uint2*>(static_cast<const char*>(p.dc_state) + (size_t)(pixel * 32u + q * 8u));
        }
    };
    auto load_x = [&](const Tile& r, auto jc) {
        constexpr int j = decltype(jc)::value;
        int inf = xinfo[j];
        asm volatile("" : "+v"(inf));
        const int hy = (inf >> 18) & 15, hx = (inf >> 22) & 63, ch = (inf >> 28) & 7;
        const int gy = r.ty0 - 1 + hy, gx = r.tx0 - 1 + hx;
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W && inf < 0) x_inside |= 1u << j;
        const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
        const unsigned pixel = (unsigned)((r.b * p.H + cy) * p.W + cx);
        const bool is_h = ch >= K::NCH0;
        // (the zero state: an h_prev chunk re-reads the pixel's first x chunk instead - a line the x chunks bring in
        // anyway - and the store phase writes zeros)
        const bool from_x = !is_h || zero_state;
        const char* base = from_x ? static_cast<const char*>(p.x) : static_cast<const char*>(p.h_prev);
        const unsigned off = is_h ? (zero_state ? pixel * (unsigned)(C0 * 2) : pixel * (unsigned)(F_ * 2) + (unsigned)(ch - K::NCH0) * 16u)
                                  : pixel * (unsigned)(C0 * 2) + (unsigned)ch * 16u;
        sx[j] = *reinterpret_cast<const uint4*>(base + (size_t)off);
    };

    // ---- weight-gradient work of this wave: a group of (tap, row tile) pairs x one 32-column half of dY ----
    const int wg_n = wave & 1, wg_grp = wave >> 1;
    // the tiles are dealt to the 4 wave pairs in runs: the first NWT % 4 pairs take one more
    constexpr int kBase = K::NWT / 4, kExtra = K::NWT % 4;
    const int wg_first = wg_grp * kBase + min(wg_grp, kExtra), wg_cnt = kBase + (wg_grp < kExtra ? 1 : 0);
    const bool wg_full = wg_cnt == K::NSLOT;  // the last slot is used by the larger groups only
    static_assert(!K::BIAS_MFMA || (K::NWT == 14 && kBase == 3 && kExtra == 2), "the ones-fed tile is tile 13 = slot 2 of group 3");
    // (lanes 16-31 and 48-63 feed rows 16-31 of the A operand)
    const bool ones_rows = K::BIAS_MFMA && wg_grp == K::BIAS_GROUP && ((lane >> 4) & 1) != 0;
    f32x16 wacc[K::NSLOT];
#pragma unroll
    for (int i = 0; i < K::NSLOT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[i][r] = 0.0f;
    float bacc[16];  // bias gradient: this thread's dY columns 16 q .. 16 q + 15, summed over its own pixels
#pragma unroll
    for (int i = 0; i < 16; ++i) bacc[i] = 0.0f;
    // this lane's fixed read positions in the dY interior image and, per slot, in the [x | h_prev] halo image
    const char* wg_dy;
    const char* wg_in[K::NSLOT];
    {
        const int li = lane & 15, g16 = lane >> 4;
        const int csub = 16 * (g16 & 1) + 4 * (li & 3);      // channel offset inside the 32-channel region
        const int lane_px = 8 * half + (li >> 2);            // pixel inside the 16-pixel k-step (second read: + 4)
        wg_dy = lds_dyh + (HALO_W + 1 + lane_px) * K::DY_PSTRIDE + wg_n * 64 + csub * 2;
#pragma unroll
        for (int i = 0; i < K::NSLOT; ++i) {
            const int idx = wg_first + min(i, wg_cnt - 1);
            int tap, mt, chan;
            if constexpr (K::HMERGE) {
                // tiles 0-8: x rows of tap idx; tiles 9-13: h rows of taps 2 j (this lane's rows 0-15) and 2 j + 1
                // (rows 16-31; the ninth tap has no partner: its upper rows repeat it and are dropped at the end)
                const bool htile = idx >= 9;
                tap = htile ? min(2 * (idx - 9) + (g16 & 1), 8) : idx;
                mt = htile ? 1 : 0;
                chan = htile ? 4 * (li & 3) : csub;
            } else {
                tap = MT == 1 ? idx : idx >> 1, mt = MT == 1 ? 0 : idx & 1, chan = csub;
            }
            const int dy = tap / 3, dx = tap - dy * 3;
            wg_in[i] = lds_in + mt * K::IN_REGION + (dy * HALO_W + dx + lane_px) * K::RSTRIDE + chan * 2;
        }
    }

    // (one unit per wave: tile row x row tile; two when an 8-row tile has two row tiles: then both of the wave's row)
    const int dg_row = (MT == 1 || K::NMT == 2) ? wave : wave >> 1;
    const int dg_mt = (MT == 1 || K::NMT == 2) ? 0 : wave & 1;

    // ---------------- the tile pipeline ----------------
    // Three tiles are in flight.  Between two workgroup barriers a wave
    //   - runs phase B of tile n out of dY image n % 2 and the [x | h_prev] image (matrix pipe), and
    //   - turns the staged operands of tile n + 1 into dY image (n + 1) % 2 (vector ALU; dc_prev leaves for
    //     global memory), requesting the operands of tile n + 2 into each staging register as it is freed.
    // Neither touches what the other reads, so their order inside the interval is free: waves 0-3 (one per SIMD)
    // take the gate backward first, waves 4-7 phase B first, and on every SIMD one wave's vector arithmetic runs
    // under the other's MFMAs.  A short store phase after the closing barrier writes the [x | h_prev] tile of
    // tile n + 1 (in registers since the interval before) over that of tile n.
    uint4 hold[HOLD ? K::NPW : 1][2];  // HOLD: the last gate backward's dY, on its way into the image
    bool hold_live = false;
    bool m_live = false;     // facts of the tile the last gate backward worked on, taken from the staging words
    int m_hp[4] = {-1, 0, -1, 0};

    auto gate_backward = [&](const Tile& c, const Tile& req, char* dyh) __attribute__((always_inline)) {
        m_live = c.valid && (p.durations == nullptr || p.t < __builtin_amdgcn_readfirstlane(st_dur));
#pragma unroll
        for (int i = 0; i < 4; ++i) m_hp[i] = __builtin_amdgcn_readfirstlane(st_hp[i]);
        const unsigned inside_bits = pw_inside;
        pw_inside = 0u;
        load_meta(req);
        hold_live = m_live;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            ([&] {
                constexpr int j = I;
                int inf = pinfo[j];
                asm volatile("" : "+v"(inf));
                if (m_live && inf < 0) {
                    const int hy = inf & 0xff, hx = (inf >> 8) & 0xff;
                    const bool inside = (inside_bits >> j) & 1u;
                    const bool own = hy >= 1 && hy <= TR && hx >= 1 && hx <= TW;  // a pixel of the tile itself
                    const int gy = c.ty0 - 1 + hy, gx = c.tx0 - 1 + hx;
                    unsigned ow[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // dY of 4 channels x 4 gates, bf16 pairs
                    float dcp[4] = {bf_lo(sdc[j].x), bf_hi(sdc[j].x), bf_lo(sdc[j].y), bf_hi(sdc[j].y)};
#ifdef ISG_BWD_NO_PW  // diagnostic build: the loads and the LDS traffic without the gate arithmetic
                    if (false) {
#else
                    if (inside) {
#endif
                        // word 4 (ch >> 1) + k holds value k (A, B, f, o) of channels ch & ~1 (low half) and ch | 1 (high)
                        const unsigned fw[8] = {sg[j][0].x, sg[j][0].y, sg[j][0].z, sg[j][0].w,
                                                sg[j][1].x, sg[j][1].y, sg[j][1].z, sg[j][1].w};
                        float cpv[4];
                        if constexpr (CB) {
                            cpv[0] = bf_lo(scp[j].x), cpv[1] = bf_hi(scp[j].x), cpv[2] = bf_lo(scp[j].y), cpv[3] = bf_hi(scp[j].y);
                        } else {
                            cpv[0] = __uint_as_float(scp[j].x), cpv[1] = __uint_as_float(scp[j].y);
                            cpv[2] = __uint_as_float(scp[j].z), cpv[3] = __uint_as_float(scp[j].w);
                        }
                        const float hv[4] = {bf_lo(shc[j].x), bf_hi(shc[j].x), bf_lo(shc[j].y), bf_hi(shc[j].y)};
                        const float da[4] = {bf_lo(sda[j].x), bf_hi(sda[j].x), bf_lo(sda[j].y), bf_hi(sda[j].y)};
                        const float ds[4] = {bf_lo(sds[j].x), bf_hi(sds[j].x), bf_lo(sds[j].y), bf_hi(sds[j].y)};
                        float out[16];
#pragma unroll
                        for (int ch = 0; ch < 4; ++ch) {
                            float fk[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                fk[k] = (ch & 1) ? bf_hi(fw[4 * (ch >> 1) + k]) : bf_lo(fw[4 * (ch >> 1) + k]);
                            const float fg = fk[2], og = fk[3], h = hv[ch];
                            // o (1 - tanh(c)^2) with tanh(c) = h / o  (o = 0 only where h = 0 too: the guard keeps 0 * inf away).
                            // h and o arrive rounded to bf16: for a saturated cell (|tanh c| near 1) the difference
                            // cancels and may come out slightly negative or above o - the median keeps it in [0, o].
                            const float e = __builtin_amdgcn_fmed3f(
                                og - h * (h * __builtin_amdgcn_rcpf(fmaxf(og, 1e-30f))), 0.f, og);
                            const float dh = da[ch] * inv_pool + ds[ch];
                            const float dct = dcp[ch] + dh * e;
                            out[4 * ch] = dct * fk[0];
                            out[4 * ch + 1] = dct * fk[1];
                            out[4 * ch + 2] = dct * (cpv[ch] * (fg - fg * fg));
                            out[4 * ch + 3] = dh * (h - h * og);
                            dcp[ch] = dct * fg;
                        }
                        // the bias gradient: dY summed over the pixels this tile owns (halo pixels belong to a neighbour)
                        if constexpr (!K::BIAS_MFMA) {
                            const float ownf = own ? 1.0f : 0.0f;
#pragma unroll
                            for (int i = 0; i < 16; ++i) bacc[i] = fmaf(ownf, out[i], bacc[i]);
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) ow[i] = pack_bf2(out[2 * i], out[2 * i + 1]);
                    }
                    const uint4 o0 = make_uint4(ow[0], ow[1], ow[2], ow[3]), o1 = make_uint4(ow[4], ow[5], ow[6], ow[7]);
                    if constexpr (HOLD) {
                        hold[j][0] = o0, hold[j][1] = o1;
                    } else {
                        char* dh_dst = dyh + (hy * HALO_W + hx) * K::DY_PSTRIDE + q * 32;
                        *reinterpret_cast<uint4*>(dh_dst) = o0;
                        *reinterpret_cast<uint4*>(dh_dst + 16) = o1;
                    }
                    if (own && inside) {
                        const unsigned pixel = (unsigned)((c.b * p.H + gy) * p.W + gx);
                        if (p.dc_prev)  // (null: nobody wants the gradient of the state before this step)
                            *reinterpret_cast<bf16x4*>(static_cast<char*>(p.dc_prev) + (size_t)(pixel * 32u + q * 8u)) =
                                bf16x4{(T)dcp[0], (T)dcp[1], (T)dcp[2], (T)dcp[3]};
                        if (p.hist > 0) {
                            // dY at the pixels the history taps reach, kept for the tap weights' gradient
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int tx = m_hp[2 * i] - gx + 1, ty = m_hp[2 * i + 1] - gy + 1;
                                if (i < p.hist && m_hp[2 * i] >= 0 && (unsigned)tx < 3u && (unsigned)ty < 3u) {
                                    T* hd = static_cast<T*>(p.hist_dy) + ((((size_t)c.b * p.hist + i) * 9 + ty * 3 + tx) * NG) + q * 16;
                                    *reinterpret_cast<uint4*>(hd) = o0;
                                    *reinterpret_cast<uint4*>(hd + 8) = o1;
                                }
                            }
                        }
                    }
                }
                if constexpr (!(K::DEFER_LAST && j == K::NPW - 1)) load_pw(req, std::integral_constant<int, j>{});
            }(), ...);
        }(std::make_integer_sequence<int, K::NPW>{});
    };

    // HOLD: the held dY of the last gate backward into the (one) image; called where no wave reads the image
    auto flush_hold = [&]() __attribute__((always_inline)) {
        if constexpr (HOLD) {
            if (hold_live) {
                [&]<int... I>(std::integer_sequence<int, I...>) {
                    ([&] {
                        constexpr int j = I;
                        int inf = pinfo[j];
                        asm volatile("" : "+v"(inf));
                        if (inf < 0) {
                            const int hy = inf & 0xff, hx = (inf >> 8) & 0xff;
                            char* dh_dst = lds_dyh + (hy * HALO_W + hx) * K::DY_PSTRIDE + q * 32;
                            *reinterpret_cast<uint4*>(dh_dst) = hold[j][0];
                            *reinterpret_cast<uint4*>(dh_dst + 16) = hold[j][1];
                        }
                    }(), ...);
                }(std::make_integer_sequence<int, K::NPW>{});
            }
        }
    };

    // Store phase of tile `c`: its [x | h_prev] chunks (in sx) into the LDS image, then (pipelined schedule) the
    // request for those of the tile after it.
    auto request_x = [&](const Tile& r) __attribute__((always_inline)) {
        x_inside = 0u;
        [&]<int... I>(std::integer_sequence<int, I...>) { (load_x(r, std::integral_constant<int, I>{}), ...); }
        (std::make_integer_sequence<int, K::NXP>{});
    };
    auto store_phase = [&](const Tile& c, bool live, const Tile& after) __attribute__((always_inline)) {
        if (live) {
            [&]<int... I>(std::integer_sequence<int, I...>) {
                ([&] {
                    constexpr int j = I;
                    int inf = xinfo[j];
                    asm volatile("" : "+v"(inf));
                    if (inf < 0) {
                        const bool inside = ((x_inside >> j) & 1u) && !(zero_state && ((inf >> 28) & 7) >= K::NCH0);
                        *reinterpret_cast<uint4*>(lds_in + (inf & 0x3ffff)) = inside ? sx[j] : make_uint4(0u, 0u, 0u, 0u);
                    }
                }(), ...);
            }(std::make_integer_sequence<int, K::NXP>{});
        } else if (c.valid) {
            // A masked sample (t >= its length): the state gradients pass through as copies and dx is zero.  Rare,
            // so it reads what it needs itself instead of widening the staged path above.
            for (int i = tid; i < TR * TW * 8; i += kBwdThreads) {
                const int pix = i >> 3, qq = i & 7;
                const int gy = c.ty0 + pix / TW, gx = c.tx0 + (pix & (TW - 1));
                if (gy < p.H && gx < p.W) {
                    const unsigned pixel = (unsigned)((c.b * p.H + gy) * p.W + gx);
                    const size_t so = (size_t)(pixel * 32u + qq * 4u);
                    if (p.dc_prev) {
                        *reinterpret_cast<unsigned*>(static_cast<char*>(p.dc_prev) + so) =
                            no_carry ? 0u : *reinterpret_cast<const unsigned*>(static_cast<const char*>(p.dc_state) + so);
                        *reinterpret_cast<unsigned*>(static_cast<char*>(p.dh_prev) + so) =
                            no_carry ? 0u : *reinterpret_cast<const unsigned*>(static_cast<const char*>(p.dh_state) +
                                                                               (size_t)(pixel * ds_stride + qq * 4u));
                    }
                    if (p.dx && !p.dx_accumulate)
                        for (int ch = 2 * qq; ch < p.dx_channels; ch += 16)
                            *reinterpret_cast<unsigned*>(static_cast<T*>(p.dx) + (size_t)pixel * p.dx_cstride + ch) = 0u;
                }
            }
        }
        if constexpr (PIPE) request_x(after);
    };

#ifdef ISG_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // phase B of tile `c`: data gradient of this wave's tile row (+ its stores), then the weight gradient
    auto phase_b = [&](int dyo, bool e_in, unsigned epixel_u, int tb, int ty0, int tx0) __attribute__((always_inline)) {
        (void)tb, (void)ty0, (void)tx0, (void)e_in, (void)epixel_u;
        BSTAMP(s2);
        // (32-bit byte offsets from the tensors' bases, as everywhere in this kernel: the bases then stay in scalar
        // registers instead of vector register pairs)
        const unsigned dx_off = epixel_u * (unsigned)(p.dx_cstride * 2);
        // the running dx sum the epilogue adds to, requested ahead of the k-loop
        // Requested by every lane of an accumulating launch, on every path (a lane whose channels lie past
        // dx_channels re-reads the first bytes of dx and drops them): a load under a condition is one the compiler
        // cannot count, and it then waits for everything in flight wherever the register comes up again.
        // (ACC: the host picks the instantiation, p.dx != null and p.dx_accumulate)
        const char* pre_base = static_cast<const char*>(p.dx);
        // Row tile 0 of the weight image is stored permuted (dgrad_row_to_channel): a lane's 16 accumulator registers
        // are channels 16 half .. 16 half + 15, two 16-byte pieces.  A wave whose row tile is a later one has four
        // 8-byte pieces (channels 8 a + 4 half .. + 3 of the tile).
        uint4 pre[K::NMT][2];
        // (the kernel on 16-row tiles: lane l holds pixels 16 hh + l % 16 of the tile row, x channels 8 (l / 16) .. + 7)
        const int l16 = lane & 15, kb = lane >> 4;
        unsigned cdx_off[2];
        bool c_in[2];
        if constexpr (K::WCOMPACT) {
            const int gy2 = ty0 + dg_row;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int gx2 = tx0 + 16 * hh + l16;
                c_in[hh] = gy2 < p.H && gx2 < p.W;
                const unsigned pixel = (unsigned)((tb * p.H + min(gy2, p.H - 1)) * p.W + min(gx2, p.W - 1));
                cdx_off[hh] = pixel * (unsigned)(p.dx_cstride * 2) + (unsigned)(kb * 16);
            }
        }
        // (requested after the weight gradient's k-loop, whose fragment buffers need the registers; the data
        // gradient's k-loop is time enough for them to arrive)
        auto request_pre_compact = [&]() __attribute__((always_inline)) {
            if constexpr (ACC && K::WCOMPACT) {
                const bool has = 8 * kb < p.dx_channels;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    pre[0][hh] = *reinterpret_cast<const uint4*>(pre_base + (size_t)(has ? cdx_off[hh] : 0u));
            }
        };
        if constexpr (ACC && !K::WCOMPACT) {
#pragma unroll
            for (int u = 0; u < K::NMT; ++u) {
                const int row0 = 32 * (dg_mt + u);
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (32 * u + 16 * q < C0 || K::NMT == 1) {  // (compile-time: row groups that can hold dx channels at all)
                        if (row0 == 0) {
                            const int cc0 = 16 * half + 8 * q;
                            const bool has = cc0 < C0 && cc0 < p.dx_channels;
                            pre[u][q] = *reinterpret_cast<const uint4*>(pre_base + (size_t)(has ? dx_off + (unsigned)(cc0 * 2) : 0u));
                        } else {
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const int cc0 = row0 + 8 * (2 * q + k) + 4 * half;
                                const bool has = cc0 < C0 && cc0 < p.dx_channels;
                                const uint2 v = *reinterpret_cast<const uint2*>(pre_base + (size_t)(has ? dx_off + (unsigned)(cc0 * 2) : 0u));
                                if (k == 0) pre[u][q].x = v.x, pre[u][q].y = v.y;
                                else pre[u][q].z = v.x, pre[u][q].w = v.y;
                            }
                        }
                    }
            }
        }
        // (the weight gradient first: what the data gradient's epilogue waits for - the old dx values, and with
        // them every load requested before them - then has the whole of phase B to arrive)
#ifndef ISG_BWD_NO_WGRAD
        {
            // k-step = 16 consecutive pixels of a tile row.  Both operands come out of pixel-major LDS images
            // through the transposing read; every address is this lane's fixed base plus a compile-time
            // offset, and the fragments of the next k-step are requested before this one's MFMAs.
            // (fragment sets: the 5-slot kernel has registers for one, the others request WG_AHEAD k-steps ahead)
            constexpr int NBUF = (MT == 1 || K::HMERGE) ? K::WG_AHEAD + 1 : 1;
            bf16x8 bq[NBUF], aq[NBUF][K::NSLOT];
            auto tr_pair = [](const char* a0, auto stride) {  // 8 consecutive pixels of one k-step half
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(a0 + 4 * decltype(stride)::value));
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                return __builtin_bit_cast(bf16x8, both);
            };
            const char* dyb = wg_dy + dyo;
            auto fetch = [&](int ks, int slot) {
                const int r = ks >> 1, x0 = 16 * (ks & 1);
                bq[slot] = tr_pair(dyb + (r * HALO_W + x0) * K::DY_PSTRIDE, std::integral_constant<int, K::DY_PSTRIDE>{});
#pragma unroll
                for (int i = 0; i < K::NSLOT; ++i)
                    if (i < K::NSLOT - 1 || wg_full) {
                        aq[slot][i] = tr_pair(wg_in[i] + (r * HALO_W + x0) * K::RSTRIDE, std::integral_constant<int, K::RSTRIDE>{});
                        if constexpr (K::BIAS_MFMA) {
                            if (i == K::BIAS_SLOT) {  // 1.0 in bf16, eight times
                                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                                const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
                                aq[slot][i] = ones_rows ? __builtin_bit_cast(bf16x8, ones) : aq[slot][i];
                            }
                        }
                    }
            };
            if constexpr (NBUF > 1) {
#pragma unroll
                for (int ks = 0; ks < NBUF - 1; ++ks) fetch(ks, ks);
            }
#pragma unroll
            for (int ks = 0; ks < 2 * TR; ++ks) {
                const int cs = ks % NBUF;
                if constexpr (NBUF > 1) {
                    if (ks + NBUF - 1 < 2 * TR) fetch(ks + NBUF - 1, (ks + NBUF - 1) % NBUF);
                } else {
                    fetch(ks, 0);
                }
#pragma unroll
                for (int i = 0; i < K::NSLOT; ++i)
                    if (i < K::NSLOT - 1 || wg_full)
                        wacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[cs][i], bq[cs], wacc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        BSTAMP(s4);
        BSTAMP_ADD(4, s2, s4);  // weight-gradient k-loop
        request_pre_compact();
        // (a wave with two row tiles of its tile row takes them one after the other: both at once would need a second
        // set of accumulators the kernel has no registers for; the dY fragments are read again instead)
#pragma unroll
        for (int u = 0; u < (K::WCOMPACT ? 0 : K::NMT); ++u) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#ifndef ISG_BWD_NO_DGRAD  // diagnostic builds leave a phase out to see what the others take
            {
                const char* bb = lds_dyh + dyo + (dg_row * HALO_W + px_l) * K::DY_PSTRIDE + half * 16;
                // A fragments: one 16-byte slot per lane and k-step; the compact second row tile one slot per
                // (k half, row % 16)
                const bool second = K::NMT == 2 && u == 1;
                const char* aw = !second ? lds_w + (size_t)dg_mt * K::NS * 1024 + lane * 16
                                 : K::WCOMPACT ? lds_w + K::NS * 1024 + ((lane >> 5) * 16 + (lane & 15)) * 16
                                               : lds_w + K::NS * 1024 + lane * 16;
                const int astep = second ? K::W1_STEP : 1024;
                // One MFMA (32 cycles) per k-step against an LDS round trip of 64 - 130: the fragments are requested
                // DG_AHEAD k-steps ahead, as far as the kernel's registers allow.
                constexpr int D = K::DG_AHEAD, NB = D + 1;
                bf16x8 a[NB], b[NB];
                auto fetch = [&](int s, int slot) {
                    const int tap = s >> 2, qk = s & 3;
                    a[slot] = *reinterpret_cast<const bf16x8*>(aw + s * astep);
                    b[slot] = *reinterpret_cast<const bf16x8*>(bb + ((tap / 3) * HALO_W + (tap % 3)) * K::DY_PSTRIDE + qk * 32);
                };
#pragma unroll
                for (int s = 0; s < D; ++s) fetch(s, s);
#pragma unroll
                for (int s = 0; s < K::NS; ++s) {
                    if (s + D < K::NS) fetch(s + D, (s + D) % NB);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % NB], b[s % NB], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
            // The old dx values are added to ALL accumulators before the first store goes out: a store between two
            // of these uses would make the wait for the next old value also wait for that store's completion (the
            // counter is in order), one memory round trip per piece of dx.
            const int row0 = 32 * (dg_mt + u);
            if constexpr (ACC) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (32 * u + 16 * q < C0 || K::NMT == 1) {
                        const uint4 o = pre[u][q];
                        const unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int cc0 = row0 == 0 ? 16 * half + 8 * q : row0 + 8 * (2 * q + k) + 4 * half;
                            const bool has = cc0 < C0 && cc0 < p.dx_channels;
                            const int r0 = 8 * q + 4 * k;
                            acc[r0] += has ? bf_lo(w[2 * k]) : 0.0f, acc[r0 + 1] += has ? bf_hi(w[2 * k]) : 0.0f;
                            acc[r0 + 2] += has ? bf_lo(w[2 * k + 1]) : 0.0f, acc[r0 + 3] += has ? bf_hi(w[2 * k + 1]) : 0.0f;
                        }
                    }
            }
            // (the sums exist HERE, in registers: without this the additions sink into the conditional stores below)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(acc[r]));
            __builtin_amdgcn_sched_barrier(0);
            if (row0 == 0) {
                // 16 consecutive channels per lane, all of them x channels or all of them h_prev channels (C0 is a
                // multiple of 16): the destination is picked per lane and both halves of the wave store together
                const bool to_dx = C0 >= 32 || half == 0;
                char* dst = to_dx ? static_cast<char*>(p.dx) + (size_t)(dx_off + (unsigned)(32 * half))
                                  : static_cast<char*>(p.dh_prev) + (size_t)(epixel_u * (unsigned)(F_ * 2) + (unsigned)(32 * half - 2 * C0));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cc0 = 16 * half + 8 * q;
                    const bool ok = to_dx ? (p.dx != nullptr && cc0 < p.dx_channels) : (p.dh_prev != nullptr && cc0 < C0 + F_);
                    if (e_in && ok) {
                        uint4 o;
                        o.x = pack_bf2(acc[8 * q], acc[8 * q + 1]), o.y = pack_bf2(acc[8 * q + 2], acc[8 * q + 3]);
                        o.z = pack_bf2(acc[8 * q + 4], acc[8 * q + 5]), o.w = pack_bf2(acc[8 * q + 6], acc[8 * q + 7]);
                        *reinterpret_cast<uint4*>(dst + 16 * q) = o;
                    }
                }
            } else if (e_in) {
#pragma unroll
                for (int a4 = 0; a4 < 4; ++a4) {
                    float v[4] = {acc[4 * a4], acc[4 * a4 + 1], acc[4 * a4 + 2], acc[4 * a4 + 3]};
                    const int cc0 = row0 + 8 * a4 + 4 * half;
                    if (cc0 < C0) {
                        if (p.dx && cc0 < p.dx_channels)
                            store4(reinterpret_cast<T*>(static_cast<char*>(p.dx) + (size_t)(dx_off + (unsigned)(cc0 * 2))), v);
                    } else if (cc0 < C0 + F_ && p.dh_prev) {
                        store4(reinterpret_cast<T*>(static_cast<char*>(p.dh_prev) + (size_t)(epixel_u * (unsigned)(F_ * 2) + (unsigned)((cc0 - C0) * 2))), v);
                    }
                }
            }
        }
        if constexpr (K::WCOMPACT) {
            // 48 real rows (32 x channels, 16 h_prev channels) as three 16-row tiles on 16 x 16 x 32 MFMAs, two per
            // tile and 32-deep k-step (pixels 0-15 and 16-31 of the tile row).  Against a 32-row tile plus a 16-row
            // one this reads each dY fragment once for all 48 rows: 5 KB of LDS per k-step and wave instead of 7, in a
            // kernel whose phases are bounded by what LDS delivers.  The weight image is in the instruction's A
            // layout: lane l reads 16 bytes at l * 16 of each 1 KB k-step (row l % 16, k = 8 (l / 16) .. + 7).
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 acc2[3][2];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) acc2[j][hh] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const bool want_dh = p.dh_prev != nullptr;  // (the third tile is dh_prev alone: skipped when nobody wants it)
#ifndef ISG_BWD_NO_DGRAD
            {
                const char* bb = lds_dyh + dyo + (dg_row * HALO_W + l16) * K::DY_PSTRIDE + kb * 16;
                const char* aw = lds_w + lane * 16;
                // (dY fragments double-buffered; a weight fragment is requested again as soon as its two MFMAs of this
                // k-step have issued - one register set for the three of them)
                bf16x8 a[3], b[2][2];
                auto fetch_a = [&](int s32, int j) {
                    a[j] = *reinterpret_cast<const bf16x8*>(aw + j * (K::NS * 512) + s32 * 1024);
                };
                auto fetch_b = [&](int s32, int slot) {
                    const int tap = s32 >> 1;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        b[slot][hh] = *reinterpret_cast<const bf16x8*>(
                            bb + ((tap / 3) * HALO_W + (tap % 3) + 16 * hh) * K::DY_PSTRIDE + (s32 & 1) * 64);
                };
                fetch_b(0, 0);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < 2 || want_dh) fetch_a(0, j);
#pragma unroll
                for (int s32 = 0; s32 < K::NS / 2; ++s32) {
                    if (s32 + 1 < K::NS / 2) fetch_b(s32 + 1, (s32 + 1) & 1);
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (j < 2 || want_dh) {
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh)
                                acc2[j][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], b[s32 & 1][hh], acc2[j][hh], 0, 0, 0);
                            if (s32 + 1 < K::NS / 2) fetch_a(s32 + 1, j);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
            }
#endif
            // lane l: pixel 16 hh + l % 16 of the tile row; x channels 8 (l / 16) .. + 7 (tiles 0 and 1), h_prev
            // channels 4 (l / 16) .. + 3 (tile 2).  The old dx values are added to all accumulators before the first
            // store goes out (the memory counter is in order).
            float dxv[2][8];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dxv[hh][i] = acc2[0][hh][i], dxv[hh][4 + i] = acc2[1][hh][i];
                if constexpr (ACC) {
                    const bool has = 8 * kb < p.dx_channels;
                    const unsigned w[4] = {pre[0][hh].x, pre[0][hh].y, pre[0][hh].z, pre[0][hh].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        dxv[hh][2 * i] += has ? bf_lo(w[i]) : 0.0f, dxv[hh][2 * i + 1] += has ? bf_hi(w[i]) : 0.0f;
                }
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(dxv[hh][i]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (c_in[hh] && p.dx != nullptr && 8 * kb < p.dx_channels) {
                    uint4 o;
                    o.x = pack_bf2(dxv[hh][0], dxv[hh][1]), o.y = pack_bf2(dxv[hh][2], dxv[hh][3]);
                    o.z = pack_bf2(dxv[hh][4], dxv[hh][5]), o.w = pack_bf2(dxv[hh][6], dxv[hh][7]);
                    *reinterpret_cast<uint4*>(static_cast<char*>(p.dx) + (size_t)cdx_off[hh]) = o;
                }
            }
            if (want_dh) {
                const int gy2 = ty0 + dg_row;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int gx2 = tx0 + 16 * hh + l16;
                    if (c_in[hh]) {
                        const unsigned pixel = (unsigned)((tb * p.H + gy2) * p.W + gx2);
                        float v[4] = {acc2[2][hh][0], acc2[2][hh][1], acc2[2][hh][2], acc2[2][hh][3]};
                        store4(reinterpret_cast<T*>(static_cast<char*>(p.dh_prev) + (size_t)(pixel * (unsigned)(F_ * 2) + (unsigned)(kb * 8))), v);
                    }
                }
            }
        }
        BSTAMP(s5);
        BSTAMP_ADD(3, s4, s5);  // data gradient: k-loops and stores
    };

    Tile cur = tile_ref(0), nxt = tile_ref(1);
    __syncthreads();  // the zero fill of the [x | h_prev] image is done before any wave's store phase writes into it
    load_meta(cur);
    [&]<int... I>(std::integer_sequence<int, I...>) { (load_pw(cur, std::integral_constant<int, I>{}), ...); }
    (std::make_integer_sequence<int, K::NPW>{});
    [&]<int... I>(std::integer_sequence<int, I...>) { (load_x(cur, std::integral_constant<int, I>{}), ...); }
    (std::make_integer_sequence<int, K::NXP>{});
    bool cur_live = false;
    int cur_buf = 0;                // dY image of tile `cur`
    auto run_b = [&]() __attribute__((always_inline)) {
        if (cur_live) {
            const int egy = cur.ty0 + dg_row, egx = cur.tx0 + px_l;
            const bool e_in = egy < p.H && egx < p.W;
            const unsigned epixel = (unsigned)((cur.b * p.H + min(egy, p.H - 1)) * p.W + min(egx, p.W - 1));
            phase_b(cur_buf * K::DYH_BYTES, e_in, epixel, cur.b, cur.ty0, cur.tx0);
        }
    };

    BSTAMP(t_begin);
    if constexpr (PIPE) {
        gate_backward(cur, nxt, lds_dyh);  // dY image 0 of tile 0; the operands of tile 1 requested
        cur_live = m_live;
        const bool m_first = wave < 4;  // one wave of each kind per SIMD (waves w and w + 4 share one)

        // The join between two intervals: every wave is done with the LDS images of tile `cur`; the next tile
        // becomes `cur`, its [x | h_prev] chunks go to LDS and those of the tile after it are requested.
        Tile nxt2 = nxt;
        auto join = [&]() __attribute__((always_inline)) {
            BSTAMP(s6);
            __syncthreads();
            BSTAMP(s7);
            cur = nxt, nxt = nxt2, cur_live = m_live, cur_buf ^= K::NDY - 1;
            flush_hold();
            store_phase(cur, cur_live, nxt);
            BSTAMP(s8);
            __syncthreads();
            BSTAMP(s9);
            BSTAMP_ADD(5, s6, s7);  // barrier after phase B / the gate backward
            BSTAMP_ADD(0, s7, s8);  // store phase
            BSTAMP_ADD(1, s8, s9);  // barrier after it
        };
        // Both orders out of ONE copy of the gate backward (two copies cost more registers than there are): the
        // loop body is gate backward -> phase B for every wave, and what differs is where the join falls.  Waves
        // 0-3 join after phase B: their interval is [gate backward of tile k + 1, phase B of tile k].  Waves 4-7
        // join between the two, so that phase B of a pass already works on tile k + 1: their interval is [phase B
        // of tile k, gate backward of tile k + 1].  Every wave passes the same number of barriers.
        // DEFER_LAST: the operands of the last gate-backward pass are requested after phase B instead of inside the
        // gate backward before it - phase B is where registers are shortest, and that pass still has the passes
        // before it (and, for waves 0-3, the join) to arrive.  The tile is the one the last gate backward requested.
        auto request_last_pass = [&](const Tile& r) __attribute__((always_inline)) {
            if constexpr (K::DEFER_LAST) load_pw(r, std::integral_constant<int, K::NPW - 1>{});
        };
        flush_hold();
        store_phase(cur, cur_live, nxt);
        __syncthreads();  // (also the weight image)
        if (!m_first) run_b();
        request_last_pass(nxt);
        for (int k = 0; k < n_iter; ++k) {
            nxt2 = tile_ref(k + 2);
            BSTAMP(s1c);
            gate_backward(nxt, nxt2, lds_dyh + (cur_buf ^ (K::NDY - 1)) * K::DYH_BYTES);
            BSTAMP(s1d);
            BSTAMP_ADD(2, s1c, s1d);  // gate backward of the next tile
            if (m_first) {
                run_b();
                request_last_pass(nxt2);
            }
            join();
            if (!m_first) {
                run_b();
                request_last_pass(nxt);
            }
        }
    } else {
        for (int k = 0; k < n_iter; ++k) {
            BSTAMP(s0);
            gate_backward(cur, nxt, lds_dyh);  // this tile's dY; the operands of the next tile requested
            cur_live = m_live;
            BSTAMP(s1);
            store_phase(cur, cur_live, nxt);
            BSTAMP(s2);
            __syncthreads();  // (first pass: also the weight image)
            BSTAMP(s3);
            run_b();
            // (requested here, not in the store phase: phase B is where registers are shortest, and the chunks still
            // have the next tile's gate backward to arrive)
            request_x(nxt);
            BSTAMP(s6);
            __syncthreads();
            BSTAMP(s7);
            BSTAMP_ADD(2, s0, s1);  // gate backward
            BSTAMP_ADD(0, s1, s2);  // store phase
            BSTAMP_ADD(1, s2, s3);  // barrier after it
            BSTAMP_ADD(5, s6, s7);  // barrier after phase B
            cur = nxt, nxt = tile_ref(k + 2);
        }
    }
#ifdef ISG_STAMPS
    BSTAMP(t_end);
    stamp_sum[6] = t_end - t_begin;
    stamp_sum[7] = n_iter;
    if (g_bwd_stamp_buf != nullptr && lane == 0 && (wave == 0 || wave == 7))
        for (int i = 0; i < 8; ++i) g_bwd_stamp_buf[(blockIdx.x * 2 + (wave == 7)) * 8 + i] = stamp_sum[i];
#endif

    // ---- this workgroup's partial sums ----
    float* slab = p.slab + (size_t)blockIdx.x * 9 * (MT * 32) * NG;
#pragma unroll
    for (int i = 0; i < K::NSLOT; ++i) {
        if (i < wg_cnt) {
            const int idx = wg_first + i;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // row of the 32-row tile
                int tap, cc;
                if constexpr (K::HMERGE) {
                    const bool htile = idx >= 9;
                    tap = htile ? 2 * (idx - 9) + (row >> 4) : idx;
                    cc = htile ? 32 + (row & 15) : row;
                } else {
                    tap = MT == 1 ? idx : idx >> 1, cc = (MT == 1 ? 0 : 32 * (idx & 1)) + row;
                }
                if (tap < 9) {
                    float* dst = slab + ((size_t)tap * (MT * 32) + cc) * NG + 32 * wg_n + (lane & 31);
                    *dst = p.slab_accumulate ? *dst + wacc[i][reg] : wacc[i][reg];
                }
            }
        }
    }
    if constexpr (K::BIAS_MFMA) {
        // register 8 of the ones-fed tile = its row 16 + 4 half: the bias sums of this wave's 32 dY columns
        if (wg_grp == K::BIAS_GROUP && half == 0) {
            float* dst = p.bias_slab + (size_t)blockIdx.x * NG + 32 * wg_n + lane;
            const float v = wacc[K::BIAS_SLOT][8];
            *dst = p.slab_accumulate ? *dst + v : v;
        }
        return;
    }
    // the bias sums of the 128 threads that share q, through LDS (every wave is past its last tile), in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds_dyh);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[i * kBwdThreads + tid] = bacc[i];
    __syncthreads();
    if (tid < NG) {
        const int qq = tid >> 4, i = tid & 15;
        float v = 0.0f;
        for (int k = 0; k < kBwdThreads / 4; ++k) v += red[i * kBwdThreads + 4 * k + qq];
        float* dst = p.bias_slab + (size_t)blockIdx.x * NG + tid;
        *dst = p.slab_accumulate ? *dst + v : v;
    }
}

// dw_x[tap][C + i][:] (+)= sum over samples of the dY rows the fused kernel filed per (sample, frame, tap).
// Two launches: kHistGroups x (9 hist) workgroups each sum a slice of the batch (one workgroup per row walking
// thousands of samples in series was latency-bound: 0.19 ms at B = 2,048), then one workgroup per row adds the
// slices in a fixed order.
constexpr int kHistGroups = 32;

__host__ __device__ inline long long hist_dy_bytes(int B, int hist) {
    return (((long long)B * hist * 9 * NG * 2) + 255) & ~255LL;
}

__global__ __launch_bounds__(256) void hist_side_partial_kernel(const __bf16* __restrict__ hist_dy, int B, int hist,
                                                                float* __restrict__ partial) {
    __shared__ float red[4][NG];
    const int row = blockIdx.x, g = blockIdx.y;  // row = tap * hist + i
    const int tap = row / hist, i = row - tap * hist;
    const int k = threadIdx.x & (NG - 1), sg = threadIdx.x >> 6;
    float sum = 0.0f;
    for (int b = g * 4 + sg; b < B; b += 4 * kHistGroups) sum += (float)hist_dy[(((size_t)b * hist + i) * 9 + tap) * NG + k];
    red[sg][k] = sum;
    __syncthreads();
    if (sg == 0) partial[((size_t)g * 9 * hist + row) * NG + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

__global__ __launch_bounds__(NG) void hist_side_finish_kernel(const float* __restrict__ partial, int hist, int C,
                                                              int accumulate, float* __restrict__ dw_x) {
    const int row = blockIdx.x, tap = row / hist, i = row - tap * hist, k = threadIdx.x;
    float sum = 0.0f;
    for (int g = 0; g < kHistGroups; ++g) sum += partial[((size_t)g * 9 * hist + row) * NG + k];
    float* dst = dw_x + ((size_t)tap * (C + hist) + C + i) * NG + (k & 3) * F_ + (k >> 2);
    *dst = accumulate ? *dst + sum : sum;
}

template <int C0, int MT, int TR, bool PIPE, bool ACC, int EDGE, bool HOLD, bool CB = false>
int launch_bwd_edge(BwdParams p, hipStream_t stream) {
    using K = FB<C0, MT, TR, PIPE, HOLD>;
    if constexpr (!CB) {  // c_prev stored as bf16: instantiations of their own
        if (p.cprev_bf16) return launch_bwd_edge<C0, MT, TR, PIPE, ACC, EDGE, HOLD, true>(p, stream);
    }
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TR - 1) / TR;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convlstm_bwd_fused_kernel<C0, MT, TR, PIPE, ACC, EDGE, HOLD, CB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    // one slab per workgroup, as many as isg_convlstm_wgrad_finish folds: its count goes by 8-row tiles
    const int ntiles8 = p.tiles_x * ((p.H + TH - 1) / TH) * p.B;
    const int grid = ntiles8 < kWgradMaxBlocks ? ntiles8 : kWgradMaxBlocks;
    hipLaunchKernelGGL((convlstm_bwd_fused_kernel<C0, MT, TR, PIPE, ACC, EDGE, HOLD, CB>), dim3(grid), dim3(kBwdThreads),
                       K::LDS_BYTES, stream, p);
    return isg::check_launch("convlstm_bwd_fused_kernel");
}

template <int C0, int MT, int TR, bool PIPE, bool ACC, bool HOLD>
int launch_bwd_acc(const BwdParams& p, hipStream_t stream) {
    const int edge = (p.dh_state == nullptr ? 1 : 0) | (p.c_prev == nullptr ? 2 : 0);
    switch (edge) {
        case 0: return launch_bwd_edge<C0, MT, TR, PIPE, ACC, 0, HOLD>(p, stream);
        case 1: return launch_bwd_edge<C0, MT, TR, PIPE, ACC, 1, HOLD>(p, stream);
        case 2: return launch_bwd_edge<C0, MT, TR, PIPE, ACC, 2, HOLD>(p, stream);
        default: return launch_bwd_edge<C0, MT, TR, PIPE, ACC, 3, HOLD>(p, stream);
    }
}

template <int C0, int MT, int TR, bool PIPE, bool HOLD = false>
int launch_bwd(const BwdParams& p, int num_cus, hipStream_t stream) {
    (void)num_cus;
    return p.dx != nullptr && p.dx_accumulate ? launch_bwd_acc<C0, MT, TR, PIPE, true, HOLD>(p, stream)
                                             : launch_bwd_acc<C0, MT, TR, PIPE, false, HOLD>(p, stream);
}

int bwd_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

bool bwd_supported(int Cs, int F, int dtype, int B, int H, int W, int hist) {
    return F == F_ && dtype == ISG_BF16 && (Cs == 16 || Cs == 32 || Cs == 48) && hist >= 0 && hist <= 2 && B >= 1 && H >= 1 && W >= 1 &&
           (long long)B * H * W * 128 < (1LL << 32);
}

}  // namespace

#ifdef ISG_STAMPS
extern "C" int isg_debug_set_bwd_stamp_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int isg_convlstm_bwd_fused_supported(int Cs, int F, int dtype, int B, int H, int W, int hist) {
    return bwd_supported(Cs, F, dtype, B, H, W, hist) ? 1 : 0;
}

extern "C" long long isg_convlstm_bwd_fused_hist_bytes(int B, int hist) {
    // the dY rows per (sample, frame, tap), bf16, then the batch slices' partial sums
    return hist_dy_bytes(B, hist) + (long long)kHistGroups * 9 * hist * NG * 4;
}

namespace {
int bwd_fused_impl(const void* gates, const void* c_prev_v, const void* h_cur, const void* dh_above,
                   long long dh_above_batch_stride, int dh_above_pool, const void* dh_state, const void* dc_state,
                   const void* x, const void* h_prev, const void* packed_dgrad, int B, int H, int W, int C, int Cs, int F,
                   int dtype, const int32_t* durations, int t, const int32_t* hist_points, int hist,
                   long long hist_batch_stride, void* hist_scratch, void* dc_prev, void* dh_prev, void* dx,
                   int dx_channels, int dx_channel_stride, int dx_accumulate, void* workspace, int mode, float* dw_x,
                   void* stream, int h_cur_pixel_stride, int dh_state_pixel_stride, int dh_above_pixel_stride) {
    {
        const int strides[3] = {h_cur_pixel_stride, dh_state_pixel_stride, dh_above_pixel_stride};
        for (int st : strides)
            ISG_REQUIRE(st >= F_ && st <= 64 && st % 8 == 0 && (Cs == 48 || st == F_), ISG_EINVAL,
                        "isg_convlstm_bwd_fused: pixel stride %d of h_cur / dh_state / dh_above: a multiple of 8 in "
                        "[16, 64], and 16 unless Cs = 48", st);
    }
    // (ISG_HIST_DEFER: the caller zeroes the history scratch once and folds it once, see isg_convlstm_bwd_fused_hist_finish)
    const bool hist_defer = (mode & ISG_HIST_DEFER) != 0;
    const bool cprev_bf16 = (mode & ISG_CPREV_BF16) != 0;
    mode &= ~(ISG_HIST_DEFER | ISG_CPREV_BF16);
    const float* c_prev = static_cast<const float*>(c_prev_v);  // (typed as float in the parameter block)
    ISG_REQUIRE(bwd_supported(Cs, F, dtype, B, H, W, hist), ISG_EINVAL,
                "isg_convlstm_bwd_fused: needs bf16, F = 16, Cs 16, 32 or 48, hist <= 2 (got Cs=%d F=%d dtype=%d hist=%d): ask "
                "isg_convlstm_bwd_fused_supported first", Cs, F, dtype, hist);
    ISG_REQUIRE(gates && h_cur && x && packed_dgrad && workspace && dh_above, ISG_EINVAL,
                "isg_convlstm_bwd_fused: null pointer");
    ISG_REQUIRE((dh_state == nullptr) == (dc_state == nullptr), ISG_EINVAL,
                "isg_convlstm_bwd_fused: dh_state / dc_state are both given or both NULL (no gradient carried in)");
    ISG_REQUIRE((c_prev == nullptr) == (h_prev == nullptr) && (dc_prev == nullptr) == (dh_prev == nullptr), ISG_EINVAL,
                "isg_convlstm_bwd_fused: c_prev / h_prev are both given or both NULL (the step out of the zero state), "
                "and so are dc_prev / dh_prev (NULL: not wanted)");
    ISG_REQUIRE(dc_prev == nullptr || (dc_prev != dc_state && dh_prev != dh_state), ISG_EINVAL,
                "isg_convlstm_bwd_fused: dc_prev / dh_prev must not alias dc_state / dh_state (neighbouring tiles read "
                "the incoming gradients of each other's border pixels)");
    ISG_REQUIRE(mode == ISG_WGRAD_DEFER_FIRST || mode == ISG_WGRAD_DEFER, ISG_EINVAL,
                "isg_convlstm_bwd_fused: the partial sums stay in the workspace (mode ISG_WGRAD_DEFER_FIRST or ISG_WGRAD_DEFER)");
    ISG_REQUIRE(C >= 1 && C <= Cs, ISG_EINVAL, "isg_convlstm_bwd_fused: bad channel count");
    ISG_REQUIRE(dh_above_pool >= 1 && (dh_above_pool & (dh_above_pool - 1)) == 0 && H % dh_above_pool == 0 &&
                    W % dh_above_pool == 0, ISG_EINVAL,
                "isg_convlstm_bwd_fused: pool %d must be a power of two that divides the %dx%d grid", dh_above_pool, H, W);
    ISG_REQUIRE(!dx || (long long)B * H * W * dx_channel_stride * 2 < (1LL << 32), ISG_EINVAL,
                "isg_convlstm_bwd_fused: dx is addressed with 32-bit byte offsets");
    int pool_shift = 0;
    while ((1 << pool_shift) < dh_above_pool) ++pool_shift;
    ISG_REQUIRE(!dx || (dx_channels >= 8 && dx_channels % 8 == 0 && dx_channels <= Cs && dx_channel_stride % 8 == 0 &&
                        dx_channel_stride >= dx_channels),
                ISG_EINVAL, "isg_convlstm_bwd_fused: dx channels/stride must be multiples of 8 within the input width");
    ISG_REQUIRE(hist == 0 || (hist_points && hist_scratch && dw_x), ISG_EINVAL,
                "isg_convlstm_bwd_fused: history points, scratch and dw_x are needed when hist > 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int MT = (Cs + F_ + 31) / 32;
    BwdParams p = {};
    p.gates = gates, p.c_prev = c_prev, p.cprev_bf16 = cprev_bf16 ? 1 : 0, p.h_cur = h_cur;
    p.h_cur_pstride = h_cur_pixel_stride * 2, p.dh_state_pstride = dh_state_pixel_stride * 2;
    p.dh_above_pstride = dh_above_pixel_stride * 2;
    p.dh_above = dh_above, p.dh_above_bstride = dh_above_batch_stride, p.pool_shift = pool_shift;
    p.dh_state = dh_state, p.dc_state = dc_state, p.x = x, p.h_prev = h_prev;
    p.B = B, p.H = H, p.W = W, p.durations = durations, p.t = t, p.wimg = packed_dgrad;
    p.dc_prev = dc_prev, p.dh_prev = dh_prev;
    p.dx = dx, p.dx_channels = dx ? dx_channels : 0, p.dx_cstride = dx_channel_stride, p.dx_accumulate = dx_accumulate;
    p.slab = static_cast<float*>(workspace);
    p.bias_slab = p.slab + (size_t)kWgradMaxBlocks * 9 * MT * 32 * NG;
    p.slab_accumulate = mode == ISG_WGRAD_DEFER;
    p.hist_points = hist_points, p.hist = hist, p.hist_bstride = hist_batch_stride, p.hist_dy = hist_scratch;
    if (hist > 0 && !hist_defer) {
        hipError_t e = hipMemsetAsync(hist_scratch, 0, (size_t)hist_dy_bytes(B, hist), s);
        ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
#ifdef ISG_BWD_L0_TR4  // diagnostic build: the 48-row kernel on pipelined 4-row tiles
    int rc = Cs == 16   ? launch_bwd<16, 1, 8, true>(p, bwd_num_cus(), s)
             : Cs == 32 ? launch_bwd<32, 2, 4, true>(p, bwd_num_cus(), s)
                        : launch_bwd<48, 2, 4, true>(p, bwd_num_cus(), s);
#else
#ifdef ISG_BWD_L0_NOHOLD  // diagnostic build: the 48-row kernel on the schedule without overlap
    int rc = Cs == 16   ? launch_bwd<16, 1, 8, true>(p, bwd_num_cus(), s)
             : Cs == 32 ? launch_bwd<32, 2, 8, false>(p, bwd_num_cus(), s)
                        : launch_bwd<48, 2, 4, true>(p, bwd_num_cus(), s);
#else
    int rc = Cs == 16   ? launch_bwd<16, 1, 8, true>(p, bwd_num_cus(), s)
             : Cs == 32 ? launch_bwd<32, 2, 8, true, true>(p, bwd_num_cus(), s)
                        : launch_bwd<48, 2, 4, true>(p, bwd_num_cus(), s);
#endif
#endif
    if (rc != ISG_OK) return rc;
    if (hist > 0 && !hist_defer) {
        float* partial = reinterpret_cast<float*>(static_cast<char*>(hist_scratch) + hist_dy_bytes(B, hist));
        hipLaunchKernelGGL(hist_side_partial_kernel, dim3(9 * hist, kHistGroups), dim3(256), 0, s,
                           static_cast<const __bf16*>(hist_scratch), B, hist, partial);
        rc = isg::check_launch("hist_side_partial_kernel");
        if (rc != ISG_OK) return rc;
        hipLaunchKernelGGL(hist_side_finish_kernel, dim3(9 * hist), dim3(NG), 0, s, partial, hist, C,
                           mode == ISG_WGRAD_DEFER ? 1 : 0, dw_x);
        rc = isg::check_launch("hist_side_finish_kernel");
    }
    return rc;
}
}  // namespace

extern "C" int isg_convlstm_bwd_fused(const void* gates, const void* c_prev, const void* h_cur, const void* dh_above,
                                      long long dh_above_batch_stride, int dh_above_pool, const void* dh_state,
                                      const void* dc_state, const void* x, const void* h_prev, const void* packed_dgrad,
                                      int B, int H, int W, int C, int Cs, int F, int dtype, const int32_t* durations,
                                      int t, const int32_t* hist_points, int hist, long long hist_batch_stride,
                                      void* hist_scratch, void* dc_prev, void* dh_prev, void* dx, int dx_channels,
                                      int dx_channel_stride, int dx_accumulate, void* workspace, int mode, float* dw_x,
                                      void* stream) {
    return bwd_fused_impl(gates, c_prev, h_cur, dh_above, dh_above_batch_stride, dh_above_pool, dh_state, dc_state, x,
                          h_prev, packed_dgrad, B, H, W, C, Cs, F, dtype, durations, t, hist_points, hist,
                          hist_batch_stride, hist_scratch, dc_prev, dh_prev, dx, dx_channels, dx_channel_stride,
                          dx_accumulate, workspace, mode, dw_x, stream, F_, F_, F_);
}

extern "C" int isg_convlstm_bwd_fused_strided(const void* gates, const void* c_prev, const void* h_cur,
                                              int h_cur_pixel_stride, const void* dh_above,
                                              long long dh_above_batch_stride, int dh_above_pixel_stride,
                                              int dh_above_pool, const void* dh_state,
                                              int dh_state_pixel_stride, const void* dc_state, const void* x,
                                              const void* h_prev, const void* packed_dgrad, int B, int H, int W, int C,
                                              int Cs, int F, int dtype, const int32_t* durations, int t, void* dc_prev,
                                              void* dh_prev, void* dx, int dx_channels, int dx_channel_stride,
                                              int dx_accumulate, void* workspace, int mode, void* stream) {
    return bwd_fused_impl(gates, c_prev, h_cur, dh_above, dh_above_batch_stride, dh_above_pool, dh_state, dc_state, x,
                          h_prev, packed_dgrad, B, H, W, C, Cs, F, dtype, durations, t, nullptr, 0, 0, nullptr, dc_prev,
                          dh_prev, dx, dx_channels, dx_channel_stride, dx_accumulate, workspace, mode, nullptr, stream,
                          h_cur_pixel_stride, dh_state_pixel_stride, dh_above_pixel_stride);
}

extern "C" int isg_convlstm_bwd_fused_hist_begin(void* hist_scratch, int samples, int hist, void* stream) {
    ISG_REQUIRE(hist_scratch && samples > 0 && hist > 0 && hist <= 2, ISG_EINVAL, "isg_convlstm_bwd_fused_hist_begin: bad argument");
    hipError_t e = hipMemsetAsync(hist_scratch, 0, (size_t)hist_dy_bytes(samples, hist), static_cast<hipStream_t>(stream));
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    return ISG_OK;
}

extern "C" int isg_convlstm_bwd_fused_hist_finish(void* hist_scratch, int samples, int hist, int C, int accumulate,
                                                  float* dw_x, void* stream) {
    ISG_REQUIRE(hist_scratch && dw_x && samples > 0 && hist > 0 && hist <= 2 && C > 0, ISG_EINVAL,
                "isg_convlstm_bwd_fused_hist_finish: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(hist_scratch) + hist_dy_bytes(samples, hist));
    hipLaunchKernelGGL(hist_side_partial_kernel, dim3(9 * hist, kHistGroups), dim3(256), 0, s,
                       static_cast<const __bf16*>(hist_scratch), samples, hist, partial);
    int rc = isg::check_launch("hist_side_partial_kernel");
    if (rc != ISG_OK) return rc;
    hipLaunchKernelGGL(hist_side_finish_kernel, dim3(9 * hist), dim3(NG), 0, s, partial, hist, C, accumulate ? 1 : 0, dw_x);
    return isg::check_launch("hist_side_finish_kernel");
}
