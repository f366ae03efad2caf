// bf16 fast path of the two convolution-shaped ConvLSTM kernels (forward gates + LSTM, data gradient).
//
// Why a second kernel: at 224 x 224 the cell is HBM-bound (about 380 B move per pixel-step against
// 55 kFLOP), so what matters is keeping loads in flight while the matrix pipe and the VALU both work.
// One 512-thread workgroup per CU holds the weight image in LDS once and runs TWO groups of 4 waves
// in ping-pong on separate input buffers: while group A issues its 9 x SPT k-steps of MFMA on tile
// a_i, group B runs the epilogue of b_(i-1) (gate math, stores) and stages b_i; a workgroup barrier
// flips the roles.  Every SIMD hosts one wave of each group, so MFMA of one overlaps VALU/VMEM of the
// other.  Staging is split: the global loads of the next tile are issued BEFORE the epilogue and
// their LDS writes come after it, which keeps ~40 KB per CU in flight during the gate math.
// The k-loop is fully unrolled with the next step's fragments read while the current MFMAs issue.
#include <utility>

#include "isg_convlstm_shared.h"

using namespace isgc;

namespace {

constexpr int kPPThreads = 512;

// Diagnostic build only (-DISG_STAMPS, see tools/stamp_conv.py): per-phase cycle sums of wave 0 of each
// group, written to a buffer that nothing else reads.  In the shipped library no stamp executes.
#ifdef ISG_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(var) const unsigned long long var = stamp()
#define STAMP_ADD(slot, a, b) stamp_sum[slot] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(slot, a, b)
#endif

struct TileRef {
    int b, ty0, tx0;
    bool valid, live;  // live is known once the tile's staged words have landed (stage_store)
    int hp[4];         // history points (x0, y0, x1, y1) of the tile's sample, same
};

// Read-only, wave-uniform words (sequence lengths, history points) go through the scalar cache: a
// vector load here would also make the next s_waitcnt drain every store still in flight.
__device__ __forceinline__ int scalar_load(const int* base, long long index) {
    const long long i = (long long)__builtin_amdgcn_readfirstlane((int)index);
    return *reinterpret_cast<const __attribute__((address_space(4))) int*>(
        (const __attribute__((address_space(4))) char*)(unsigned long long)(base) + i * 4);
}

// NPT = tile rows per wave: 2 -> the 8 x 32 tile; 1 -> a 4 x 32 tile, whose smaller halo image leaves LDS room
// for a two-row-tile weight image over 64 staged channels (the 48-row data gradient of layer 0 in ONE launch; the
// forward cell on a 48-channel input, which has no pooled output: the pool's reduction goes by 8-row tiles)
template <int C0, int C1, int MT, int MODE, int NPT = 2>
struct PP {
    static constexpr int TH_ = 4 * NPT;
    static constexpr int NPX_ = (TH_ + 2) * HALO_W;
    static constexpr int CPT = C0 + C1;          // staged channels per pixel
    static constexpr int NCH0 = C0 / 8;          // 16-byte chunks of in0 per pixel
    static constexpr int NCH = CPT / 8;
    static constexpr int SPT = CPT / 16;         // k-steps per tap
    static constexpr int NS = 9 * SPT;
    // The two-row-tile data gradient on 8-row pixel tiles only fits the 160 KB of LDS without padding: its staged
    // pixels are 128 bytes apart and the 16-byte chunks of a pixel are stored XOR-ed with the pixel index's low
    // three bits instead, which spreads eight neighbouring pixels' reads of one chunk over all banks as well.
    static constexpr bool SWZ = MODE == 1 && MT == 2 && NPT == 2 && C0 == 64 && C1 == 0;
    static constexpr int PSTRIDE = SWZ ? CPT * 2 : CPT * 2 + 16;  // LDS bytes per staged pixel (pad spreads banks)
    static constexpr int IN_BYTES = (NPX_ * PSTRIDE + 15) & ~15;
    static constexpr int W_BYTES = NS * MT * 1024;
    static constexpr int NCH1 = NCH - NCH0;
    // staged 16-byte chunks per thread per tile: the chunks of in0 first, then those of in1, each list
    // rounded up to whole passes of the group's 256 threads so that a pass reads one array only
    static constexpr int NL0 = (NPX_ * NCH0 + 255) / 256;
    static constexpr int NL1 = (NPX_ * NCH1 + 255) / 256;
    static constexpr int NLOAD = NL0 + NL1;
    static constexpr int TAB_FLOATS = NG + 9 * 2 * NG;     // bias + history taps (hist <= 2)
    static constexpr int POOL_FLOATS = 2 * 4 * 4 * F_;     // per group and wave: 4 windows x F partial sums
    static constexpr int LDS_BYTES = W_BYTES + 2 * IN_BYTES + (MODE == 0 ? (TAB_FLOATS + POOL_FLOATS) * 4 : 0);
};

// EARLY: request the tile after next already in the finish phase (two phases of flight, ahead of the
// phase's stores) instead of at the start of the compute phase (one phase of flight).  Measured at
// 224 x 224 with the present staging code: no instantiation gains from it, so none is launched with it.
// ZS (forward only): the step out of the zero state - in1 (h_prev) and c_prev are null, nothing is read for them and
// zeros take their place.  An instantiation of its own: as run-time facts the few extra scalars spilled a dozen more
// scalar registers in the loops of every launch (+2 - 7 % per launch).
// CB (forward only): the cell state (c_prev in, c_out out) is stored as bf16 instead of float - 64 fewer bytes per
// pixel of a launch that runs at the memory system's pace.  The arithmetic stays float; c is rounded once per step.
// SF (forward only): what the step keeps for backward is the backward factors (ISG_SAVE_FACTORS), else the gate
// activations - a compile-time fact: as a run-time one it cost seven selects per channel pair in the finish phase,
// which is bound by the vector ALU.
template <int C0, int C1, int MT, int MODE, bool EARLY, int NPT = 2, bool ZS = false, bool CB = false, bool SF = false>
__global__ __launch_bounds__(kPPThreads) void conv3x3_pp_kernel(ConvParams p) {
    using K = PP<C0, C1, MT, MODE, NPT>;
    using T = __bf16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);  // wave-uniform: 0 = group A, 1 = group B
    const int gtid = tid & 255, lane = tid & 63, wave = gtid >> 6;
    const int px_l = lane & 31, half = lane >> 5;
    char* lds_w = smem;
    char* lds_in = smem + K::W_BYTES + grp * K::IN_BYTES;
    float* lds_tab = reinterpret_cast<float*>(smem + K::W_BYTES + 2 * K::IN_BYTES);
    float* lds_pool = lds_tab + K::TAB_FLOATS;

    {  // weight image and tables, once per workgroup
        const uint4* src = static_cast<const uint4*>(p.wimg);
        uint4* dst = reinterpret_cast<uint4*>(lds_w);
        for (int i = tid; i < K::W_BYTES / 16; i += kPPThreads) dst[i] = src[i];
        if constexpr (MODE == 0) {
            if (tid < NG) lds_tab[tid] = p.bias[(tid & 3) * F_ + (tid >> 2)];  // [channel][gate] order
            for (int i = tid; i < 9 * p.hist * NG; i += kPPThreads) lds_tab[NG + i] = p.hist_table[i];
        }
    }

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    const int stride = 2 * gridDim.x;
    const int n_iter = (ntiles + stride - 1) / stride;

    auto tile_ref = [&](int k) {
        TileRef r;
        const int tile = k * stride + 2 * blockIdx.x + grp;
        r.valid = k < n_iter && tile < ntiles;
        r.b = 0, r.ty0 = 0, r.tx0 = 0, r.live = false;
        r.hp[0] = r.hp[2] = -1, r.hp[1] = r.hp[3] = 0;
        if (r.valid) {
            r.b = tile / tiles_per_img;
            const int rem = tile - r.b * tiles_per_img;
            const int tyi = rem / p.tiles_x;
            r.ty0 = tyi * K::TH_, r.tx0 = (rem - tyi * p.tiles_x) * TW;
        }
        return r;
    };

    // Staging is split in two: the loads are issued unconditionally from clamped (always valid)
    // addresses so that nothing has to look at the data until the LDS write one phase later, where
    // chunks that lie outside the image are replaced by zeros (the SAME padding).
    // The sample's sequence length and history points ride along with the staged data as ordinary
    // vector loads (same wave-uniform address in every lane): they are consumed phases later through
    // readfirstlane, so neither a scalar-cache round trip nor a drained store queue is ever waited for.
    uint4 st[K::NLOAD];
    int st_meta[5] = {0x7fffffff, -1, 0, -1, 0};
    unsigned st_inside = 0u;  // bit j: chunk j of this thread lies inside the image
    // Where chunk j of this thread sits in the halo tile never changes: its LDS byte offset (bits 0-15), halo
    // row (16-19) and column (20-25), 16 x its chunk index within the pixel (26-30; bits 4-6 once shifted) and
    // whether it exists at all (31) are worked out once, and a tile only adds its origin.
    unsigned info[K::NLOAD];
#pragma unroll
    for (int j = 0; j < K::NLOAD; ++j) {
        const bool second = j >= K::NL0;
        const int nch = second ? K::NCH1 : K::NCH0;
        const int idx = gtid + 256 * (second ? j - K::NL0 : j);
        const bool exists = idx < K::NPX_ * nch;
        const int cidx = exists ? idx : K::NPX_ * nch - 1;
        const int px = cidx / nch, chl = cidx - px * nch;
        const int hy = px / HALO_W, hx = px - hy * HALO_W;
        const int chunk = (second ? K::NCH0 : 0) + chl;
        const unsigned ldso = px * K::PSTRIDE + (K::SWZ ? chunk ^ (px & 7) : chunk) * 16;
        info[j] = ldso | (unsigned)hy << 16 | (unsigned)hx << 20 | (unsigned)chl << 26 | (exists ? 0x80000000u : 0u);
    }
    auto stage_load = [&](const TileRef& r) {
        if (!r.valid) return;
        st_inside = 0u;
        st_meta[0] = 0x7fffffff;
        if (p.durations != nullptr) st_meta[0] = p.durations[r.b];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < 2 * p.hist) st_meta[1 + i] = p.hist_points[(size_t)r.b * p.hist_bstride + i];
        }
        const int row0 = r.b * p.H;
#pragma unroll
        for (int j = 0; j < K::NLOAD; ++j) {
            unsigned inf = info[j];
            // with this many passes the unpacked fields must not be kept in registers of their own across the k-loop
            if constexpr (K::SWZ) asm volatile("" : "+v"(inf));
            const int hy = (inf >> 16) & 15, hx = (inf >> 20) & 63;
            const int gy = r.ty0 - 1 + hy, gx = r.tx0 - 1 + hx;
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) st_inside |= 1u << j;
            const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
            const unsigned pixel = (unsigned)((row0 + cy) * p.W + cx);
            const unsigned chunk_bytes = (inf >> 22) & 0x70u;
            // byte offsets fit 32 bits (checked on the host): a scalar base plus one vector offset per load
#ifdef ISG_DIAG_NOLOAD  // diagnostic build: the launch without its tile loads
            if (true) { st[j] = make_uint4(pixel, chunk_bytes, 0u, 0u); continue; }
#endif
            if (j < K::NL0)
                st[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.in0) +
                                                        (size_t)(pixel * (unsigned)(C0 * 2) + chunk_bytes));
            else if constexpr (ZS)
                st[j] = make_uint4(0u, 0u, 0u, 0u);
            else
                st[j] = *reinterpret_cast<const uint4*>(static_cast<const char*>(p.in1) +
                                                        (size_t)(pixel * (unsigned)(C1 * 2) + chunk_bytes));
        }
    };
    auto stage_store = [&](TileRef& r) {
        if (!r.valid) return;
        r.live = p.t < __builtin_amdgcn_readfirstlane(st_meta[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) r.hp[i] = __builtin_amdgcn_readfirstlane(st_meta[1 + i]);
        if (!r.live) return;
#pragma unroll
        for (int j = 0; j < K::NLOAD; ++j) {
            unsigned inf = info[j];
            if constexpr (K::SWZ) asm volatile("" : "+v"(inf));  // as in stage_load
            const bool in = ((st_inside >> j) & 1u) && !(ZS && j >= K::NL0);
            const uint4 v = in ? st[j] : make_uint4(0u, 0u, 0u, 0u);
            if (inf & 0x80000000u) *reinterpret_cast<uint4*>(lds_in + (inf & 0xFFFFu)) = v;
        }
    };

    f32x16 acc[MT][NPT];
    auto mainloop = [&]() {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                if constexpr (MODE == 0) {
                    // the accumulators start from the bias: the finish phase then has no table reads of its own
#pragma unroll
                    for (int a4 = 0; a4 < 4; ++a4) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lds_tab + 32 * half + 16 * t + 4 * a4);
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[t][pt][4 * a4 + g] = b4[g];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][pt][r] = 0.0f;
                }
            }
        const int lane_px = (NPT * wave) * HALO_W + px_l;
        const char* bb = lds_in + lane_px * K::PSTRIDE + (K::SWZ ? 0 : half * 16);
        const char* aw = lds_w + lane * 16;
        // swizzled image: chunk 2 q + half of the pixel lane_px + c sits at byte ((2 q + half) ^ ((lane_px + c) & 7)) * 16
        int lane_px7 = lane_px & 7;
        const int half_bit = half << 4;
        if constexpr (K::SWZ) asm volatile("" : "+v"(lane_px7));  // else every fetch address of the k-loop is kept
                                                                    // in a register of its own across tiles
#ifndef ISG_PP_AHEAD
#define ISG_PP_AHEAD 1
#endif
        // k-steps the fragments are requested ahead of their MFMAs (four 32-cycle MFMAs per k-step cover an LDS round
        // trip: 2 measured within 1 % of 1 for 16 more registers)
        constexpr int AHEAD = ISG_PP_AHEAD, NBUF = AHEAD + 1;
        bf16x8 a[NBUF][MT], bf[NBUF][NPT];
        auto fetch = [&](int s, int slot) {
            const int tap = s / K::SPT, q = s - tap * K::SPT;
            const int off = ((tap / 3) * HALO_W + (tap % 3)) * K::PSTRIDE + (K::SWZ ? 0 : q * 32);
            // the forward image interleaves the row tiles per k-step; the data-gradient image keeps each row
            // tile's k-steps together (a launch may also take a single tile of it)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#ifdef ISG_DIAG_FEWER_LDS  // diagnostic build (wrong results): the k-loop with half of its LDS reads
                if (s >= NBUF) { asm volatile("" : "+v"(a[slot][t])); continue; }
#endif
                a[slot][t] = *reinterpret_cast<const bf16x8*>(aw + (MODE == 1 ? t * K::NS + s : s * MT + t) * 1024);
            }
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
            {
                if constexpr (K::SWZ) {
                    const int c7 = (pt * HALO_W + (tap / 3) * HALO_W + (tap % 3)) & 7;
                    bf[slot][pt] = *reinterpret_cast<const bf16x8*>(bb + pt * HALO_W * K::PSTRIDE + off +
                                                                    ((q << 5) ^ half_bit ^ (((lane_px7 + c7) & 7) << 4)));
                } else {
                    bf[slot][pt] = *reinterpret_cast<const bf16x8*>(bb + pt * HALO_W * K::PSTRIDE + off);
                }
            }
        };
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) fetch(s, s);
#pragma unroll
        for (int s = 0; s < K::NS; ++s) {
            const int cur = s % NBUF;
            if (s + AHEAD < K::NS) fetch(s + AHEAD, (s + AHEAD) % NBUF);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
                    acc[t][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][t], bf[cur][pt], acc[t][pt], 0, 0, 0);
            // keep the prefetch distance where it is: without this fence the scheduler hoists many
            // steps of LDS reads ahead and the register file overflows into scratch
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Operands the epilogue needs from global memory (previous cell state, or the running dx sum when
    // accumulating) are requested at the START of the compute phase and consumed one phase later, so
    // no load latency is exposed between the MFMAs and the stores.
    constexpr int NPRE = MODE == 0 ? 4 : NPT * MT * 4;
    using PreT = std::conditional_t<MODE == 0, uint4, uint2>;  // 16 bytes of c_prev / 8 bytes of dx per entry
    PreT pre[NPRE];
    auto prefetch_epilogue = [&](const TileRef& r) {
        if (!(r.valid && r.live)) return;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int gy = min(r.ty0 + NPT * wave + pt, p.H - 1), gx = min(r.tx0 + px_l, p.W - 1);
            const size_t pixel = ((size_t)r.b * p.H + gy) * p.W + gx;
            if constexpr (MODE == 0) {
                if constexpr (ZS) {
                    pre[2 * pt] = pre[2 * pt + 1] = make_uint4(0u, 0u, 0u, 0u);  // 0.0f: the zero cell state
                } else if constexpr (CB) {
                    pre[2 * pt] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.c_prev) + pixel * F_ + 8 * half);
                } else {
                    const uint4* src = reinterpret_cast<const uint4*>(p.c_prev + pixel * F_ + 8 * half);
                    pre[2 * pt] = src[0];
                    pre[2 * pt + 1] = src[1];
                }
            } else {
                if (p.dx && p.dx_accumulate) {
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int a4 = 0; a4 < 4; ++a4) {
                            const int cc0 = dgrad_frag_channel(p.row_base + 32 * t, a4, half);
                            uint2 v = make_uint2(0u, 0u);
                            if (cc0 < p.row_split && cc0 < p.dx_channels)
                                v = *reinterpret_cast<const uint2*>(static_cast<const T*>(p.dx) + pixel * p.dx_cstride + cc0);
                            pre[(pt * MT + t) * 4 + a4] = v;
                        }
                }
            }
        }
    };

    auto epilogue = [&](const TileRef& r) {
        if (!r.valid) return;
        if (!r.live) {
            if constexpr (MODE == 0) lstm_copy_through_tile<T, std::conditional_t<CB, T, float>>(p, r.b, r.ty0, r.tx0, gtid, 256, K::TH_);
            else dgrad_zero_dx_tile<T>(p, r.b, r.ty0, r.tx0, gtid, 256, K::TH_);
            return;
        }
        // Everything this wave has requested is waited for HERE, ahead of the first store (s_waitcnt vmcnt(0); the
        // staged tile was needed by the LDS writes just before anyway): the memory counter is in order, so a wait
        // for a prefetched operand placed behind a store would also wait for that store to complete.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // History taps touch the 3 x 3 pixels around a history point: a wave whose rows are nowhere near one
        // (nearly every wave) skips them on a scalar test.
        const int hpx[2] = {p.hist > 0 ? r.hp[0] : -1, p.hist > 1 ? r.hp[2] : -1};
        const int hpy[2] = {r.hp[1], r.hp[3]};
        bool near_point = false;
        if constexpr (MODE == 0) {
            const int wy0 = r.ty0 + NPT * wave;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                near_point |= hpx[i] >= 0 && hpy[i] >= wy0 - 1 && hpy[i] <= wy0 + NPT && hpx[i] >= r.tx0 - 1 &&
                              hpx[i] <= r.tx0 + TW;
        }
        float psum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // this lane's channels summed over its rows
        // Two copies of the per-pixel code: the (rare) wave near a history point adds the taps to its pre-activations,
        // every other wave runs a copy without that code - under a run-time test inside one copy, every channel of
        // every wave first copied its four pre-activations (4 of ~80 vector instructions per channel and pixel in
        // a phase bound by the vector ALU).
        auto pixels = [&](auto np_c) __attribute__((always_inline)) {
        constexpr bool NP = decltype(np_c)::value;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            const int gy = r.ty0 + NPT * wave + pt, gx = r.tx0 + px_l;
            if (gy >= p.H || gx >= p.W) continue;
            if constexpr (MODE == 0) {
                // one channel at a time: 4 accumulator registers + bias (+ history taps) -> gates ->
                // state; gate activations leave as 16-byte stores every second channel
                const size_t pixel = ((size_t)r.b * p.H + gy) * p.W + gx;
                const size_t so = pixel * F_ + 8 * half;
                float cp[8];
                if constexpr (CB && !ZS) {  // eight bf16 values: the low and the high half of each word
                    const unsigned w[4] = {pre[2 * pt].x, pre[2 * pt].y, pre[2 * pt].z, pre[2 * pt].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        cp[2 * k] = __uint_as_float(w[k] << 16), cp[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
                } else {
                    const unsigned w[8] = {pre[2 * pt].x,     pre[2 * pt].y,     pre[2 * pt].z,     pre[2 * pt].w,
                                           pre[2 * pt + 1].x, pre[2 * pt + 1].y, pre[2 * pt + 1].z, pre[2 * pt + 1].w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) cp[k] = __uint_as_float(w[k]);
                }
                constexpr bool factors = SF;
#ifdef ISG_DIAG_NOFSTORE  // diagnostic build: the launch without the stores of what backward keeps
                T* gd = nullptr;
#else
                T* gd = p.gates ? static_cast<T*>(p.gates) + pixel * NG + 32 * half : nullptr;
#endif
                float hn[8], cn[8], gprev[4];
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int a4 = 0; a4 < 4; ++a4) {
                        const int cl = 4 * t + a4;
                        float z[4] = {acc[t][pt][4 * a4], acc[t][pt][4 * a4 + 1], acc[t][pt][4 * a4 + 2],
                                      acc[t][pt][4 * a4 + 3]};
                        if constexpr (NP) {
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const int ty = hpy[i] - gy + 1, tx = hpx[i] - gx + 1;
                                if (hpx[i] >= 0 && (unsigned)ty < 3u && (unsigned)tx < 3u) {
                                    const f32x4 tb = *reinterpret_cast<const f32x4*>(
                                        lds_tab + NG + ((ty * 3 + tx) * p.hist + i) * NG + 32 * half + 4 * cl);
                                    z[0] += tb[0], z[1] += tb[1], z[2] += tb[2], z[3] += tb[3];
                                }
                            }
                        }
                        const float g = act_tanh<true>(z[0]), ig = act_sigmoid<true>(z[1]);
                        const float fg = act_sigmoid<true>(z[2]), og = act_sigmoid<true>(z[3]);
                        const float c = __builtin_fmaf(fg, cp[cl], ig * g);  // (as lstm_pixel: one fused form)
                        cn[cl] = c;
                        hn[cl] = og * act_tanh<true>(c);
                        // kept for backward, 16 bytes per channel pair: the gate activations (j i f o | j i f o), or
                        // the backward factors (A A B B f f o o)
                        float s0 = g, s1 = ig;
                        if (factors) lstm_backward_factors(g, ig, s0, s1);
                        if (cl & 1) {
                            if (gd) {
                                const float gates_pair[8] = {gprev[0], gprev[1], gprev[2], gprev[3], s0, s1, fg, og};
                                const float factor_pair[8] = {gprev[0], s0, gprev[1], s1, gprev[2], fg, gprev[3], og};
                                if (factors) store8(gd + 4 * (cl - 1), factor_pair);
                                else store8(gd + 4 * (cl - 1), gates_pair);
                            }
                        } else {
                            gprev[0] = s0, gprev[1] = s1, gprev[2] = fg, gprev[3] = og;
                        }
                    }
#ifdef ISG_DIAG_NOHC  // diagnostic build: the launch without its state stores (one lane in 2^20 keeps the values live)
                if (__builtin_expect(hn[0] + cn[1] != 123.456f, 1)) continue;
#endif
                if constexpr (CB) {
                    store8(reinterpret_cast<T*>(p.c_out) + so, cn);
                } else {
                    *reinterpret_cast<f32x4*>(p.c_out + so) = f32x4{cn[0], cn[1], cn[2], cn[3]};
                    *reinterpret_cast<f32x4*>(p.c_out + so + 4) = f32x4{cn[4], cn[5], cn[6], cn[7]};
                }
                // (pixel indices fit 32 bits, checked on the host: one multiply-add each for the strided outputs)
                store8(static_cast<T*>(p.h_out) + ((size_t)(unsigned)pixel * (unsigned)p.h_pstride + 8 * half), hn);
                if (p.out)
                    store8(static_cast<T*>(p.out) + (size_t)r.b * p.out_bstride +
                               ((size_t)(unsigned)(gy * p.W + gx) * (unsigned)p.out_pstride + 8 * half), hn);
                if (p.pool_out) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) psum[c] += (float)(T)hn[c];  // the pool averages the STORED values
                }
            } else {
                const size_t pixel = ((size_t)r.b * p.H + gy) * p.W + gx;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int a4 = 0; a4 < 4; ++a4) {
                        float v[4] = {acc[t][pt][4 * a4], acc[t][pt][4 * a4 + 1], acc[t][pt][4 * a4 + 2],
                                      acc[t][pt][4 * a4 + 3]};
                        const int cc0 = dgrad_frag_channel(p.row_base + 32 * t, a4, half);
                        if (p.dx && p.dx_accumulate && cc0 < p.row_split && cc0 < p.dx_channels) {
                            const uint2 o = pre[(pt * MT + t) * 4 + a4];
                            v[0] += __uint_as_float(o.x << 16), v[1] += __uint_as_float(o.x & 0xFFFF0000u);
                            v[2] += __uint_as_float(o.y << 16), v[3] += __uint_as_float(o.y & 0xFFFF0000u);
                            store4(static_cast<T*>(p.dx) + pixel * p.dx_cstride + cc0, v);
                        } else {
                            dgrad_store4<T>(p, pixel, cc0, v);
                        }
                    }
            }
        }
        };
        if constexpr (MODE == 0) {
            if (near_point) pixels(std::true_type{});
            else pixels(std::false_type{});
        } else {
            pixels(std::false_type{});
        }
        if constexpr (MODE == 0) {
            if (p.pool_out) {
                // 8 x 8 window = 8 columns (lanes) x this wave's 2 rows x the group's 4 waves: columns through
                // DPP, the waves through LDS, combined by the group at the start of its next phase
                auto dpp_add = [](float v, auto ctrl) {
                    return v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v),
                                                                                   decltype(ctrl)::value, 0xf, 0xf, true));
                };
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float v = psum[c];
                    v = dpp_add(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
                    v = dpp_add(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
                    v = dpp_add(v, std::integral_constant<int, 0x141>{});  // row_half_mirror: the other four lanes
                    psum[c] = v;
                }
                if ((px_l & 7) == 0) {
                    float* dst = lds_pool + ((grp * 4 + wave) * 4 + (px_l >> 3)) * F_ + 8 * half;
                    *reinterpret_cast<f32x4*>(dst) = f32x4{psum[0], psum[1], psum[2], psum[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{psum[4], psum[5], psum[6], psum[7]};
                }
            }
        }
    };

    // The pooled means of the tile whose epilogue ran in this group's last finish phase (the barrier in between
    // made every wave's partial sums visible): 4 windows x 16 channels = one 128-byte run.
    bool pool_pending = false;
    auto pool_flush = [&](const TileRef& r) {
        if constexpr (MODE == 0) {
            if (!pool_pending) return;
            pool_pending = false;
            if (wave != 0) return;
            const int win = lane >> 4, ch = lane & 15;
            float v = 0.0f;
            if (r.live) {
#pragma unroll
                for (int w = 0; w < 4; ++w) v += lds_pool[((grp * 4 + w) * 4 + win) * F_ + ch];
                v *= 1.0f / 64.0f;
            }
            T* dst = static_cast<T*>(p.pool_out) + (size_t)r.b * p.pool_bstride +
                     ((size_t)(r.ty0 >> 3) * (p.W >> 3) + (r.tx0 >> 3) + win) * F_ + ch;
            *dst = (T)v;
        }
    };

    // Schedule.  Half-step h: a group computes when (h + grp) is even.  Entering a compute phase it
    // first requests the NEXT tile's input and the CURRENT tile's epilogue operands, then runs the
    // k-loop; in its other phase it finishes the tile (gate math, stores) and writes the staged
    // registers into its LDS buffer.  Both groups meet at the barrier that ends each half-step.
    //   h:   0                      1                           2              3
    //   A:   compute a0             store a1; load a2; epi a0   compute a1     store a2; load a3; epi a1
    //   B:   store b0; load b1      compute b0                  store b1; load b2; epi b0   compute b1
#ifdef ISG_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TileRef cur = tile_ref(0);
    TileRef pending = cur, done = cur;
    bool have_acc = false;
    int knext = 1;
    stage_load(cur);
    if (grp == 0) {
        stage_store(cur);
        if constexpr (EARLY) {
            pending = tile_ref(knext++);
            stage_load(pending);
        }
    }
    __syncthreads();
    const int n_half = 2 * n_iter + 2;
    STAMP(t_begin);
    for (int h = 0; h < n_half; ++h) {
        STAMP(t0);
        if (((h + grp) & 1) == 0) {
            pool_flush(done);
            // The partner wave on this SIMD is in its finish phase, a long run of vector ALU work (gate math) that
            // would otherwise win most issue slots: the few instructions that put this phase's loads in flight go
            // first (measured: 3 - 10 % of a launch).
            __builtin_amdgcn_s_setprio(3);
            if constexpr (!EARLY) {
                pending = tile_ref(knext++);
                stage_load(pending);
            }
            // (the widest data-gradient kernel has no registers to hold these across its k-loop: it asks for
            // them at the start of the finish phase, ahead of that phase's LDS writes)
            if constexpr (!K::SWZ) prefetch_epilogue(cur);
            __builtin_amdgcn_s_setprio(0);
            STAMP(t1);
#ifndef ISG_NO_MFMA  // diagnostic build without the k-loop: what the staging / epilogue structure alone takes
            if (cur.valid && cur.live) mainloop();
#endif
            STAMP(t2);
            STAMP_ADD(0, t0, t1);  // issue of the prefetch loads
            STAMP_ADD(1, t1, t2);  // k-loop
            done = cur;
            have_acc = true;
        } else {
            if constexpr (K::SWZ) {
                if (have_acc) prefetch_epilogue(done);
            }
            stage_store(pending);  // requested two phases ago
            cur = pending;
            // the tile after that: requested BEFORE this phase's stores are issued, so that the
            // load destinations are never registers a store still has to read
            if constexpr (EARLY) {
                pending = tile_ref(knext++);
                stage_load(pending);
            }
            STAMP(t1);
            if (have_acc) epilogue(done);
            if constexpr (MODE == 0) pool_pending = have_acc && done.valid && p.pool_out != nullptr;
            have_acc = false;
            STAMP(t2);
            STAMP_ADD(2, t0, t1);  // wait for the staged loads + LDS writes
            STAMP_ADD(3, t1, t2);  // epilogue
        }
        STAMP(t3);
        __syncthreads();
        STAMP(t4);
        STAMP_ADD(4 + ((h + grp) & 1), t3, t4);  // barrier wait after a compute / a finish phase
    }
#ifdef ISG_STAMPS
    STAMP(t_end);
    stamp_sum[6] = t_end - t_begin;
    stamp_sum[7] = n_half;
    if (g_stamp_buf != nullptr && (tid & 255) == 0)
        for (int k = 0; k < 8; ++k) g_stamp_buf[(blockIdx.x * 2 + grp) * 8 + k] = stamp_sum[k];
#endif
}

template <int C0, int C1, int MT, int MODE, bool EARLY, int NPT = 2, bool ZS = false, bool CB = false, bool SF = false>
int launch_pp(const ConvParams& p0, int num_cus, hipStream_t stream) {
    using K = PP<C0, C1, MT, MODE, NPT>;
    if constexpr (MODE == 0 && !SF) {  // what is kept for backward: instantiations of their own
        if (p0.save_layout == ISG_SAVE_FACTORS) return launch_pp<C0, C1, MT, MODE, EARLY, NPT, ZS, CB, true>(p0, num_cus, stream);
    }
    if constexpr (MODE == 0 && !CB) {  // the cell state stored as bf16: instantiations of their own
        if (p0.state_bf16) return launch_pp<C0, C1, MT, MODE, EARLY, NPT, ZS, true, SF>(p0, num_cus, stream);
    }
    if constexpr (MODE == 0 && !ZS) {  // the step out of the zero state has an instantiation of its own
        if (p0.in1 == nullptr) return launch_pp<C0, C1, MT, MODE, EARLY, NPT, true, CB, SF>(p0, num_cus, stream);
    }
    static_assert(K::LDS_BYTES <= 160 * 1024, "ping-pong LDS budget");
    if (MODE == 0 && p0.hist > 2) return kNotSupported;
    if (MODE == 0 && NPT != 2 && p0.pool_out != nullptr) return kNotSupported;
    // the staging loads address their arrays with 32-bit byte offsets
    if ((long long)p0.B * p0.H * p0.W * (C0 > C1 ? C0 : C1) * 2 >= (1LL << 32)) return kNotSupported;
    ConvParams p = p0;
    p.tiles_y = (p.H + K::TH_ - 1) / K::TH_;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pp_kernel<C0, C1, MT, MODE, EARLY, NPT, ZS, CB, SF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    const int ntiles = p.tiles_x * p.tiles_y * p.B;
    const int pairs = (ntiles + 1) / 2;
    const int grid = pairs < num_cus ? pairs : num_cus;
    hipLaunchKernelGGL((conv3x3_pp_kernel<C0, C1, MT, MODE, EARLY, NPT, ZS, CB, SF>), dim3(grid), dim3(kPPThreads), K::LDS_BYTES,
                       stream, p);
    return isg::check_launch("conv3x3_pp_kernel");
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------
// weight gradient, bf16 fast path
// ---------------------------------------------------------------------------------------------
// slab[tap][cc][col] = sum over this workgroup's pixels of XH[pixel + tap][cc] * dY[pixel][col].
// Eight waves share the 9 taps x MT row tiles x 2 column tiles of one 8 x 32-pixel tile; the tile's
// [x | h_prev] halo image and its dY image sit in LDS pixel-major and both MFMA operands come out of
// them through the transposing read (the reduction runs over pixels).  The next tile's images are
// requested into registers before the k-loop and written to the other LDS buffer after it.
template <int C0, int MT>
struct WG {
    static constexpr int CPT = C0 + F_;
    static constexpr int NCH0 = C0 / 8;  // 16-byte chunks per pixel of x; h_prev has 2, dY has 8
    // LDS images are split into regions of 32 channels (64 bytes) per pixel with NO padding: one
    // transposing read of a 32-lane phase then covers 4 consecutive pixels x 64 bytes = 256
    // contiguous bytes, i.e. every bank once.  (A padded [pixel][all channels] image makes the four
    // pixels' 16-bank windows overlap: 2-way conflicts on most reads.)
    static constexpr int RSTRIDE = 64;
    static constexpr int IN_REGION = NPX * RSTRIDE;            // one 32-channel row tile of [x | h_prev]
    static constexpr int DY_REGION = TH * TW * RSTRIDE + 128;  // one 32-column half of dY; +128 B puts the two
                                                               // halves of a pixel on different banks for the stores
    static constexpr int IN_BYTES = MT * IN_REGION;
    static constexpr int DY_BYTES = 2 * DY_REGION;
    static constexpr int BUF_BYTES = IN_BYTES + DY_BYTES;
    // Staging enumerates x chunks, then h_prev chunks, then dY chunks, each range rounded up to a
    // whole number of 512-thread passes so that the kind of chunk is a compile-time fact per pass.
    static constexpr int N0 = NPX * NCH0, N1 = NPX * 2, N2 = TH * TW * 8;
    static constexpr int J0 = (N0 + kPPThreads - 1) / kPPThreads, J1 = (N1 + kPPThreads - 1) / kPPThreads,
                         J2 = N2 / kPPThreads;
    static constexpr int NLOAD = J0 + J1 + J2;
    static constexpr int NSLOT = MT == 2 ? 5 : 3;  // taps per wave
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static_assert(CPT <= MT * 32, "row tiles must cover the input channels");
    static_assert(N2 % kPPThreads == 0, "dY tile must fill whole passes");
    static_assert(BUF_BYTES < (1 << 20), "LDS offsets are packed into 20 bits");
};

// Work split of the 8 waves.  The pixel (reduction) axis is cut in two: waves with kh = 0 reduce
// the left 16 pixels of every tile row, kh = 1 the right 16, and a wave keeps BOTH 32-column halves
// of dY for its taps, so a k-step costs it (taps + 2) operand fragments for 2 * taps MFMAs - the
// loop is bound by LDS reads, and this split needs 40 % fewer of them than one column half per wave.
// The two kh partners are summed through LDS once, after the last tile.
template <int C0, int MT>
__global__ __launch_bounds__(kPPThreads) void conv_wgrad_pp_kernel(WgradParams p) {
    using K = WG<C0, MT>;
    using T = __bf16;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int kh = wave & 1;
    int mt, tap0, ntaps;
    if constexpr (MT == 2) {
        // waves w and w + 4 share a SIMD: one takes 5 taps, the other 4
        mt = (wave >> 1) & 1, tap0 = (wave >> 2) * 5, ntaps = (wave >> 2) ? 4 : 5;
    } else {
        const int g = wave >> 1;  // tap groups {0,1,2} {3,4} {5,6} {7,8}
        mt = 0, tap0 = g == 0 ? 0 : 2 * g + 1, ntaps = g == 0 ? 3 : 2;
    }
    const bool does_bias = wave < 2;

    f32x16 acc[K::NSLOT][2];
#pragma unroll
    for (int i = 0; i < K::NSLOT; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.0f;
    float bsum[2] = {0.0f, 0.0f};

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int ntiles = tiles_per_img * p.B;
    const int n_iter = (ntiles + gridDim.x - 1) / gridDim.x;

    // ---- per-thread facts about its chunk of every staging pass, fixed for the whole kernel ----
    // info = byte offset inside an LDS buffer | hy << 20 | hx << 24 (tile-relative row / column of the
    // chunk's pixel, halo included); negative = this thread has no chunk in the pass
    int info[K::NLOAD];
#pragma unroll
    for (int j = 0; j < K::NLOAD; ++j) {
        int hy, hx, ldso;
        bool has = true;
        if (j < K::J0 + K::J1) {
            const bool is_x = j < K::J0;
            const int i = tid + kPPThreads * (is_x ? j : j - K::J0);
            const int nch = is_x ? K::NCH0 : 2;
            const int px = i / nch, ch = i - px * nch;
            hy = px / HALO_W, hx = px - hy * HALO_W;
            has = i < (is_x ? K::N0 : K::N1);
            const int cbyte = ((is_x ? 0 : C0) + ch * 8) * 2;  // byte position in the [x | h_prev] channel row
            ldso = (cbyte / K::RSTRIDE) * K::IN_REGION + px * K::RSTRIDE + (cbyte % K::RSTRIDE);
        } else {
            const int i = tid + kPPThreads * (j - K::J0 - K::J1);
            const int px = i >> 3, ch = i & 7;
            hy = px / TW + 1, hx = (px & (TW - 1)) + 1;  // same convention as the halo image: +1 = tile interior
            ldso = K::IN_BYTES + (ch >> 2) * K::DY_REGION + px * K::RSTRIDE + (ch & 3) * 16;
        }
        info[j] = has ? (ldso | (hy << 20) | (hx << 24)) : -1;
    }
    // Element offset of pass j's chunk from the tile's first pixel: pixel offset * channels + the chunk's
    // channel offset, which is its byte position in the 64-byte LDS pixel row.  The two-row-tile kernel rebuilds
    // it from info at every use (one more register per pass would push its accumulators into scratch); the
    // one-row-tile kernel has the registers and lets the compiler keep it (6 % of a launch).
    auto rel_of = [&](int j) -> int {
        int inf = info[j];
        if constexpr (MT == 2) asm volatile("" : "+v"(inf));  // not loop-invariant as far as the compiler can tell
        const int hy = (inf >> 20) & 15, hx = (inf >> 24) & 63;
        const int pix = (hy - 1) * p.W + (hx - 1);
        const int chb = inf & 63;
        if (j < K::J0) return pix * C0 + (chb >> 1);  // x fills region 0 from byte 0 (C0 <= 32)
        if (j < K::J0 + K::J1) return pix * F_ + ((chb - (C0 * 2) % K::RSTRIDE) >> 1);
        return pix * NG + (chb >> 1) + 32 * (((inf & 0xfffff) - K::IN_BYTES) / K::DY_REGION);
    };
    static_assert(C0 <= 32, "x must fit the first 64-byte region");

    uint4 st[K::NLOAD];
    int st_dur = 0x7fffffff;
    unsigned st_inside = 0u;
    // All passes of the next tile are requested before the k-loop.  (Issuing them one per tile row from
    // inside the k-loop was measured: the waves then stall on the memory queues between MFMAs and the
    // loop takes twice as long.)
    bool ld_valid = false, ld_interior = false;
    int ld_ty0 = 0, ld_tx0 = 0;
    const T *base0 = nullptr, *base1 = nullptr, *base2 = nullptr;
    auto load_begin = [&](int k) {
        const int tile = k * gridDim.x + blockIdx.x;
        ld_valid = k < n_iter && tile < ntiles;
        if (!ld_valid) return;
        const int b = tile / tiles_per_img;
        const int rem = tile - b * tiles_per_img;
        const int tyi = rem / p.tiles_x;
        ld_ty0 = tyi * TH, ld_tx0 = (rem - tyi * p.tiles_x) * TW;
        st_dur = 0x7fffffff;
        if (p.durations != nullptr) st_dur = p.durations[b];
        const int origin = (b * p.H + ld_ty0) * p.W + ld_tx0;  // the host keeps B * H * W * 64 below 2^31
        base0 = static_cast<const T*>(p.in0) + (size_t)origin * C0;
        base1 = static_cast<const T*>(p.in1) + (size_t)origin * F_;
        base2 = static_cast<const T*>(p.dy) + (size_t)origin * NG;
        ld_interior = ld_ty0 >= 1 && ld_ty0 + TH + 1 <= p.H && ld_tx0 >= 1 && ld_tx0 + TW + 1 <= p.W;
        st_inside = ld_interior ? 0xffffffffu : 0u;
    };
    auto load_pass = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < K::NLOAD) {
            if (!ld_valid) return;
            const T* base = j < K::J0 ? base0 : (j < K::J0 + K::J1 ? base1 : base2);
            if (ld_interior) {
                // a thread without a chunk in this pass re-reads the tile's first chunk (in bounds), value dropped
                st[j] = *reinterpret_cast<const uint4*>(base + (info[j] >= 0 ? rel_of(j) : 0));
            } else {
                const int hy = (info[j] >> 20) & 15, hx = (info[j] >> 24) & 63;
                const int gy = ld_ty0 + hy - 1, gx = ld_tx0 + hx - 1;
                const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && info[j] >= 0;
                if (inside) st_inside |= 1u << j;
                st[j] = *reinterpret_cast<const uint4*>(base + (inside ? rel_of(j) : 0));
            }
        }
    };
    auto load_rest = [&](auto jc) {  // passes jc .. NLOAD-1
        constexpr int j0 = decltype(jc)::value;
        [&]<int... I>(std::integer_sequence<int, I...>) {
            (load_pass(std::integral_constant<int, j0 + I>{}), ...);
        }(std::make_integer_sequence<int, (K::NLOAD > j0 ? K::NLOAD - j0 : 0)>{});
    };
    // returns whether the staged tile takes part (its sample is still running at step t)
    auto stage_store = [&](int k, char* buf) -> bool {
        const int tile = k * gridDim.x + blockIdx.x;
        if (k >= n_iter || tile >= ntiles) return false;
        const bool live = p.t < __builtin_amdgcn_readfirstlane(st_dur);
        if (!live) return false;
#pragma unroll
        for (int j = 0; j < K::NLOAD; ++j) {
            const bool in = (st_inside >> j) & 1u;
            const uint4 v = in ? st[j] : make_uint4(0u, 0u, 0u, 0u);
            if (info[j] >= 0) *reinterpret_cast<uint4*>(buf + (info[j] & 0xfffff)) = v;
        }
        return true;
    };

    auto compute = [&](const char* buf) {
        const char* lds_in = buf + mt * K::IN_REGION;
        const char* lds_dy = buf + K::IN_BYTES;
        const int li = lane & 15, g16 = lane >> 4;
        const int q = li >> 2, pp4 = li & 3;
        const int csub = 16 * (g16 & 1) + 4 * pp4;  // channel offset inside the 32-row tile
        const int pj = 16 * kh + 8 * half + q;
#pragma unroll 1
        for (int r = 0; r < TH; ++r) {
            bf16x8 bfrag[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const char* a0 = lds_dy + n * K::DY_REGION + (r * TW + pj) * K::RSTRIDE + csub * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(a0 + 4 * K::RSTRIDE));
                const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfrag[n] = __builtin_bit_cast(bf16x8, both);
            }
            if (does_bias) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum[n] += (float)bfrag[n][j];
            }
#pragma unroll
            for (int i = 0; i < K::NSLOT; ++i) {
                if (i < ntaps) {
                    const int tap = tap0 + i;
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const char* a0 = lds_in + ((r + dy) * HALO_W + pj + dx) * K::RSTRIDE + csub * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0 + 4 * K::RSTRIDE));
                    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 afrag = __builtin_bit_cast(bf16x8, both);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag[1], acc[i][1], 0, 0, 0);
                }
            }
        }
    };

    load_begin(0);
    load_rest(std::integral_constant<int, 0>{});
    if constexpr (K::CPT < MT * 32) {
        // channels past the real ones are never staged: clear them once so that the unused
        // accumulator rows they feed stay finite
        for (int i = tid; i < 2 * MT * NPX * 4; i += kPPThreads) {
            const int bufi = i / (MT * NPX * 4), rem = i - bufi * (MT * NPX * 4);
            const int region = rem / (NPX * 4), c = rem - region * (NPX * 4);
            if (region * 4 + (c & 3) >= K::CPT / 8)
                *reinterpret_cast<uint4*>(smem + bufi * K::BUF_BYTES + region * K::IN_REGION + c * 16) =
                    make_uint4(0u, 0u, 0u, 0u);
        }
    }
    bool live = stage_store(0, smem);
    __syncthreads();
#ifdef ISG_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    STAMP(t_begin);
    for (int k = 0; k < n_iter; ++k) {
        STAMP(t0);
        load_begin(k + 1);
        load_rest(std::integral_constant<int, 0>{});
        STAMP(t1);
        if (live) compute(smem + (k & 1) * K::BUF_BYTES);
        STAMP(t2);
        live = stage_store(k + 1, smem + ((k + 1) & 1) * K::BUF_BYTES);
        STAMP(t3);
        __syncthreads();
        STAMP(t4);
        STAMP_ADD(0, t0, t1);  // issue of the next tile's loads
        STAMP_ADD(1, t1, t2);  // k-loop
        STAMP_ADD(2, t2, t3);  // wait for the loads + LDS writes
        STAMP_ADD(4, t3, t4);  // barrier
    }
#ifdef ISG_STAMPS
    STAMP(t_end);
    stamp_sum[6] = t_end - t_begin;
    stamp_sum[7] = n_iter;
    if (g_stamp_buf != nullptr && lane == 0 && (wave == 0 || wave == 7))
        for (int k = 0; k < 8; ++k) g_stamp_buf[(blockIdx.x * 2 + (wave == 7)) * 8 + k] = stamp_sum[k];
#endif

    // ---- sum the two pixel halves (kh partners are waves 2q and 2q + 1) through LDS, one tap slot at a time ----
    float* red = reinterpret_cast<float*>(smem) + (size_t)(wave >> 1) * (2 * 16 * 64);
#pragma unroll
    for (int i = 0; i < K::NSLOT; ++i) {
        if (kh == 1 && i < ntaps) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) red[(n * 16 + reg) * 64 + lane] = acc[i][n][reg];
        }
        __syncthreads();
        if (kh == 0 && i < ntaps) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) acc[i][n][reg] += red[(n * 16 + reg) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 1) red[lane] = bsum[0], red[64 + lane] = bsum[1];  // wave 1's region is wave 0's region
    __syncthreads();
    if (wave == 0) bsum[0] += red[lane], bsum[1] += red[64 + lane];
    if (kh != 0) return;

    float* slab = p.slab + (size_t)blockIdx.x * 9 * (MT * 32) * NG;
#pragma unroll
    for (int i = 0; i < K::NSLOT; ++i) {
        if (i < ntaps) {
            const int tap = tap0 + i;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cc = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    float* dst = slab + ((size_t)tap * (MT * 32) + cc) * NG + 32 * n + (lane & 31);
                    *dst = p.slab_accumulate ? *dst + acc[i][n][reg] : acc[i][n][reg];
                }
        }
    }
    if (does_bias) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float v = bsum[n] + __shfl_xor(bsum[n], 32);
            if (half == 0) {
                float* dst = p.bias_slab + (size_t)blockIdx.x * NG + 32 * n + lane;
                *dst = p.slab_accumulate ? *dst + v : v;
            }
        }
    }
}

template <int C0, int MT>
int launch_wg(const WgradParams& p, int grid, hipStream_t stream) {
    using K = WG<C0, MT>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "weight-gradient LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_pp_kernel<C0, MT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    ISG_REQUIRE(e == hipSuccess, ISG_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL((conv_wgrad_pp_kernel<C0, MT>), dim3(grid), dim3(kPPThreads), K::LDS_BYTES, stream, p);
    return isg::check_launch("conv_wgrad_pp_kernel");
}

}  // namespace

#ifdef ISG_STAMPS
extern "C" int isg_debug_set_stamp_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

namespace isgc {

int launch_conv_pp(const ConvParams& p, int mt, int mode, int num_cus, hipStream_t stream) {
    if (mode == 0) {
#ifdef ISG_PP_EARLY_FWD  // diagnostic build: the forward cell with its tiles requested one phase earlier
        constexpr bool kEarly = true;
#else
        constexpr bool kEarly = false;
#endif
        // A launch whose 8-row tiles would leave half of the CUs without a workgroup (the reference's own 28 x 28 grid
        // at a few dozen samples: one launch is then a fixed latency, not work) runs on 4-row tiles: twice the
        // workgroups, each with half the pixels.  Same arithmetic per pixel, same results.
        const int tiles8 = p.tiles_x * ((p.H + TH - 1) / TH) * p.B;
#ifndef ISG_PP_NO_SMALL_TILES  // (switch for same-box A/B builds)
        const bool small = tiles8 <= num_cus && p.pool_out == nullptr;
#else
        const bool small = false;
#endif
        if (p.c0 == 16 && p.c1 == 16)
            return small ? launch_pp<16, 16, 2, 0, false, 1>(p, num_cus, stream) : launch_pp<16, 16, 2, 0, kEarly>(p, num_cus, stream);
        if (p.c0 == 32 && p.c1 == 16)
            return small ? launch_pp<32, 16, 2, 0, false, 1>(p, num_cus, stream) : launch_pp<32, 16, 2, 0, kEarly>(p, num_cus, stream);
        if (p.c0 == 48 && p.c1 == 16) return launch_pp<48, 16, 2, 0, false, 1>(p, num_cus, stream);
        return kNotSupported;
    }
    if (p.c0 == NG && p.c1 == 0 && mt == 1) return launch_pp<NG, 0, 1, 1, false>(p, num_cus, stream);
#ifdef ISG_DGRAD_SWZ
    if (p.c0 == NG && p.c1 == 0 && mt == 2) return launch_pp<NG, 0, 2, 1, false, 2>(p, num_cus, stream);
#endif
    if (p.c0 == NG && p.c1 == 0 && mt == 2) return launch_pp<NG, 0, 2, 1, false, 1>(p, num_cus, stream);
    return kNotSupported;
}

int launch_wgrad_pp(const WgradParams& p, int mt, int grid, hipStream_t stream) {
    if (p.c1 != F_) return kNotSupported;
    if ((long long)p.B * p.H * p.W * NG >= (1LL << 31)) return kNotSupported;  // the kernel's element offsets are int
    if (p.c0 == 16 && mt == 1) return launch_wg<16, 1>(p, grid, stream);
    if (p.c0 == 32 && mt == 2) return launch_wg<32, 2>(p, grid, stream);
    return kNotSupported;
}

}  // namespace isgc
