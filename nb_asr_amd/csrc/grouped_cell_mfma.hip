// bf16 storage path (BASELINE config 4): a whole SearchCell of three grouped convolutions in ONE launch on the MATRIX cores.
//   x1 = op0(x0n) + s00 x0n;   x2 = op1(x1) + s10 x0n + s11 x1;   x3 = op2(x2) + s20 x0n + s21 x1 + s22 x2
// (reference model.py:49-59 over model.py:13-22 and ops.py:24-30 on torch.bfloat16 tensors).
//
// Why a second cell kernel: with bf16 storage the bytes halve but the vector-ALU form (grouped_cell.hip) still issues one FMA per
// product -- 2 per lane per instruction -- and is bound by that (0.31-0.51 of its roof at 32 x 1600).  Every tensor of the bf16 model
// IS a bfloat16 tensor (the reference rounds each op's output once), so the products can go to v_mfma_f32_16x16x32_bf16 unchanged:
// exact bf16 x bf16 products, fp32 accumulation -- the reference's arithmetic up to the order of the sums.
//
// Mapping (per group of CG = 6..12 channels; groups never mix):
//   * the group's tensor lives in LDS as channel-quad PLANES, tile[quad][frame][4 channels] bf16 (CP = 8 or 16 channel slots = 2 or 4
//     planes; pad channels are zero), 16 zero frames either side of every plane: a plane row is the 8 bytes one lane of the accumulator
//     fragment holds (round 5; rounds 3-4 kept frame-major rows of CP channels: every store 4-way, every skip read 2-way bank-conflicted);
//   * one MFMA = 16 fragment rows (M) x 16 frames (N) x 32 (tap, channel) pairs (K), always 2 taps per K step.  CP = 16: the rows are
//     the group's 16 (padded) channels, K = 2 taps x 16 channels.  CP = 8: only 8 rows exist, so ONE MFMA serves TWO column blocks 128
//     frames apart -- rows 0-7 x K slots 0-15 (2 taps x 8 channels of block nb), rows 8-15 x K slots 16-31 (the same taps of block
//     nb + 8), weights packed block-diagonally -- and every lane of the accumulator fragment holds real outputs.  The B operand of lane
//     (n, kb) is two ds_read_b64 one plane apart -- the 8 channels of frame f0 + n + tap * dilation - left_pad -- so dilation and padding
//     are address arithmetic; the A operand is the weights, pre-packed per (node, group, K step, lane) and held in registers for the node
//     (<= 16 registers);
//   * a wave owns 256 frames = 16 column blocks (128 = 8 for rows of <= 1024 frames: twice the waves); per block (CP = 8: per pair of
//     blocks) 3-4 MFMAs (taps padded to 6 / 8 with zero weights; a padded tap reads the lane's tap-0 window, so a NaN there surfaces at
//     this frame as it does through the real tap 0), then the node's epilogue on the accumulator fragment -- lane = (frame, 4 channels):
//     bias is the accumulator's initial value, relu + clamp, the skips in python's sum order, ONE rounding to bf16 -- written as 8 bytes
//     into the OTHER tile (ping-pong: x0n in A, x1 in B, x2 in A, x3 in B), so a node needs one barrier, not two; the wave tile that
//     holds the row's end re-zeroes the frames beyond it once per node.  Skip inputs are read at the lane's own position (x0n for the
//     last node is carried in registers);
//   * x0 comes in and x3 goes out channel-major (the tensors' layout in HBM): 16-byte global accesses by lane = (channel, 8-frame chunk),
//     a wave-private channel-major scratch image in the tile that is idle at that moment, and ds_read_b64_tr_b16 -- the hardware
//     transposing read -- between that image and the planes; the pending LayerNorm is applied on the transposed side (lane = one frame,
//     4 channels: one (mean, rstd) per lane and gather, packed arithmetic).
// No statistics by-product (the bf16 executor's cell LayerNorm is consumed by the next convolution's image writer, which computes
// them on the way); a cell whose consumer wants them runs on grouped_cell.hip.
#include "common.h"
#include "storage.h"

#include <type_traits>

#ifndef NBASR_CELLM_STAMPS
#define NBASR_CELLM_STAMPS 0
#endif
#include <cstdio>
#include <cstdlib>

namespace nbasr {

typedef __bf16 cm_bf8 __attribute__((ext_vector_type(8)));
typedef float cm_f4 __attribute__((ext_vector_type(4)));
typedef unsigned cm_u4 __attribute__((ext_vector_type(4)));

constexpr int CM_PADL = 16, CM_PADR = 16;       // zero frames either side of a tile (taps reach <= 12 frames back, <= 14 ahead)
// NBT = 16-frame column blocks per wave (template): 8, 10, 14 or 16 (128 .. 256 frames per wave); which one a launch takes: cellm_plan.
struct CellMDims {
    int channels, frames, ld, groups, batch, cg;
    int k[3], d[3], lpad[3], nstep[3];          // taps, dilation, left padding, MFMAs per column block of each node
    int skips;                                  // bit0 s00 | bit1 s10 | bit2 s11 | bit3 s20 | bit4 s21 | bit5 s22
    int nt;                                     // 256-frame wave tiles per row
#if NBASR_CELLM_STAMPS
    unsigned long long* stamps;                 // diagnostics build (tools/gpu/cellm_stamps.py): 16 clock stamps per wave
#endif
};

__device__ __forceinline__ cm_f4 cm_unpack4(u2v p) { return cm_f4{bf16_lo(p.x), bf16_hi(p.x), bf16_lo(p.y), bf16_hi(p.y)}; }

// ---- LDS image of a group's tensor (round 5): channel-quad PLANES, tile[quad][frame][4 channels] ---------------------------------
// A plane row is 8 bytes: the 4 channels of one frame -- exactly what one lane of the accumulator fragment holds (frame n, rows
// 4 kb .. 4 kb + 3), so a node's output is ONE ds_write_b64 whose 16-lane groups cover 128 contiguous bytes (the frame-major rows of
// rounds 3-4 put those lanes 32 bytes apart: 4-way bank conflicts on every store, 2-way on every skip read).  The B operand of lane
// (n, kb) -- 8 channels of frame n + shift -- is two ds_read_b64 one plane apart.  Plane q starts at q * PB + {0, 128, 128, 256}[q]
// with PB a multiple of 256: modulo the 256 bytes the 64 banks span, the planes sit at 0 / 128 / 128 / 0, so the two planes a
// 32-lane half touches in one access (operand reads: quads {0, 2} or {1, 3}; own-position reads and stores: {0, 1} or {2, 3}) never
// share banks.
__host__ __device__ inline int cm_plane_bytes(int rows) { return (rows * 8 + 255) & ~255; }
__host__ __device__ inline int cm_plane_off(int q, int pb) { return q * pb + (q == 0 ? 0 : q == 3 ? 256 : 128); }
// a tile also hosts the wave-private channel-major scratch images of the way in / out: nt x SR rows (SR = 8, or 12 for the 16-slot
// groups: 10 or 12 real channels, the fourth quad is all padding and never stored) of (NBT + 1) x 32 bytes (one
// 16-frame unit of padding per row: the 8 channel rows a 32-lane half gathers from then start on 8 different 32-byte bank groups)
__host__ __device__ inline int cm_scratch_row(int nbt) { return (nbt + 1) * 32; }
__host__ __device__ inline int cm_tile_bytes(int cp, int nt, int nbt)
{
    const int planes = (cp / 4) * cm_plane_bytes(nt * 16 * nbt + 32) + 256, scratch = nt * (cp == 16 ? 12 : 8) * cm_scratch_row(nbt);
    return ((planes > scratch ? planes : scratch) + 255) & ~255;
}

typedef short cm_s4 __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16 (gfx950): per 16-lane group a block of 4 rows x 16 columns of 16-bit elements; lane 4q + p supplies the address of
// row q, columns 4p .. 4p + 3; lane i receives column i of the 4 rows (row q in element q).  EXEC must be all ones.
__device__ __forceinline__ u2v cm_tr_read(const unsigned char* p)
{
    const cm_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((cm_s4 __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u2v, v);
}

template <int CP, int GPW, int NBT>
__global__ __launch_bounds__(1024) void grouped_cell_mfma_kernel(
    const bf16_t* __restrict__ x0, bf16_t* __restrict__ y,
    const cm_u4* __restrict__ wp0, const cm_u4* __restrict__ wp1, const cm_u4* __restrict__ wp2,
    const float* __restrict__ b0, const float* __restrict__ b1, const float* __restrict__ b2,
    const float* __restrict__ ln_stats, const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, const CellMDims a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cm_lds[];
    constexpr int CM_NB = NBT, CM_WAVE_FRAMES = 16 * NBT, Q = CP / 4;
    const int nt = a.nt;
    const int rows = nt * CM_WAVE_FRAMES + CM_PADL + CM_PADR;        // frames of a tile, pads included
    const int pb = cm_plane_bytes(rows);
    const int tile_bytes = cm_tile_bytes(CP, nt, NBT);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gi = wave / nt, ti = wave - gi * nt;
    unsigned char* const tile_a = cm_lds + gi * 2 * tile_bytes;      // x0n, then x2
    unsigned char* const tile_b = tile_a + tile_bytes;               // x1, then x3
    const int g = blockIdx.x * GPW + gi, b = blockIdx.y;             // (GPW divides the group count: host check)
    const int fb = ti * CM_WAVE_FRAMES;
    // -DNBASR_CELLM_STAMPS=1 (NBASR_EXTRA_CXXFLAGS; tools/gpu/cellm_stamps.py): every wave records the 100 MHz clock at its phase boundaries
#if NBASR_CELLM_STAMPS
    unsigned long long* const stamp_row = a.stamps ? a.stamps + ((static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave) * 16 : nullptr;
    int stamp_i = 1;
    auto stamp = [&]() {
        if (stamp_row && lane == 0) stamp_row[stamp_i] = wall_clock64();
        ++stamp_i;
    };
    if (stamp_row && lane == 0) stamp_row[0] = __builtin_amdgcn_s_getreg(4 | (31 << 11)) | (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(20 | (31 << 11))) << 32);
#else
    auto stamp = []() {};
#endif
    stamp();

    // the 16 zero frames either side of every plane of one tile of every group of the workgroup (16-byte units: 8 per pad)
    auto zero_pads = [&](int which) {
        for (int i = threadIdx.x; i < GPW * Q * 16; i += blockDim.x) {
            const int t = i / (Q * 16), r = i - t * (Q * 16), q = r >> 4, u = r & 15;
            const int byte = cm_plane_off(q, pb) + (u < 8 ? u * 16 : (rows - CM_PADR) * 8 + (u - 8) * 16);
            *reinterpret_cast<cm_u4*>(cm_lds + (t * 2 + which) * tile_bytes + byte) = cm_u4{0u, 0u, 0u, 0u};
        }
    };
    // frames beyond the row stay exactly 0: the wave tile that holds the row's end (and any tile beyond it) re-zeroes its rows of a
    // tile from `frames` on -- behind its own stores (one wave's LDS operations execute in order), before the barrier
    const bool tail = fb + CM_WAVE_FRAMES > a.frames;                // (wave-uniform)
    auto zero_tail = [&](unsigned char* tile) {
        const int f_lo = a.frames > fb ? a.frames : fb;
        const int n = fb + CM_WAVE_FRAMES - f_lo;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            unsigned char* z = tile + cm_plane_off(q, pb) + (f_lo + CM_PADL) * 8;
            for (int r = lane; r < n; r += 64) *reinterpret_cast<u2v*>(z + r * 8) = u2v{0u, 0u};
        }
    };
    zero_pads(0);

    // ---- x0 -> tile A, normalised (pending LayerNorm) and rounded ----------------------------------------------------------------------
    // x0 arrives channel-major (the tensors' layout in HBM): lane = (channel, 8-frame chunk), 16-byte global loads, stored as they are
    // into a wave-private channel-major scratch image (the wave's share of tile B, free until node 0 writes x1), then read back
    // TRANSPOSED: ds_read_b64_tr_b16 hands lane i of a 16-lane group the 4 channels of a quad at frame i -- a plane row.  The image is
    // [channel][NBT + 1 units of 16 frames]: the pad unit puts the 8 channel rows a 32-lane half gathers from on 8 different 32-byte bank
    // groups.  (Rounds 3-4 wrote 4-byte channel pairs straight into frame-major rows: the
    // chunks of a wave's lanes sat 128 bytes apart, 8-way bank conflicts on every store; and normalised 16 elements per lane with
    // per-element statistics.)
    constexpr int CH_CHUNKS = CM_WAVE_FRAMES / 8;                    // 8-frame chunks of a channel row within a wave tile
    constexpr int SR = CP == 16 ? 12 : 8;                            // channel rows of a scratch image (the 16-slot groups' fourth quad is all padding)
    constexpr int NI = (SR * CH_CHUNKS + 63) / 64;                   // global-access instructions per wave (pad channels of a real quad included)
    constexpr int NTR = Q * NBT / 4;                                 // transposed reads per wave: 4 (quad, 16-frame unit) blocks each
    constexpr int ROWB = (NBT + 1) * 32;                             // bytes of a scratch channel row (cm_scratch_row)
    const size_t group_row0 = (static_cast<size_t>(b) * a.channels + static_cast<size_t>(g) * a.cg) * a.ld;
    const int i16 = lane & 15, gq = lane >> 4;
    const int tquad = Q == 2 ? (gq & 1) : gq;                        // the quad this lane's group transposes
    const int tun = Q == 2 ? (gq >> 1) : 0;                          // and its unit within an instruction's pair (Q == 2: two units per instruction)
    constexpr int UPI = Q == 2 ? 2 : 1;                              // units per transposed-read instruction
    {
        unsigned char* const scr = tile_b + ti * (SR * ROWB);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x0 + group_row0), 0, a.cg * a.ld * 2, 0x00020000);
        cm_u4 raw[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 64 + lane, ch = item / CH_CHUNKS, chunk = item - ch * CH_CHUNKS, f0 = fb + chunk * 8;
            const int off = (ch < a.cg && f0 < a.ld) ? (ch * a.ld + f0) * 2 : 0x7ffffff0;   // out of range (pad channels, beyond the row): zeros
            raw[it] = __builtin_bit_cast(cm_u4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
        const bool has_ln = ln_stats != nullptr;                     // (workgroup-uniform)
        typedef float cm_f2 __attribute__((ext_vector_type(2)));
        float nmean[NTR], rstd[NTR];
        cm_f2 gam01{0.f, 0.f}, gam23 = gam01, bet01 = gam01, bet23 = gam01;
        if (has_ln) {
            // (mean, rstd) of the wave's frames: two 16-byte loads per lane at most, parked in the wave's own rows of tile A's first plane
            // (which x0n overwrites below) and picked up per (unit, frame) from there -- 2 NTR four-byte global loads per lane otherwise,
            // and the texture path, not the data, is what 16 waves staging at once wait for
            const float* mrow = ln_stats + static_cast<size_t>(b) * 2 * a.ld;
            float* const sarea = reinterpret_cast<float*>(tile_a + (CM_PADL + fb) * 8);          // 2 x CM_WAVE_FRAMES floats
            constexpr int QR = CM_WAVE_FRAMES / 4;                   // 16-byte pieces per statistics row
#pragma unroll
            for (int i0 = 0; i0 < 2 * QR; i0 += 64) {
                const int i = i0 + lane, row = i / QR, c = i - row * QR, f = fb + 4 * c;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < 2 * QR && f < a.ld) v = *reinterpret_cast<const float4*>(mrow + row * a.ld + f);
                if (i < 2 * QR) *reinterpret_cast<float4*>(sarea + row * CM_WAVE_FRAMES + 4 * c) = v;
            }
#pragma unroll
            for (int k = 0; k < NTR; ++k) {
                const int fl = 16 * (UPI * k + tun) + i16;
                nmean[k] = -sarea[fl]; rstd[k] = sarea[CM_WAVE_FRAMES + fl];
            }
            // pad channels of the last quad: gamma = beta = 0, so that their (zero) inputs stay zero
            const int c = 4 * tquad;
            float gm[4], bt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool real = c + j < a.cg;
                const int idx = g * a.cg + (real ? c + j : 0);
                gm[j] = real ? ln_gamma[idx] : 0.f; bt[j] = real ? ln_beta[idx] : 0.f;
            }
            gam01 = cm_f2{gm[0], gm[1]}; gam23 = cm_f2{gm[2], gm[3]}; bet01 = cm_f2{bt[0], bt[1]}; bet23 = cm_f2{bt[2], bt[3]};
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 64 + lane, ch = item / CH_CHUNKS, chunk = item - ch * CH_CHUNKS;
            if (NI * 64 == SR * CH_CHUNKS || item < SR * CH_CHUNKS) *reinterpret_cast<cm_u4*>(scr + ch * ROWB + chunk * 16) = raw[it];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool pad_quad = 4 * tquad >= SR;                       // (16-slot groups: the group of lanes that would gather the all-padding quad)
        const int tch = 4 * (pad_quad ? 0 : tquad) + ((lane >> 2) & 3);   // the channel row this lane addresses for the gather (any valid one for a padding quad)
        const unsigned char* const trow = scr + tch * ROWB + tun * 32 + (lane & 3) * 8;
        unsigned char* const drow = tile_a + cm_plane_off(tquad, pb) + (CM_PADL + fb + 16 * tun + i16) * 8;
        // (all gathers first: a store to the plane between two of them would pin the next gather behind it -- the compiler cannot tell
        // the plane from the scratch image -- and the eight LDS round trips would run one after the other)
        u2v tv[NTR];
#pragma unroll
        for (int k = 0; k < NTR; ++k) tv[k] = cm_tr_read(trow + k * (UPI * 32));
#pragma unroll
        for (int k = 0; k < NTR; ++k) {
            u2v v = tv[k];
            if (CP == 16 && pad_quad) v = u2v{0u, 0u};
            if (has_ln) {
                const cm_f2 nm{nmean[k], nmean[k]}, rs{rstd[k], rstd[k]};
                const cm_f2 lo = __builtin_elementwise_fma((cm_f2{bf16_lo(v.x), bf16_hi(v.x)} + nm) * rs, gam01, bet01);
                const cm_f2 hi = __builtin_elementwise_fma((cm_f2{bf16_lo(v.y), bf16_hi(v.y)} + nm) * rs, gam23, bet23);
                v = u2v{pack_bf16x2(lo.x, lo.y), pack_bf16x2(hi.x, hi.y)};
            }
            *reinterpret_cast<u2v*>(drow + k * (UPI * 128)) = v;
        }
        if (tail) zero_tail(tile_a);                                 // (normalising a zero beyond `frames` gives beta, not 0)
    }
    stamp();
    __syncthreads();
    stamp();
    zero_pads(1);                                                    // (tile B's pads lay under the scratch image; node 0 writes only frames of the row)

    // ---- the three nodes -------------------------------------------------------------------------------------------------------------
    const int n16 = lane & 15, kb = lane >> 4;                       // MFMA fragment coordinates: column / row-in-tile, 8-deep k block
    // CP = 16: the fragment's 16 rows are the group's (padded) 16 channels; a K step is 2 taps x 16 channels; 16 column blocks per wave.
    // CP = 8: only 8 rows exist, so ONE MFMA serves TWO column blocks 128 frames apart: rows 0-7 x K slots 0-15 (2 taps x 8 channels of
    // block nb), rows 8-15 x K slots 16-31 (the same 2 taps of block nb + 8) -- the weights are packed block-diagonally.  Same MFMA count
    // as 4 taps x 8 channels per step, but every lane of the accumulator fragment holds real outputs: half the epilogue instructions.
    constexpr int NBW = CP == 8 ? CM_NB / 2 : CM_NB;                 // column-block iterations per wave
    const int fblk = CP == 8 ? (kb >> 1) * (CM_WAVE_FRAMES / 2) : 0; // this lane's frame offset within the wave tile (second block of the pair)
    const int q4 = CP == 8 ? (kb & 1) * 4 : kb * 4;                  // first of this lane's 4 output channels in the accumulator fragment
    const int own = cm_plane_off(q4 >> 2, pb) + (fb + fblk + n16 + CM_PADL) * 8;   // this lane's (frame, channel quad) of column block 0, bytes
    constexpr int NB_STRIDE = 16 * 8;                                // bytes between column blocks (within a plane)
    const int opq = CP == 8 ? 0 : 2 * (kb & 1);                      // first of the two quads this lane's B operand takes
    const int op_plane = cm_plane_off(opq, pb), op_next = pb + 128;  // (plane opq + 1 starts pb + 128 bytes after plane opq, for opq 0 and 2)
    u2v keep0[NBW];                                                  // x0n at the lane's positions, for the last node's skip

    constexpr int NSMAX = 4;                                         // K steps per column block: 2 taps each, 6 or 8 taps (3 or 4 steps)
    auto node = [&](auto idx, const cm_u4* __restrict__ wp, const float* __restrict__ bias, const unsigned char* src, unsigned char* dst) {
        constexpr int NODE = decltype(idx)::value;
        const int K = a.k[NODE], D = a.d[NODE], LP = a.lpad[NODE], ns = a.nstep[NODE];
        const bool four = NSMAX == 4 && ns == 4;                     // (wave-uniform)
        cm_bf8 aw[NSMAX];
        int off[NSMAX];
#pragma unroll
        for (int s = 0; s < NSMAX; ++s) {
            const int tap = CP == 8 ? s * 2 + (kb & 1) : s * 2 + (kb >> 1);
            // a padded tap (tap >= K) has zero weights; its B operand is the lane's tap-0 window (finite wherever the data are)
            off[s] = op_plane + (fb + fblk + n16 + (tap < K ? tap * D - LP : -LP) + CM_PADL) * 8;
            aw[s] = __builtin_bit_cast(cm_bf8, wp[(static_cast<size_t>(g) * ns + (s < ns ? s : 0)) * 64 + lane]);
        }
        cm_f4 bv;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (q4 + r < a.cg) ? bias[g * a.cg + q4 + r] : 0.f;
        cm_u4 nxt[NSMAX];
        auto fetch = [&](int nb, auto nsc) {
            constexpr int NS = decltype(nsc)::value;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const u2v lo = *reinterpret_cast<const u2v*>(src + off[s] + nb * NB_STRIDE);
                const u2v hi = *reinterpret_cast<const u2v*>(src + off[s] + op_next + nb * NB_STRIDE);
                nxt[s] = cm_u4{lo.x, lo.y, hi.x, hi.y};
            }
        };
        // The node's skip inputs as all-ones / all-zeros MASKS on the packed bf16 pairs (wave-uniform): the column-block loop below has NO
        // branch on them, so the sixteen blocks are one straight line and the scheduler overlaps one block's epilogue with the next
        // block's MFMAs and LDS reads (with a branch per skip a block cost ~430 cycles of mostly dependent latency at 3.5 waves per SIMD).
        // An absent skip contributes an exact +0 whatever the tensor holds there (round 3 multiplied by a 0 / 1 factor: 0 x Inf = NaN
        // where the reference never reads the tensor).  A node without skips takes the loop without the reads.
        const int sk = a.skips;
        const unsigned m_a = (NODE == 0 ? (sk & 1) : NODE == 1 ? (sk & 2) : (sk & 32)) ? 0xffffffffu : 0u;      // x0n (nodes 0, 1) / x2 (node 2) from tile A
        const unsigned m_b = (NODE == 1 ? (sk & 4) : NODE == 2 ? (sk & 16) : 0) ? 0xffffffffu : 0u;               // x1 from tile B
        const unsigned m_k = (NODE == 2 && (sk & 8)) ? 0xffffffffu : 0u;                                          // x0n carried in registers
        const bool any = NODE == 0 ? (sk & 1) != 0 : NODE == 1 ? (sk & (2 | 4 | 8)) != 0 : (sk & (8 | 16 | 32)) != 0;
        // (K steps as a compile-time count: a wave-uniform branch per block on `four` splits the unrolled blocks into basic blocks
        // and pins every block's reads, MFMAs and epilogue in program order)
        auto blocks = [&](auto with_skips, auto nsc) {
            constexpr bool SK = decltype(with_skips)::value;
            constexpr int NS = decltype(nsc)::value;
            fetch(0, nsc);
            // the skip inputs of the NEXT block are read before this block's store: the compiler cannot tell the positions apart (tile A is
            // read at a position and then overwritten there by nodes 1's output), so reads placed after the store wait behind it -- an
            // exposed LDS round trip per block
            u2v sa_n{0u, 0u}, sb_n{0u, 0u};
            auto fetch_skips = [&](int nb) {
                const int pos = own + nb * NB_STRIDE;
                sa_n = *reinterpret_cast<const u2v*>(tile_a + pos);
                if constexpr (NODE != 0) sb_n = *reinterpret_cast<const u2v*>(tile_b + pos);
            };
            if constexpr (SK) fetch_skips(0);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                cm_u4 cur[NSMAX];
#pragma unroll
                for (int s = 0; s < NS; ++s) cur[s] = nxt[s];
                const u2v ta = sa_n, tb = sb_n;
                if (nb + 1 < NBW) {
                    fetch(nb + 1, nsc);                              // the next block's operands are in flight behind this block's MFMAs
                    if constexpr (SK) fetch_skips(nb + 1);
                }
                cm_f4 acc = bv;
#pragma unroll
                for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[s], __builtin_bit_cast(cm_bf8, cur[s]), acc, 0, 0, 0);
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[r]);
                const int pos = own + nb * NB_STRIDE;
                if constexpr (SK) {
                    auto add = [&](unsigned m, u2v p) {
                        const cm_f4 v = cm_unpack4(u2v{p.x & m, p.y & m});
                        o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
                    };
                    if constexpr (NODE == 0) {
                        add(m_a, ta);
                    } else if constexpr (NODE == 1) {
                        keep0[nb] = ta;
                        add(m_a, ta);
                        add(m_b, tb);
                    } else {
                        add(m_k, keep0[nb]);
                        add(m_b, tb);
                        add(m_a, ta);
                    }
                }
                *reinterpret_cast<u2v*>(dst + pos) = u2v{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        };
        using N3 = std::integral_constant<int, 3>; using N4 = std::integral_constant<int, 4>;
        if (any) { if (four) blocks(std::true_type{}, N4{}); else blocks(std::true_type{}, N3{}); }
        else     { if (four) blocks(std::false_type{}, N4{}); else blocks(std::false_type{}, N3{}); }
        if (tail) zero_tail(dst);                                    // (round 5: was a compare + 4 selects on every column block of every wave)
        stamp();
        __syncthreads();
        stamp();
    };
    node(std::integral_constant<int, 0>{}, wp0, b0, tile_a, tile_b);
    node(std::integral_constant<int, 1>{}, wp1, b1, tile_b, tile_a);
    node(std::integral_constant<int, 2>{}, wp2, b2, tile_a, tile_b);

    // ---- x3 (tile B) -> y, channel-major: the way in, backwards ---------------------------------------------------------------------------
    // Transposed gather from the planes (rows = 4 frames, 4 channels each; lane i receives channel i & 3 at frames 4 (i >> 2) .. + 3), an
    // 8-byte store into the wave-private channel-major scratch image (its share of tile A: nobody reads x2 after the barrier above), then
    // lane = (channel, 8-frame chunk): 16-byte LDS reads, 16-byte global stores.
    {
        int ln2 = lane;                                              // (a fresh value: keeps the offsets of the way in from being carried -- spilled -- across the nodes)
        asm volatile("" : "+v"(ln2));
        unsigned char* const scr = tile_a + ti * (SR * ROWB);
        const unsigned char* const prow = tile_b + cm_plane_off(tquad, pb) + (CM_PADL + fb + 16 * tun + 4 * (ln2 & 3) + ((ln2 >> 2) & 3)) * 8;
        const int och = 4 * tquad + (ln2 & 3);
        unsigned char* const srow = scr + och * ROWB + tun * 32 + ((ln2 >> 2) & 3) * 8;
        u2v tv[NTR];
#pragma unroll
        for (int k = 0; k < NTR; ++k) tv[k] = cm_tr_read(prow + k * (UPI * 128));
#pragma unroll
        for (int k = 0; k < NTR; ++k)
            if (CP != 16 || 4 * tquad < SR) *reinterpret_cast<u2v*>(srow + k * (UPI * 32)) = tv[k];    // (the all-padding quad has no scratch rows)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y + group_row0, 0, a.cg * a.ld * 2, 0x00020000);
        cm_u4 rr[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 64 + ln2, ch = item / CH_CHUNKS, chunk = item - ch * CH_CHUNKS;
            rr[it] = *reinterpret_cast<const cm_u4*>(scr + (item < SR * CH_CHUNKS ? ch * ROWB + chunk * 16 : 0));
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 64 + ln2, ch = item / CH_CHUNKS, chunk = item - ch * CH_CHUNKS, f0 = fb + chunk * 8;
            const int off = (item < SR * CH_CHUNKS && ch < a.cg && f0 < a.ld) ? (ch * a.ld + f0) * 2 : 0x7ffffff0;   // out of range: the store is dropped
            __builtin_amdgcn_raw_buffer_store_b128(rr[it], yr, off, 0, 2);
        }
    }
    stamp();
}

static size_t cellm_lds_bytes(int cp, int nt, int gpw, int nbt)
{
    return static_cast<size_t>(gpw) * 2 * cm_tile_bytes(cp, nt, nbt);
}
static int cellm_cp(int cg) { return cg <= 8 ? 8 : 16; }
// The tiling of a launch: NBT (column blocks per wave; nt = waves per row follows) and groups per workgroup, picked by a small cost model
// of what the per-wave phase stamps show (tools/gpu/cellm_stamps.py, profiles/NOTES_r05.md): a wave lives a fixed ~7 us (loads, the way
// in and out, four barriers) plus ~0.3 us per column-block step and node, whatever else runs beside it -- the kernel is bound by
// per-wave latency, so a launch takes (rounds of workgroups over the CU slots) x (life of a wave).  What varies with the tiling: the
// frames a row's last wave computes for nothing (1600 frames = 100 blocks: 7 x 16 wastes 12, 8 x 14 as well but runs 16 waves per CU
// instead of 14, 5 x 10 at 800 frames wastes none and fits three workgroups per CU), and the round count.  Rounds 3-4 fixed NBT = 16
// (8 for 16-slot groups at <= 1024 frames) and took the most groups per workgroup that left two workgroups per CU.
struct CellMPlan { int nbt, gpw, nt; };
static CellMPlan cellm_plan(int cg, int ld, int groups, int batch)
{
    static const int kNbt[4] = {16, 14, 10, 8};
    const int cp = cellm_cp(cg), nblk = (ld + 15) / 16;
    CellMPlan best{0, 0, 0};
    double best_cost = 0.0;
    for (int nbt : kNbt) {
        // (16-slot groups: the 10- and 14-block instances need more than 128 registers -- x0n carried for the last node's skip is 2 per
        // block -- and measured slower spilling than 8 blocks do with a workgroup less per CU: 93 vs 90 us at 32 x 1000 x 800)
        if (cp == 16 && (nbt == 10 || nbt == 14)) continue;
        const int nt = (nblk + nbt - 1) / nbt;
        for (int gpw = 1; gpw <= 4; gpw <<= 1) {                    // (ties go to the smaller barrier domain)
            if (groups % gpw || gpw * nt > 16) continue;
            const size_t lds = cellm_lds_bytes(cp, nt, gpw, nbt);
            if (lds > 160 * 1024) continue;
            const int by_lds = static_cast<int>(160 * 1024 / lds), by_waves = 16 / (gpw * nt);       // (128 registers: 4 waves per SIMD)
            const int per_cu = by_lds < by_waves ? by_lds : by_waves;
            const long wgs = static_cast<long>(groups / gpw) * (batch > 0 ? batch : 32);
            const long rounds = (wgs + 256L * per_cu - 1) / (256L * per_cu);
            // fixed part: ~4.8 us of loads and the way in / out + four barriers whose skew grows with the waves behind them
            double life = 4.8 + 0.3 * (gpw * nt) + 3 * 0.3 * (cp == 8 ? nbt / 2 : nbt);
            if (per_cu == 1) life *= 1.1;                            // (nothing to run beside a workgroup that waits)
            const double cost = rounds * life;
            if (best.nbt == 0 || cost < best_cost - 1e-9) { best = CellMPlan{nbt, gpw, nt}; best_cost = cost; }
        }
    }
    return best;
}
static int cellm_nstep(int /*cp*/, int kernel) { return (kernel + 1) / 2; }      // 2 taps per K step either way

template <int CP, int GPW, int NBT>
static int launch_cellm(const bf16_t* x0, bf16_t* y, const void* const* wp, const float* const* bias, const LnRef& l, const CellMDims& a, hipStream_t stream)
{
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_cell_mfma_kernel<CP, GPW, NBT>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) {
        set_error("nbasr_grouped_cell_mfma: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
        return static_cast<int>(attr);
    }
    hipLaunchKernelGGL((grouped_cell_mfma_kernel<CP, GPW, NBT>), dim3(a.groups / GPW, a.batch), dim3(64 * GPW * a.nt), cellm_lds_bytes(CP, a.nt, GPW, NBT), stream,
                       x0, y, static_cast<const cm_u4*>(wp[0]), static_cast<const cm_u4*>(wp[1]), static_cast<const cm_u4*>(wp[2]),
                       bias[0], bias[1], bias[2], l.stats, l.gamma, l.beta, a);
    return launch_status("nbasr_grouped_cell_mfma");
}

// w (channels, cg, kernel) fp32 -> [group][K step][lane] x 8 bf16: the A fragments of the node's MFMAs (zero outside the group's
// cg x cg x kernel block).  Lane (m = fragment row, kb): CP = 16: out channel m, tap = 2 s + (kb >> 1), channels 8 (kb & 1) .. + 7;
// CP = 8: out channel m & 7, tap = 2 s + (kb & 1), channels 0..7, and ZERO unless (m >> 3) == (kb >> 1): rows 0-7 multiply the K slots of
// the first column block of a pair, rows 8-15 those of the second.
__global__ __launch_bounds__(256) void pack_cell_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int groups, int cg,
                                                                int kernel, int cp, int nstep)
{
    const int total = groups * nstep * 64 * 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int e = i;
        const int j = e % 8; e /= 8;
        const int lane = e % 64; e /= 64;
        const int s = e % nstep; e /= nstep;
        const int g = e, m = lane & 15, kb = lane >> 4;
        const int tap = cp == 8 ? s * 2 + (kb & 1) : s * 2 + (kb >> 1);
        const int c = cp == 8 ? j : (kb & 1) * 8 + j;
        const int co = cp == 8 ? (m & 7) : m;
        const bool mine = cp != 8 || (m >> 3) == (kb >> 1);
        float v = 0.f;
        if (mine && co < cg && c < cg && tap < kernel) v = w[((static_cast<size_t>(g) * cg + co) * cg + c) * kernel + tap];
        packed[i] = static_cast<unsigned short>(pack_bf16x2(v, 0.f) & 0xffffu);
    }
}

}  // namespace nbasr

using namespace nbasr;

extern "C" size_t nbasr_grouped_cell_mfma_weights_bytes(int channels, int groups, int kernel)
{
    if (channels <= 0 || groups <= 0 || channels % groups || (kernel != 5 && kernel != 7)) return 0;
    const int cg = channels / groups;
    if (cg != 6 && cg != 8 && cg != 10 && cg != 12) return 0;
    return static_cast<size_t>(groups) * cellm_nstep(cellm_cp(cg), kernel) * 64 * 16;
}

extern "C" int nbasr_grouped_cell_mfma_pack(const float* w, void* packed, int channels, int groups, int kernel, nbasr_stream_t stream)
{
    clear_error();
    const size_t bytes = nbasr_grouped_cell_mfma_weights_bytes(channels, groups, kernel);
    NBASR_REQUIRE(bytes != 0, NBASR_EINVAL, "nbasr_grouped_cell_mfma_pack: channels=%d groups=%d kernel=%d unsupported (channels/groups in {6, 8, 10, 12}, kernel in {5, 7})",
                  channels, groups, kernel);
    NBASR_REQUIRE(w && packed, NBASR_ENULL, "nbasr_grouped_cell_mfma_pack: NULL pointer");
    NBASR_REQUIRE(aligned16(packed), NBASR_EALIGN, "nbasr_grouped_cell_mfma_pack: packed must be 16-byte aligned");
    const int cg = channels / groups, cp = cellm_cp(cg);
    hipLaunchKernelGGL(pack_cell_weights_kernel, dim3(256), dim3(256), 0, as_stream(stream), w, static_cast<unsigned short*>(packed), groups, cg, kernel,
                       cp, cellm_nstep(cp, kernel));
    return launch_status("nbasr_grouped_cell_mfma_pack");
}

extern "C" int nbasr_grouped_cell_mfma_fits(int channels, int frames_ld, int groups)
{
    if (channels <= 0 || groups <= 0 || channels % groups || frames_ld <= 0 || frames_ld % 8) return 0;
    const int cg = channels / groups;
    if (cg != 6 && cg != 8 && cg != 10 && cg != 12) return 0;
    return cellm_plan(cg, frames_ld, groups, 0).gpw;
}

extern "C" int nbasr_grouped_cell_mfma(const void* x0, const void* wp0, const float* b0, int k0, int d0,
                                       const void* wp1, const float* b1, int k1, int d1,
                                       const void* wp2, const float* b2, int k2, int d2, int skip_mask, void* y,
                                       int batch, int channels, int frames, int ld, int groups,
                                       const nbasr_deferred_ln* ln, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "nbasr_grouped_cell_mfma: bad sizes batch=%d channels=%d frames=%d groups=%d", batch, channels, frames, groups);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x0 && wp0 && b0 && wp1 && b1 && wp2 && b2 && y, NBASR_ENULL, "nbasr_grouped_cell_mfma: NULL pointer");
    NBASR_REQUIRE(ld >= frames && ld % 8 == 0 && aligned16(x0) && aligned16(y) && aligned16(wp0) && aligned16(wp1) && aligned16(wp2), NBASR_EALIGN,
                  "nbasr_grouped_cell_mfma: ld=%d must be >= frames=%d and a multiple of 8; x0, y and the packed weights 16-byte aligned", ld, frames);
    NBASR_REQUIRE(batch <= 65535 && skip_mask >= 0 && skip_mask < 64, NBASR_EINVAL, "nbasr_grouped_cell_mfma: bad batch / skip mask");
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta && aligned16(ln->stats)), NBASR_ENULL,
                  "nbasr_grouped_cell_mfma: deferred LayerNorm needs stats (16-byte aligned), gamma and beta");
    NBASR_REQUIRE(nbasr_grouped_cell_mfma_fits(channels, ld, groups) != 0, NBASR_EINVAL,
                  "nbasr_grouped_cell_mfma: a row of %d frames x %d channels per group does not fit one workgroup (channels/groups in {6, 8, 10, 12}, "
                  "two bf16 tiles per group within 160 KiB of LDS); use nbasr_grouped_cell_fused or the per-node launches", ld, channels / groups);
    const int ks[3] = {k0, k1, k2}, ds[3] = {d0, d1, d2};
    CellMDims a{};
    a.channels = channels; a.frames = frames; a.ld = ld; a.groups = groups; a.batch = batch; a.cg = channels / groups;
    const int cp = cellm_cp(a.cg);
    for (int i = 0; i < 3; ++i) {
        NBASR_REQUIRE((ks[i] == 5 || ks[i] == 7) && (ds[i] == 1 || ds[i] == 2), NBASR_EINVAL,
                      "nbasr_grouped_cell_mfma: node %d has (kernel=%d, dilation=%d); the search space has k in {5, 7}, d in {1, 2}", i, ks[i], ds[i]);
        a.k[i] = ks[i]; a.d[i] = ds[i]; a.lpad[i] = pad_left(ks[i], ds[i], 1); a.nstep[i] = cellm_nstep(cp, ks[i]);
    }
    a.skips = skip_mask;
    CellMPlan plan = cellm_plan(a.cg, ld, groups, batch);
    if (const char* e = getenv("NBASR_CELLM_TILING")) {              // diagnostics (tools/ubench/bench_cell_mfma.py --sweep): "nbt,gpw"
        int nbt_e = 0, gpw_e = 0;
        if (sscanf(e, "%d,%d", &nbt_e, &gpw_e) == 2 && (nbt_e == 8 || nbt_e == 16 || (cp == 8 && (nbt_e == 10 || nbt_e == 14))) && (gpw_e == 1 || gpw_e == 2 || gpw_e == 4)) {
            const int nt_e = ((ld + 15) / 16 + nbt_e - 1) / nbt_e;
            NBASR_REQUIRE(groups % gpw_e == 0 && gpw_e * nt_e <= 16 && cellm_lds_bytes(cp, nt_e, gpw_e, nbt_e) <= 160 * 1024, NBASR_EINVAL,
                          "nbasr_grouped_cell_mfma: NBASR_CELLM_TILING=%s does not fit this row", e);
            plan = CellMPlan{nbt_e, gpw_e, nt_e};
        }
    }
    const int nbt = plan.nbt, gpw = plan.gpw;
    a.nt = plan.nt;
#if NBASR_CELLM_STAMPS
    { const char* e = getenv("NBASR_CELLM_STAMPS"); a.stamps = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
    const void* const wp[3] = {wp0, wp1, wp2};
    const float* const bias[3] = {b0, b1, b2};
    const LnRef l = ln_ref(ln, true);
    const bf16_t* xin = static_cast<const bf16_t*>(x0);
    bf16_t* yout = static_cast<bf16_t*>(y);
    hipStream_t s = as_stream(stream);
#define NBASR_CELLM_LAUNCH(CP_, NBT_)                                                            \
    do {                                                                                        \
        if (gpw == 4) return launch_cellm<CP_, 4, NBT_>(xin, yout, wp, bias, l, a, s);          \
        if (gpw == 2) return launch_cellm<CP_, 2, NBT_>(xin, yout, wp, bias, l, a, s);          \
        return launch_cellm<CP_, 1, NBT_>(xin, yout, wp, bias, l, a, s);                        \
    } while (0)
    if (cp == 8) {
        if (nbt == 8) NBASR_CELLM_LAUNCH(8, 8);
        if (nbt == 10) NBASR_CELLM_LAUNCH(8, 10);
        if (nbt == 14) NBASR_CELLM_LAUNCH(8, 14);
        NBASR_CELLM_LAUNCH(8, 16);
    }
    if (nbt == 8) NBASR_CELLM_LAUNCH(16, 8);
    NBASR_CELLM_LAUNCH(16, 16);
#undef NBASR_CELLM_LAUNCH
}
