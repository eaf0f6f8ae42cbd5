// A whole SearchCell whose three node operations are grouped convolutions, in ONE launch
// (reference model.py:49-59 SearchCell.forward over model.py:13-22 Node.forward and ops.py:24-30 PadConvRelu):
//     x1 = op0(x0n) + s00 x0n;   x2 = op1(x1) + s10 x0n + s11 x1;   x3 = op2(x2) + s20 x0n + s21 x1 + s22 x2
// with x0n = the cell input, optionally still carrying its LayerNorm (applied while loading, nbasr.h), and -- round 3 -- the
// LayerNorm statistics of x3 as a by-product (the cell's own LayerNorm is then applied by whoever loads x3).
//
// Channel groups never mix inside such a cell, so x1 and x2 never have to touch HBM: HBM traffic per cell drops from 6 tensor passes
// (+ skip re-reads) to 1 read + 1 write, and the node op -- HBM-bound as three launches (0.55 of 8 TB/s) -- becomes bound by the
// vector ALU.  The arithmetic is the per-node kernel's, in the same order: bit-identical to three nbasr_grouped_conv1d_node launches
// (+ nbasr_grouped_stats_finalize), asserted in tests/.
//
// Round 3 form (round 2's: one workgroup per (utterance, group) row with TWO LDS tiles, 1-3 waves per SIMD, no statistics -- a win in
// block 0 only, and the separate statistics pass took that back):
//  * workgroup = (utterance, 4 or 2 groups) x the whole row: GPW groups x nt waves (nt = row length / 256 frames, <= 4): with 4
//    groups the statistics partials are the node kernel's ([group quad][batch][2][ld], merged by stats_finalize_kernel); neighbouring
//    tiles of a row exchange their halo through the LDS tile itself;
//  * ONE LDS tile per group ([CG][row + zero pads] floats) holds the INPUT of the node being computed; a node's output stays in the
//    accumulator registers (a lane owns 4 frames x CG channels), gets its skip sum there, and is written to the tile only as the next
//    node's input.  x0n and x2 as skip inputs are read back from the tile while it still holds them, x1 for node 2 (s21) is kept in a
//    second register set (template KEEP1), x0n for nodes 1 / 2 is re-read from HBM;
//  * the cell input is loaded as each lane's OWN chunk of every channel (one coalesced 1 KiB wave access per row, all CG rows in
//    flight at once), normalised ONCE per element and dealt out through the tile -- the per-node kernel loads (and normalises) every
//    element once per window chunk;
//  * the window of input channel ci + 1 is read from LDS while channel ci multiplies.
//
// Round 4 -- the vector stream is the FMAs and little else (41-48 % of the vector instructions were v_pk_fma_f32,
// 16-21 % v_readlane / v_writelane of spilled scalar registers, 14-21 % moves):
//  * a packed FMA covers TWO OUTPUT CHANNELS of one frame (round 3: two frames of one channel).  Its weight operand is then a scalar
//    register PAIR -- the weights arrive as [group][ci][tap][co] (nbasr_pack_grouped_weights), so the CG weights of a (ci, tap) are
//    one wide scalar load -- and its x operand ONE window register broadcast to both halves by op_sel: any register of the window
//    is addressable (no alignment, no shifted copies; hipcc only folds a splat of an EVEN register, hence the inline assembly);
//  * one input channel per trip: its CG K scalars (30-84) are the only weights alive -- nothing spills --, requested together with
//    the channel's window (ds_read_b128) and waited for once; the other waves of the SIMD cover that wait;
//  * the same sums in the same order as before (bias, then ci-major, tap-minor, one fma each): still bit-identical to the node kernels.
//
// Algorithmic bytes credited per launch (bench.py): x0 in, y out, the weights once (round 3 credited the three node operations).
#include "storage.h"

#include <algorithm>
#include <cstddef>
#include <cstdlib>

#ifndef NBASR_CELL_STAMPS
#define NBASR_CELL_STAMPS 0
#endif
#ifndef NBASR_CELL_BATCH
#define NBASR_CELL_BATCH 60          // most scalar weights requested (and waited for) at once
#endif
#include <type_traits>

namespace nbasr {

template <int K, int D>
struct Win {
    static constexpr int LPAD = pad_left(K, D, 1);
    static constexpr int SPAN = (K - 1) * D;            // taps reach frames [t - LPAD, t - LPAD + SPAN]
    static constexpr int QL = (LPAD + 3) / 4;           // whole chunks left of the lane's own chunk
    static constexpr int QR = (SPAN - LPAD + 3) / 4;    // whole chunks right of it
    static constexpr int NCH = QL + 1 + QR;
    static constexpr int BASE = 4 * QL - LPAD;          // window index of (r = 0, tap = 0)
};

constexpr int CELL_PADL = 2, CELL_PADR = 1;             // the most zero chunks a tile row can need on either side (QL <= 2, QR <= 1: the conv's padding)
// A tile row holds exactly the row's ld / 4 chunks plus the pad chunks THIS cell's three (taps, dilation) pairs reach (round 4; rounds
// 2-3: 64 chunks per wave + 2 + 2).  At 1000 frames that is 251-253 chunks instead of 260 -- and five instead of four one-group
// workgroups in a CU's 160 KiB at 8 channels per group (block 1), eight instead of seven at 10 (block 2)
__host__ __device__ constexpr int cell_ql(int kd) { return kd == 0 ? Win<5, 1>::QL : kd == 1 ? Win<5, 2>::QL : kd == 2 ? Win<7, 1>::QL : Win<7, 2>::QL; }
__host__ __device__ constexpr int cell_qr(int kd) { return kd == 0 ? Win<5, 1>::QR : kd == 1 ? Win<5, 2>::QR : kd == 2 ? Win<7, 1>::QR : Win<7, 2>::QR; }

struct CellDims {
    int channels, frames, ld, groups, batch;
    int kd0, kd1, kd2;          // 0: k5 d1, 1: k5 d2, 2: k7 d1, 3: k7 d2
    int skips;                  // bit0 s00 | bit1 s10 | bit2 s11 | bit3 s20 | bit4 s21 | bit5 s22
    int nt;                     // 64-chunk tiles (waves) per row
    int padl, padr;             // zero chunks left / right of a tile row: the largest QL / QR of the three nodes
    int rowq;                   // data chunks of a tile row
#if NBASR_CELL_STAMPS
    unsigned long long* stamps; // [workgroup][16]: HW_ID, XCC_ID, then the 100 MHz clock at the phase boundaries of wave 0
#endif
};

typedef float cell_f4 __attribute__((ext_vector_type(4)));
typedef unsigned cell_u4 __attribute__((ext_vector_type(4)));

// Bounds-checked 16-byte accesses to ONE row (raw buffer: out-of-range lanes read zeros / store nothing).  No predicate, no branch:
// hipcc waits for vmcnt(0) at every control-flow join, so a lane's CG loads (or stores) each behind their own `if (in_row)` would be
// issued one round trip after the other (the first version of this kernel: 12 exec-masked branches around 12 loads).
__device__ __forceinline__ float4 cell_load4(const float* row, int row_bytes, int byte_offset)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_bytes, 0x00020000);
    const cell_f4 f = __builtin_bit_cast(cell_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_offset, 0, 0));
    return make_float4(f[0], f[1], f[2], f[3]);
}
// a lane's 4 frames of a row in either storage type (fp32: 16 bytes; bf16: 8 bytes, widened exactly); `chunk` = the lane's chunk index
__device__ __forceinline__ float4 cell_load_frames(const float* row, int ld, int chunk) { return cell_load4(row, ld * 4, chunk * 16); }
__device__ __forceinline__ float4 cell_load_frames(const bf16_t* row, int ld, int chunk)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(row), 0, ld * 2, 0x00020000);
    const u2v t = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(rs, chunk * 8, 0, 0));
    return make_float4(bf16_lo(t.x), bf16_hi(t.x), bf16_lo(t.y), bf16_hi(t.y));
}
// streaming store of a lane's 4 frames (bf16: ONE rounding, here); row_len = the row's pitch, or 0 to drop the store
__device__ __forceinline__ void cell_store_frames(float* row, int row_len, int chunk, const float (&o)[4])
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, row_len * 4, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cell_u4, cell_f4{o[0], o[1], o[2], o[3]}), rs, chunk * 16, 0, 2);   // aux 2 = nt
}
__device__ __forceinline__ void cell_store_frames(bf16_t* row, int row_len, int chunk, const float (&o)[4])
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, row_len * 2, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b64(u2v{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])}, rs, chunk * 8, 0, 2);
}
// what a stored-and-reloaded value is: itself (fp32) or its bfloat16 rounding (the bf16 path rounds every tensor once, when stored)
template <typename T> __device__ __forceinline__ void cell_round(float (&o)[4])
{
    if constexpr (sizeof(T) == 2) {
        const unsigned a = pack_bf16x2(o[0], o[1]), b = pack_bf16x2(o[2], o[3]);
        o[0] = bf16_lo(a); o[1] = bf16_hi(a); o[2] = bf16_lo(b); o[3] = bf16_hi(b);
    }
}

typedef float cell_f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float* cell_cptr;      // wave-uniform read-only data: scalar loads
__device__ __forceinline__ cell_cptr cell_const(const float* p) { return (cell_cptr)p; }

// N (even, <= 40) consecutive floats at a wave-uniform address -> scalar registers, by ONE statement that also waits for them.
// hipcc, left to place the scalar loads of a weight batch itself, schedules for the scalar-register pressure it sees BEFORE register
// allocation (the kernel's ~50 long-lived scalars count in full) and fetched eight weights at a time with a full wait after each: 7-11
// exposed scalar-cache round trips per input channel.  Written out, the batch is requested at once and waited for once, and the
// allocator moves long-lived scalars out of the way around the loop (a few v_writelane / v_readlane per node, none inside).
typedef float cell_f8 __attribute__((ext_vector_type(8)));
typedef float cell_f16 __attribute__((ext_vector_type(16)));
template <int N>
struct WeightBatch {
    static_assert(N % 2 == 0 && N >= 2 && N <= 62, "an even number of scalars, at most 3 x 16 + 8 + 4 + 2");
    static constexpr int N16 = N / 16, N8 = (N % 16) / 8, N4 = (N % 8) / 4, N2 = (N % 4) / 2;
    cell_f16 q0, q1, q2; cell_f8 o; cell_f4 f; cell_f2 t;
    __device__ __forceinline__ cell_f2 pair(int i) const     // scalars (2i, 2i + 1); i is a constant after unrolling
    {
        int e = 2 * i;
        if (e < 16 * N16) {
            const int w = e & 15;
            return e < 16 ? cell_f2{q0[w], q0[w + 1]} : e < 32 ? cell_f2{q1[w], q1[w + 1]} : cell_f2{q2[w], q2[w + 1]};
        }
        e -= 16 * N16;
        if (N8 && e < 8) return cell_f2{o[e & 7], o[(e & 7) + 1]};
        e -= 8 * N8;
        if (N4 && e < 4) return cell_f2{f[e & 3], f[(e & 3) + 1]};
        return t;
    }
    __device__ __forceinline__ void load(cell_cptr p)
    {
        constexpr int O8 = 64 * N16, O4 = O8 + 32 * N8, O2 = O4 + 16 * N4;      // byte offsets of the 8- / 4- / 2-dword pieces
#define NBASR_WB_CASE(n16, n8, n4, n2, text, ...)                                                    \
        if constexpr (N16 == n16 && N8 == n8 && N4 == n4 && N2 == n2)                                \
            asm volatile(text "\n\ts_waitcnt lgkmcnt(0)" : __VA_ARGS__ : [p] "s"(p), [o8] "n"(O8), [o4] "n"(O4), [o2] "n"(O2));
#define L16A "s_load_dwordx16 %[q0], %[p], 0x0"
#define L16B "\n\ts_load_dwordx16 %[q1], %[p], 0x40"
#define L16C "\n\ts_load_dwordx16 %[q2], %[p], 0x80"
#define L8 "\n\ts_load_dwordx8 %[o], %[p], %[o8]"
#define L4 "\n\ts_load_dwordx4 %[f], %[p], %[o4]"
#define L2 "\n\ts_load_dwordx2 %[t], %[p], %[o2]"
#define Q0 [q0] "=&s"(q0)
#define Q1 [q1] "=&s"(q1)
#define Q2 [q2] "=&s"(q2)
#define O_ [o] "=&s"(o)
#define F_ [f] "=&s"(f)
#define T_ [t] "=&s"(t)
        NBASR_WB_CASE(0, 1, 1, 0, "s_load_dwordx8 %[o], %[p], 0x0" L4, O_, F_)                       // 12
        NBASR_WB_CASE(1, 0, 0, 0, L16A, Q0)                                                            // 16
        NBASR_WB_CASE(1, 0, 0, 1, L16A L2, Q0, T_)                                                     // 18
        NBASR_WB_CASE(1, 0, 1, 0, L16A L4, Q0, F_)                                                     // 20
        NBASR_WB_CASE(1, 1, 0, 0, L16A L8, Q0, O_)                                                     // 24
        NBASR_WB_CASE(1, 1, 1, 1, L16A L8 L4 L2, Q0, O_, F_, T_)                                       // 30
        NBASR_WB_CASE(2, 0, 0, 0, L16A L16B, Q0, Q1)                                                   // 32
        NBASR_WB_CASE(2, 0, 1, 0, L16A L16B L4, Q0, Q1, F_)                                            // 36
        NBASR_WB_CASE(2, 1, 0, 0, L16A L16B L8, Q0, Q1, O_)                                            // 40
        NBASR_WB_CASE(2, 1, 0, 1, L16A L16B L8 L2, Q0, Q1, O_, T_)                                     // 42
        NBASR_WB_CASE(3, 0, 0, 0, L16A L16B L16C, Q0, Q1, Q2)                                          // 48
        NBASR_WB_CASE(3, 0, 0, 1, L16A L16B L16C L2, Q0, Q1, Q2, T_)                                   // 50
        NBASR_WB_CASE(3, 1, 0, 0, L16A L16B L16C L8, Q0, Q1, Q2, O_)                                   // 56
        NBASR_WB_CASE(3, 1, 1, 0, L16A L16B L16C L8 L4, Q0, Q1, Q2, O_, F_)                            // 60
        static_assert(N == 12 || N == 16 || N == 18 || N == 20 || N == 24 || N == 30 || N == 32 || N == 36 || N == 40 || N == 42 || N == 48 ||
                      N == 50 || N == 56 || N == 60, "batch size without a load pattern");
#undef NBASR_WB_CASE
#undef L16A
#undef L16B
#undef L16C
#undef L8
#undef L4
#undef L2
#undef Q0
#undef Q1
#undef Q2
#undef O_
#undef F_
#undef T_
    }
};

// taps J0 .. J0 + NT - 1 of one input channel: its NT * CG weights (wt: [tap][co]) in one batch, then the FMAs.  One packed FMA: the
// weight pair from scalar registers, x = word (e & 1) of the aligned register pair that holds window element e (after unrolling e
// is a constant and one of the two statements remains)
template <int CG, int K, int D, int J0, int NT>
__device__ __forceinline__ void conv_taps(cell_f2 (&acc)[CG / 2][4], cell_cptr wt, const cell_f4 (&xq)[Win<K, D>::NCH])
{
    using W = Win<K, D>;
    WeightBatch<NT * CG> wb;
    wb.load(wt);
#pragma unroll
    for (int j = J0; j < J0 + NT; ++j) {
#pragma unroll
        for (int p = 0; p < CG / 2; ++p) {
            const cell_f2 wv = wb.pair((j - J0) * (CG / 2) + p);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = W::BASE + r + j * D;
                const cell_f2 xp = {xq[e / 4][(e & 3) & ~1], xq[e / 4][(e & 3) | 1]};
                if (e & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[p][r]) : "s"(wv), "v"(xp));
                else       asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc[p][r]) : "s"(wv), "v"(xp));
            }
        }
    }
}

// Round 6 -- a wave that computes only CH of the group's CG output channels (the output-channel split of small batches, see the
// kernel): of every (input channel, tap) it needs CH consecutive weights out of CG, i.e. K pieces at a pitch of CG floats.  All K
// taps of an input channel are requested by one statement and waited for once, like WeightBatch (K * CH <= 42 scalars).
template <int K, int CH, int CG>
struct StridedBatch {
    static_assert((CH == 6 && CG == 12) || (CH == 4 && CG == 8), "output-channel halves: 6 of 12 or 4 of 8");
    static_assert(K == 5 || K == 7, "5 or 7 taps");
    cell_f4 a[K]; cell_f2 b[K];                              // tap j: channels 0-3 in a[j], 4-5 (CH == 6) in b[j]
    __device__ __forceinline__ cell_f2 pair(int i) const     // output channels (2 pp, 2 pp + 1) of tap jj; i = jj * (CH / 2) + pp
    {
        const int jj = i / (CH / 2), pp = i % (CH / 2);
        return pp < 2 ? cell_f2{a[jj][2 * pp], a[jj][2 * pp + 1]} : b[jj];
    }
    __device__ __forceinline__ void load(cell_cptr p)
    {
#define A_(j) [a##j] "=&s"(a[j])
#define B_(j) [b##j] "=&s"(b[j])
        if constexpr (CH == 6 && K == 5)
            asm volatile("s_load_dwordx4 %[a0], %[p], 0x0\n\ts_load_dwordx2 %[b0], %[p], 0x10\n\ts_load_dwordx4 %[a1], %[p], 0x30\n\ts_load_dwordx2 %[b1], %[p], 0x40\n\t"
                         "s_load_dwordx4 %[a2], %[p], 0x60\n\ts_load_dwordx2 %[b2], %[p], 0x70\n\ts_load_dwordx4 %[a3], %[p], 0x90\n\ts_load_dwordx2 %[b3], %[p], 0xa0\n\t"
                         "s_load_dwordx4 %[a4], %[p], 0xc0\n\ts_load_dwordx2 %[b4], %[p], 0xd0\n\ts_waitcnt lgkmcnt(0)"
                         : A_(0), B_(0), A_(1), B_(1), A_(2), B_(2), A_(3), B_(3), A_(4), B_(4) : [p] "s"(p));
        if constexpr (CH == 6 && K == 7)
            asm volatile("s_load_dwordx4 %[a0], %[p], 0x0\n\ts_load_dwordx2 %[b0], %[p], 0x10\n\ts_load_dwordx4 %[a1], %[p], 0x30\n\ts_load_dwordx2 %[b1], %[p], 0x40\n\t"
                         "s_load_dwordx4 %[a2], %[p], 0x60\n\ts_load_dwordx2 %[b2], %[p], 0x70\n\ts_load_dwordx4 %[a3], %[p], 0x90\n\ts_load_dwordx2 %[b3], %[p], 0xa0\n\t"
                         "s_load_dwordx4 %[a4], %[p], 0xc0\n\ts_load_dwordx2 %[b4], %[p], 0xd0\n\ts_load_dwordx4 %[a5], %[p], 0xf0\n\ts_load_dwordx2 %[b5], %[p], 0x100\n\t"
                         "s_load_dwordx4 %[a6], %[p], 0x120\n\ts_load_dwordx2 %[b6], %[p], 0x130\n\ts_waitcnt lgkmcnt(0)"
                         : A_(0), B_(0), A_(1), B_(1), A_(2), B_(2), A_(3), B_(3), A_(4), B_(4), A_(5), B_(5), A_(6), B_(6) : [p] "s"(p));
        if constexpr (CH == 4 && K == 5)
            asm volatile("s_load_dwordx4 %[a0], %[p], 0x0\n\ts_load_dwordx4 %[a1], %[p], 0x20\n\ts_load_dwordx4 %[a2], %[p], 0x40\n\ts_load_dwordx4 %[a3], %[p], 0x60\n\t"
                         "s_load_dwordx4 %[a4], %[p], 0x80\n\ts_waitcnt lgkmcnt(0)"
                         : A_(0), A_(1), A_(2), A_(3), A_(4) : [p] "s"(p));
        if constexpr (CH == 4 && K == 7)
            asm volatile("s_load_dwordx4 %[a0], %[p], 0x0\n\ts_load_dwordx4 %[a1], %[p], 0x20\n\ts_load_dwordx4 %[a2], %[p], 0x40\n\ts_load_dwordx4 %[a3], %[p], 0x60\n\t"
                         "s_load_dwordx4 %[a4], %[p], 0x80\n\ts_load_dwordx4 %[a5], %[p], 0xa0\n\ts_load_dwordx4 %[a6], %[p], 0xc0\n\ts_waitcnt lgkmcnt(0)"
                         : A_(0), A_(1), A_(2), A_(3), A_(4), A_(5), A_(6) : [p] "s"(p));
#undef A_
#undef B_
    }
};

// all K taps of one input channel for a wave's CH output channels (wt: [tap][CG], already at the wave's first channel)
template <int CG, int CH, int K, int D>
__device__ __forceinline__ void conv_taps_strided(cell_f2 (&acc)[CH / 2][4], cell_cptr wt, const cell_f4 (&xq)[Win<K, D>::NCH])
{
    using W = Win<K, D>;
    StridedBatch<K, CH, CG> wb;
    wb.load(wt);
#pragma unroll
    for (int j = 0; j < K; ++j) {
#pragma unroll
        for (int p = 0; p < CH / 2; ++p) {
            const cell_f2 wv = wb.pair(j * (CH / 2) + p);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = W::BASE + r + j * D;
                const cell_f2 xp = {xq[e / 4][(e & 3) & ~1], xq[e / 4][(e & 3) | 1]};
                if (e & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[p][r]) : "s"(wv), "v"(xp));
                else       asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc[p][r]) : "s"(wv), "v"(xp));
            }
        }
    }
}

// acc = bias + conv over one group's CG input channels, the input read from this group's LDS tile (row pitch rl floats; a lane's own
// chunk of channel ci at row[ci * rl + 4 * col]).  acc[p][r] = output channels (2p, 2p + 1) of the wave's CH at the lane's frame r;
// wg = this group's weights as [ci][tap][co], bg its bias, both already at the wave's first output channel (CH == CG: the whole group)
template <int CG, int CH, int K, int D>
__device__ __forceinline__ void conv_from_tile(cell_f2 (&acc)[CH / 2][4], cell_cptr wg, cell_cptr bg, const float* tile, int rl, int col)
{
    using W = Win<K, D>;
#pragma unroll
    for (int p = 0; p < CH / 2; ++p) {
        const cell_f2 bv = {bg[2 * p], bg[2 * p + 1]};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[p][r] = bv;
    }
    const float* win = tile + 4 * (col - W::QL);
    // the weights of an input channel in batches of TB taps, each at most 40 scalars: what is alive beside the kernel's own
    // scalars (pointers, sizes) then fits the 102 scalar registers
    constexpr int NB = (K * CG + NBASR_CELL_BATCH - 1) / NBASR_CELL_BATCH, TB = (K + NB - 1) / NB;
    static_assert(NB <= 3, "at most three weight batches per input channel");
#pragma unroll 1
    for (int ci = 0; ci < CG; ++ci) {
        cell_f4 xq[W::NCH];
#pragma unroll
        for (int c = 0; c < W::NCH; ++c) xq[c] = *reinterpret_cast<const cell_f4*>(win + ci * rl + 4 * c);
        const cell_cptr wc = wg + ci * (K * CG);
        if constexpr (CH != CG) conv_taps_strided<CG, CH, K, D>(acc, wc, xq);
        else {
        conv_taps<CG, K, D, 0, (TB < K ? TB : K)>(acc, wc, xq);
        if constexpr (NB >= 2) {
            __builtin_amdgcn_sched_barrier(0);      // (this batch's loads stay behind the previous batch's FMAs)
            conv_taps<CG, K, D, TB, (2 * TB < K ? TB : K - TB)>(acc, wc + TB * CG, xq);
        }
        if constexpr (NB >= 3) {
            __builtin_amdgcn_sched_barrier(0);
            conv_taps<CG, K, D, 2 * TB, K - 2 * TB>(acc, wc + 2 * TB * CG, xq);
        }
        }
    }
}

// GPW: groups per workgroup -- 4 (one-wave rows: the statistics partials are then the node kernel's group quads, bit for bit), 2 (kept
// for one-wave rows whose four tiles would not leave room for two workgroups per CU), or 1 (rows of several waves, round 4: see
// cell_gpw); the partials are then per group pair / per group, merged by stats_finalize_kernel with groups_per_part = 2 / 1 -- the
// same statistics to rounding, not bit for bit.
// NTB: the largest number of tiles per row this instantiation is launched with (1, 2 or 4): its register budget is that of a
// 64 * GPW * NTB-thread workgroup (128 registers at 1024 threads; the narrower forms may use more)
// T: storage type of the cell input and output (float, or bf16_t: the bf16 path -- x1 and x2 are then rounded to bfloat16 exactly
// where the three-launch form stores them, the statistics describe x3 before its rounding)
// The kernel's arguments as ONE by-value struct (the kernarg segment).  Pointers that are needed only late in the kernel -- the
// weights and biases of nodes 1 and 2, y, the LayerNorm vectors, the partials -- are NOT read through it: kept alive across the three
// convolution loops they left the loops ~16 scalar registers for weights, and hipcc, scheduling for that pressure, fetched the
// weights eight at a time with a full wait after each.  cell_arg<>() re-reads such a pointer from the kernarg segment where it is used.
template <typename T>
struct CellArgs {
    const T* x0; T* y;
    const float* w0; const float* w1; const float* w2;
    const float* b0; const float* b1; const float* b2;
    const float* ln_stats; const float* ln_gamma; const float* ln_beta;
    float* part;
    CellDims a;
};
// (an asm result counts as divergent and a re-read pointer as generic: hipcc would fetch the weights behind it by flat vector loads.
// v_readfirstlane makes it uniform again; the users cast it to the constant address space, i.e. to scalar loads)
#define NBASR_CELL_ARG(T_, field) reinterpret_cast<decltype(CellArgs<T_>::field)>(cell_arg<offsetof(CellArgs<T_>, field)>())
template <int OFFSET>
__device__ __forceinline__ uintptr_t cell_arg()
{
    uint64_t p;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(p) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFFSET));
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(p)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(p >> 32));
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// OS (round 6): waves per (group, row tile) -- 2 = the output channels of a group split over two waves (CG 12 -> 6 + 6, 8 -> 4 + 4).
// At 8-16 utterances a one-wave row in block 3 leaves the chip with less than one wave per SIMD, and a wave's life is one dependent
// chain of LDS windows and FMAs; two waves per row halve the FMAs of each (both still read every input channel's window).  Every
// output channel's sum is formed by one wave in the same order as before and the statistics are taken over all CG channels in channel
// order by the first wave of the pair (the second hands its x3 over through the tile): bit-identical to OS = 1.
template <typename T, int CG, bool KEEP1, int NTB, int GPW, int OS = 1>
__global__ __launch_bounds__(64 * GPW * NTB * OS) void grouped_cell_kernel(const CellArgs<T> A)
{
    constexpr int CH = CG / OS;                              // output channels of this wave
    static_assert(CG % OS == 0 && CH % 2 == 0, "a wave's output channels come in pairs");
    extern __shared__ __attribute__((aligned(16))) float cell_tiles[];
    const CellDims& a = A.a;
    const int nt = a.nt;
    const int nq = a.ld >> 2;                                // chunks of a row
    const int rl = (a.rowq + a.padl + a.padr) * 4;           // tile row length in floats
    const int lane = threadIdx.x & 63;
    const int wave_raw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int oh = OS > 1 ? wave_raw % OS : 0;               // which part of the output channels (neighbouring waves: different SIMDs)
    const int c_base = oh * CH;                              // this wave's first channel within the group
    const int wave = wave_raw / OS;
    const int gi = wave / nt, ti = wave - gi * nt;           // group within the quad, tile within the row
    float* const tile = cell_tiles + gi * (CG * rl);         // this group's tile: [CG][rl]

    const int g_raw = blockIdx.x * GPW + gi, b = blockIdx.y; // wave-uniform: weights / bias / gamma / beta come through s_load
    const bool g_ok = g_raw < a.groups;                       // a surplus wave of the last quad recomputes the last group, stores nothing
    const int g = g_ok ? g_raw : a.groups - 1;
    const int q = ti * 64 + lane;                            // chunk within the row
    const int col = a.padl + q;                              // its column (in chunks) in the tile row
    const bool in_row = q < nq;
    const int t0 = q * 4;
    const size_t row0 = (static_cast<size_t>(b) * a.channels + static_cast<size_t>(g) * CG) * a.ld;
    const bool has_ln = A.ln_stats != nullptr;

    // -DNBASR_CELL_STAMPS=1 (NBASR_EXTRA_CXXFLAGS; tools/gpu/cell_stamps.py): wave 0 of every workgroup records the 100 MHz clock at its
    // phase boundaries -- how round 4 found that a workgroup spends half of its life outside the convolution loops
#if NBASR_CELL_STAMPS
    unsigned long long* const stamp_row = a.stamps ? a.stamps + (static_cast<size_t>(blockIdx.x) + static_cast<size_t>(gridDim.x) * blockIdx.y) * 16 : nullptr;
    int stamp_i = 2;
    auto stamp = [&]() {
        if (stamp_row && threadIdx.x == 0) stamp_row[stamp_i] = wall_clock64();
        ++stamp_i;
    };
    if (stamp_row && threadIdx.x == 0) {
        stamp_row[0] = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_ID
        stamp_row[1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));      // XCC_ID
    }
#else
    auto stamp = []() {};
#endif
    stamp();
    // the zero pads of every tile row (never written again)
    const int npad = a.padl + a.padr;
    for (int i = threadIdx.x; i < GPW * CG * npad; i += blockDim.x) {
        const int row = i / npad, p = i - row * npad;
        const int c = p < a.padl ? p : a.rowq + p;
        *reinterpret_cast<cell_f4*>(cell_tiles + row * rl + 4 * c) = cell_f4{0.f, 0.f, 0.f, 0.f};
    }

    const int row_bytes = a.ld * 4, boff = q * 16;  // a statistics row as a bounds-checked buffer; this lane's chunk in it (beyond the row: zeros)
    // statistics of this lane's own 4 frames (pending LayerNorm of the cell input); beyond the row rstd = 0, i.e. normalised = 0
    // kept as frame PAIRS (-mean, rstd) so that the normalisation is packed arithmetic: (x + -mean) * rstd, fma(., gamma, beta)
    // rounds exactly like ln_apply -- 1.5 instructions per element instead of 4.  ln_apply's "exactly 0 where rstd == 0" (the
    // frames beyond the row) is NOT in here: every user passes its result through mask_tail (below), which zeroes those frames
    typedef float cell_f2 __attribute__((ext_vector_type(2)));
    cell_f2 nm01{0.f, 0.f}, nm23 = nm01, sr01 = nm01, sr23 = nm01;
    if (has_ln) {                                   // (workgroup-uniform)
        const float* mrow = A.ln_stats + static_cast<size_t>(b) * 2 * a.ld;
        const float4 sm = cell_load4(mrow, row_bytes, boff);
        const float4 sr = cell_load4(mrow + a.ld, row_bytes, boff);
        nm01 = cell_f2{-sm.x, -sm.y}; nm23 = cell_f2{-sm.z, -sm.w};
        sr01 = cell_f2{sr.x, sr.y};   sr23 = cell_f2{sr.z, sr.w};
    }
    // (gamma / beta / the input rows through pointers handed in: the late users pass freshly re-read ones)
    auto normalise = [&](float4 v, int co, const float* gamma, const float* beta) -> float4 {
        if (has_ln) {
            const float gam = cell_const(gamma)[g * CG + co], bet = cell_const(beta)[g * CG + co];
            const cell_f2 g2{gam, gam}, b2{bet, bet};
            const cell_f2 lo = __builtin_elementwise_fma((cell_f2{v.x, v.y} + nm01) * sr01, g2, b2);
            const cell_f2 hi = __builtin_elementwise_fma((cell_f2{v.z, v.w} + nm23) * sr23, g2, b2);
            v = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        return v;
    };
    auto x0_raw = [&](const T* __restrict__ xg, int co) -> float4 {   // the cell input at this lane's frames, channel co, from HBM (zeros beyond the row)
        return cell_load_frames(xg + static_cast<size_t>(co) * a.ld, a.ld, q);
    };
    auto tile_own = [&](int co) -> float4 {         // this lane's chunk of the tensor the tile holds
        const cell_f4 v = *reinterpret_cast<const cell_f4*>(tile + co * rl + 4 * col);
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    // only the wave tile that holds the row's end has frames to zero: a wave-uniform branch saves the other waves 4 selects per output
    // channel and node
    const bool tail_wave = (ti + 1) * 256 > a.frames;
    auto mask_tail = [&](float (&o)[4]) {
        if (tail_wave) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (!in_row || t0 + r >= a.frames) o[r] = 0.f;
        }
    };

    // between the phases of a node: the nt waves of a group row read each other's halo chunks, so they meet at a workgroup barrier;
    // a one-wave row (NTB == 1) is wave-private -- LDS operations of one wave execute in order -- and its waves run free
    auto phase_sync = [&]() {
        if constexpr (NTB > 1 || OS > 1) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    cell_f2 acc[CH / 2][4];                         // the convolution's accumulators: this wave's output channels (2p, 2p + 1) at frame r
    float out[CH][4];                               // a node's finished output (after the epilogue): channel c_base + c at frame r
    float keep1[KEEP1 ? CH : 1][4];

    // ---- the cell input: own chunks of all CG channels (all loads in flight together), normalised once, into the tile ------------
    {
        const T* __restrict__ xg = A.x0 + row0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {              // (OS > 1: each wave of the pair brings in its own channels)
            const float4 v = x0_raw(xg, c_base + c);
            out[c][0] = v.x; out[c][1] = v.y; out[c][2] = v.z; out[c][3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ci = c_base + c;
            const float4 v = normalise(make_float4(out[c][0], out[c][1], out[c][2], out[c][3]), ci, A.ln_gamma, A.ln_beta);
            float o[4] = {v.x, v.y, v.z, v.w};
            if (has_ln) mask_tail(o);               // (beta, not 0, beyond the row otherwise)
            if (q < a.rowq) *reinterpret_cast<cell_f4*>(tile + ci * rl + 4 * col) = cell_f4{o[0], o[1], o[2], o[3]};   // (a tile row ends with the row's last chunk + pads)
        }
    }
    stamp();
    __syncthreads();                                // (also orders the zero pads, which other waves wrote)
    stamp();

    using I5 = std::integral_constant<int, 5>; using I7 = std::integral_constant<int, 7>;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto conv = [&](auto kc, auto dc, const float* w, const float* bias) {
        constexpr int K = decltype(kc)::value, D = decltype(dc)::value;
        conv_from_tile<CG, CH, K, D>(acc, cell_const(w) + static_cast<size_t>(g) * (CG * CG * K) + c_base, cell_const(bias) + g * CG + c_base, tile, rl, col);
    };
#define NBASR_KD_SWITCH(kd, W_, B_)                                                                    \
    switch (kd) {                                                                                      \
        case 0: conv(I5{}, I1{}, W_, B_); break; case 1: conv(I5{}, I2{}, W_, B_); break;              \
        case 2: conv(I7{}, I1{}, W_, B_); break; default: conv(I7{}, I2{}, W_, B_); break;             \
    }
    auto tile_write = [&]() {                       // the node output in `out` becomes the next node's input
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (q < a.rowq) *reinterpret_cast<cell_f4*>(tile + (c_base + c) * rl + 4 * col) = cell_f4{out[c][0], out[c][1], out[c][2], out[c][3]};
    };

    // ---- node 0: x1 = op0(x0n) + s00 x0n (x0n read back from the tile) -------------------------------------------------------------
    NBASR_KD_SWITCH(a.kd0, A.w0, A.b0)
    stamp();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[c >> 1][r][c & 1]);
        if (a.skips & 1) { const float4 v = tile_own(c_base + c); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
        mask_tail(o);
        cell_round<T>(o);
#pragma unroll
        for (int r = 0; r < 4; ++r) { out[c][r] = o[r]; if (KEEP1) keep1[KEEP1 ? c : 0][r] = o[r]; }
    }
    phase_sync();                                   // every read of x0n is done
    tile_write();
    phase_sync();
    stamp();

    // ---- node 1: x2 = op1(x1) + s10 x0n + s11 x1 --------------------------------------------------------------------------------------
    {
        const float* __restrict__ w1 = NBASR_CELL_ARG(T, w1);
        const float* __restrict__ b1 = NBASR_CELL_ARG(T, b1);
        NBASR_KD_SWITCH(a.kd1, w1, b1)
    }
    stamp();
    {
        // (the skip input x0n comes back from HBM: its four-channel groups are requested together behind ONE wave-uniform branch each)
        const bool s10 = a.skips & 2;
        const T* __restrict__ xg = nullptr; const float* __restrict__ gamma = nullptr; const float* __restrict__ beta = nullptr;
        if (s10) { xg = NBASR_CELL_ARG(T, x0) + row0; if (has_ln) { gamma = NBASR_CELL_ARG(T, ln_gamma); beta = NBASR_CELL_ARG(T, ln_beta); } }
#pragma unroll
        for (int c0 = 0; c0 < CH; c0 += 4) {
            float4 u[4];
            if (s10) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c0 + c < CH) u[c] = x0_raw(xg, c_base + c0 + c);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cl = c0 + c;                  // channel within this wave's CH; co within the group
                if (cl >= CH) break;
                const int co = c_base + cl;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[cl >> 1][r][cl & 1]);
                if (s10) { const float4 v = normalise(u[c], co, gamma, beta); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
                if (a.skips & 4) { const float4 v = tile_own(co); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
                mask_tail(o);
                cell_round<T>(o);
#pragma unroll
                for (int r = 0; r < 4; ++r) out[cl][r] = o[r];
            }
        }
    }
    phase_sync();
    tile_write();
    phase_sync();
    stamp();

    // ---- node 2: x3 = op2(x2) + s20 x0n + s21 x1 + s22 x2 -> HBM ------------------------------------------------------------------------
    {
        const float* __restrict__ w2 = NBASR_CELL_ARG(T, w2);
        const float* __restrict__ b2 = NBASR_CELL_ARG(T, b2);
        NBASR_KD_SWITCH(a.kd2, w2, b2)
    }
#undef NBASR_KD_SWITCH
    stamp();
    {
        const int store_len = g_ok ? a.ld : 0;      // a surplus wave's stores are dropped by the bounds check
        const bool s20 = a.skips & 8;
        T* __restrict__ yg = NBASR_CELL_ARG(T, y) + row0;
        const T* __restrict__ xg = nullptr; const float* __restrict__ gamma = nullptr; const float* __restrict__ beta = nullptr;
        if (s20) { xg = NBASR_CELL_ARG(T, x0) + row0; if (has_ln) { gamma = NBASR_CELL_ARG(T, ln_gamma); beta = NBASR_CELL_ARG(T, ln_beta); } }
#pragma unroll
        for (int c0 = 0; c0 < CH; c0 += 4) {
            float4 u[4];
            if (s20) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c0 + c < CH) u[c] = x0_raw(xg, c_base + c0 + c);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int cl = c0 + c;
                if (cl >= CH) break;
                const int co = c_base + cl;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = relu_clamp(acc[cl >> 1][r][cl & 1]);
                if (s20) { const float4 v = normalise(u[c], co, gamma, beta); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
                if (KEEP1) {
                    if (a.skips & 16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] += keep1[KEEP1 ? cl : 0][r];
                    }
                }
                if (a.skips & 32) { const float4 v = tile_own(co); o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w; }
                mask_tail(o);
                cell_store_frames(yg + static_cast<size_t>(co) * a.ld, store_len, q, o);        // (no predicate: lanes beyond the row store nothing)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[cl][r] = o[r];    // the final values, for the statistics
            }
        }
    }
    stamp();
    float* __restrict__ part = NBASR_CELL_ARG(T, part);
    if (part == nullptr) return;                    // (workgroup-uniform)

    // ---- LayerNorm statistics of x3: per lane (mean, M2) over this group's CG channels, exact two-pass in registers; the wave of the
    // quad's first group merges the four groups (same arithmetic and order as the node kernel's statistics epilogue) -------------------
    // (frame pairs as packed arithmetic: per frame the same operations in the same order as the scalar form)
    float pm[4], p2[4];
    float all_[OS > 1 ? CG : 1][4];                 // OS > 1: x3 of the whole group at this lane's frames (first wave of the pair)
    if constexpr (OS > 1) {
        // the other wave's channels come through the tile: every read of x2 done -> each wave writes its x3 -> the first wave reads all
        __syncthreads();
        tile_write();
        __syncthreads();
#pragma unroll
        for (int co = 0; co < CG; ++co) {           // (both waves: the second one's copy is never stored)
            const float4 v = tile_own(co);
            all_[co][0] = v.x; all_[co][1] = v.y; all_[co][2] = v.z; all_[co][3] = v.w;
        }
    }
    auto x3 = [&](int co, int r) -> float { if constexpr (OS > 1) return all_[co][r]; else return out[co][r]; };
    {
        cell_f2 s01{0.f, 0.f}, s23 = s01;
#pragma unroll
        for (int co = 0; co < CG; ++co) { s01 += cell_f2{x3(co, 0), x3(co, 1)}; s23 += cell_f2{x3(co, 2), x3(co, 3)}; }
        const cell_f2 m01 = s01 * (1.0f / CG), m23 = s23 * (1.0f / CG);
        cell_f2 q01{0.f, 0.f}, q23 = q01;
#pragma unroll
        for (int co = 0; co < CG; ++co) {
            const cell_f2 d01 = cell_f2{x3(co, 0), x3(co, 1)} - m01, d23 = cell_f2{x3(co, 2), x3(co, 3)} - m23;
            q01 = __builtin_elementwise_fma(d01, d01, q01); q23 = __builtin_elementwise_fma(d23, d23, q23);
        }
        pm[0] = m01.x; pm[1] = m01.y; pm[2] = m23.x; pm[3] = m23.y;
        p2[0] = q01.x; p2[1] = q01.y; p2[2] = q23.x; p2[3] = q23.y;
    }
    if constexpr (GPW == 1) {                       // one group per workgroup: the lane's own (mean, M2) IS the partial -- no exchange, no barrier
        if (in_row && oh == 0) {
            float* prow = part + (static_cast<size_t>(blockIdx.x) * gridDim.y + b) * 2 * a.ld + t0;
            *reinterpret_cast<float4*>(prow) = make_float4(pm[0], pm[1], pm[2], pm[3]);
            *reinterpret_cast<float4*>(prow + a.ld) = make_float4(p2[0], p2[1], p2[2], p2[3]);
        }
        stamp();
        return;
    }
    __syncthreads();                                // every read of x2 is done: the tiles become the exchange buffer [GPW][nt][8][64]
    float* const sp = cell_tiles;
    if (oh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sp[((gi * nt + ti) * 8 + r) * 64 + lane] = pm[r];
            sp[((gi * nt + ti) * 8 + 4 + r) * 64 + lane] = p2[r];
        }
    }
    __syncthreads();
    if (gi == 0 && oh == 0 && in_row) {
        const int nw = min(GPW, a.groups - static_cast<int>(blockIdx.x) * GPW);  // groups (waves) that hold real data
        float om[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mean = 0.f;
            for (int k = 0; k < nw; ++k) mean += sp[((k * nt + ti) * 8 + r) * 64 + lane];
            mean /= static_cast<float>(nw);
            float m2 = 0.f;
            for (int k = 0; k < nw; ++k) { const float d = sp[((k * nt + ti) * 8 + r) * 64 + lane] - mean; m2 += sp[((k * nt + ti) * 8 + 4 + r) * 64 + lane] + CG * d * d; }
            om[r] = mean; o2[r] = m2;
        }
        float* prow = part + (static_cast<size_t>(blockIdx.x) * gridDim.y + b) * 2 * a.ld + t0;
        *reinterpret_cast<float4*>(prow) = make_float4(om[0], om[1], om[2], om[3]);
        *reinterpret_cast<float4*>(prow + a.ld) = make_float4(o2[0], o2[1], o2[2], o2[3]);
    }
    stamp();
}

// LDS of a workgroup: gpw tiles of cg rows of (chunks + pads) x 16 bytes (fits / groups-per-workgroup decisions: the widest pads)
static size_t cell_lds_bytes(int cg, int chunks, int gpw, int pads = CELL_PADL + CELL_PADR) { return static_cast<size_t>(gpw) * cg * (chunks + pads) * 16; }
// groups per workgroup: 4 while at least two such workgroups fit a CU's 160 KiB and the workgroup its 16 waves, else 2
// groups per workgroup.  Rows of ONE wave (<= 256 frames): 4 -- the waves of a workgroup never meet at a barrier and the partials are
// the node kernel's group quads, bit for bit.  Longer rows (round 4): ONE -- a workgroup is the nt waves of one group row.  Phase stamps
// showed a workgroup of 2 groups x 4 waves outside its convolution loops for half of its life, much of it at the barriers of its
// node boundaries (the slowest of 8 waves) and in the statistics exchange (two more barriers, half of the waves idle); with one group
// per workgroup a barrier joins 4 waves, twice as many workgroups interleave their phases on a CU, and the lane's own (mean, M2) IS
// the partial: 95 / 162 / 140 -> 90 / 142 / 132 us per cell in blocks 0-2 at 64 x 1000.  Price: per-group partials (100 instead of
// 50 rows per utterance for nbasr_grouped_stats_finalize: 0.17 -> 0.21 ms per forward); end to end +0.9 % (same-box A/B, twice).
static int cell_gpw(int cg, int nt)
{
    if (nt >= 2) return 1;
    // (one-wave rows as workgroups of ONE wave -- 13 instead of 12 rows in a CU's LDS, 1.92 instead of 2.08 rounds at 64 x 250 -- measured
    // slower, 112 -> 116 us per cell, with per-group partials on top: not taken)
    return 2 * cell_lds_bytes(cg, nt * 64, 4) <= 160 * 1024 ? 4 : 2;
}

template <typename T, int CG, bool KEEP1, int NTB, int GPW, int OS = 1>
static int launch_cell_kernel(const CellArgs<T>& p, hipStream_t stream)
{
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(grouped_cell_kernel<T, CG, KEEP1, NTB, GPW, OS>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) {
        set_error("nbasr_grouped_cell_fused: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
        return static_cast<int>(attr);
    }
    const CellDims& a = p.a;
    // (the tiles double as the statistics exchange buffer [GPW][nt][8][64] floats of a multi-group workgroup)
    const size_t lds = std::max(cell_lds_bytes(CG, a.rowq, GPW, a.padl + a.padr), GPW > 1 ? static_cast<size_t>(GPW) * a.nt * 8 * 64 * 4 : 0);
    hipLaunchKernelGGL((grouped_cell_kernel<T, CG, KEEP1, NTB, GPW, OS>), dim3((a.groups + GPW - 1) / GPW, a.batch), dim3(GPW * a.nt * OS * 64), lds, stream, p);
    return launch_status("nbasr_grouped_cell_fused");
}

// Two waves per (group, row tile) -- the output-channel split -- while the launch would otherwise leave SIMDs without a wave (1024 on
// the chip): block 3 (12 channels per group, one-wave rows) up to 10 utterances.  Measured (tools/ubench/cell_os.py, 1200 channels x
// 250 frames): 43.6 -> 38.1 us per launch at 8 utterances, 40.5 -> 34.6 at 4, no difference from 16 on; phase stamps at 8 utterances
// (tools/gpu/cell_stamps.py): a convolution loop 5.0 -> 3.8 us, the node boundaries 1.0-1.3 -> 1.6 us (a barrier of 8 waves).  A whole
// forward at 8 utterances does not notice (5 409 vs 5 354 utterances/s, within the run-to-run spread): the six block-3 cells are
// 0.2 of its 1.47 ms.  NBASR_CELL_OS = 0 / 1 forces the split off / on wherever an instantiation exists (fp32, 12 or 8 channels per
// group), NBASR_CELL_OS_WAVES moves the threshold.
// (read at every launch: the tests flip it between calls)
static bool cell_os_split(const CellDims& a)
{
    const char* e = getenv("NBASR_CELL_OS");
    if (e && *e) return atoi(e) != 0;
    const char* w = getenv("NBASR_CELL_OS_WAVES");
    return static_cast<long long>(a.batch) * a.groups * a.nt <= (w && *w ? atoi(w) : 1024);
}

template <typename T, int CG, bool KEEP1>
static int launch_cell_nt(const CellArgs<T>& p, hipStream_t stream)
{
    const CellDims& a = p.a;
    if constexpr (std::is_same<T, float>::value && (CG == 12 || CG == 8)) {
        if (cell_os_split(a)) {
            if (cell_gpw(CG, a.nt) == 1 && a.nt == 2) return launch_cell_kernel<T, CG, KEEP1, 2, 1, 2>(p, stream);
            if (cell_gpw(CG, a.nt) == 1 && a.nt <= 4) return launch_cell_kernel<T, CG, KEEP1, 4, 1, 2>(p, stream);
            if (cell_gpw(CG, a.nt) == 4 && a.nt == 1) return launch_cell_kernel<T, CG, KEEP1, 1, 4, 2>(p, stream);
        }
    }
    if (cell_gpw(CG, a.nt) == 1) {
        if (a.nt == 2) return launch_cell_kernel<T, CG, KEEP1, 2, 1>(p, stream);
        if (a.nt <= 4) return launch_cell_kernel<T, CG, KEEP1, 4, 1>(p, stream);
        return launch_cell_kernel<T, CG, KEEP1, 8, 1>(p, stream);
    }
    if (cell_gpw(CG, a.nt) == 4) {
        if (a.nt == 1) return launch_cell_kernel<T, CG, KEEP1, 1, 4>(p, stream);
        if (a.nt == 2) return launch_cell_kernel<T, CG, KEEP1, 2, 4>(p, stream);
        return launch_cell_kernel<T, CG, KEEP1, 4, 4>(p, stream);
    }
    if (a.nt == 1) return launch_cell_kernel<T, CG, KEEP1, 1, 2>(p, stream);
    if (a.nt == 2) return launch_cell_kernel<T, CG, KEEP1, 2, 2>(p, stream);
    if (a.nt <= 4) return launch_cell_kernel<T, CG, KEEP1, 4, 2>(p, stream);
    return launch_cell_kernel<T, CG, KEEP1, 8, 2>(p, stream);
}

template <typename T, int CG>
static int launch_cell(const CellArgs<T>& p, hipStream_t stream)
{
    return (p.a.skips & 16) ? launch_cell_nt<T, CG, true>(p, stream) : launch_cell_nt<T, CG, false>(p, stream);
}

static int kd_code(int kernel, int dilation)
{
    if (kernel == 5 && dilation == 1) return 0;
    if (kernel == 5 && dilation == 2) return 1;
    if (kernel == 7 && dilation == 1) return 2;
    if (kernel == 7 && dilation == 2) return 3;
    return -1;
}

template <typename T>
static int cell_fused_impl(const T* x0, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2, const float* b2,
                           T* y, const CellDims& a, const nbasr_deferred_ln* ln, float* stats_ws, hipStream_t s)
{
    const LnRef l = ln_ref(ln, true);
    const CellArgs<T> p{x0, y, w0, w1, w2, b0, b1, b2, l.stats, l.gamma, l.beta, stats_ws, a};
    switch (a.channels / a.groups) {
        case 6:  return launch_cell<T, 6>(p, s);
        case 8:  return launch_cell<T, 8>(p, s);
        case 10: return launch_cell<T, 10>(p, s);
        default: return launch_cell<T, 12>(p, s);
    }
}

}  // namespace nbasr

using namespace nbasr;

extern "C" int nbasr_grouped_cell_fits(int channels, int frames_ld, int groups)
{
    if (channels <= 0 || groups <= 0 || channels % groups || frames_ld <= 0 || frames_ld % 4) return 0;
    const int cg = channels / groups;
    if (cg != 6 && cg != 8 && cg != 10 && cg != 12) return 0;
    const int nt = (frames_ld / 4 + 63) / 64;
    if (nt > 8) return 0;                                              // 2 groups x 8 waves = the 1024 threads of a workgroup
    const int gpw = cell_gpw(cg, nt);
    return cell_lds_bytes(cg, frames_ld / 4, gpw) <= 160 * 1024 ? gpw : 0;       // the groups per statistics partial (nbasr_grouped_stats_finalize)
}

extern "C" int nbasr_grouped_cell_fused(const void* x0, const float* w0, const float* b0, int k0, int d0,
                                        const float* w1, const float* b1, int k1, int d1,
                                        const float* w2, const float* b2, int k2, int d2, int skip_mask, void* y,
                                        int batch, int channels, int frames, int ld, int groups,
                                        const nbasr_deferred_ln* ln, float* stats_ws, int dtype, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(dtype == NBASR_F32 || dtype == NBASR_BF16, NBASR_EINVAL, "nbasr_grouped_cell_fused: dtype %d is neither NBASR_F32 nor NBASR_BF16", dtype);
    NBASR_REQUIRE(batch >= 0 && channels > 0 && frames >= 0 && groups > 0 && channels % groups == 0, NBASR_EINVAL,
                  "nbasr_grouped_cell_fused: bad sizes batch=%d channels=%d frames=%d groups=%d", batch, channels, frames, groups);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(x0 && w0 && b0 && w1 && b1 && w2 && b2 && y, NBASR_ENULL, "nbasr_grouped_cell_fused: NULL pointer");
    const int pitch = dtype == NBASR_BF16 ? 8 : 4;
    NBASR_REQUIRE(ld >= frames && ld % pitch == 0 && aligned16(x0) && aligned16(y) && aligned16(stats_ws), NBASR_EALIGN,
                  "nbasr_grouped_cell_fused: ld=%d must be >= frames=%d and a multiple of %d; x0, y, stats_ws 16-byte aligned", ld, frames, pitch);
    NBASR_REQUIRE(batch <= 65535 && skip_mask >= 0 && skip_mask < 64, NBASR_EINVAL, "nbasr_grouped_cell_fused: bad batch / skip mask");
    NBASR_REQUIRE(!ln || (ln->stats && ln->gamma && ln->beta && aligned16(ln->stats)), NBASR_ENULL,
                  "nbasr_grouped_cell_fused: deferred LayerNorm needs stats (16-byte aligned), gamma and beta");
    NBASR_REQUIRE(nbasr_grouped_cell_fits(channels, ld, groups), NBASR_EINVAL,
                  "nbasr_grouped_cell_fused: a row of %d frames x %d channels per group does not fit one workgroup "
                  "(<= 2048 frames, channels/groups in {6, 8, 10, 12}, the group tiles <= 160 KiB of LDS); use the per-node launches", ld, channels / groups);
    CellDims a{};
    a.channels = channels; a.frames = frames; a.ld = ld; a.groups = groups; a.batch = batch;
    a.kd0 = kd_code(k0, d0); a.kd1 = kd_code(k1, d1); a.kd2 = kd_code(k2, d2);
    NBASR_REQUIRE(a.kd0 >= 0 && a.kd1 >= 0 && a.kd2 >= 0, NBASR_EINVAL,
                  "nbasr_grouped_cell_fused: node ops must be conv5 / conv5d2 / conv7 / conv7d2 (got k=%d,%d,%d d=%d,%d,%d)", k0, k1, k2, d0, d1, d2);
    a.skips = skip_mask;
    a.nt = (ld / 4 + 63) / 64;
    a.padl = std::max(cell_ql(a.kd0), std::max(cell_ql(a.kd1), cell_ql(a.kd2)));
    a.padr = std::max(cell_qr(a.kd0), std::max(cell_qr(a.kd1), cell_qr(a.kd2)));
    a.rowq = ld / 4;            // (same-box A/B against rows of 64 nt + 2 + 2 chunks: block 2 140 -> 132.5 us per cell at 64 x 1000, the others unchanged)
#if NBASR_CELL_STAMPS
    { const char* e = getenv("NBASR_CELL_STAMPS"); a.stamps = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr; }
#endif
    hipStream_t s = as_stream(stream);
    if (dtype == NBASR_F32)
        return cell_fused_impl<float>(static_cast<const float*>(x0), w0, b0, w1, b1, w2, b2, static_cast<float*>(y), a, ln, stats_ws, s);
    return cell_fused_impl<bf16_t>(static_cast<const bf16_t*>(x0), w0, b0, w1, b1, w2, b2, static_cast<bf16_t*>(y), a, ln, stats_ws, s);
}
