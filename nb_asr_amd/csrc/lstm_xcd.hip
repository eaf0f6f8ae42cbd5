// The LSTM recurrence (reference model.py:100, 118-121: nn.LSTM(1200 -> 500), zero initial state, gates i, f, g, o) as ONE resident
// launch whose per-frame exchange never leaves an XCD, on the 16-bit matrix cores (round 6).
//
// Why.  The recurrence is `frames` dependent steps of a tiny GEMM (gates(2000) x utterances(16) x hidden(500)).  Rounds 2-5 ran it
// as one launch per frame (4.4-6.5 us per frame) or as a resident grid of 252 workgroups over the whole chip that exchange h_t in
// tagged write-through granules (3.85 us per frame: ~0.85 of fp32 MFMA, the rest the fabric round trip of an `sc1` store that drops
// its line from L2 + an `sc1` load that has to fetch it from the memory side).  At 8 utterances per GPU -- one rank's share of the
// north star's 8-GPU run -- the chain (0.8-1.0 ms) was as long as the whole encoder, and at 64 its 250 launches took 0.64 ms from the
// encoder they were supposed to hide behind.  Here:
//   * a tile of 16 utterances is owned by the (<= 32) workgroups of ONE XCD: workgroup = 16 hidden units x 4 gates (4 MFMA row
//     tiles), all of K = hidden <= 512.  Producers and consumers of a tile share one L2, so h_t is published with PLAIN stores (the
//     line stays in that L2) and polled with `sc1` loads (they bypass the CU's L1 and are served by L2): one L2 round trip per frame
//     instead of two fabric trips (/opt/skills/guides/MI355X_MICROARCH.md, "stores of each flavour", handoff rows);
//   * placement is not assumed, it is READ: a workgroup takes its XCD from HW_REG_XCC_ID and its slice from a per-XCD arrival ticket;
//     the first arrival of an XCD claims whole tiles from a global ticket once its XCD has all its slices, publishes the claim, and
//     its peers follow.  Whatever the dispatcher does, the workgroups that exchange a tile's h ARE on one XCD; an XCD that never
//     collects its slices claims nothing; surplus workgroups leave at once.  HIP's round-robin dealing of workgroups over the XCDs
//     (observed, not promised) only decides how many XCDs take part.  The last workgroup to leave checks that every tile was
//     computed and raises the status word otherwise;
//   * the recurrent product runs on v_mfma_f32_16x16x32_f16 with fp32-accurate operands: w_hh (scaled by one power of two) and h
//     (|h| <= 1) are two fp16 terms each, value = hi + lo * 2^-11, and a product is hi*hi (first accumulator) + (hi*lo' + lo'*hi)
//     (second accumulator, folded in with its 2^-11 at the end): 3 MFMAs per 32 k, 1/5 of the fp32 MFMA time.  The dropped lo*lo
//     term is <= 2^-24 of the product; the representation error of an operand is 2^-23 relative (as in the dense convs and the
//     input projection, gemm_conv_split.hip) -- an fp32 evaluation in another summation order, not a narrower one;
//   * h_t travels as it is consumed: dword = (fp16 hi << 16) | fp16 lo', four units per 16-byte granule, bit 30 of every dword
//     (= bit 14 of hi: free, |hi| <= 1) carries the tag of the step as in lstm_seq_kernel: no flag, no drain, no second location.
//     A non-finite h travels as lo' = NaN, which every product it meets turns into NaN: a diverged utterance stays visible;
//   * every wait is bounded (1 s of the 100 MHz clock); a timeout raises the status word (nbasr_lstm_seq_status, or the
//     executor's asynchronous read-back) and fills the rest of that slice's h rows with NaN.
// Per frame and workgroup: poll 4 granule loads per lane (2 k-steps x 2) -> 16 v_perm + 24 MFMAs per wave -> partial tiles through
// LDS, one barrier -> waves 0-3: gates, cell, h for 4 units x 16 utterances each -> granule store.
#include "common.h"

#include <mutex>

namespace nbasr {

typedef float xfloat4 __attribute__((ext_vector_type(4)));
typedef _Float16 xhalf8 __attribute__((ext_vector_type(8)));
typedef unsigned xuint4 __attribute__((ext_vector_type(4)));

constexpr int LX_WAVES = 8;                      // K = 512 split eight ways: a wave owns 2 k-steps of 32
constexpr int LX_UNITS = 16;                     // hidden units per workgroup = 4 row tiles of (4 units x 4 gates)
constexpr int LX_MT = LX_UNITS / 4;
constexpr int LX_KSTEPS = 16;                    // hidden <= 512
constexpr int LX_IMAGE_BYTES = LX_KSTEPS * 2 * 64 * 16;    // one exchange image of one tile: [k-step][granule][lane] x 16 B = 32 KiB
constexpr int LX_MAX_XCD = 16;
constexpr int LX_MAX_TILES = 256;                // batch <= 4096
// header of the workspace (32-bit words); [0] is the status word nbasr_lstm_seq_status reads
constexpr int LX_W_STATUS = 0, LX_W_NEXT_TILE = 1, LX_W_TILES_DONE = 2, LX_W_EXITS = 3;
constexpr int LX_W_ARRIVALS = 16;                // + xcd * 16 (a counter per 64-byte line)
constexpr int LX_W_CLAIMS = LX_W_ARRIVALS + LX_MAX_XCD * 16;          // + xcd * (LX_MAX_TILES + 16) + seq
constexpr int LX_CLAIM_STRIDE = LX_MAX_TILES + 16;
constexpr int LX_HEADER_WORDS = LX_W_CLAIMS + LX_MAX_XCD * LX_CLAIM_STRIDE;
constexpr unsigned LX_DONE = 0xffffffffu;
constexpr unsigned long long LX_TIMEOUT_TICKS = 100000000ull;         // 1 s of the constant 100 MHz clock
constexpr int LX_PACK_HEADER_BYTES = 256;        // packed w_hh: [0] 2^-e, [1] 2^-(e+11) (floats), [2] max |w| bits (pack-time scratch)

// Diagnostic build (-DNBASR_LX_STAMPS=1, tools/ubench/lstm_xcd_stamps.py): wave 0 of every slice of tile 0 stamps the shader clock at its
// phase boundaries of every frame into a region behind the exchange images.  Never in the shipped library.
#ifndef NBASR_LX_STAMPS
#define NBASR_LX_STAMPS 0
#endif
constexpr int LX_STAMP_POINTS = 8, LX_STAMP_FRAMES = 256;
constexpr size_t LX_STAMP_BYTES = NBASR_LX_STAMPS ? static_cast<size_t>(32) * LX_STAMP_FRAMES * LX_STAMP_POINTS * 8 : 0;
#if NBASR_LX_STAMPS
#define LX_STAMP(i) do { if (stamping) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LX_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float lx_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ unsigned lx_tag(int t) { return (static_cast<unsigned>((t >> 1) + 1) & 1u) << 30; }

__device__ __forceinline__ unsigned lx_load_sc1(const unsigned* p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void lx_store_sc1(unsigned* p, unsigned v)
{
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

// ---- packing -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lx_absmax_kernel(const float* __restrict__ w, size_t n, unsigned* header)
{
    float m = 0.f;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) m = fmaxf(m, finite_abs(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(header + 2, __float_as_uint(m));          // non-negative floats order like their bits
}

// packed[256 B header][slice][wave][kk (2)][mt (4)][part (hi, lo')][lane] x 16 B: the A fragment of v_mfma_f32_16x16x32_f16 for row tile mt of
// the slice (row i = lane & 15: unit slice*16 + mt*4 + (i >> 2), gate i & 3) and k-step 2*wave + kk (k = kstep*32 + (lane >> 4)*8 + j)
__global__ __launch_bounds__(256) void lx_pack_kernel(const float* __restrict__ w_hh, unsigned char* __restrict__ packed, int hidden, int slices)
{
    float* const hdr = reinterpret_cast<float*>(packed);
    const float amax = __uint_as_float(reinterpret_cast<const unsigned*>(packed)[2]);
    int e = 0;
    if (amax > 0.f) { int ex; (void)frexpf(amax, &ex); e = 14 - ex; }                // amax * 2^e in [2^13, 2^14)
    const float scale = ldexpf(1.0f, e);
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = ldexpf(1.0f, -e); hdr[1] = ldexpf(1.0f, -e - 11); }
    xuint4* const out = reinterpret_cast<xuint4*>(packed + LX_PACK_HEADER_BYTES);
    const size_t total = static_cast<size_t>(slices) * LX_WAVES * 2 * LX_MT * 2 * 64;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += static_cast<size_t>(gridDim.x) * blockDim.x) {
        size_t r = idx;
        const int lane = r % 64; r /= 64;
        const int part = r % 2; r /= 2;
        const int mt = r % LX_MT; r /= LX_MT;
        const int kk = r % 2; r /= 2;
        const int wave = r % LX_WAVES; r /= LX_WAVES;
        const int slice = static_cast<int>(r);
        const int i = lane & 15, kq = lane >> 4;
        const int unit = slice * LX_UNITS + mt * 4 + (i >> 2), gate = i & 3;
        const int k0 = (2 * wave + kk) * 32 + kq * 8;
        unsigned short h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (unit < hidden && k0 + j < hidden) v = w_hh[(static_cast<size_t>(gate) * hidden + unit) * hidden + k0 + j] * scale;
            const _Float16 hi = static_cast<_Float16>(v);
            const _Float16 lo = static_cast<_Float16>((v - static_cast<float>(hi)) * 2048.0f);
            h[j] = __builtin_bit_cast(unsigned short, part == 0 ? hi : lo);
        }
        out[idx] = xuint4{static_cast<unsigned>(h[0]) | (static_cast<unsigned>(h[1]) << 16), static_cast<unsigned>(h[2]) | (static_cast<unsigned>(h[3]) << 16),
                          static_cast<unsigned>(h[4]) | (static_cast<unsigned>(h[5]) << 16), static_cast<unsigned>(h[6]) | (static_cast<unsigned>(h[7]) << 16)};
    }
}

// ---- the recurrence ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * LX_WAVES) void lstm_xcd_kernel(
    const float* __restrict__ gates_in,     // (frames, batch, 4*hidden)
    const unsigned char* __restrict__ wp,   // packed w_hh (lx_pack_kernel)
    float* __restrict__ cell, float* __restrict__ h_out, unsigned* ws,
    int batch, int frames, int hidden, int slices, int n_tiles, int total_wgs, int flags)
{
    __shared__ xfloat4 red[2][LX_WAVES][LX_MT][64];       // [step parity][wave][row tile][lane]: a lane's 4 accumulator registers (= gates)
    __shared__ unsigned s_role[4];                         // [0] xcd, [1] rank, [2] claim of the current round, [3] stop

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    unsigned char* const images = reinterpret_cast<unsigned char*>(ws + LX_HEADER_WORDS);

    // ---- who am I: XCD from the hardware register, slice from the XCD's arrival ticket ----
    if (threadIdx.x == 0) {
        const unsigned xcd = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xfu;        // HW_REG_XCC_ID
        unsigned rank = atomicAdd(ws + LX_W_ARRIVALS + xcd * 16, 1u);
        // NBASR_LSTM_SEQ_INJECT_FAULT (tests): the second arrival of every XCD takes its ticket and leaves -- what compute units taken
        // away by another process look like to its peers: the XCD never completes, claims nothing, and the last leaver finds tiles missing
        if ((flags & NBASR_LSTM_SEQ_INJECT_FAULT) && rank == 1) rank = 0x7fffffffu;
        s_role[0] = xcd; s_role[1] = rank; s_role[3] = 0;
    }
    __syncthreads();
    const unsigned xcd = s_role[0], rank = s_role[1];
    unsigned* const claims = ws + LX_W_CLAIMS + xcd * LX_CLAIM_STRIDE;
    const bool member = rank < static_cast<unsigned>(slices);
    const int slice = static_cast<int>(rank);

    xuint4 wfrag[2][LX_MT][2];                             // [kk][row tile][hi, lo']: 64 registers, loaded with the first claimed tile
    bool have_w = false;
    float inv = 1.f;

    for (int seq = 0; member; ++seq) {
        // ---- which tile: the XCD's first arrival claims, the others follow its published claim ----
        if (threadIdx.x == 0) {
            unsigned claim = LX_DONE;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            if (rank == 0) {
                bool complete = seq > 0;
                while (!complete) {                        // all slices of this XCD present?  (or nothing left to claim / out of time)
                    if (lx_load_sc1(ws + LX_W_ARRIVALS + xcd * 16) >= static_cast<unsigned>(slices)) { complete = true; break; }
                    if (lx_load_sc1(ws + LX_W_NEXT_TILE) >= static_cast<unsigned>(n_tiles)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > LX_TIMEOUT_TICKS) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                if (complete) {
                    const unsigned tile = atomicAdd(ws + LX_W_NEXT_TILE, 1u);
                    if (tile < static_cast<unsigned>(n_tiles)) claim = tile + 1;
                }
                lx_store_sc1(claims + seq, claim);
            } else {
                for (;;) {
                    claim = lx_load_sc1(claims + seq);
                    if (claim != 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 2 * LX_TIMEOUT_TICKS) { claim = LX_DONE; break; }    // (the first arrival gives up after 1 s)
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            s_role[2] = claim;
        }
        __syncthreads();                                   // (also: the partials of the previous tile's last step have been read)
        const unsigned claim = s_role[2];
        if (claim == LX_DONE) break;
        const int tile = static_cast<int>(claim) - 1;
        const int b0 = tile * 16;

        if (!have_w) {
            have_w = true;
            const float* hdr = reinterpret_cast<const float*>(wp);
            inv = hdr[0];
            const xuint4* src = reinterpret_cast<const xuint4*>(wp + LX_PACK_HEADER_BYTES) + (static_cast<size_t>(slice) * LX_WAVES + wave) * (2 * LX_MT * 2 * 64) + lane;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int mt = 0; mt < LX_MT; ++mt)
#pragma unroll
                    for (int p = 0; p < 2; ++p) wfrag[kk][mt][p] = src[((kk * LX_MT + mt) * 2 + p) * 64];
        }

        unsigned char* const image = images + static_cast<size_t>(tile) * 2 * LX_IMAGE_BYTES;
        // the granules this lane consumes: utterance n16 of the tile, k-steps 2*wave + kk, units kstep*32 + kq*8 + 4*g .. +3
        bool live[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int g = 0; g < 2; ++g) live[kk][g] = (b0 + n16) < batch && ((2 * wave + kk) * 32 + kq * 8 + 4 * g) < hidden;

        // epilogue role of waves 0..3: wave = row tile, lane = (unit kq of the tile, utterance n16); 4 accumulator registers = 4 gates
        const int eu = slice * LX_UNITS + (wave & 3) * 4 + kq, eb = b0 + n16;
        const bool e_ok = wave < LX_MT && eu < hidden && eb < batch;
        const size_t gate_off = static_cast<size_t>(min(eb, batch - 1)) * (4 * hidden) + min(eu, hidden - 1);
        const bool q_ok = wave < LX_MT && kq == 0 && (slice * LX_UNITS + wave * 4) < hidden && eb < batch;     // lanes 0..15 publish the tile's 4 units
        // where this wave's granule goes: k-step slice/2, k-octet 2*(slice&1) + (wave>>1), granule wave&1
        const int pub_off = ((((slice >> 1) * 2 + (wave & 1)) * 4 + 2 * (slice & 1) + ((wave & 3) >> 1)) * 16 + n16) * 16;
        float c_state = 0.f;

        // gate pre-activations two frames ahead (HBM / last-level cache latency is longer than a step)
        float pre_a[4] = {0.f, 0.f, 0.f, 0.f}, pre_b[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave < LX_MT) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pre_a[g] = gates_in[gate_off + g * hidden];
                if (frames > 1) pre_b[g] = gates_in[static_cast<size_t>(batch) * (4 * hidden) + gate_off + g * hidden];
            }
        }

#if NBASR_LX_STAMPS
        const bool stamping = tile == 0 && wave == 0 && lane == 0;
        unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>(images + static_cast<size_t>(n_tiles) * 2 * LX_IMAGE_BYTES) +
                                               static_cast<size_t>(slice) * LX_STAMP_FRAMES * LX_STAMP_POINTS;
        unsigned long long st[LX_STAMP_POINTS] = {};
#endif
        int t = 0;
        bool failed = false;
        for (; t < frames; ++t) {
            LX_STAMP(0);
            float pre_c[4] = {0.f, 0.f, 0.f, 0.f};
            if (wave < LX_MT && t + 2 < frames) {
                const float* gin = gates_in + static_cast<size_t>(t + 2) * batch * (4 * hidden) + gate_off;
#pragma unroll
                for (int g = 0; g < 4; ++g) pre_c[g] = gin[g * hidden];
            }
            xfloat4 acc0[LX_MT], acc1[LX_MT];
#pragma unroll
            for (int mt = 0; mt < LX_MT; ++mt) { acc0[mt] = xfloat4{0.f, 0.f, 0.f, 0.f}; acc1[mt] = xfloat4{0.f, 0.f, 0.f, 0.f}; }
            if (t > 0) {
                const unsigned char* img = image + ((t - 1) & 1) * LX_IMAGE_BYTES;
                const unsigned want = lx_tag(t - 1);
                xuint4 raw[2][2];
                const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
                bool ok = false;
                for (;;) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const unsigned char* gp = img + ((((2 * wave + kk) * 2 + g) * 64) + lane) * 16;
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(raw[kk][g]) : "v"(gp) : "memory");   // sc1: past this CU's L1, served by the XCD's L2
                        }
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1]) :: "memory");
                    bool mine = true;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            unsigned tags = 0x40000000u;
#pragma unroll
                            for (int e = 0; e < 4; ++e) tags &= raw[kk][g][e] ^ ~want;          // bit 30 stays set while every dword's tag == want
                            mine = mine && (!live[kk][g] || (tags & 0x40000000u) != 0);
                        }
                    ok = __all(mine);
                    if (ok || __builtin_amdgcn_s_memrealtime() - t_start > LX_TIMEOUT_TICKS) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!ok && lane == 0) { s_role[3] = 1; __hip_atomic_store(ws + LX_W_STATUS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                LX_STAMP(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    // 8 dwords (hi << 16 | lo') of 8 units -> the two B fragments: 8 hi halves, 8 lo' halves (element j = unit j of the octet)
                    unsigned d[8];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[g * 4 + e] = live[kk][g] ? raw[kk][g][e] : 0u;
                    xuint4 bh, bl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bh[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x07060302u) & 0xbfffbfffu;      // (tag bits off)
                        bl[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x05040100u);
                    }
                    const xhalf8 hh = __builtin_bit_cast(xhalf8, bh), hl = __builtin_bit_cast(xhalf8, bl);
#pragma unroll
                    for (int mt = 0; mt < LX_MT; ++mt) {
                        const xhalf8 wh = __builtin_bit_cast(xhalf8, wfrag[kk][mt][0]), wl = __builtin_bit_cast(xhalf8, wfrag[kk][mt][1]);
                        acc0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hh, acc0[mt], 0, 0, 0);
                        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hl, acc1[mt], 0, 0, 0);
                        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, hh, acc1[mt], 0, 0, 0);
                    }
                }
            }
            const int par = t & 1;
            LX_STAMP(2);
#pragma unroll
            for (int mt = 0; mt < LX_MT; ++mt) red[par][wave][mt][lane] = acc0[mt] + acc1[mt] * 0.00048828125f;     // hi*hi + 2^-11 (hi*lo' + lo'*hi)
            LX_STAMP(3);
            // ONE barrier per step; the partials are double-buffered by step parity (a wave can only write those of step t + 2 after the
            // barrier of step t + 1, which the epilogue waves reach after they have read those of step t)
            __syncthreads();
            LX_STAMP(4);
            if (s_role[3]) { failed = true; break; }       // (workgroup-uniform: read behind the barrier)
            if (wave < LX_MT) {
                xfloat4 s = red[par][0][wave][lane];
#pragma unroll
                for (int w = 1; w < LX_WAVES; ++w) s += red[par][w][wave][lane];
#if NBASR_LX_STAMPS
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s) :: "memory");
#endif
                LX_STAMP(5);
                const float p0 = __builtin_fmaf(s[0], inv, pre_a[0]), p1 = __builtin_fmaf(s[1], inv, pre_a[1]);
                const float p2 = __builtin_fmaf(s[2], inv, pre_a[2]), p3 = __builtin_fmaf(s[3], inv, pre_a[3]);
                const float c_new = lx_sigmoid(p1) * c_state + lx_sigmoid(p0) * tanhf(p2);
                const float h_new = lx_sigmoid(p3) * tanhf(c_new);
                c_state = c_new;
#if NBASR_LX_STAMPS
                { float hh = h_new; asm volatile("" : "+v"(hh)); }
#endif
                LX_STAMP(6);
                // h as it is consumed: fp16 hi, fp16 lo' = (h - hi) * 2^11; a value that cannot be an LSTM output travels as lo' = NaN
                const _Float16 hi = static_cast<_Float16>(h_new);
                const _Float16 lo = static_cast<_Float16>((h_new - static_cast<float>(hi)) * 2048.0f);
                unsigned dw = (static_cast<unsigned>(__builtin_bit_cast(unsigned short, hi)) << 16) | __builtin_bit_cast(unsigned short, lo);
                if (!(fabsf(h_new) <= 1.0f)) dw = 0x00007e00u;
                dw |= lx_tag(t);
                const xuint4 gran = {static_cast<unsigned>(__shfl(static_cast<int>(dw), n16)), static_cast<unsigned>(__shfl(static_cast<int>(dw), n16 + 16)),
                                     static_cast<unsigned>(__shfl(static_cast<int>(dw), n16 + 32)), static_cast<unsigned>(__shfl(static_cast<int>(dw), n16 + 48))};
                if (q_ok) *reinterpret_cast<xuint4*>(image + (t & 1) * LX_IMAGE_BYTES + pub_off) = gran;      // PLAIN store: the line stays in this XCD's L2
                if (e_ok) h_out[(static_cast<size_t>(eb) * frames + t) * hidden + eu] = h_new;
#pragma unroll
                for (int g = 0; g < 4; ++g) { pre_a[g] = pre_b[g]; pre_b[g] = pre_c[g]; }
                LX_STAMP(7);
#if NBASR_LX_STAMPS
                if (stamping && t < LX_STAMP_FRAMES)
                    for (int i = 0; i < LX_STAMP_POINTS; ++i) stamp_base[t * LX_STAMP_POINTS + i] = st[i];
#endif
            }
        }
        if (failed) {                                      // timed out: make the failure visible in the output too
            if (e_ok) {
                const float nan = __builtin_nanf("");
                for (int u = t; u < frames; ++u) h_out[(static_cast<size_t>(eb) * frames + u) * hidden + eu] = nan;
            }
            break;
        }
        if (e_ok) cell[static_cast<size_t>(eb) * hidden + eu] = c_state;
        if (rank == 0 && threadIdx.x == 0) atomicAdd(ws + LX_W_TILES_DONE, 1u);
    }

    // ---- the last workgroup to leave checks that every tile was computed (an XCD that never collected its slices claims nothing) ----
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = atomicAdd(ws + LX_W_EXITS, 1u);
        if (before + 1 == static_cast<unsigned>(total_wgs)) {
            const unsigned done = atomicAdd(ws + LX_W_TILES_DONE, 0u);
            if (done != static_cast<unsigned>(n_tiles)) __hip_atomic_store(ws + LX_W_STATUS, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace nbasr

using namespace nbasr;

static inline int lx_slices(int hidden) { return (hidden + LX_UNITS - 1) / LX_UNITS; }

extern "C" size_t nbasr_lstm_packed_whh16_bytes(int hidden)
{
    if (hidden <= 0 || hidden > LX_KSTEPS * 32) return 0;
    return LX_PACK_HEADER_BYTES + static_cast<size_t>(lx_slices(hidden)) * LX_WAVES * 2 * LX_MT * 2 * 64 * 16;
}

extern "C" int nbasr_lstm_pack_whh16(const float* w_hh, void* packed, int hidden, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(hidden > 0 && hidden % 4 == 0 && hidden <= LX_KSTEPS * 32, NBASR_EINVAL,
                  "nbasr_lstm_pack_whh16: hidden=%d must be a positive multiple of 4, at most %d", hidden, LX_KSTEPS * 32);
    NBASR_REQUIRE(w_hh && packed, NBASR_ENULL, "nbasr_lstm_pack_whh16: NULL pointer");
    NBASR_REQUIRE(aligned16(packed), NBASR_EALIGN, "nbasr_lstm_pack_whh16: packed must be 16-byte aligned");
    hipError_t e = hipMemsetAsync(packed, 0, LX_PACK_HEADER_BYTES, as_stream(stream));
    if (e != hipSuccess) { set_error("nbasr_lstm_pack_whh16: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    hipLaunchKernelGGL(lx_absmax_kernel, dim3(256), dim3(256), 0, as_stream(stream), w_hh, static_cast<size_t>(4) * hidden * hidden,
                       static_cast<unsigned*>(packed));
    hipLaunchKernelGGL(lx_pack_kernel, dim3(512), dim3(256), 0, as_stream(stream), w_hh, static_cast<unsigned char*>(packed), hidden, lx_slices(hidden));
    return launch_status("nbasr_lstm_pack_whh16");
}

// The grid: one workgroup per compute unit of the device (every XCD then gets the workgroups of all its units under the round-robin
// dealing, i.e. all `slices` it needs -- and nothing depends on that).  A device with fewer than `slices` units cannot run the form.
static int lx_grid(int hidden)
{
    static std::mutex m;
    static int cached[64] = {};
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(m);
    if (cached[device] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) { (void)hipGetLastError(); return 0; }
        cached[device] = cus;
    }
    return cached[device] >= lx_slices(hidden) ? cached[device] : 0;
}

extern "C" size_t nbasr_lstm_xcd_workspace_bytes(int batch, int hidden)
{
    if (batch <= 0 || hidden <= 0 || hidden % 4 || hidden > LX_KSTEPS * 32) return 0;
    const size_t tiles = (batch + 15) / 16;
    if (tiles > LX_MAX_TILES) return 0;
    return LX_HEADER_WORDS * sizeof(unsigned) + tiles * 2 * LX_IMAGE_BYTES + LX_STAMP_BYTES;
}

extern "C" int nbasr_lstm_recurrence_xcd(const float* gates_ws, const void* packed_whh16, float* cell_ws, float* h_out, void* xcd_ws,
                                         int batch, int frames, int hidden, int flags, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && frames >= 0 && hidden > 0, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: bad sizes");
    NBASR_REQUIRE(hidden % 4 == 0, NBASR_EALIGN, "nbasr_lstm_recurrence_xcd: hidden=%d must be a multiple of 4", hidden);
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(gates_ws && packed_whh16 && cell_ws && h_out && xcd_ws, NBASR_ENULL, "nbasr_lstm_recurrence_xcd: NULL pointer");
    NBASR_REQUIRE(aligned16(packed_whh16) && aligned16(xcd_ws), NBASR_EALIGN, "nbasr_lstm_recurrence_xcd: packed_whh16, xcd_ws must be 16-byte aligned");
    const size_t ws_bytes = nbasr_lstm_xcd_workspace_bytes(batch, hidden);
    NBASR_REQUIRE(ws_bytes != 0, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: batch=%d hidden=%d does not fit the form (hidden <= %d, batch <= %d); "
                  "use nbasr_lstm_recurrence_packed", batch, hidden, LX_KSTEPS * 32, LX_MAX_TILES * 16);
    const int grid = lx_grid(hidden);
    NBASR_REQUIRE(grid > 0, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: the device has fewer compute units than the %d slices of hidden=%d",
                  lx_slices(hidden), hidden);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    hipError_t e = hipStreamIsCapturing(as_stream(stream), &capturing);
    if (e != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
    const bool captured = capturing != hipStreamCaptureStatusNone;
    int device = 0;
    e = hipGetDevice(&device);
    NBASR_REQUIRE(e == hipSuccess && device >= 0 && device < 64, NBASR_EINVAL, "nbasr_lstm_recurrence_xcd: device %d out of range", device);
    // Two of these grids can sit on the chip together; a third one could split an XCD's units with them so that none of the three
    // collects its slices (they would time out, bounded, and say so).  Launches from different streams of the process are therefore
    // chained, stream-ordered (each waits for the event behind the previous one; no host synchronisation).  A stream under capture
    // cannot take part in that chain: a captured launch (memset + kernel nodes) is ordered by its graph alone, and keeping replays of
    // such a graph from overlapping with other resident recurrences is the caller's business (the executor replays a graph on the
    // stream its other forwards run on).
    static std::mutex chain_mutex;
    static hipEvent_t chain_done[64] = {};
    std::unique_lock<std::mutex> lock(chain_mutex);
    hipEvent_t& done = chain_done[device];
    e = hipSuccess;
    if (!captured) {
        if (done == nullptr) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        else e = hipStreamWaitEvent(as_stream(stream), done, 0);
    }
    if (e == hipSuccess) e = hipMemsetAsync(xcd_ws, 0, ws_bytes, as_stream(stream));
    if (e != hipSuccess) { set_error("nbasr_lstm_recurrence_xcd: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    const int n_tiles = (batch + 15) / 16;
    hipLaunchKernelGGL(lstm_xcd_kernel, dim3(grid), dim3(64 * LX_WAVES), 0, as_stream(stream), gates_ws, static_cast<const unsigned char*>(packed_whh16),
                       cell_ws, h_out, static_cast<unsigned*>(xcd_ws), batch, frames, hidden, lx_slices(hidden), n_tiles, grid, flags);
    const int rc = launch_status("nbasr_lstm_recurrence_xcd");
    if (rc != NBASR_OK || captured) return rc;
    e = hipEventRecord(done, as_stream(stream));
    if (e != hipSuccess) { set_error("nbasr_lstm_recurrence_xcd: hipEventRecord: %s", hipGetErrorString(e)); return static_cast<int>(e); }
    return NBASR_OK;
}
