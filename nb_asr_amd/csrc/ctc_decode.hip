// The reference's validation decode (reference training/torch/trainer.py:229-247 Trainer.decode):
//   CTCBeamDecoder(beam_width = 12, log_probs_input = True).decode(log_probs, output_len)   [ctcdecode, third-party C++]
//   -> PhonemeEncoder.fold_encoded(., 39)  -> torch_edit_distance.compute_wer(..., blank = [0], sep = [])  -> mean
// as two kernels: the CTC prefix beam search, and the label folding + blank removal + Levenshtein distance.
//
// Beam search (ctcdecode ctc_beam_search_decoder.cpp, no language model): per frame the vocabulary is cut to the
// `top_n` most probable classes; every live prefix keeps log P(ending in blank) and log P(ending in a non-blank); a prefix
// extended by class c is the SAME prefix as a live one that spells the same tokens (the trie of the C++ code), otherwise it
// is new; the `width` best of {live prefixes, new prefixes} by (score descending, last class ascending) survive.
//
// The frames of an utterance are a serial chain (250 at T = 1000), so the kernel is built for LATENCY: one wavefront per
// utterance, everything in registers.  Lane j holds live prefix j, lane c holds class c of the frame; values cross lanes by
// v_readlane (wave-uniform index) and DPP, not through LDS (a dependent LDS round trip costs ~130 cycles, an LDS atomic over
// the wavefront thousands).  Each lane tracks the best two candidates of its column (class); the survivors are picked by
// `width` rounds of a DPP wavefront maximum over 64-bit keys.  Prefix identity is a 64-bit hash of the token string (what
// the trie of ctcdecode provides; a re-created prefix must meet its still-live extensions again), the token strings
// themselves are (parent node, class) pairs in a per-utterance pool in global memory, walked backwards once at the end.
// All arithmetic is fp32 with ctcdecode's finite "-infinity" (-FLT_MAX).  The utterances of a batch run on different CUs.
#include "common.h"

#include <cfloat>

namespace nbasr {

constexpr int BEAM_MAX = 32;                   // beam width limit (reference: 12)
constexpr int BEAM_CLASSES = 64;               // classes limit = lanes (reference: 49)
constexpr float NEG = -FLT_MAX;

__device__ __forceinline__ float log_sum_exp(float x, float y)
{
    if (x <= NEG) return y;
    if (y <= NEG) return x;
    const float m = fmaxf(x, y);
    return logf(expf(x - m) + expf(y - m)) + m;
}

// order-preserving key: larger = better.  score first, then the SMALLER last class (+1: the root's -1 becomes 0), then the
// smaller table slot (makes the choice between exact ties deterministic; ctcdecode leaves it to nth_element)
__device__ __forceinline__ unsigned long long beam_key(float score, int last_plus1, int slot)
{
    unsigned u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return (static_cast<unsigned long long>(u) << 32) | (static_cast<unsigned long long>(0xFFFF - last_plus1) << 16) |
           static_cast<unsigned long long>(0xFFFF - slot);
}

// Vocabulary pruning of ctcdecode (get_pruned_log_probs with cutoff_prob = 1): per frame only the top_n most probable classes
// take part (ties: lower class first).  Every frame is independent, so this runs as a wide pre-pass -- one wavefront per
// frame, lane = class -- instead of sitting in the serial frame chain of the search: pruned entries become -FLT_MAX.
__global__ __launch_bounds__(256) void ctc_prune_kernel(const float* __restrict__ log_probs, float* __restrict__ pruned,
                                                        long long n_frames, int classes, int top_n)
{
    const long long f = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= n_frames) return;
    const float lp = lane < classes ? log_probs[f * classes + lane] : NEG;
    int rank = 0;
    for (int c = 0; c < classes; ++c) {
        const float o = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(lp), c));      // c is wave-uniform
        rank += (o > lp || (o == lp && c < lane)) ? 1 : 0;
    }
    if (lane < classes) pruned[f * classes + lane] = rank < top_n ? lp : NEG;
}

__device__ __forceinline__ int rl(int v, int src) { return __builtin_amdgcn_readlane(v, src); }                 // src wave-uniform
__device__ __forceinline__ float rl(float v, int src) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src)); }
__device__ __forceinline__ unsigned long long rl(unsigned long long v, int src)
{
    const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(v), src), hi = __builtin_amdgcn_readlane(static_cast<unsigned>(v >> 32), src);
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src)
{
    const unsigned lo = __shfl(static_cast<unsigned>(v), src), hi = __shfl(static_cast<unsigned>(v >> 32), src);
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<unsigned>(v)), CTRL, 0xF, 0xF, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<unsigned>(v >> 32)), CTRL, 0xF, 0xF, false);
    const unsigned long long o = (static_cast<unsigned long long>(hi) << 32) | lo;
    return o > v ? o : v;
}

// maximum over the wavefront, wave-uniform result: butterfly inside each row of 16 lanes by DPP (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four rows by v_readlane
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v)
{
    v = dpp_max<0xB1>(v);
    v = dpp_max<0x4E>(v);
    v = dpp_max<0x141>(v);
    v = dpp_max<0x140>(v);
    const unsigned long long a = rl(v, 0), b = rl(v, 16), c = rl(v, 32), d = rl(v, 48);
    const unsigned long long ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_max(unsigned v)
{
    const unsigned o = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max(unsigned v)
{
    v = dpp_max<0xB1>(v);
    v = dpp_max<0x4E>(v);
    v = dpp_max<0x141>(v);
    v = dpp_max<0x140>(v);
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// the best 64-bit key of the wavefront (0: none).  The score occupies the high word, so a 32-bit maximum finds it; only when
// two lanes tie on the score exactly do the low words (last class, slot) have to be compared
__device__ __forceinline__ unsigned long long wave_best_key(unsigned long long mine)
{
    const unsigned hi = static_cast<unsigned>(mine >> 32);
    const unsigned top_hi = wave_max(hi);
    if (top_hi == 0u) return 0ull;
    const unsigned long long holders = __ballot(hi == top_hi);
    if (__popcll(holders) == 1) return rl(mine, __builtin_ctzll(holders));
    return wave_max(hi == top_hi ? mine : 0ull);
}

// rotate a value by one lane around the wavefront (DPP wave_rol:1)
__device__ __forceinline__ int wave_rotate(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x134, 0xF, 0xF, false); }
__device__ __forceinline__ float wave_rotate(float v) { return __int_as_float(wave_rotate(__float_as_int(v))); }

__device__ __forceinline__ unsigned long long extend_hash(unsigned long long h, int c)
{
    h = (h ^ static_cast<unsigned long long>(c + 1)) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

__global__ __launch_bounds__(64) void ctc_beam_search_kernel(
    const float* __restrict__ log_probs, const int* __restrict__ lengths, int2* __restrict__ pool_all, int* __restrict__ beams,
    float* __restrict__ scores, int* __restrict__ beam_lens, int frames, int classes, int width, int blank)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int len = lengths ? min(max(lengths[b], 0), frames) : frames;
    int2* __restrict__ pool = pool_all + static_cast<size_t>(b) * (static_cast<size_t>(frames) * width + 1);
    const float* __restrict__ lp_b = log_probs + static_cast<size_t>(b) * frames * classes;

    // live prefix `lane` (lanes >= n_live: unused).  The empty prefix: P(blank-ending) = 1, hash 0, pool node 0.
    float p_b = lane == 0 ? 0.f : NEG, p_nb = NEG, p_score = lane == 0 ? 0.f : NEG;
    int p_last = -1, p_node = 0, p_len = 0;
    int p_mp = -1;                                  // index of the live prefix that is this one minus its last token, or -1
    unsigned long long p_hash = 0ull, p_phash = ~0ull;
    // The candidates (live prefix i + class c) are spread over the lanes by ROTATING the per-class values (log-probability,
    // class id, merge mask) one lane per live prefix: neither the extensions of one strong prefix nor those by one strong
    // class pile up in a single lane (a lane that wins more than twice in a frame has to rescan its candidates).
    unsigned merged = 0u;                           // class lane c: bit i set = (live prefix i + class c) is itself a live prefix
    int n_live = 1, n_nodes = 1;
    if (lane == 0) pool[0] = make_int2(-1, -1);

    float lp_next = (len > 0 && lane < classes) ? lp_b[lane] : NEG;
    for (int t = 0; t < len; ++t) {
        const float lp = lp_next;                                                                       // class `lane` of frame t
        if (t + 1 < len && lane < classes) lp_next = lp_b[static_cast<size_t>(t + 1) * classes + lane];  // in flight during this frame
        const bool keep = lane < classes && lp > NEG;               // pruned by the pre-pass (or impossible): -FLT_MAX
        const float lpk = keep ? lp : NEG;
        const float lp_blank = rl(lpk, blank);

        // ---- live prefixes (lane j): stay on blank / repeat the last class / absorb the extension that spells the same tokens
        const float lp_last = __shfl(lpk, p_last >= 0 ? p_last : 0);
        const int par = p_mp >= 0 ? p_mp : 0;
        const float par_score = __shfl(p_score, par), par_b = __shfl(p_b, par);
        const int par_last = __shfl(p_last, par);
        float n_b = NEG, n_nb = NEG, n_score = NEG;
        unsigned long long live_key = 0ull;
        if (lane < n_live) {
            n_b = lp_blank > NEG ? lp_blank + p_score : NEG;
            if (p_last >= 0 && lp_last > NEG) {
                n_nb = lp_last + p_nb;
                if (p_mp >= 0) n_nb = log_sum_exp(n_nb, p_last == par_last ? (par_b > NEG ? lp_last + par_b : NEG) : lp_last + par_score);
            }
            n_score = log_sum_exp(n_b, n_nb);
            live_key = beam_key(n_score, p_last + 1, width * BEAM_CLASSES + lane);
        }

        // ---- candidates of column `lane`: the live prefix of this lane and the extensions of every live prefix by class `lane`;
        //      the best two are tracked, `taken` (bit i: extension of prefix i, bit 32: the live prefix) excludes picked ones
        unsigned long long taken = 0ull;
        auto column_scan = [=](unsigned long long excluded) -> ulonglong2 {
            unsigned long long m1 = (excluded >> 32) & 1ull ? 0ull : live_key, m2 = 0ull;
            float lp_c = lpk;                                        // log-probability of class (lane + i * step) mod 64
            int cls = lane, mrg = static_cast<int>(merged);
            for (int i = 0; i < n_live; ++i) {
                const float sc = rl(p_score, i), bp = rl(p_b, i);
                const int last = rl(p_last, i);
                const float v = (cls == last) ? (bp > NEG ? lp_c + bp : NEG) : lp_c + sc;
                unsigned long long k = beam_key(v, cls + 1, i * BEAM_CLASSES + cls);
                if (!(lp_c > NEG) || cls == blank || (((mrg | static_cast<int>(excluded)) >> i) & 1)) k = 0ull;
                const unsigned long long lo = k < m1 ? k : m1;
                m1 = k > m1 ? k : m1;
                m2 = lo > m2 ? lo : m2;
                lp_c = wave_rotate(lp_c);
                cls = wave_rotate(cls);
                mrg = wave_rotate(mrg);
            }
            return make_ulonglong2(m1, m2);
        };
        ulonglong2 best = column_scan(0ull);
        unsigned long long mine = best.x, second = best.y;
        bool second_known = true;

        // ---- the `width` best of everything, best first: lane r keeps the winner of round r
        unsigned long long my_pick = 0ull;
        int n_next = 0;
        for (int r = 0; r < width; ++r) {
            const unsigned long long top = wave_best_key(mine);
            if (top == 0ull) break;                                    // fewer candidates than the beam is wide
            if (lane == r) my_pick = top;
            bool rescan = false;
            if (mine == top) {                                         // exactly one lane: the slot makes keys unique
                const int row = (0xFFFF - static_cast<int>(top & 0xFFFFu)) / BEAM_CLASSES;
                taken |= 1ull << (row == width ? 32 : row);
                if (second_known) { mine = second; second_known = false; }
                else rescan = true;
            }
            if (__any(rescan)) {                                       // by ALL lanes: the scan rotates values across the wavefront
                best = column_scan(taken);
                mine = best.x; second = best.y; second_known = true;
            }
            ++n_next;
        }

        // ---- the survivors become the live prefixes of the next frame: lane r gathers entry r from its source lane
        const int slot = 0xFFFF - static_cast<int>(my_pick & 0xFFFFu), row = slot / BEAM_CLASSES, col = slot % BEAM_CLASSES;
        const bool mine_valid = lane < n_next, survivor = row == width;
        const int src = mine_valid ? (survivor ? col : row) : 0;
        const float s_nb_new = __shfl(n_nb, src), s_b_new = __shfl(n_b, src), s_score_new = __shfl(n_score, src);
        const float s_b = __shfl(p_b, src), s_score = __shfl(p_score, src);
        const int s_last = __shfl(p_last, src), s_node = __shfl(p_node, src), s_len = __shfl(p_len, src);
        const unsigned long long s_hash = shfl64(p_hash, src), s_phash = shfl64(p_phash, src);
        const float lp_col = __shfl(lpk, mine_valid && !survivor ? col : 0);
        if (mine_valid && survivor) {
            p_b = s_b_new; p_nb = s_nb_new; p_score = s_score_new;
            p_last = s_last; p_node = s_node; p_len = s_len; p_hash = s_hash; p_phash = s_phash;
        } else if (mine_valid) {
            const float v = (col == s_last) ? (s_b > NEG ? lp_col + s_b : NEG) : lp_col + s_score;
            p_b = NEG; p_nb = v; p_score = v;
            p_last = col; p_len = s_len + 1; p_phash = s_hash; p_hash = extend_hash(s_hash, col);
            p_node = -1 - s_node;                                       // parent's node, until this prefix gets its own
        } else {
            p_b = p_nb = p_score = NEG; p_last = -1; p_node = 0; p_len = 0; p_hash = 0ull; p_phash = ~0ull;
        }
        // new pool nodes in rank order
        const bool is_new = mine_valid && p_node < 0;
        const unsigned long long new_mask = __ballot(is_new);
        if (is_new) {
            const int id = n_nodes + __popcll(new_mask & ((1ull << lane) - 1ull));
            pool[id] = make_int2(-1 - p_node, p_last);
            p_node = id;
        }
        n_nodes += __popcll(new_mask);
        n_live = n_next;
        // which live prefix is this one minus its last token (same length - 1, same token string)
        p_mp = -1;
        for (int i = 0; i < n_live; ++i)
            if (rl(p_hash, i) == p_phash && rl(p_len, i) + 1 == p_len) p_mp = i;
        if (lane >= n_live || p_last < 0) p_mp = -1;
        // class lanes: the extensions that are live prefixes themselves
        merged = 0u;
        for (int j = 0; j < n_live; ++j) {
            const int mp = rl(p_mp, j), last = rl(p_last, j);
            if (mp >= 0 && lane == last) merged |= 1u << mp;
        }
    }

    // results, best first (the selection of the last frame already ordered them; a zero-length utterance has the empty prefix)
    __syncthreads();                                                   // pool entries written by other lanes
    if (lane < width) {
        int* out = beams + (static_cast<size_t>(b) * width + lane) * frames;
        const bool live = lane < n_live;
        const int n = live ? p_len : 0;
        if (live) {
            int node = p_node;
            for (int k = n - 1; k >= 0; --k) {
                const int2 e = pool[node];
                out[k] = e.y;
                node = e.x;
            }
        }
        for (int k = n; k < frames; ++k) out[k] = 0;
        scores[static_cast<size_t>(b) * width + lane] = live ? -p_score : FLT_MAX;     // ctcdecode returns -log P
        beam_lens[static_cast<size_t>(b) * width + lane] = n;
    }
}

// ---- CTC loss (forward value) -----------------------------------------------------------------------------------------------
// The reference's loss (training/torch/trainer.py:36-42): F.ctc_loss(log_probs (T, B, C), targets, output_len, targets_len,
// reduction='none', zero_infinity=True) / output_len, then the mean over the batch (the mean is left to the caller).
// Standard alpha recursion in log space over the blank-extended label sequence l' (2L + 1 positions), as ATen's ctc_loss does
// it: alpha_t(s) = logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), alpha_{t-1}(s-2) if l'_s != blank and l'_s != l'_{s-2}) +
// log_probs[t][l'_s];  nll = -logsumexp(alpha_{T-1}(2L), alpha_{T-1}(2L-1)).  One wavefront per utterance, positions strided
// over the lanes, alpha double-buffered in LDS (a single wavefront needs no barrier: its LDS operations execute in issue
// order, wave_sync only keeps the compiler from reordering them).
constexpr int CTC_MAX_LABELS = 1024;

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void ctc_loss_kernel(const float* __restrict__ log_probs, const int* __restrict__ lengths,
                                                      const int* __restrict__ targets, const int* __restrict__ target_lengths,
                                                      float* __restrict__ losses, int frames, int classes, int ld_targets, int blank,
                                                      int divide_by_length)
{
    __shared__ float s_alpha[2][2 * CTC_MAX_LABELS + 1];
    __shared__ int s_label[2 * CTC_MAX_LABELS + 1];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int len = min(max(lengths[b], 0), frames);
    const int n_lab = min(max(target_lengths[b], 0), ld_targets);
    const int n_pos = 2 * n_lab + 1;
    const float ninf = -INFINITY;
    const float* __restrict__ lp_b = log_probs + static_cast<size_t>(b) * frames * classes;
    bool bad = false;
    for (int s = lane; s < n_pos; s += 64) {
        int lab = blank;
        if (s & 1) {
            lab = targets[static_cast<size_t>(b) * ld_targets + (s >> 1)];
            if (lab < 0 || lab >= classes) { bad = true; lab = blank; }
        }
        s_label[s] = lab;
    }
    wave_sync();
    float nll = INFINITY;
    if (len > 0) {
        // t = 0: only the first blank and the first label are reachable
        for (int s = lane; s < n_pos; s += 64) s_alpha[0][s] = s < 2 ? lp_b[s_label[s]] : ninf;
        wave_sync();
        int cur = 0;
        for (int t = 1; t < len; ++t) {
            const float* __restrict__ row = lp_b + static_cast<size_t>(t) * classes;
            const float* prev = s_alpha[cur];
            float* next = s_alpha[cur ^ 1];
            for (int s = lane; s < n_pos; s += 64) {
                const int lab = s_label[s];
                const float lp = row[lab];
                const float a = prev[s];
                const float a1 = s > 0 ? prev[s - 1] : ninf;
                const float a2 = (s > 1 && lab != blank && lab != s_label[s - 2]) ? prev[s - 2] : ninf;
                const float m = fmaxf(a, fmaxf(a1, a2));
                next[s] = m == ninf ? ninf : logf(expf(a - m) + expf(a1 - m) + expf(a2 - m)) + m + lp;
            }
            cur ^= 1;
            wave_sync();
        }
        if (lane == 0) {
            const float l1 = s_alpha[cur][n_pos - 1], l2 = n_pos > 1 ? s_alpha[cur][n_pos - 2] : ninf;
            const float m = fmaxf(l1, l2);
            nll = m == ninf ? INFINITY : -(logf(expf(l1 - m) + expf(l2 - m)) + m);
        }
    } else if (n_lab == 0) {
        nll = 0.f;                                                // nothing to emit in no frames: probability 1
    }
    const bool any_bad = __any(bad);
    if (lane == 0) {
        float out = (nll == INFINITY) ? 0.f : nll;               // zero_infinity=True
        if (divide_by_length) out = out / static_cast<float>(lengths[b]);       // the reference divides by output_len as given
        losses[b] = any_bad ? NAN : out;
    }
}

// ---- CTC loss gradient (first step of the backward pass) -------------------------------------------------
// d L / d logits for L = mean_b( nll_b / len_b ), log_probs = log_softmax(logits)  (reference trainer.py:36-42, 217-222: this is
// what `_regu_loss.backward()` propagates into the model, without the weight-norm term).  With alpha as above and beta the
// mirrored recursion from the last frame,
//     grad[b][t][c] = ( exp(lp[t][c]) - exp( log sum_{s: l'_s = c} exp(alpha_t(s) + beta_t(s)) + nll - lp[t][c] ) ) / (B len_b)
// for t < len_b, 0 beyond (ATen's ctc_loss backward composed with log_softmax's; Graves et al. 2006, eq. 16); 0 for utterances
// whose loss is infinite (zero_infinity).  One wavefront per utterance: alpha of every frame goes to a global workspace in
// the forward sweep, the backward sweep keeps beta in LDS and turns each frame into its gradient row: lane c sums, in a fixed
// order, the positions that carry class c (deterministic; classes <= 64).
__global__ __launch_bounds__(64) void ctc_grad_kernel(const float* __restrict__ log_probs, const int* __restrict__ lengths,
                                                      const int* __restrict__ targets, const int* __restrict__ target_lengths,
                                                      float* __restrict__ alpha_ws, float* __restrict__ losses, float* __restrict__ grad,
                                                      int batch, int frames, int classes, int ld_targets, int blank)
{
    __shared__ float s_beta[2][2 * CTC_MAX_LABELS + 1];
    __shared__ float s_ab[2 * CTC_MAX_LABELS + 1];
    __shared__ int s_label[2 * CTC_MAX_LABELS + 1];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int len = min(max(lengths[b], 0), frames);
    const int n_lab = min(max(target_lengths[b], 0), ld_targets);
    const int n_pos = 2 * n_lab + 1;
    const float ninf = -INFINITY;
    const float* __restrict__ lp_b = log_probs + static_cast<size_t>(b) * frames * classes;
    float* __restrict__ g_b = grad + static_cast<size_t>(b) * frames * classes;
    float* __restrict__ al = alpha_ws + static_cast<size_t>(b) * frames * (2 * static_cast<size_t>(ld_targets) + 1);
    const int pitch = 2 * ld_targets + 1;
    bool bad = false;
    for (int s = lane; s < n_pos; s += 64) {
        int lab = blank;
        if (s & 1) {
            lab = targets[static_cast<size_t>(b) * ld_targets + (s >> 1)];
            if (lab < 0 || lab >= classes) { bad = true; lab = blank; }
        }
        s_label[s] = lab;
    }
    wave_sync();
    const bool any_bad = __any(bad);
    // ---- forward sweep: alpha_t for every frame --------------------------------------------------------------------
    float nll = INFINITY;
    if (len > 0) {
        for (int s = lane; s < n_pos; s += 64) al[s] = s < 2 ? lp_b[s_label[s]] : ninf;
        __threadfence_block();
        for (int t = 1; t < len; ++t) {
            const float* __restrict__ row = lp_b + static_cast<size_t>(t) * classes;
            const float* prev = al + static_cast<size_t>(t - 1) * pitch;
            float* next = al + static_cast<size_t>(t) * pitch;
            __syncthreads();                                           // previous row (global memory) written by other lanes
            for (int s = lane; s < n_pos; s += 64) {
                const int lab = s_label[s];
                const float a = prev[s];
                const float a1 = s > 0 ? prev[s - 1] : ninf;
                const float a2 = (s > 1 && lab != blank && lab != s_label[s - 2]) ? prev[s - 2] : ninf;
                const float m = fmaxf(a, fmaxf(a1, a2));
                next[s] = m == ninf ? ninf : logf(expf(a - m) + expf(a1 - m) + expf(a2 - m)) + m + row[lab];
            }
        }
        __syncthreads();
        const float* last = al + static_cast<size_t>(len - 1) * pitch;
        const float l1 = last[n_pos - 1], l2 = n_pos > 1 ? last[n_pos - 2] : ninf;
        const float m = fmaxf(l1, l2);
        nll = m == ninf ? INFINITY : -(logf(expf(l1 - m) + expf(l2 - m)) + m);
    } else if (n_lab == 0) {
        nll = 0.f;
    }
    const bool finite = nll != INFINITY && !any_bad && len > 0;
    if (lane == 0) losses[b] = any_bad ? NAN : (nll == INFINITY ? 0.f : nll / static_cast<float>(lengths[b]));
    const float scale = finite ? 1.0f / (static_cast<float>(batch) * static_cast<float>(lengths[b])) : 0.f;
    // ---- backward sweep: beta in LDS, one gradient row per frame --------------------------------------------------------
    int cur = 0;
    for (int t = frames - 1; t >= 0; --t) {
        float* grow = g_b + static_cast<size_t>(t) * classes;
        if (t >= len || !finite) {                                      // beyond the utterance, or no gradient at all
            if (lane < classes) grow[lane] = 0.f;
            for (int c = 64 + lane; c < classes; c += 64) grow[c] = 0.f;
            continue;
        }
        const float* __restrict__ row = lp_b + static_cast<size_t>(t) * classes;
        const float* arow = al + static_cast<size_t>(t) * pitch;
        float* beta = s_beta[cur];
        const float* bnext = s_beta[cur ^ 1];
        for (int s = lane; s < n_pos; s += 64) {
            const int lab = s_label[s];
            float bt;
            if (t == len - 1) {
                bt = (s >= n_pos - 2) ? row[lab] : ninf;               // only the last blank and the last label can end the path
            } else {
                const float b0 = bnext[s];
                const float b1 = s + 1 < n_pos ? bnext[s + 1] : ninf;
                const float b2 = (s + 2 < n_pos && s_label[s + 2] != blank && s_label[s + 2] != lab) ? bnext[s + 2] : ninf;
                const float m = fmaxf(b0, fmaxf(b1, b2));
                bt = m == ninf ? ninf : logf(expf(b0 - m) + expf(b1 - m) + expf(b2 - m)) + m + row[lab];
            }
            beta[s] = bt;
            s_ab[s] = arow[s] + bt;                                     // alpha and beta both contain lp[t][l'_s]: divided out below
        }
        wave_sync();
        for (int c = lane; c < classes; c += 64) {
            float m = ninf;
            for (int s = (c == blank ? 0 : 1); s < n_pos; s += 2) if (s_label[s] == c) m = fmaxf(m, s_ab[s]);      // blanks sit at even positions
            float sum = 0.f;
            if (m != ninf)
                for (int s = (c == blank ? 0 : 1); s < n_pos; s += 2) if (s_label[s] == c) sum += expf(s_ab[s] - m);
            const float lp = row[c];
            const float lcab = m == ninf ? ninf : logf(sum) + m;
            grow[c] = (expf(lp) - (lcab == ninf ? 0.f : expf(lcab + nll - lp))) * scale;
        }
        cur ^= 1;
        wave_sync();
    }
}

// ---- label table + blank removal + Levenshtein distance, one workgroup per utterance -------------------------------------
// D[i][j] over anti-diagonals d = i + j: the cells of a diagonal are independent, three diagonals rotate through LDS
// (indexed by i).  Integer arithmetic: exact.
constexpr int EDIT_MAX = 2048;                 // tokens per sequence after blank removal

__global__ __launch_bounds__(256) void token_errors_kernel(
    const int* __restrict__ hyp, const int* __restrict__ hyp_len, int ld_hyp, const int* __restrict__ ref,
    const int* __restrict__ ref_len, int ld_ref, const int* __restrict__ table, int n_table, int blank, int* __restrict__ out)
{
    __shared__ int s_h[EDIT_MAX], s_r[EDIT_MAX];
    __shared__ int s_diag[3][EDIT_MAX + 1];
    __shared__ int s_count[2];
    __shared__ int s_bad;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) { s_count[0] = s_count[1] = 0; s_bad = 0; }
    __syncthreads();
    // compaction in sequence order: wave 0 handles the hypothesis, wave 1 the reference (ballot prefix over 64-token chunks)
    if (tid < 128) {
        const int which = tid >> 6, lane = tid & 63;
        const int* src = which ? ref + static_cast<size_t>(b) * ld_ref : hyp + static_cast<size_t>(b) * ld_hyp;
        const int n = which ? min(max(ref_len[b], 0), ld_ref) : min(max(hyp_len[b], 0), ld_hyp);
        int* dst = which ? s_r : s_h;
        int count = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            int v = blank;
            if (i < n) {
                v = src[i];
                if (table) {
                    if (v < 0 || v >= n_table) { s_bad = 1; v = blank; }
                    else v = table[v];
                }
            }
            const bool keep = i < n && v != blank;
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < EDIT_MAX) dst[pos] = v;
            }
            count += __popcll(m);
        }
        if (lane == 0) s_count[which] = count;
    }
    __syncthreads();
    const int m = s_count[0], n = s_count[1];
    if (m > EDIT_MAX || n > EDIT_MAX || s_bad) {
        if (tid == 0) { out[2 * b] = -1; out[2 * b + 1] = s_bad ? -2 : -1; }
        return;
    }
    // diagonal d holds D[i][d - i] for max(0, d - n) <= i <= min(d, m)
    for (int d = 0; d <= m + n; ++d) {
        int* curd = s_diag[d % 3];
        const int* p1 = s_diag[(d + 2) % 3];      // d - 1
        const int* p2 = s_diag[(d + 1) % 3];      // d - 2
        const int lo = max(0, d - n), hi = min(d, m);
        for (int i = lo + tid; i <= hi; i += 256) {
            const int j = d - i;
            int v;
            if (i == 0) v = j;
            else if (j == 0) v = i;
            else {
                const int sub = p2[i - 1] + (s_h[i - 1] != s_r[j - 1] ? 1 : 0);
                v = min(min(p1[i - 1] + 1, p1[i] + 1), sub);        // D[i-1][j] + 1, D[i][j-1] + 1, D[i-1][j-1] + cost
            }
            curd[i] = v;
        }
        __syncthreads();
    }
    if (tid == 0) { out[2 * b] = s_diag[(m + n) % 3][m]; out[2 * b + 1] = n; }
}

}  // namespace nbasr

using namespace nbasr;

static size_t beam_pool_bytes(int batch, int frames, int beam_width)
{
    return static_cast<size_t>(batch) * (static_cast<size_t>(frames) * beam_width + 1) * sizeof(int2);
}

extern "C" size_t nbasr_ctc_beam_workspace_bytes(int batch, int frames, int classes, int beam_width)
{
    if (batch <= 0 || frames < 0 || classes <= 0 || beam_width <= 0) return 0;
    return beam_pool_bytes(batch, frames, beam_width) + static_cast<size_t>(batch) * frames * classes * sizeof(float);
}

extern "C" int nbasr_ctc_beam_search(const float* log_probs, const int* lengths, void* ws, int* beams, float* scores, int* beam_lens,
                                     int batch, int frames, int classes, int beam_width, int blank, int cutoff_top_n,
                                     nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && frames >= 0 && classes > 0 && blank >= 0 && blank < classes && cutoff_top_n > 0, NBASR_EINVAL,
                  "nbasr_ctc_beam_search: bad sizes (batch=%d frames=%d classes=%d blank=%d cutoff_top_n=%d)", batch, frames, classes, blank, cutoff_top_n);
    NBASR_REQUIRE(classes <= BEAM_CLASSES && beam_width >= 1 && beam_width <= BEAM_MAX, NBASR_EINVAL,
                  "nbasr_ctc_beam_search: classes=%d (limit %d) / beam_width=%d (limit %d) unsupported", classes, BEAM_CLASSES, beam_width, BEAM_MAX);
    if (batch == 0) return NBASR_OK;
    NBASR_REQUIRE(ws && scores && beam_lens && (frames == 0 || (log_probs && beams)), NBASR_ENULL, "nbasr_ctc_beam_search: NULL pointer");
    NBASR_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7u) == 0, NBASR_EALIGN, "nbasr_ctc_beam_search: workspace must be 8-byte aligned");
    hipStream_t s = as_stream(stream);
    const float* src = log_probs;
    if (cutoff_top_n < classes && frames > 0) {
        float* pruned = reinterpret_cast<float*>(static_cast<char*>(ws) + beam_pool_bytes(batch, frames, beam_width));
        const long long n_frames = static_cast<long long>(batch) * frames;
        hipLaunchKernelGGL(ctc_prune_kernel, dim3(static_cast<unsigned>((n_frames + 3) / 4)), dim3(256), 0, s, log_probs, pruned, n_frames,
                           classes, cutoff_top_n);
        src = pruned;
    }
    hipLaunchKernelGGL(ctc_beam_search_kernel, dim3(batch), dim3(64), 0, s, src, lengths, static_cast<int2*>(ws), beams, scores,
                       beam_lens, frames, classes, beam_width, blank);
    return launch_status("nbasr_ctc_beam_search");
}

extern "C" int nbasr_ctc_loss(const float* log_probs, const int* lengths, const int* targets, const int* target_lengths, float* losses,
                              int batch, int frames, int classes, int ld_targets, int blank, int divide_by_length, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && frames >= 0 && classes > 0 && ld_targets >= 0 && blank >= 0 && blank < classes, NBASR_EINVAL,
                  "nbasr_ctc_loss: bad sizes (batch=%d frames=%d classes=%d ld_targets=%d blank=%d)", batch, frames, classes, ld_targets, blank);
    NBASR_REQUIRE(ld_targets <= CTC_MAX_LABELS, NBASR_EINVAL, "nbasr_ctc_loss: at most %d labels per utterance, got ld_targets=%d", CTC_MAX_LABELS, ld_targets);
    if (batch == 0) return NBASR_OK;
    NBASR_REQUIRE(lengths && target_lengths && losses && (frames == 0 || log_probs) && (ld_targets == 0 || targets), NBASR_ENULL,
                  "nbasr_ctc_loss: NULL pointer");
    hipLaunchKernelGGL(ctc_loss_kernel, dim3(batch), dim3(64), 0, as_stream(stream), log_probs, lengths, targets, target_lengths, losses,
                       frames, classes, ld_targets, blank, divide_by_length);
    return launch_status("nbasr_ctc_loss");
}

extern "C" size_t nbasr_ctc_grad_workspace_bytes(int batch, int frames, int ld_targets)
{
    if (batch <= 0 || frames <= 0 || ld_targets < 0) return 0;
    return static_cast<size_t>(batch) * frames * (2 * static_cast<size_t>(ld_targets) + 1) * sizeof(float);
}

extern "C" int nbasr_ctc_loss_grad(const float* log_probs, const int* lengths, const int* targets, const int* target_lengths, void* ws,
                                   float* losses, float* grad_logits, int batch, int frames, int classes, int ld_targets, int blank,
                                   nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && frames >= 0 && classes > 0 && ld_targets >= 0 && blank >= 0 && blank < classes, NBASR_EINVAL,
                  "nbasr_ctc_loss_grad: bad sizes (batch=%d frames=%d classes=%d ld_targets=%d blank=%d)", batch, frames, classes, ld_targets, blank);
    NBASR_REQUIRE(ld_targets <= CTC_MAX_LABELS, NBASR_EINVAL, "nbasr_ctc_loss_grad: at most %d labels per utterance, got ld_targets=%d", CTC_MAX_LABELS, ld_targets);
    if (batch == 0) return NBASR_OK;
    NBASR_REQUIRE(lengths && target_lengths && losses && (frames == 0 || (log_probs && grad_logits && ws)) && (ld_targets == 0 || targets), NBASR_ENULL,
                  "nbasr_ctc_loss_grad: NULL pointer");
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(batch), dim3(64), 0, as_stream(stream), log_probs, lengths, targets, target_lengths,
                       static_cast<float*>(ws), losses, grad_logits, batch, frames, classes, ld_targets, blank);
    return launch_status("nbasr_ctc_loss_grad");
}

extern "C" int nbasr_token_error_counts(const int* hyp, const int* hyp_len, int ld_hyp, const int* ref, const int* ref_len, int ld_ref,
                                        const int* table, int n_table, int blank, int* counts, int batch, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && ld_hyp >= 0 && ld_ref >= 0 && n_table >= 0, NBASR_EINVAL, "nbasr_token_error_counts: bad sizes");
    if (batch == 0) return NBASR_OK;
    NBASR_REQUIRE(hyp_len && ref_len && counts && (ld_hyp == 0 || hyp) && (ld_ref == 0 || ref) && (n_table == 0 || table), NBASR_ENULL,
                  "nbasr_token_error_counts: NULL pointer");
    hipLaunchKernelGGL(token_errors_kernel, dim3(batch), dim3(256), 0, as_stream(stream), hyp, hyp_len, ld_hyp, ref, ref_len, ld_ref,
                       n_table ? table : nullptr, n_table, blank, counts);
    return launch_status("nbasr_token_error_counts");
}
