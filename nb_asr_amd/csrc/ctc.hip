// The step right after the forward pass (reference training/torch/trainer.py:217-219, 229-247):
//   log_probs = log_softmax(logits, dim = classes);   output_len = audio_len // 4;   decode.
// This file: log_softmax, the length mapping and GREEDY CTC decoding (per-frame argmax, collapse repeats, drop the blank =
// class 0, F.ctc_loss's default blank).  The beam search the reference's Trainer.decode uses (ctcdecode), the phoneme folding
// and the error rate are in ctc_decode.hip.
//
// One 256-thread workgroup per utterance: a thread owns a frame (49 contiguous floats), computes max / log-sum-exp / argmax
// in registers, then the surviving tokens are compacted with a workgroup prefix sum so the output order is the frame order.
#include "common.h"

namespace nbasr {

__global__ __launch_bounds__(256) void ctc_postprocess_kernel(
    const float* __restrict__ logits, const int* __restrict__ lengths, float* __restrict__ log_probs,
    int* __restrict__ tokens, int* __restrict__ token_counts, int frames, int classes, int blank)
{
    __shared__ int s_scan[256];
    __shared__ int s_last;     // argmax of the last frame of the previous chunk
    __shared__ int s_base;     // tokens emitted by previous chunks
    const int b = blockIdx.x;
    const int len = lengths ? min(max(lengths[b], 0), frames) : frames;
    if (threadIdx.x == 0) { s_last = -1; s_base = 0; }
    __syncthreads();

    for (int t0 = 0; t0 < frames; t0 += 256) {
        const int t = t0 + threadIdx.x;
        int best = -1;
        if (t < frames) {
            const float* row = logits + (static_cast<size_t>(b) * frames + t) * classes;
            float m = row[0];
            best = 0;
            for (int c = 1; c < classes; ++c) {
                const float v = row[c];
                if (v > m) { m = v; best = c; }                 // first maximum wins, like torch.argmax
            }
            if (log_probs) {
                float s = 0.f;
                for (int c = 0; c < classes; ++c) s += expf(row[c] - m);
                const float lse = m + logf(s);
                float* out = log_probs + (static_cast<size_t>(b) * frames + t) * classes;
                for (int c = 0; c < classes; ++c) out[c] = row[c] - lse;
            }
        }
        // neighbour's argmax (previous frame) through LDS
        s_scan[threadIdx.x] = best;
        __syncthreads();
        const int prev = threadIdx.x == 0 ? s_last : s_scan[threadIdx.x - 1];
        const int keep = (tokens != nullptr && t < len && best != blank && best != prev) ? 1 : 0;
        __syncthreads();
        if (threadIdx.x == 255) s_last = best;                   // t0 + 255 < frames whenever another chunk follows
        // inclusive prefix sum of the keep flags (Hillis-Steele over 256 entries)
        s_scan[threadIdx.x] = keep;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = static_cast<int>(threadIdx.x) >= off ? s_scan[threadIdx.x - off] : 0;
            __syncthreads();
            s_scan[threadIdx.x] += add;
            __syncthreads();
        }
        if (keep) tokens[static_cast<size_t>(b) * frames + s_base + s_scan[threadIdx.x] - 1] = best;
        __syncthreads();
        if (threadIdx.x == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (tokens) {
        const int n = s_base;
        for (int i = n + threadIdx.x; i < frames; i += 256) tokens[static_cast<size_t>(b) * frames + i] = -1;
        if (threadIdx.x == 0 && token_counts) token_counts[b] = n;
    }
}

}  // namespace nbasr

using namespace nbasr;

extern "C" int nbasr_ctc_postprocess(const float* logits, const int* lengths, float* log_probs, int* tokens, int* token_counts,
                                     int batch, int frames, int classes, int blank, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && frames >= 0 && classes > 0 && blank >= 0 && blank < classes, NBASR_EINVAL,
                  "nbasr_ctc_postprocess: bad sizes (batch=%d frames=%d classes=%d blank=%d)", batch, frames, classes, blank);
    if (batch == 0 || frames == 0) return NBASR_OK;
    NBASR_REQUIRE(logits, NBASR_ENULL, "nbasr_ctc_postprocess: logits must be non-NULL");
    NBASR_REQUIRE(log_probs || tokens, NBASR_ENULL, "nbasr_ctc_postprocess: nothing to compute (log_probs and tokens are both NULL)");
    NBASR_REQUIRE(!tokens || token_counts, NBASR_ENULL, "nbasr_ctc_postprocess: tokens needs token_counts");
    hipLaunchKernelGGL(ctc_postprocess_kernel, dim3(batch), dim3(256), 0, as_stream(stream), logits, lengths, log_probs, tokens,
                       token_counts, frames, classes, blank);
    return launch_status("nbasr_ctc_postprocess");
}
