/*
 * nbasr.h -- C ABI of libnbasr_hip.so: the MI355X (gfx950) kernels behind the NAS-Bench-ASR
 * acoustic-model forward pass.
 *
 * Every entry point replaces one piece of the reference's PyTorch forward (files under
 * /root/reference/nasbench_asr/model/torch/ -- cited per function as file:line).  The reference
 * has no FFI of its own for this path (it calls torch ATen), so this ABI is the boundary a
 * `hip` backend package binds with ctypes; see INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - plain pointers and ints only; all pointers are DEVICE pointers owned by the caller
 *     (torch tensors' data_ptr()); no allocation, no ownership transfer, no host sync;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - activations use the reference's (batch, channels, frames) order with frames contiguous and
 *     a row pitch `ld` (elements) that must be a multiple of 4; columns frames..ld-1 of every
 *     activation buffer are kept at ZERO by every kernel (and must be zero on input);
 *   - weights are in the layouts torch stores them in (state_dict tensors, contiguous);
 *   - return value: 0 on success, a negative NBASR_E* code for argument errors, or a positive
 *     hipError_t; nbasr_last_error() gives a thread-local message for the last failure;
 *   - re-entrant and thread-safe (no global mutable state except the thread-local message).
 */
#ifndef NBASR_H
#define NBASR_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBASR_ABI_VERSION 6

#define NBASR_OK 0
#define NBASR_EINVAL (-1)   /* bad size / unsupported shape */
#define NBASR_EALIGN (-2)   /* pointer or pitch alignment   */
#define NBASR_ENULL (-3)    /* required pointer is NULL     */

typedef void* nbasr_stream_t;

/* Storage type of an activation tensor for the entry points that exist for both precisions (`dtype` argument).
 * NBASR_BF16 = bfloat16 storage (BASELINE config 4): values are converted to fp32 when loaded, every product, sum,
 * LayerNorm statistic and recurrent state is fp32, and a tensor is rounded to bf16 ONCE when it is written back.
 * bf16 rows are pitched to multiples of 8 frames (16 bytes), like fp32 rows to multiples of 4. */
#define NBASR_F32 0
#define NBASR_BF16 1

/* ABI version of the loaded library (== NBASR_ABI_VERSION it was built with). */
int nbasr_version(void);

/* Identifies the sources this library was compiled from: the first 16 hex digits of a SHA-256 over csrc and include files and the
 * compiler flags (nb_asr_amd/build.py:source_hash computes the same value from a checkout), "unknown" for a build that
 * bypassed build.py.  Static storage; never NULL. */
const char* nbasr_build_id(void);

/* Message for the most recent failing call on this thread ("" if none). */
const char* nbasr_last_error(void);

/* Zero-padding rule of PadConvRelu (reference ops.py:12-17, look-ahead context = 4 frames).
 * Pure host arithmetic; usable without a GPU. */
int nbasr_pad_amounts(int kernel, int dilation, int stride, int* left, int* right);

/* Number of output frames of the model for `frames` input frames (two stride-2 downsample
 * convs, reference model.py:76): ceil(ceil(T/2)/2).  Host arithmetic. */
int nbasr_output_frames(int frames);

/* ABI 6 -- a non-blocking stream of the library's own on the current device (hipStreamCreateWithFlags, default priority), for the life
 * of the process.  For a host whose framework hands out streams from a small recycled pool (PyTorch: 32 per priority): the streams a
 * pipelined forward's tail and the chains of `forward_many` run on are chosen by probing which candidates execute beside the streams
 * they must overlap with (a stream sits on one of a few hardware queues from its creation on; nb_asr_amd/streams.py), and a dozen
 * probes into the pool a "new" stream is one that is already in use.  No reference counterpart (the reference runs on one stream). */
int nbasr_stream_create(nbasr_stream_t* stream);

/* One entry point per operation.  Tensors x / y / skips are `dtype` tensors (NBASR_F32 | NBASR_BF16) where an entry point takes a
 * `dtype`; weights, biases, gamma / beta and statistics are fp32 everywhere.  A pending LayerNorm travels as a nbasr_deferred_ln
 * ("normalise on load"; NULL = none). */
typedef struct nbasr_deferred_ln {
    const float* stats;   /* (batch, 2, ld): row 0 = mean, row 1 = rstd = 1/sqrt(var + eps); both 0 in columns >= frames */
    const float* gamma;   /* (channels) */
    const float* beta;    /* (channels) */
} nbasr_deferred_ln;

/* LayerNorm over the channel dimension of (batch, channels, ld), biased variance
 * (reference model.py:92 + 125-128 and model.py:46-47 + 55-58; eps = 1e-3 there).  x of in_dtype -> y of out_dtype:
 * f32 -> f32, bf16 -> bf16, or bf16 -> f32 (the encoder output handed to the fp32 LSTM); in place (y == x) only when the two types
 * are equal.  absmax (may be NULL; f32 -> f32 only): absmax[b] = max |y[b, :, :]| (device pointer, `batch` floats, overwritten),
 * the range information the 2-way fp16 dense convolution below needs, produced while the normalised tensor is being written. */
int nbasr_layernorm_channels(const void* x, const float* gamma, const float* beta, void* y, float* absmax,
                             int batch, int channels, int frames, int ld, float eps, int in_dtype, int out_dtype,
                             nbasr_stream_t stream);

/* Dense PadConvRelu (groups = 1) on the fp32 matrix cores, fused bias + relu + clamp (+ skips):
 *   kernel == 8: the four downsample convs (reference model.py:82-89, ops.py:24-30), stride 1|2;
 *   kernel == 1: the `linear` node op (reference ops.py:42-50), stride must be 1.
 * x: (batch, c_in, ld_in) with frames_in valid frames (ld_in need not be a multiple of 4 here:
 * the model input has whatever length the caller gives); y: (batch, c_out, ld_out) with
 * ceil(frames_in/stride) valid frames; w: (c_out, c_in, kernel); skips as above (c_out, ld_out). */
int nbasr_dense_conv1d_fused(const float* x, const float* w, const float* bias,
                             const float* skip0, const float* skip1, const float* skip2,
                             float* y, int batch, int c_in, int frames_in, int ld_in,
                             int c_out, int ld_out, int kernel, int stride,
                             const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, nbasr_stream_t stream);

/* ---- the k = 8 dense PadConvRelu on the 16-bit matrix cores: ONE convolution entry point, three operand schemes ----------------
 * (reference model.py:82-89, ops.py:24-30).  Weights are split / re-laid-out once per weight version:
 *   bytes = nbasr_packed_dense_weights_bytes(scheme, c_out, c_in, 8, row_tile);
 *   nbasr_pack_dense_weights(scheme, w, packed, c_out, c_in, 8, stride, row_tile, stream);
 * (the packed image depends on the stride of the convolution that will consume it and on the row tile; images of different
 * schemes are NOT interchangeable), then nbasr_dense_conv1d_packed(scheme, ...) takes `packed` in place of w.
 *
 * NBASR_DENSE_BF16X3  fp32-accurate on the bf16 matrix cores (16x the fp32 MFMA rate): operands are split exactly into three bf16
 *     terms and six cross products are accumulated in fp32 (dropped terms <= 2^-24 of a product); fp32's exponent range, NaN / Inf
 *     propagate like in the reference.  x: a plain fp32 tensor in the activation layout (16-byte aligned, ld_in % 4 == 0, columns
 *     frames_in..ld_in-1 zero: the kernel fetches aligned 4-frame quads; nbasr_repitch brings a caller tensor into it); `ln`: its
 *     pending LayerNorm; skips allowed; row_tile 128.
 * NBASR_DENSE_F16X2   the same convolution with HALF the matrix instructions: operands are split into two fp16 terms, v = hi + lo
 *     (11 + 11 significand bits plus the rounding sign cover fp32's 24; the residual may be an fp16 subnormal, exact to 2^-24), and
 *     three cross products are accumulated in fp32 (dropped term <= 2^-24 of a product) as a two-level blocked sum: one block per
 *     16 input channels x 8 taps, then the total.  fp16's narrow exponent range is handled by exact power-of-two scalings that the
 *     epilogue undoes: each weight row is normalised at pack time, and each utterance of x by 2^k derived from x_absmax[b], a
 *     caller-supplied UPPER BOUND of max|x[b, :, :]| (device pointer, `batch` floats, finite; nbasr_layernorm_channels writes it as
 *     a by-product).  A bound that is too small makes fp16 overflow possible (undefined results); a loose bound only costs precision
 *     of elements more than 2^-16 below it (they keep an absolute accuracy of 2^-39 of the bound).
 *     x: a plain fp32 tensor as above (x_is_image = 0; skips allowed; row_tile 128), or (x_is_image = 1; no skips; row_tile 64, 128
 *     or 160) the pre-split fp16 OPERAND IMAGE that nbasr_layernorm_split_image / nbasr_split_image_ranged write,
 *       image[b][16-channel group][split][8-channel half][1 + ld rows][8 ch]   (row 0 zero, frame t at row t + 1;
 *       nbasr_split_image_bytes(batch, channels, ld) bytes),
 *     which the GEMM gathers by LDS-DMA (no vector work in the GEMM).  row_tile = output channels per workgroup: 160 pays where 128
 *     leaves a mostly empty last row tile (c_out = 800: 7 tiles, the last a quarter full, vs 5 full ones) or a partial last round of
 *     workgroups; 64 doubles the workgroups of a small batch.  frame_tile = output frames per workgroup: 256, or (ABI 5; image path
 *     only) 128 -- twice the workgroups, half the matrix work per K-step, where a small batch would leave most compute units without a
 *     workgroup.  Results are bit-identical between tiles: the K order of every output is the same.
 * NBASR_DENSE_BF16    the bf16 storage path (BASELINE config 4): ONE v_mfma_f32_16x16x32_bf16 per 32 products, fp32 accumulation,
 *     bias + relu + min(20) in fp32, one rounding to the bf16 output y (ld_out % 8 == 0).  x: nbasr_bf16_image's operand image
 *     (x_is_image = 1); weights: the fp32 values of the bf16 parameter; row_tile 128 or 160; frame_tile 256 or (ABI 6) 512 -- this
 *     flavour is bound by LDS fragment reads and a 512-frame tile re-uses a weight fragment 8 times instead of 4; bit-identical
 *     results; no bound, range, LayerNorm or skips.
 *
 * Per-utterance routing of caller data (x_range != NULL in place of x_absmax; the model input, whose range this library does not
 * control): nbasr_input_range writes, per utterance, range[4 b] = { max finite |x|, the quietest non-silent frame's max |x| over
 * channels, non-zero if any sample is Inf / NaN, unused } (x: (batch, channels, ld) fp32, any ld >= frames).  An utterance is
 * EXTREME when it holds a non-finite sample or a frame more than 2^12 below its loudest sample: the scaled fp16 scheme would then
 * lose precision in (or, for Inf, flush) the quiet part.  The two schemes are launched back to back on the same output with the same
 * x_range: NBASR_DENSE_F16X2 computes the ordinary utterances (scale from range[4 b]) and skips the extreme ones,
 * NBASR_DENSE_BF16X3 computes exactly those.  No host synchronisation.
 *
 * stats_part (may be NULL; fp32 schemes, no skips; ABI 5): the launch also emits the partial LayerNorm statistics of y from its own
 * epilogue -- per frame the (mean, M2) over every 16 output channels, part[16-channel unit][batch][2][ld_out] floats (16-byte aligned,
 * NBASR_DENSE_STATS_UNIT = 16; ceil(c_out / 16) * batch * 2 * ld_out floats) -- which nbasr_grouped_stats_finalize(stats_part, stats,
 * batch, c_out, frames_out, ld_out, groups = c_out, groups_per_part = 16, eps) merges into (mean, rstd) rows: what nbasr_channel_stats(y)
 * computes in a pass of its own over y (model.py:92, 125-128: the block LayerNorm behind every downsample convolution).  The unit does
 * not depend on row_tile: the statistics are the same arithmetic whatever tile the launch runs with.  With x_range routing both
 * launches take the same stats_part (each writes the utterances it computes). */
#define NBASR_DENSE_STATS_UNIT 16
#define NBASR_DENSE_BF16X3 0
#define NBASR_DENSE_F16X2 1
#define NBASR_DENSE_BF16 2
size_t nbasr_packed_dense_weights_bytes(int scheme, int c_out, int c_in, int kernel, int row_tile);
int nbasr_pack_dense_weights(int scheme, const float* w, void* packed, int c_out, int c_in, int kernel, int stride, int row_tile,
                             nbasr_stream_t stream);
int nbasr_dense_conv1d_packed(int scheme, const void* x, int x_is_image, const float* x_absmax, const float* x_range,
                              const void* packed_w, const float* bias,
                              const float* skip0, const float* skip1, const float* skip2, void* y,
                              int batch, int c_in, int frames_in, int ld_in, int c_out, int ld_out, int kernel, int stride,
                              int row_tile, int frame_tile, const nbasr_deferred_ln* ln, float* stats_part, nbasr_stream_t stream);
int nbasr_input_range(const float* x, float* range, int batch, int channels, int frames, int ld, nbasr_stream_t stream);

/* The operand images of the fp16 scheme.  nbasr_layernorm_split_image normalises x (batch, channels, ld) and writes it as the
 * image, scaled per utterance by the power of two that brings bound[b] into [2^14, 2^15), where bound[b] (batch floats, written
 * here; the convolution's x_absmax) is an upper bound of max|LayerNorm(x)[b]| computed together with the statistics (stats:
 * (batch, 2, ld) as nbasr_channel_stats).  Two kernels, 2 reads (the second L2-warm) + 1 write: the same traffic as
 * nbasr_layernorm_channels.  nbasr_split_image_ranged writes x -- the model input -- as the image, every utterance scaled by the
 * power of two that its x_range[4 b] implies (conv 0 then runs like convs 1-3). */
size_t nbasr_split_image_bytes(int batch, int channels, int ld);
int nbasr_layernorm_split_image(const float* x, const float* gamma, const float* beta, float* stats, float* bound,
                                void* image, int batch, int channels, int frames, int ld, float eps,
                                nbasr_stream_t stream);
int nbasr_split_image_ranged(const float* x, const float* x_range, void* image, int batch, int channels, int frames, int ld,
                             nbasr_stream_t stream);

/* nn.LSTM(input_size=c_in, hidden_size=hidden, batch_first) forward with zero initial state
 * (reference model.py:100 and 118-121): gates i,f,g,o; biases b_ih + b_hh, as two calls so that a caller can run the input
 * projection (one large GEMM) on one stream and the latency-bound recurrence (frames dependent steps) on another:
 *   gates_ws(frames, batch, 4*hidden) = x . w_ih^T + b_ih + b_hh ;   then h_out from gates_ws and the recurrent weight.
 * x: (batch, c_in, ld) encoder layout (the reference's permute is folded into the loader), `ln` its pending LayerNorm;
 * w_ih: (4*hidden, c_in); w_hh: (4*hidden, hidden); h_out: (batch, frames, hidden).
 * Workspaces (caller-owned): gates_ws (batch*frames*4*hidden floats), cell_ws (batch*hidden; holds c_T afterwards). */
int nbasr_lstm_input_projection(const float* x, const float* w_ih, const float* b_ih, const float* b_hh,
                                float* gates_ws, int batch, int c_in, int frames, int ld, int hidden,
                                const nbasr_deferred_ln* ln, nbasr_stream_t stream);
/* The recurrence, one launch per frame, on a fragment-ordered copy of w_hh (nbasr_lstm_packed_whh_bytes / nbasr_lstm_pack_whh, once
 * per weight version): every operand load of a wave is 1 KiB contiguous instead of 16 rows x 64 B of the (4H, H) matrix. */
size_t nbasr_lstm_packed_whh_bytes(int hidden);
int nbasr_lstm_pack_whh(const float* w_hh, void* packed, int hidden, nbasr_stream_t stream);
int nbasr_lstm_recurrence_packed(const float* gates_ws, const void* packed_whh, float* cell_ws, float* h_out,
                                 int batch, int frames, int hidden, nbasr_stream_t stream);
/* The same recurrence, all frames in ONE launch (reference model.py:100,118-121: the LSTM's time loop): the grid of the per-frame
 * kernel stays resident, every workgroup keeps its slice of w_hh in registers and its cell state in registers, h_t is exchanged
 * through a double-buffered image in `seq_ws` whose 16-byte granules carry their own step tag (bit 30 of every fp32 -- free because
 * |h| <= 1 -- written write-through, polled with L1-bypassing loads; no flags, no drains); utterance tiles never synchronise with each
 * other.  Bit-identical h_out to nbasr_lstm_recurrence_packed.  For the single forward (latency); a pipelined caller whose next
 * encoder needs the CUs keeps the per-frame launches.  nbasr_lstm_seq_workspace_bytes returns 0 where the form does not apply
 * (hidden > 512, or more workgroups = ceil(hidden / 8) * ceil(batch / 16) than the CURRENT device holds at once -- its compute units x
 * the kernel's occupancy, at most 256: every workgroup must be resident).  The launch is cooperative where the device offers it
 * (ABI 4): a grid the device cannot hold at once is refused with an error instead of started in part.
 * Every wait is bounded (1 s): on a timeout (compute units taken by another process) the kernel raises the status word -- the first
 * 32-bit word of seq_ws -- and fills the rest of h_out with NaN; nbasr_lstm_seq_status (synchronises `stream`) returns NBASR_EINVAL
 * then, and a caller that must not block copies that word back asynchronously and looks at it later (what nb_asr_amd's executor does
 * behind every launch: a timed-out forward raises at the next call, and the one-launch form is switched off for the plan).
 * Launches from different streams of one process are chained by an event; a stream under graph capture is refused (use the
 * per-frame form there).  flags: 0, or NBASR_LSTM_SEQ_INJECT_FAULT (tests: one workgroup per tile never starts). */
#define NBASR_LSTM_SEQ_INJECT_FAULT 1
size_t nbasr_lstm_seq_workspace_bytes(int batch, int hidden);
int nbasr_lstm_recurrence_seq(const float* gates_ws, const void* packed_whh, float* cell_ws, float* h_out, void* seq_ws,
                              int batch, int frames, int hidden, int flags, nbasr_stream_t stream);
int nbasr_lstm_seq_status(const void* seq_ws, nbasr_stream_t stream);
/* ABI 6 -- the recurrence as ONE resident launch whose per-frame exchange stays inside an XCD, on the 16-bit matrix cores (reference
 * model.py:100,118-121; csrc/lstm_xcd.hip).  A tile of 16 utterances is owned by the ceil(hidden / 16) workgroups of ONE XCD (16 hidden
 * units x 4 gates each, w_hh resident in registers as two fp16 terms per weight, row-independent power-of-two scale): h_t is published
 * with plain stores and polled with L1-bypassing loads, both served by that XCD's L2 -- one L2 round trip per frame instead of two
 * fabric trips.  A workgroup reads its XCD from the hardware id and takes its slice from a per-XCD arrival ticket; the first arrival of
 * an XCD that has collected all its slices claims whole tiles from a global ticket; XCDs that never complete claim nothing, surplus
 * workgroups leave at once: correctness does not depend on how the dispatcher places workgroups.  The product w_hh . h_(t-1) is
 * fp32-accurate (3 v_mfma_f32_16x16x32_f16 per 32 k: hi*hi + hi*lo + lo*hi, fp32 accumulation; operand error 2^-23 relative, as in
 * nbasr_dense_conv1d_packed's NBASR_DENSE_F16X2): h_out agrees with nbasr_lstm_recurrence_packed to fp32 round-off, not bit for bit.
 * packed_whh16: nbasr_lstm_packed_whh16_bytes / nbasr_lstm_pack_whh16 (once per weight version; hidden <= 512, else 0 bytes / EINVAL).
 * xcd_ws: nbasr_lstm_xcd_workspace_bytes (0 where the form does not apply: hidden > 512 or batch > 4096), zeroed by the call.
 * Every wait is bounded (1 s); the status word -- the first 32-bit word of xcd_ws, as for nbasr_lstm_recurrence_seq, read by
 * nbasr_lstm_seq_status -- is 1 after a timeout (that slice's remaining h rows are NaN) and 2 when a tile was never computed.
 * Launches from different streams of one process are chained by an event (more than two such grids at once could split an XCD's
 * units among them so that none collects its slices); a launch on a stream under capture becomes plain graph nodes outside that chain.
 * flags: 0, or NBASR_LSTM_SEQ_INJECT_FAULT (tests: one slice of every XCD never starts). */
size_t nbasr_lstm_packed_whh16_bytes(int hidden);
int nbasr_lstm_pack_whh16(const float* w_hh, void* packed, int hidden, nbasr_stream_t stream);
size_t nbasr_lstm_xcd_workspace_bytes(int batch, int hidden);
int nbasr_lstm_recurrence_xcd(const float* gates_ws, const void* packed_whh16, float* cell_ws, float* h_out, void* xcd_ws,
                              int batch, int frames, int hidden, int flags, nbasr_stream_t stream);
/* The same arithmetic as ONE LAUNCH PER FRAME (bit-identical h_out to nbasr_lstm_recurrence_xcd): what a caller runs where a resident
 * grid is unwelcome -- beside other work that needs every XCD (a pipelined tail next to the following batch's encoder) -- and what a
 * plan falls back to after a failed status word.  Same packed weights, same workspace (only its images are used; no status word to
 * read); the chain of `frames` launches is replayed as one cached graph where the call recurs, like nbasr_lstm_recurrence_packed. */
int nbasr_lstm_recurrence_frames16(const float* gates_ws, const void* packed_whh16, float* cell_ws, float* h_out, void* xcd_ws,
                                   int batch, int frames, int hidden, nbasr_stream_t stream);

/* CTC head nn.Linear(features -> classes) (reference model.py:101 / 122-124):
 * logits(rows, classes) = h(rows, features) . w(classes, features)^T + bias. */
int nbasr_linear_head(const float* h, const float* w, const float* bias, float* logits,
                      int rows, int features, int classes, nbasr_stream_t stream);

/* Head for the use_rnn=False model (reference model.py:103): input is the encoder output
 * x (batch, features, ld), `ln` its pending LayerNorm; logits (batch, frames, classes). */
int nbasr_linear_head_bct(const float* x, const float* w, const float* bias, float* logits,
                          int batch, int features, int frames, int ld, int classes,
                          const nbasr_deferred_ln* ln, nbasr_stream_t stream);

/* ---- deferred LayerNorm ("normalise on load") ---------------------------------------------------------------
 * Instead of materialising LayerNorm(x) (2 reads + 1 write of the tensor), nbasr_channel_stats makes ONE read pass
 * and stores per-frame statistics (stats: (batch, 2, ld) floats, see nbasr_deferred_ln); every consumer of the normalised
 * tensor then applies
 *     xn[b][c][t] = (x[b][c][t] - mean[b][t]) * rstd[b][t] * gamma[c] + beta[c]
 * while loading (zero padding and pitch columns stay exactly zero).  Same arithmetic as nbasr_layernorm_channels
 * (reference model.py:92 + 125-128, model.py:46-47 + 55-58).  Entry points that take `ln` also take flags saying which operand
 * carries it: the main input x and/or skip0 (inside a cell only the cell input is a LayerNorm output, and it is always skip0
 * when it is a skip). */
int nbasr_channel_stats(const void* x, float* stats, int batch, int channels, int frames, int ld, float eps, int dtype,
                        nbasr_stream_t stream);

/* LayerNorm statistics of a node's output from the convolution's own epilogue (what nbasr_channel_stats(y) would give):
 * workgroups write per-part partial (mean, M2) to stats_ws (nbasr_grouped_stats_workspace_bytes), nbasr_grouped_stats_finalize
 * merges them into stats_out (batch, 2, ld).  A part covers `groups_per_part` groups: 4 (nbasr_grouped_conv1d_node,
 * nbasr_grouped_cell_fused on rows of one wave), 2 or 1 (the value nbasr_grouped_cell_fits returns for the shape), or -- for the partials
 * of nbasr_dense_conv1d_packed(stats_part), whose "groups" are single channels -- NBASR_DENSE_STATS_UNIT; <= 128 parts either way. */
size_t nbasr_grouped_stats_workspace_bytes(int batch, int ld, int groups);
int nbasr_grouped_stats_finalize(const float* stats_ws, float* stats_out, int batch, int channels, int frames, int ld,
                                 int groups, int groups_per_part, float eps, nbasr_stream_t stream);

/* Node operation, grouped-convolution flavour, fused with the node's skip-sum
 * (reference ops.py:24-30 with groups=100 from the table ops.py:73-76, Node.forward model.py:13-22):
 *     y = min(relu(conv1d(zero_pad(x), w, bias, dilation, groups)), 20) + skip0 + skip1 + skip2
 * x, y, skip*: (batch, channels, ld) `dtype` tensors; w: (channels, channels/groups, kernel); bias: (channels).
 * NULL skips are absent (a Zero branch).  Supported: kernel in {5,7}, dilation in {1,2},
 * channels/groups in {6,8,10,12}.  Skips are added left to right (python `sum` order).  `ln` / ln_on_x / ln_on_skip0: pending
 * LayerNorm of x and / or skip0.  stats_ws (may be NULL): also emit the partial LayerNorm statistics of y (see above).
 * variant: 0 = 4 frames per lane, weights as torch stores them; NBASR_GC_FPL8 = 8 frames per lane (fp32: ld % 8 == 0);
 * NBASR_GC_WPERM = `w` is the [group][ci][tap][co] copy made by nbasr_pack_grouped_weights (channels * channels/groups *
 * kernel floats), whose per-input-channel weights are contiguous for the scalar loads.  Results do not depend on the variant. */
#define NBASR_GC_FPL8 1
#define NBASR_GC_WPERM 2
#define NBASR_GC_OSPLIT 4       /* fp32, on its own, no stats_ws: a wave owns half of a group's output channels (short rows / small batches) */
#define NBASR_GC_PIPE 8         /* fp32, alone (stats_ws allowed) or with NBASR_GC_OSPLIT (no stats_ws): software-pipelined window loads (buffer loads, zero fill by the bounds check) */
#define NBASR_GC_RING 16       /* fp32, alone (stats_ws allowed): input windows staged through LDS by LDS-DMA -- a wave requests all input rows of its tile up front, consumes them behind counted waits (grouped_conv_ring.hip) */
int nbasr_grouped_conv1d_node(const void* x, const float* w, const float* bias,
                              const void* skip0, const void* skip1, const void* skip2, void* y,
                              int batch, int channels, int frames, int ld, int groups, int kernel, int dilation,
                              const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, float* stats_ws,
                              int dtype, int variant, nbasr_stream_t stream);
int nbasr_pack_grouped_weights(const float* w, float* packed, int channels, int groups, int kernel, nbasr_stream_t stream);

/* A whole SearchCell whose three node operations are grouped convolutions, in one launch (reference model.py:49-59 over
 * model.py:13-22 and ops.py:24-30): x1 = op0(x0) + s00 x0; x2 = op1(x1) + s10 x0 + s11 x1; x3 = op2(x2) + s20 x0 + s21 x1 + s22 x2.
 * The intermediates never leave the compute unit (registers + one LDS tile per group); the result is bit-identical to three
 * nbasr_grouped_conv1d_node launches.  skip_mask: bit0 s00 | bit1 s10 | bit2 s11 | bit3 s20 | bit4 s21 | bit5 s22.  `ln` (may be
 * NULL): pending LayerNorm of x0.  `stats_ws` (may be NULL; nbasr_grouped_stats_workspace_bytes): the launch also emits the partial
 * LayerNorm statistics of x3, exactly as nbasr_grouped_conv1d_node does for a cell's last node (merge: nbasr_grouped_stats_finalize).
 * `dtype` = NBASR_F32 | NBASR_BF16: storage type of x0 and y (weights, biases, statistics, gamma / beta are fp32 either way; with
 * bf16 storage x1 and x2 are rounded to bfloat16 exactly where the three-launch form stores them).
 * w0, w1, w2 (ABI 4): the [group][ci][tap][co] copies made by nbasr_pack_grouped_weights -- a packed FMA of the kernel covers two
 * output channels of one frame, its weight operand is a scalar-register pair loaded from two adjacent floats.
 * nbasr_grouped_cell_fits tells whether a (channels, ld, groups) row fits one workgroup -- <= 2048 frames (<= 8 waves per group row),
 * channels / groups in {6, 8, 10, 12}, the group tiles within 160 KiB of LDS: 0 = no, else the number of groups one statistics partial
 * covers (4, 2 or 1: the groups_per_part of nbasr_grouped_stats_finalize; 1 = one group row per workgroup, rows of more than 256 frames). */
int nbasr_grouped_cell_fits(int channels, int frames_ld, int groups);
int nbasr_grouped_cell_fused(const void* x0, const float* w0, const float* b0, int k0, int d0,
                             const float* w1, const float* b1, int k1, int d1,
                             const float* w2, const float* b2, int k2, int d2, int skip_mask, void* y,
                             int batch, int channels, int frames, int ld, int groups,
                             const nbasr_deferred_ln* ln, float* stats_ws, int dtype, nbasr_stream_t stream);
/* The same cell for the bf16 storage path on the MATRIX cores (grouped_cell_mfma.hip): every tensor of the bf16 model is a bfloat16
 * tensor, so the products go to v_mfma_f32_16x16x32_bf16 unchanged -- exact bf16 x bf16 products, fp32 accumulation: the reference's
 * arithmetic up to the order of the sums (NOT bit-identical to the vector-ALU kernels; within an fp32 rounding of them before the one
 * bf16 rounding per tensor).  x0, y: bf16 (batch, channels, ld), ld % 8 == 0; the normalised cell input is rounded to bf16 like every
 * other tensor (the reference's LayerNorm output is one).  Weights: the fp32 values of the bf16 parameters, re-laid-out once per
 * weight version as MFMA fragments (nbasr_grouped_cell_mfma_weights_bytes / nbasr_grouped_cell_mfma_pack, one image per node).
 * No statistics by-product.  nbasr_grouped_cell_mfma_fits: 0 = the row does not fit a workgroup (two bf16 tiles per group within
 * 160 KiB of LDS), else the groups per workgroup of the tiling a launch of 32 utterances takes (a launch picks its tiling -- frames
 * per wave, groups per workgroup -- from the sizes it is given, the batch included). */
size_t nbasr_grouped_cell_mfma_weights_bytes(int channels, int groups, int kernel);
int nbasr_grouped_cell_mfma_pack(const float* w, void* packed, int channels, int groups, int kernel, nbasr_stream_t stream);
int nbasr_grouped_cell_mfma_fits(int channels, int frames_ld, int groups);
int nbasr_grouped_cell_mfma(const void* x0, const void* wp0, const float* b0, int k0, int d0,
                            const void* wp1, const float* b1, int k1, int d1,
                            const void* wp2, const float* b2, int k2, int d2, int skip_mask, void* y,
                            int batch, int channels, int frames, int ld, int groups,
                            const nbasr_deferred_ln* ln, nbasr_stream_t stream);

/* Node whose main op is `zero` (reference ops.py:67-68): y = 0 + skip0 + skip1 + skip2 (`dtype` tensors; `ln`: pending LayerNorm of
 * skip0 when ln_on_skip0). */
int nbasr_skip_sum(const void* skip0, const void* skip1, const void* skip2, void* y, int batch, int channels, int frames,
                   int ld, const nbasr_deferred_ln* ln, int ln_on_skip0, int dtype, nbasr_stream_t stream);

/* Post-logits step of the reference's trainer (training/torch/trainer.py:217-219, 229-247), SURVEY.md 8 row f2:
 *   log_probs(batch, frames, classes) = log_softmax(logits, classes)                      (NULL: not wanted)
 *   greedy CTC decoding: per-frame argmax over the first lengths[b] frames (lengths NULL: all frames; the trainer uses
 *   audio_len // 4), repeats collapsed, `blank` dropped (F.ctc_loss default blank = 0) -> tokens(batch, frames) int32,
 *   padded with -1, token_counts(batch).  (tokens NULL: not wanted.)  The beam search the reference decodes with is
 *   nbasr_ctc_beam_search below. */
int nbasr_ctc_postprocess(const float* logits, const int* lengths, float* log_probs, int* tokens, int* token_counts,
                          int batch, int frames, int classes, int blank, nbasr_stream_t stream);

/* ---- validation decode (replaces reference training/torch/trainer.py:229-247 Trainer.decode) ----------
 * nbasr_ctc_beam_search: CTC prefix beam search without a language model, the algorithm of the reference's decoder
 *   CTCBeamDecoder(vocab, beam_width=12, log_probs_input=True) (ctcdecode, third-party C++, trainer.py:71,237; its defaults
 *   cutoff_top_n = 40, cutoff_prob = 1.0, blank_id = 0).  log_probs(batch, frames, classes): log-probabilities
 *   (nbasr_ctc_postprocess's log_probs); lengths(batch) or NULL: frames of each utterance that count.  Outputs, best beam
 *   first: beams(batch, beam_width, frames) int32 token sequences padded with 0, scores(batch, beam_width) = -log P of each
 *   beam (ctcdecode's convention; FLT_MAX for beams that do not exist), beam_lens(batch, beam_width).  ws: scratch of
 *   nbasr_ctc_beam_workspace_bytes (8-byte aligned, no state between calls).  Limits: classes <= 64, beam_width <= 32.
 *   Log-probabilities <= -FLT_MAX (ctcdecode's "-infinity") count as pruned classes.
 * nbasr_token_error_counts: per utterance, map both label sequences through `table` (n_table entries; NULL/0: identity --
 *   the reference folds 48 -> 39 phonemes with PhonemeEncoder.fold_encoded, encoder.py:64-75), drop `blank`, and compute the
 *   Levenshtein distance: counts(batch, 2) int32 = (distance, reference length after blank removal); the error rate of
 *   torch_edit_distance.compute_wer is distance / length.  hyp(batch, ld_hyp), ref(batch, ld_ref) int32 with their lengths.
 *   counts = (-1, -1) for an utterance with more than 2048 tokens after blank removal, (-1, -2) for a label outside the table. */
/* nbasr_ctc_loss: the reference's loss value (training/torch/trainer.py:36-42, reported by its validation loop):
 *   F.ctc_loss(log_probs, targets, output_len, targets_len, reduction='none', zero_infinity=True) [/ output_len when
 *   divide_by_length != 0]; the mean over the batch is left to the caller.  log_probs(batch, frames, classes) batch-major (the
 *   reference permutes to (T, B, C) for torch; no permute here), lengths(batch) valid frames, targets(batch, ld_targets) int32
 *   labels with target_lengths(batch) (ld_targets <= 1024), losses(batch) out.  Forward value only (no gradient: f4).
 *   A label outside [0, classes) gives NaN for that utterance. */
int nbasr_ctc_loss(const float* log_probs, const int* lengths, const int* targets, const int* target_lengths, float* losses,
                   int batch, int frames, int classes, int ld_targets, int blank, int divide_by_length, nbasr_stream_t stream);
/* nbasr_ctc_loss_grad: the loss of nbasr_ctc_loss (divided by the lengths) AND the gradient of its batch mean with respect to the
 *   LOGITS that log_probs = log_softmax(logits) came from -- what the reference's `loss.backward()` (trainer.py:220-223, without
 *   the weight-norm term) hands to the model: grad_logits(batch, frames, classes), zero beyond each utterance's length and for
 *   utterances whose loss is infinite (zero_infinity).  The first step of the backward pass; nothing in
 *   this library consumes it yet.  ws: nbasr_ctc_grad_workspace_bytes (alpha of every frame), no state between calls. */
size_t nbasr_ctc_grad_workspace_bytes(int batch, int frames, int ld_targets);
int nbasr_ctc_loss_grad(const float* log_probs, const int* lengths, const int* targets, const int* target_lengths, void* ws,
                        float* losses, float* grad_logits, int batch, int frames, int classes, int ld_targets, int blank,
                        nbasr_stream_t stream);
size_t nbasr_ctc_beam_workspace_bytes(int batch, int frames, int classes, int beam_width);
int nbasr_ctc_beam_search(const float* log_probs, const int* lengths, void* ws, int* beams, float* scores, int* beam_lens,
                          int batch, int frames, int classes, int beam_width, int blank, int cutoff_top_n,
                          nbasr_stream_t stream);
int nbasr_token_error_counts(const int* hyp, const int* hyp_len, int ld_hyp, const int* ref, const int* ref_len, int ld_ref,
                             const int* table, int n_table, int blank, int* counts, int batch, nbasr_stream_t stream);

/* Copy (batch, channels, frames) `dtype` rows with pitch ld_src into pitch ld_dst, zero-filling columns
 * frames..ld_dst-1 (used to bring caller tensors into the pitched internal layout). */
int nbasr_repitch(const void* src, void* dst, int rows, int frames, int ld_src, int ld_dst, int dtype, nbasr_stream_t stream);

/* ---- per-frame linear maps on the fp16 matrix cores (fp32-accurate two-way operand split, see NBASR_DENSE_F16X2) ------
 * The `linear` node op (reference ops.py:42-50 + the node's skip sum, model.py:13-22) and the LSTM input projection.  The
 * activation is split once by a streaming pre-pass into `ws` (nbasr_pointwise_workspace_bytes; scratch, no state between
 * calls) with one exact power-of-two scale per (utterance, 256-frame tile); the weights are packed once per weight version
 * (nbasr_pointwise_packed_weights_bytes / nbasr_pack_pointwise_weights, row-wise power-of-two scales).  No range contract:
 * both scalings are computed from the data.  Arguments otherwise as nbasr_dense_conv1d_fused with kernel = 1 and as
 * nbasr_lstm_input_projection. */
size_t nbasr_pointwise_packed_weights_bytes(int c_out, int c_in);
size_t nbasr_pointwise_workspace_bytes(int batch, int c_in, int ld);
int nbasr_pack_pointwise_weights(const float* w, void* packed, int c_out, int c_in, nbasr_stream_t stream);
int nbasr_linear_fused_packed(const float* x, void* ws, const void* packed_w, const float* bias,
                              const float* skip0, const float* skip1, const float* skip2, float* y,
                              int batch, int channels_in, int frames, int ld, int channels_out,
                              const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, nbasr_stream_t stream);
int nbasr_lstm_input_projection_packed(const float* x, void* ws, const void* packed_w_ih, const float* b_ih,
                                       const float* b_hh, float* gates_ws, int batch, int c_in, int frames, int ld,
                                       int hidden, const nbasr_deferred_ln* ln, nbasr_stream_t stream);
/* ABI 6 -- the same projection into the rows of a LARGER gate tensor: gates_ws is (frames, batch_total, 4 * hidden) and this call
 * writes utterances batch_offset .. batch_offset + batch - 1 of it.  For a host that runs ONE recurrence over the gates of several
 * forwards (nb_asr_amd: `ASRModel.forward_many(tail_group=...)` -- at 8 utterances a frame of the recurrence costs what it costs at 32,
 * and no utterance's result depends on the batch it is computed in). */
int nbasr_lstm_input_projection_packed_into(const float* x, void* ws, const void* packed_w_ih, const float* b_ih,
                                            const float* b_hh, float* gates_ws, int batch, int c_in, int frames, int ld,
                                            int hidden, const nbasr_deferred_ln* ln, int batch_total, int batch_offset,
                                            nbasr_stream_t stream);
/* The same two maps for the bf16 storage path (reference ops.py:42-50 and model.py:100,118-121 under model.to(torch.bfloat16)) on the
 * bf16 matrix cores, ONE v_mfma_f32_16x16x32_bf16 per 32 products (round 4): x, y, skips are bf16 (batch, channels, ld) rows, ld % 8
 * == 0; the weights are the fp32 values of the bf16 parameter, packed once per version (nbasr_pointwise_bf16_weights_bytes /
 * nbasr_pack_pointwise_weights_bf16); `ws` (nbasr_pointwise_bf16_workspace_bytes; scratch) takes the operand image -- x, with its
 * pending LayerNorm applied and rounded to bf16 as the reference's LayerNorm module rounds, in the GEMM's LDS order.  Exact bf16 x bf16
 * products, fp32 accumulation, bias, relu / min(20), the skips added in fp32 in python's sum order, ONE rounding of the node's output;
 * the LSTM gates stay fp32, (frames, batch, 4 hidden). */
size_t nbasr_pointwise_bf16_weights_bytes(int c_out, int c_in);
size_t nbasr_pointwise_bf16_workspace_bytes(int batch, int c_in, int ld);
int nbasr_pack_pointwise_weights_bf16(const float* w, void* packed, int c_out, int c_in, nbasr_stream_t stream);
int nbasr_linear_fused_bf16(const void* x, void* ws, const void* packed_w, const float* bias,
                            const void* skip0, const void* skip1, const void* skip2, void* y,
                            int batch, int channels_in, int frames, int ld, int channels_out,
                            const nbasr_deferred_ln* ln, int ln_on_x, int ln_on_skip0, nbasr_stream_t stream);
int nbasr_lstm_input_projection_bf16(const void* x, void* ws, const void* packed_w_ih, const float* b_ih,
                                     const float* b_hh, float* gates_ws, int batch, int c_in, int frames, int ld,
                                     int hidden, const nbasr_deferred_ln* ln, nbasr_stream_t stream);

/* ---- feature front-end (reference training/torch/timit.py:78-97) --------------------------------------
 * torchaudio MelSpectrogram(16 kHz, n_fft = win = 400, hop 160, 80 mels, power 2, centred reflect-padded frames, periodic
 * Hann window, HTK mel scale) -> log -> (x - mean) / (variance + eps), produced directly in the model's input layout
 * (batch, 80, frames).  The DFT and the mel filterbank are plain matrices applied per frame (nbasr_pointwise_linear); the
 * host side (nb_asr_amd/frontend.py) builds them.  `lengths` (may be NULL = every utterance has `samples` samples): int32
 * device array of per-utterance sample counts; an utterance has lengths[b] / hop + 1 frames, later frames are zero.
 *   frames(batch, win, ld_frames)[b][k][t] = wave[b][reflect(t*hop + k - win/2)]
 *   y(batch, c_out, ld_out) = w(c_out, c_in) . x(batch, c_in, ld_in) + bias          (no activation; c_in % 4 == 0)
 *   power(batch, rows_out, ld)[b][f][t] = spec[b][f][t]^2 + spec[b][bins + f][t]^2   (rows bins..rows_out-1 zero)
 *   feats(batch, n_mels, ld)[b][m][t] = (log(mel[b][m][t]) - mean[m]) * inv_scale[m] for t < frames(b), else 0 */
int nbasr_frame_signal(const float* wave, const int* lengths, float* frames, int batch, int samples, int ld_wave,
                       int win, int hop, int ld_frames, nbasr_stream_t stream);
int nbasr_pointwise_linear(const float* x, const float* w, const float* bias, float* y, int batch, int c_in,
                           int frames, int ld_in, int c_out, int ld_out, nbasr_stream_t stream);
int nbasr_power_spectrum(const float* spec, float* power, int batch, int bins, int rows_out, int ld,
                         nbasr_stream_t stream);
int nbasr_log_normalize(const float* mel, const int* lengths, const float* mean, const float* inv_scale, float* feats,
                        int batch, int samples, int hop, int n_mels, int ld, nbasr_stream_t stream);

/* ---- bf16 path (BASELINE config 4: activations and GEMM operands stored as bfloat16) ------------------------------------------
 * Mirrors `model.to(torch.bfloat16)(x.bfloat16())` of the reference (ops.py:24-30, model.py:116-131 run on bf16 tensors).
 * Storage is bf16, arithmetic is fp32, a tensor is rounded once when it is written: the entry points above that take a `dtype`
 * serve both storage types (x / y / skips are `dtype` tensors, everything else -- weights of the node op, bias, gamma, beta,
 * statistics -- stays fp32).  bf16 rows: ld % 8 == 0, 16-byte aligned, pitch columns zero. */
/* Element-wise conversion between the two storage types (n % 8 == 0, both pointers 16-byte aligned): the bridge to the
 * operators that exist in fp32 only. */
int nbasr_convert(const void* x, void* y, long long n, int in_dtype, int out_dtype, nbasr_stream_t stream);
/* Operand image of the bf16 dense convolution: [batch][16-channel group][8-channel half][1 + ld rows][8 ch] bfloat16, row 0
 * zero, frame t at row t + 1 (nbasr_bf16_image_bytes bytes).  gamma != NULL: image of LayerNorm(x) -- `stats`
 * (batch, 2, ld) receives the per-frame statistics on the way; gamma == NULL: plain re-layout of x (the model input,
 * cells without LayerNorm).  x is a `dtype` tensor (batch, channels, ld). */
size_t nbasr_bf16_image_bytes(int batch, int channels, int ld);
int nbasr_bf16_image(const void* x, const float* gamma, const float* beta, float* stats, void* image, int batch, int channels,
                     int frames, int ld, float eps, int dtype, nbasr_stream_t stream);
/* (the bf16 convolution itself: nbasr_dense_conv1d_packed with NBASR_DENSE_BF16) */

/* ---- backward building blocks (bottom-up; fp32) -------------------------------------------------------------------
 * The node op z = min(relu(conv1d(zero_pad(x), w, b, dilation, groups)), 20) (reference ops.py:24-30) given dz = dL/dz:
 *   the relu / clamp_max_ masks are taken from the op's OUTPUT z (0 < z < 20), so no pre-activation has to be kept;
 *   dx (batch, channels, ld)  -- may be NULL;   dw (channels, channels/groups, kernel) and db (channels) -- both or neither, they need
 *   `workspace` (nbasr_grouped_conv1d_backward_workspace_bytes) and x.  The weight gradient is accumulated per utterance on
 *   the fp32 matrix cores and summed over utterances in a fixed order: no atomics, bit-reproducible.  dz's pitch columns must be 0.
 * LayerNorm over channels (reference model.py:46-47, 55-58, 92) given dy: `stats` = the (batch, 2, ld) statistics of x
 * (nbasr_channel_stats); dx, dgamma (channels), dbeta (channels); workspace of nbasr_layernorm_backward_workspace_bytes. */
size_t nbasr_grouped_conv1d_backward_workspace_bytes(int batch, int channels, int groups, int kernel);
int nbasr_grouped_conv1d_backward(const float* x, const float* w, const float* z, const float* dz, float* dx, float* dw, float* db,
                                  float* workspace, int batch, int channels, int frames, int ld, int groups, int kernel, int dilation,
                                  nbasr_stream_t stream);
size_t nbasr_layernorm_backward_workspace_bytes(int batch, int channels, int ld);
int nbasr_layernorm_channels_backward(const float* x, const float* stats, const float* gamma, const float* dy, float* dx,
                                      float* dgamma, float* dbeta, float* workspace, int batch, int channels, int frames, int ld,
                                      nbasr_stream_t stream);

/* Building blocks of the GEMM-shaped backward passes (dense downsample convs, `linear` node op; SURVEY.md 8 row f4, bottom-up).  The
 * products run on the exact-fp32 MFMA GEMMs of the forward (nbasr_pointwise_linear, nbasr_dense_conv1d_linear); these put the
 * operands into the layouts those GEMMs read:
 *   nbasr_relu_clamp_backward   dz = dy where 0 < y < 20, else 0 (n floats, n % 4 == 0)
 *   nbasr_zero_stuff            up[r][shift + t * stride] = dz[r][t], zeros elsewhere (rows x frames_up, pitch ld_up): the input
 *                               gradient of a strided conv is a stride-1 conv of this with the flipped, channel-transposed kernel
 *   nbasr_dense_conv1d_linear   that conv: k = 8, stride 1, caller-chosen left padding, no activation
 *   nbasr_conv_cols             cols[b * t_pad + t][ci * taps + j] = xpad[b][ci][t * stride + j - lpad], one column of ones behind
 *                               them (then zeros up to ld_cols >= c_in * taps + 1), rows t >= frames_out zero
 *   nbasr_rows_of_channels      rows[co][b * t_pad + t] = dz[b][co][t] (0 for t >= frames)
 * so that  nbasr_pointwise_linear(x = cols as (1, batch * t_pad, ld_cols), w = rows)  yields (dw | db) in one GEMM.
 *   nbasr_conv_fold             the input gradient again, for the split 16-bit GEMM: cols (frames_out, batch, c_in * 8), time-major (what
 *                               nbasr_lstm_input_projection_packed stores for w^T as (c_in * 8, c_out) rows (ci, tap) and x = the masked
 *                               output gradient) -> dx[b][ci][u] = sum over taps j with u + lpad - j = t * stride of cols[t][b][ci * 8 + j];
 *                               8 taps, stride 1 | 2; pitch columns of dx are zeroed */
int nbasr_relu_clamp_backward(const float* y, const float* dy, float* dz, long long n, nbasr_stream_t stream);
int nbasr_zero_stuff(const float* dz, float* up, int rows, int frames, int ld, int frames_up, int ld_up, int stride, int shift,
                     nbasr_stream_t stream);
int nbasr_dense_conv1d_linear(const float* x, const float* w, const float* bias, float* y, int batch, int c_in, int frames, int ld_in,
                              int c_out, int ld_out, int kernel, int lpad, nbasr_stream_t stream);
int nbasr_conv_cols(const float* x, float* cols, int batch, int c_in, int frames_in, int ld_in, int frames_out, int t_pad, int taps,
                    int stride, int lpad, int ld_cols, nbasr_stream_t stream);
int nbasr_rows_of_channels(const float* dz, float* rows, int batch, int channels, int frames, int ld, int t_pad, nbasr_stream_t stream);
int nbasr_conv_fold(const float* cols, float* dx, int batch, int c_in, int frames_in, int ld_in, int frames_out, int taps, int stride,
                    int lpad, nbasr_stream_t stream);

/* LSTM backward (BPTT; reference model.py:100,118-121 under autograd), correctness first.  Tensors are (rows, frames, ldb) with the
 * utterances innermost (ldb = batch rounded up to 4), rows = 4 * hidden (PyTorch gate order i, f, g, o) or hidden:
 *   nbasr_lstm_gate_scan       pre (4H, T, ldb) = input projection + w_hh . h_(t-1) for ALL frames (one GEMM on the saved h) is
 *                              overwritten by the gate activations, cells (H, T, ldb) <- c_t
 *   nbasr_lstm_backward_step   frame t of the reverse recurrence: dh_out (H, T, ldb) = dL/d(output), w_hh_t (H, 4H) the transposed recurrent
 *                              weight (the term w_hh^T . dpre[:, t+1, :] is formed in the kernel), dc (H, ldb) the carried dL/dc_t ->
 *                              dpre[:, t, :] and the new carry; call for t = T-1 .. 0, or ONCE with t = -1: all frames T-1 .. 0 as one
 *                              chain of launches (ABI 6), replayed as a cached graph where the call recurs with the same buffers
 * The GEMMs in between (nbasr_pointwise_linear) and the orchestration are in nb_asr_amd/autograd.py. */
int nbasr_lstm_gate_scan(float* pre, float* cells, int hidden, int frames, int batch, int ldb, nbasr_stream_t stream);
int nbasr_lstm_backward_step(const float* dh_out, const float* w_hh_t, float* dc, const float* acts, const float* cells, float* dpre,
                             int hidden, int frames, int batch, int ldb, int t, nbasr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NBASR_H */
