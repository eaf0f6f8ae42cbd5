// Feature front-end of the reference's TIMIT pipeline (reference training/torch/timit.py:78-97):
//   torchaudio MelSpectrogram(sample_rate 16 kHz, n_fft = win_length = 400, hop 160, 80 mel bands, power 2, centred frames
//   with reflect padding, periodic Hann window, HTK mel scale, no filterbank normalisation)  ->  log  ->
//   (x - mean) / (variance + eps)      [the torch trainer divides by the VARIANCE, not its square root: timit.py:83]
// as four launches whose output is the model's input layout (batch, 80, frames), frames contiguous:
//   1. nbasr_frame_signal:   waveform -> frames (batch, 400, T) [sample-in-window major, frames contiguous]
//   2. nbasr_pointwise_linear (gemm_conv.hip, exact-fp32 MFMA GEMM): DFT as a 402 x 400 matrix (cos rows, then sin rows; the
//      window is folded into the matrix on the host)            -> (batch, 402, T)
//   3. nbasr_power_spectrum: re^2 + im^2                         -> (batch, 204, T)   (201 bins + 3 zero rows: K % 4 == 0)
//   4. nbasr_pointwise_linear: mel filterbank 80 x 204           -> (batch, 80, T)
//   5. nbasr_log_normalize:  log, shift, scale; frames beyond an utterance's own length are set to 0 (the reference pads
//      the FEATURES of a batch with zeros, timit.py:54-69)
// The DFT is a GEMM (20 GFLOP for 64 utterances of 10 s) rather than an FFT: at n_fft = 400 the matrix cores finish it in
// ~0.2 ms, less than a radix-mixed FFT's passes over the same data would take on HBM.
#include "common.h"

namespace nbasr {

// frames[b][k][t] = wave[b][reflect(t * hop + k - win / 2)], 0 for t >= n_frames(b); lanes along t
__global__ __launch_bounds__(256) void frame_signal_kernel(const float* __restrict__ wave, const int* __restrict__ lengths,
                                                           float* __restrict__ frames, int samples, int ld_wave, int win,
                                                           int hop, int ld_frames)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (t >= ld_frames) return;
    const int len = lengths ? lengths[b] : samples;
    const int n_frames = len > 0 ? len / hop + 1 : 0;
    const float* __restrict__ wb = wave + static_cast<size_t>(b) * ld_wave;
    for (int k = blockIdx.y; k < win; k += gridDim.y) {
        float v = 0.f;
        if (t < n_frames) {
            int i = t * hop + k - win / 2;
            if (i < 0) i = -i;
            if (i >= len) i = 2 * (len - 1) - i;
            v = wb[i];
        }
        frames[(static_cast<size_t>(b) * win + k) * ld_frames + t] = v;
    }
}

// power[b][f][t] = spec[b][f][t]^2 + spec[b][bins + f][t]^2 for f < bins, 0 for bins <= f < rows_out
__global__ __launch_bounds__(256) void power_spectrum_kernel(const float* __restrict__ spec, float* __restrict__ power,
                                                             int bins, int rows_out, int nq)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (q >= nq) return;
    const float4* __restrict__ sb = reinterpret_cast<const float4*>(spec) + static_cast<size_t>(b) * 2 * bins * nq;
    float4* __restrict__ pb = reinterpret_cast<float4*>(power) + static_cast<size_t>(b) * rows_out * nq;
    for (int f = blockIdx.y; f < rows_out; f += gridDim.y) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < bins) {
            const float4 re = sb[static_cast<size_t>(f) * nq + q], im = sb[static_cast<size_t>(bins + f) * nq + q];
            o = make_float4(re.x * re.x + im.x * im.x, re.y * re.y + im.y * im.y, re.z * re.z + im.z * im.z, re.w * re.w + im.w * im.w);
        }
        pb[static_cast<size_t>(f) * nq + q] = o;
    }
}

// feats[b][m][t] = (log(mel[b][m][t]) - mean[m]) * inv_scale[m] for t < n_frames(b), else 0 (in place allowed)
__global__ __launch_bounds__(256) void log_normalize_kernel(const float* mel, const int* __restrict__ lengths,
                                                            const float* __restrict__ mean, const float* __restrict__ inv_scale,
                                                            float* feats, int samples, int hop, int n_mels, int ld)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (t >= ld) return;
    const int len = lengths ? lengths[b] : samples;
    const int n_frames = len > 0 ? len / hop + 1 : 0;
    for (int m = blockIdx.y; m < n_mels; m += gridDim.y) {
        const size_t off = (static_cast<size_t>(b) * n_mels + m) * ld + t;
        feats[off] = t < n_frames ? (logf(mel[off]) - mean[m]) * inv_scale[m] : 0.f;
    }
}

}  // namespace nbasr

using namespace nbasr;

extern "C" int nbasr_frame_signal(const float* wave, const int* lengths, float* frames, int batch, int samples, int ld_wave,
                                  int win, int hop, int ld_frames, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && samples >= 0 && win > 0 && hop > 0 && ld_wave >= samples, NBASR_EINVAL, "nbasr_frame_signal: bad sizes");
    NBASR_REQUIRE(ld_frames % 4 == 0 && ld_frames >= (samples > 0 ? samples / hop + 1 : 0), NBASR_EALIGN,
                  "nbasr_frame_signal: ld_frames=%d must be a multiple of 4 and hold %d frames", ld_frames, samples > 0 ? samples / hop + 1 : 0);
    NBASR_REQUIRE(samples == 0 || samples > win / 2, NBASR_EINVAL, "nbasr_frame_signal: reflect padding needs more than %d samples", win / 2);
    if (batch == 0 || ld_frames == 0) return NBASR_OK;
    NBASR_REQUIRE(wave && frames, NBASR_ENULL, "nbasr_frame_signal: NULL pointer");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_frame_signal: batch %d > 65535", batch);
    hipLaunchKernelGGL(frame_signal_kernel, dim3((ld_frames + 255) / 256, win < 64 ? win : 64, batch), dim3(256), 0, as_stream(stream),
                       wave, lengths, frames, samples, ld_wave, win, hop, ld_frames);
    return launch_status("nbasr_frame_signal");
}

extern "C" int nbasr_power_spectrum(const float* spec, float* power, int batch, int bins, int rows_out, int ld, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && bins > 0 && rows_out >= bins, NBASR_EINVAL, "nbasr_power_spectrum: bad sizes");
    NBASR_REQUIRE(ld % 4 == 0, NBASR_EALIGN, "nbasr_power_spectrum: ld=%d must be a multiple of 4", ld);
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(spec && power, NBASR_ENULL, "nbasr_power_spectrum: NULL pointer");
    NBASR_REQUIRE(aligned16(spec) && aligned16(power), NBASR_EALIGN, "nbasr_power_spectrum: buffers must be 16-byte aligned");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_power_spectrum: batch %d > 65535", batch);
    const int nq = ld / 4;
    hipLaunchKernelGGL(power_spectrum_kernel, dim3((nq + 255) / 256, rows_out < 64 ? rows_out : 64, batch), dim3(256), 0, as_stream(stream),
                       spec, power, bins, rows_out, nq);
    return launch_status("nbasr_power_spectrum");
}

extern "C" int nbasr_log_normalize(const float* mel, const int* lengths, const float* mean, const float* inv_scale, float* feats,
                                   int batch, int samples, int hop, int n_mels, int ld, nbasr_stream_t stream)
{
    clear_error();
    NBASR_REQUIRE(batch >= 0 && samples >= 0 && hop > 0 && n_mels > 0 && ld >= 0, NBASR_EINVAL, "nbasr_log_normalize: bad sizes");
    if (batch == 0 || ld == 0) return NBASR_OK;
    NBASR_REQUIRE(mel && mean && inv_scale && feats, NBASR_ENULL, "nbasr_log_normalize: NULL pointer");
    NBASR_REQUIRE(batch <= 65535, NBASR_EINVAL, "nbasr_log_normalize: batch %d > 65535", batch);
    hipLaunchKernelGGL(log_normalize_kernel, dim3((ld + 255) / 256, n_mels < 80 ? n_mels : 80, batch), dim3(256), 0, as_stream(stream),
                       mel, lengths, mean, inv_scale, feats, samples, hop, n_mels, ld);
    return launch_status("nbasr_log_normalize");
}
