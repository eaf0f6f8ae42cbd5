"""ctypes binding of libnbasr_hip.so (C ABI declared in include/nbasr.h).

PyTorch is used here only as plumbing: device memory (``tensor.data_ptr()``) and the current HIP
stream.  Every wrapper validates device / dtype / contiguity on the host, passes raw pointers
and sizes, and raises ``HipError`` with the library's message on a non-zero return code.

There is NO fallback: if the library is missing or the tensors are not on a HIP device the
call fails loudly (build with ``python -m nb_asr_amd.build``).
"""
import ctypes
import os
import threading
import pathlib

import torch

LIB_PATH = pathlib.Path(__file__).resolve().parent / 'lib' / 'libnbasr_hip.so'
ABI_VERSION = 6

_c_float_p = ctypes.c_void_p      # device pointers travel as opaque addresses
_c_int = ctypes.c_int
_c_stream = ctypes.c_void_p

class DeferredLN(ctypes.Structure):
    """struct nbasr_deferred_ln: per-frame statistics + affine parameters of a pending LayerNorm."""
    _fields_ = [('stats', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p)]


_c_ln_p = ctypes.POINTER(DeferredLN)

# name -> (restype, argtypes); must list every symbol declared in include/nbasr.h
SIGNATURES = {
    'nbasr_version': (_c_int, []),
    'nbasr_build_id': (ctypes.c_char_p, []),
    'nbasr_last_error': (ctypes.c_char_p, []),
    'nbasr_pad_amounts': (_c_int, [_c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    'nbasr_output_frames': (_c_int, [_c_int]),
    'nbasr_stream_create': (_c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    # LayerNorm, statistics, node ops (storage-type generic: a trailing dtype code)
    'nbasr_layernorm_channels': (_c_int, [_c_float_p] * 5 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_int, _c_stream]),
    'nbasr_channel_stats': (_c_int, [_c_float_p] * 2 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_stream]),
    'nbasr_grouped_stats_workspace_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_grouped_stats_finalize': (_c_int, [_c_float_p] * 2 + [_c_int] * 6 + [ctypes.c_float, _c_stream]),
    'nbasr_grouped_conv1d_node': (_c_int, [_c_float_p] * 7 + [_c_int] * 7 + [_c_ln_p, _c_int, _c_int, _c_float_p, _c_int, _c_int, _c_stream]),
    'nbasr_pack_grouped_weights': (_c_int, [_c_float_p] * 2 + [_c_int] * 3 + [_c_stream]),
    'nbasr_grouped_cell_fits': (_c_int, [_c_int] * 3),
    'nbasr_grouped_cell_fused': (_c_int, [_c_float_p, _c_float_p, _c_float_p, _c_int, _c_int, _c_float_p, _c_float_p, _c_int, _c_int,
                                          _c_float_p, _c_float_p, _c_int, _c_int, _c_int, _c_float_p] + [_c_int] * 5 + [_c_ln_p, _c_float_p, _c_int, _c_stream]),
    'nbasr_grouped_cell_mfma_weights_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_grouped_cell_mfma_pack': (_c_int, [_c_float_p] * 2 + [_c_int] * 3 + [_c_stream]),
    'nbasr_grouped_cell_mfma_fits': (_c_int, [_c_int] * 3),
    'nbasr_grouped_cell_mfma': (_c_int, [_c_float_p, _c_float_p, _c_float_p, _c_int, _c_int, _c_float_p, _c_float_p, _c_int, _c_int,
                                         _c_float_p, _c_float_p, _c_int, _c_int, _c_int, _c_float_p] + [_c_int] * 5 + [_c_ln_p, _c_stream]),
    'nbasr_skip_sum': (_c_int, [_c_float_p] * 4 + [_c_int] * 4 + [_c_ln_p, _c_int, _c_int, _c_stream]),
    'nbasr_repitch': (_c_int, [_c_float_p] * 2 + [_c_int] * 5 + [_c_stream]),
    'nbasr_convert': (_c_int, [_c_float_p] * 2 + [ctypes.c_longlong, _c_int, _c_int, _c_stream]),
    # dense convolutions
    'nbasr_dense_conv1d_fused': (_c_int, [_c_float_p] * 7 + [_c_int] * 8 + [_c_ln_p, _c_int, _c_int, _c_stream]),
    'nbasr_packed_dense_weights_bytes': (ctypes.c_size_t, [_c_int] * 5),
    'nbasr_pack_dense_weights': (_c_int, [_c_int] + [_c_float_p] * 2 + [_c_int] * 5 + [_c_stream]),
    'nbasr_dense_conv1d_packed': (_c_int, [_c_int, _c_float_p, _c_int] + [_c_float_p] * 8 + [_c_int] * 10 + [_c_ln_p, _c_float_p, _c_stream]),
    'nbasr_input_range': (_c_int, [_c_float_p] * 2 + [_c_int] * 4 + [_c_stream]),
    'nbasr_split_image_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_layernorm_split_image': (_c_int, [_c_float_p] * 6 + [_c_int] * 4 + [ctypes.c_float, _c_stream]),
    'nbasr_split_image_ranged': (_c_int, [_c_float_p] * 3 + [_c_int] * 4 + [_c_stream]),
    'nbasr_bf16_image_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_bf16_image': (_c_int, [_c_float_p] * 5 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_stream]),
    # per-frame linear maps
    'nbasr_pointwise_packed_weights_bytes': (ctypes.c_size_t, [_c_int] * 2),
    'nbasr_pointwise_workspace_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_pack_pointwise_weights': (_c_int, [_c_float_p] * 2 + [_c_int] * 2 + [_c_stream]),
    'nbasr_linear_fused_packed': (_c_int, [_c_float_p] * 8 + [_c_int] * 5 + [_c_ln_p, _c_int, _c_int, _c_stream]),
    'nbasr_pointwise_linear': (_c_int, [_c_float_p] * 4 + [_c_int] * 6 + [_c_stream]),
    # LSTM + head
    'nbasr_lstm_input_projection': (_c_int, [_c_float_p] * 5 + [_c_int] * 5 + [_c_ln_p, _c_stream]),
    'nbasr_lstm_input_projection_packed': (_c_int, [_c_float_p] * 6 + [_c_int] * 5 + [_c_ln_p, _c_stream]),
    'nbasr_lstm_input_projection_packed_into': (_c_int, [_c_float_p] * 6 + [_c_int] * 5 + [_c_ln_p, _c_int, _c_int, _c_stream]),
    'nbasr_pointwise_bf16_weights_bytes': (ctypes.c_size_t, [_c_int] * 2),
    'nbasr_pointwise_bf16_workspace_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_pack_pointwise_weights_bf16': (_c_int, [_c_float_p] * 2 + [_c_int] * 2 + [_c_stream]),
    'nbasr_linear_fused_bf16': (_c_int, [_c_float_p] * 8 + [_c_int] * 5 + [_c_ln_p, _c_int, _c_int, _c_stream]),
    'nbasr_lstm_input_projection_bf16': (_c_int, [_c_float_p] * 6 + [_c_int] * 5 + [_c_ln_p, _c_stream]),
    'nbasr_lstm_packed_whh_bytes': (ctypes.c_size_t, [_c_int]),
    'nbasr_lstm_pack_whh': (_c_int, [_c_float_p] * 2 + [_c_int, _c_stream]),
    'nbasr_lstm_recurrence_packed': (_c_int, [_c_float_p] * 4 + [_c_int] * 3 + [_c_stream]),
    'nbasr_lstm_seq_workspace_bytes': (ctypes.c_size_t, [_c_int] * 2),
    'nbasr_lstm_recurrence_seq': (_c_int, [_c_float_p] * 5 + [_c_int] * 4 + [_c_stream]),
    'nbasr_lstm_seq_status': (_c_int, [_c_float_p, _c_stream]),
    'nbasr_lstm_packed_whh16_bytes': (ctypes.c_size_t, [_c_int]),
    'nbasr_lstm_pack_whh16': (_c_int, [_c_float_p] * 2 + [_c_int, _c_stream]),
    'nbasr_lstm_xcd_workspace_bytes': (ctypes.c_size_t, [_c_int] * 2),
    'nbasr_lstm_recurrence_xcd': (_c_int, [_c_float_p] * 5 + [_c_int] * 4 + [_c_stream]),
    'nbasr_lstm_recurrence_frames16': (_c_int, [_c_float_p] * 5 + [_c_int] * 3 + [_c_stream]),
    'nbasr_linear_head': (_c_int, [_c_float_p] * 4 + [_c_int] * 3 + [_c_stream]),
    'nbasr_linear_head_bct': (_c_int, [_c_float_p] * 4 + [_c_int] * 5 + [_c_ln_p, _c_stream]),
    # post-logits step
    'nbasr_ctc_postprocess': (_c_int, [_c_float_p] * 5 + [_c_int] * 4 + [_c_stream]),
    'nbasr_ctc_loss': (_c_int, [_c_float_p] * 5 + [_c_int] * 6 + [_c_stream]),
    'nbasr_ctc_grad_workspace_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_ctc_loss_grad': (_c_int, [_c_float_p] * 7 + [_c_int] * 5 + [_c_stream]),
    'nbasr_ctc_beam_workspace_bytes': (ctypes.c_size_t, [_c_int] * 4),
    'nbasr_ctc_beam_search': (_c_int, [_c_float_p] * 6 + [_c_int] * 6 + [_c_stream]),
    'nbasr_token_error_counts': (_c_int, [_c_float_p, _c_float_p, _c_int, _c_float_p, _c_float_p, _c_int, _c_float_p, _c_int, _c_int,
                                          _c_float_p, _c_int, _c_stream]),
    # front-end
    'nbasr_frame_signal': (_c_int, [_c_float_p, ctypes.c_void_p, _c_float_p] + [_c_int] * 6 + [_c_stream]),
    'nbasr_power_spectrum': (_c_int, [_c_float_p] * 2 + [_c_int] * 4 + [_c_stream]),
    'nbasr_log_normalize': (_c_int, [_c_float_p, ctypes.c_void_p] + [_c_float_p] * 3 + [_c_int] * 5 + [_c_stream]),
    # backward building blocks
    'nbasr_grouped_conv1d_backward_workspace_bytes': (ctypes.c_size_t, [_c_int] * 4),
    'nbasr_grouped_conv1d_backward': (_c_int, [_c_float_p] * 8 + [_c_int] * 7 + [_c_stream]),
    'nbasr_layernorm_backward_workspace_bytes': (ctypes.c_size_t, [_c_int] * 3),
    'nbasr_layernorm_channels_backward': (_c_int, [_c_float_p] * 8 + [_c_int] * 4 + [_c_stream]),
    'nbasr_relu_clamp_backward': (_c_int, [_c_float_p] * 3 + [ctypes.c_longlong, _c_stream]),
    'nbasr_zero_stuff': (_c_int, [_c_float_p] * 2 + [_c_int] * 7 + [_c_stream]),
    'nbasr_conv_fold': (_c_int, [_c_float_p] * 2 + [_c_int] * 8 + [_c_stream]),
    'nbasr_dense_conv1d_linear': (_c_int, [_c_float_p] * 4 + [_c_int] * 8 + [_c_stream]),
    'nbasr_conv_cols': (_c_int, [_c_float_p] * 2 + [_c_int] * 10 + [_c_stream]),
    'nbasr_rows_of_channels': (_c_int, [_c_float_p] * 2 + [_c_int] * 5 + [_c_stream]),
    'nbasr_lstm_gate_scan': (_c_int, [_c_float_p] * 2 + [_c_int] * 4 + [_c_stream]),
    'nbasr_lstm_backward_step': (_c_int, [_c_float_p] * 6 + [_c_int] * 5 + [_c_stream]),
}

# entry points that only answer on the host (never recorded on a launch tape); every other one enqueues work on a stream
_HOST_ONLY = frozenset({'nbasr_version', 'nbasr_build_id', 'nbasr_last_error', 'nbasr_pad_amounts', 'nbasr_output_frames', 'nbasr_stream_create',
                        'nbasr_grouped_cell_fits', 'nbasr_grouped_cell_mfma_fits'})
_ENQUEUES = frozenset(name for name in SIGNATURES if name not in _HOST_ONLY and not name.endswith('_bytes')
                      and '_bytes_' not in name)

F32, BF16 = 0, 1                 # NBASR_F32 / NBASR_BF16
GC_FPL8, GC_WPERM, GC_OSPLIT, GC_PIPE, GC_RING = 1, 2, 4, 8, 16         # NBASR_GC_* variant bits of nbasr_grouped_conv1d_node


class HipError(RuntimeError):
    """``code``: the entry point's return value (a negative NBASR_E* argument error or a positive hipError_t), None for errors raised
    on the python side."""
    code = None




_lib = None
_tls = threading.local()


class _TapingLibrary:
    """Stand-in for the loaded library while a ``ForwardPlan`` records a launch tape (executor.LaunchTape): every entry
    point that takes a stream -- i.e. enqueues work -- is executed AND appended to ``entries`` as ``[function, [args]]``;
    host-only queries (sizes, error text, version) pass straight through."""

    def __init__(self, lib, entries):
        self._lib, self._entries = lib, entries

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in _ENQUEUES:
            return fn
        entries = self._entries

        def call(*args):
            entries.append([fn, list(args)])
            return fn(*args)
        return call


def start_tape(entries):
    """Record this thread's enqueueing library calls into ``entries`` until ``stop_tape()``."""
    _tls.taping = _TapingLibrary(load_library(), entries)


def stop_tape():
    _tls.taping = None


def load_library(path=None):
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if path is None:
        taping = getattr(_tls, 'taping', None)
        if taping is not None:
            return taping
        if _lib is not None:
            return _lib
    p = pathlib.Path(path) if path is not None else LIB_PATH
    if not p.exists():
        raise HipError(f'{p} not found: the HIP extension has not been built '
                       f'(run `python -m nb_asr_amd.build`); there is no CPU fallback')
    lib = ctypes.CDLL(str(p))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.nbasr_version()
    if got != ABI_VERSION:
        raise HipError(f'{p}: ABI version {got}, binding expects {ABI_VERSION}; rebuild the library')
    if path is None:
        _lib = lib
    return lib


def build_id():
    """Build id compiled into the loaded library (see ``nb_asr_amd.build.source_hash``)."""
    return load_library().nbasr_build_id().decode('ascii', 'replace')


def _check(rc, what):
    if rc != 0:
        msg = load_library().nbasr_last_error().decode('utf-8', 'replace')
        err = HipError(f'{what} failed with code {rc}: {msg}')
        err.code = int(rc)
        raise err


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipError(f'{name} must be a tensor on a HIP device (got {getattr(t, "device", type(t))}); '
                       f'this package has no CPU path')
    if t.dtype != torch.float32:
        raise HipError(f'{name} must be float32 (got {t.dtype})')
    if not t.is_contiguous():
        raise HipError(f'{name} must be contiguous')
    return t.data_ptr()


def _opt(t, name):
    return None if t is None else _dev(t, name)


def dtype_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise HipError(f'activations must be float32 or bfloat16 (got {dtype})')


def row_pitch(frames, dtype=torch.float32):
    """Row pitch of an activation tensor: whole 16-byte chunks (4 fp32 / 8 bf16 frames)."""
    return (frames + 7) & ~7 if dtype == torch.bfloat16 else (frames + 3) & ~3


def _act(t, name, dtype=None):
    """Device pointer of an activation tensor of either storage type (float32 / bfloat16), contiguous."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise HipError(f'{name} must be a tensor on a HIP device (got {getattr(t, "device", type(t))}); '
                       f'this package has no CPU path')
    if t.dtype not in (torch.float32, torch.bfloat16) or (dtype is not None and t.dtype != dtype):
        raise HipError(f'{name} must be {dtype or "float32 or bfloat16"} (got {t.dtype})')
    if not t.is_contiguous():
        raise HipError(f'{name} must be contiguous')
    return t.data_ptr()


def _act_opt(t, name, dtype):
    return None if t is None else _act(t, name, dtype)


def _ln(ln):
    """(stats, gamma, beta) tensors of a pending LayerNorm -> pointer to a nbasr_deferred_ln (or NULL)."""
    if ln is None:
        return None
    stats, gamma, beta = ln
    return ctypes.byref(DeferredLN(_dev(stats, 'ln.stats'), _dev(gamma, 'ln.gamma'), _dev(beta, 'ln.beta')))


# ---------------------------------------------------------------------------------------------
# host-only helpers (usable without a GPU)
# ---------------------------------------------------------------------------------------------
def pad_amounts(kernel, dilation, stride):
    left, right = _c_int(), _c_int()
    _check(load_library().nbasr_pad_amounts(kernel, dilation, stride, ctypes.byref(left), ctypes.byref(right)),
           'nbasr_pad_amounts')
    return left.value, right.value


def output_frames(frames):
    return load_library().nbasr_output_frames(frames)


def round_up4(n):
    return (n + 3) & ~3


# ---------------------------------------------------------------------------------------------
# device entry points; activations are (batch, channels, ld) float32 tensors whose last dimension
# is the row pitch ld (a multiple of 4) and `frames` <= ld the number of valid frames
# ---------------------------------------------------------------------------------------------
def grouped_conv1d_fused(x, weight, bias, skips, y, frames, groups, kernel, dilation, ln=None, ln_on_x=False,
                         ln_on_skip0=False, stats_out=None, stats_ws=None, eps=0.0):
    """The fp32 node op (the default kernel of grouped_conv1d_node).  `ln` = (stats, gamma, beta) of a pending LayerNorm carried by x
    (ln_on_x) and/or skips[0] (ln_on_skip0).  `stats_ws` (grouped_stats_workspace): also emit the partial LayerNorm statistics of y;
    with `stats_out` (B, 2, ld) they are merged at once (grouped_stats_finalize)."""
    if stats_out is not None and stats_ws is None:
        raise HipError('grouped_conv1d_fused: stats_out needs the partial-statistics workspace stats_ws')
    grouped_conv1d_node(x, weight, bias, skips, y, frames, groups, kernel, dilation, ln, ln_on_x, ln_on_skip0, stats_ws, 0)
    if stats_out is not None and x.shape[0] and x.shape[2]:
        grouped_stats_finalize(stats_ws, stats_out, x.shape[1], frames, groups, eps)
    return y


def grouped_cell_fits(channels, ld, groups):
    """0 when a (channels, ld, groups) cell cannot run as one launch, else the groups per statistics partial of that launch (4 or 2)."""
    return int(load_library().nbasr_grouped_cell_fits(channels, ld, groups))


def grouped_cell_fused(x0, nodes, skip_mask, y, frames, groups, ln=None, stats_ws=None):
    """nodes: three (weight, bias, kernel, dilation), ``weight`` = the [group][ci][tap][co] copy of ``pack_grouped_weights`` (ABI 4);
    skip_mask: bit0 s00 | bit1 s10 | bit2 s11 | bit3 s20 | bit4 s21 | bit5 s22.
    ``stats_ws`` (grouped_stats_workspace): also emit the partial LayerNorm statistics of the result (merge: grouped_stats_finalize)."""
    b, c, ld = x0.shape
    args = []
    for w, bias, k, d in nodes:
        args += [_dev(w, 'weight'), _dev(bias, 'bias'), k, d]
    dtype = x0.dtype
    _check(load_library().nbasr_grouped_cell_fused(_act(x0, 'x0', dtype), *args, skip_mask, _act(y, 'y', dtype), b, c, frames, ld, groups,
                                                   _ln(ln), _opt(stats_ws, 'stats_ws'), dtype_code(dtype), _stream(x0)), 'nbasr_grouped_cell_fused')
    return y


def grouped_cell_mfma_fits(channels, ld, groups):
    """0 when a bf16 (channels, ld, groups) cell cannot run as one matrix-core launch, else its groups per workgroup."""
    return int(load_library().nbasr_grouped_cell_mfma_fits(channels, ld, groups))


def grouped_cell_mfma_pack(weight, groups):
    """(C, C/groups, k) fp32 weight (the values of a bf16 parameter) -> the MFMA fragment image grouped_cell_mfma takes for that node."""
    c, _, k = weight.shape
    nbytes = load_library().nbasr_grouped_cell_mfma_weights_bytes(c, groups, k)
    if nbytes == 0:
        raise HipError(f'grouped_cell_mfma_pack: weight shape {tuple(weight.shape)} with {groups} groups is not a node op of the search space')
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    _check(load_library().nbasr_grouped_cell_mfma_pack(_dev(weight, 'weight'), packed.data_ptr(), c, groups, k, _stream(weight)),
           'nbasr_grouped_cell_mfma_pack')
    return packed


def grouped_cell_mfma(x0, nodes, skip_mask, y, frames, groups, ln=None):
    """The bf16 cell on the matrix cores.  nodes: three (packed weight from grouped_cell_mfma_pack, fp32 bias, kernel, dilation)."""
    b, c, ld = x0.shape
    args = []
    for packed, bias, k, d in nodes:
        if not packed.is_cuda or packed.dtype != torch.uint8 or packed.numel() != load_library().nbasr_grouped_cell_mfma_weights_bytes(c, groups, k):
            raise HipError('grouped_cell_mfma: node weights must be the uint8 tensors of grouped_cell_mfma_pack for this (channels, groups, kernel)')
        args += [packed.data_ptr(), _dev(bias, 'bias'), k, d]
    _check(load_library().nbasr_grouped_cell_mfma(_act(x0, 'x0', torch.bfloat16), *args, skip_mask, _act(y, 'y', torch.bfloat16), b, c, frames, ld,
                                                  groups, _ln(ln), _stream(x0)), 'nbasr_grouped_cell_mfma')
    return y


def grouped_stats_finalize(stats_ws, stats_out, channels, frames, groups, eps, groups_per_part=4):
    """Merge the partial statistics a node / fused-cell launch left in ``stats_ws`` into (mean, rstd) rows.  ``groups_per_part``: 4 for
    the node kernels, ``grouped_cell_fits(...)`` (4 or 2) for a fused cell."""
    b, _, ld = stats_out.shape
    _check(load_library().nbasr_grouped_stats_finalize(_dev(stats_ws, 'stats_ws'), _dev(stats_out, 'stats_out'), b, channels,
                                                       frames, ld, groups, int(groups_per_part), float(eps), _stream(stats_out)),
           'nbasr_grouped_stats_finalize')
    return stats_out


def grouped_stats_workspace(batch, ld, groups, device):
    n = load_library().nbasr_grouped_stats_workspace_bytes(batch, ld, groups) // 4
    return torch.empty(max(n, 4), dtype=torch.float32, device=device)


def skip_sum(skips, y, frames, ln=None, ln_on_skip0=False):
    """Node whose main op is `zero`: y = sum of the skips (float32 or bfloat16 tensors, all of y's type)."""
    b, c, ld = y.shape
    dt = y.dtype
    s = list(skips) + [None] * (3 - len(skips))
    _check(load_library().nbasr_skip_sum(_act_opt(s[0], 'skip0', dt), _act_opt(s[1], 'skip1', dt), _act_opt(s[2], 'skip2', dt),
                                         _act(y, 'y'), b, c, frames, ld, _ln(ln), int(ln_on_skip0), dtype_code(dt), _stream(y)),
           'nbasr_skip_sum')
    return y


def channel_stats(x, stats, frames, eps):
    """One read pass over x (B, C, ld), float32 or bfloat16: stats (B, 2, ld) <- per-frame (mean, 1/sqrt(var + eps)) over channels."""
    b, c, ld = x.shape
    _check(load_library().nbasr_channel_stats(_act(x, 'x'), _dev(stats, 'stats'), b, c, frames, ld, float(eps), dtype_code(x.dtype),
                                              _stream(x)), 'nbasr_channel_stats')
    return stats


def layernorm_channels(x, gamma, beta, y, frames, eps, absmax=None):
    """LayerNorm over channels, x -> y (float32 -> float32, bfloat16 -> bfloat16 or bfloat16 -> float32); with ``absmax`` (a (B,)
    float32 device tensor; float32 tensors only) also max|y[b]| per utterance."""
    b, c, ld = x.shape
    if absmax is not None and absmax.numel() != b:
        raise HipError('absmax must hold one float per utterance')
    _check(load_library().nbasr_layernorm_channels(_act(x, 'x'), _dev(gamma, 'gamma'), _dev(beta, 'beta'), _act(y, 'y'),
                                                   _opt(absmax, 'absmax'), b, c, frames, ld, float(eps), dtype_code(x.dtype),
                                                   dtype_code(y.dtype), _stream(x)), 'nbasr_layernorm_channels')
    return y


def split_image(batch, channels, ld, device):
    """Buffer for the pre-split fp16 image of a (batch, channels, ld) activation (layernorm_split_image)."""
    return torch.empty(max(load_library().nbasr_split_image_bytes(batch, channels, ld), 16), dtype=torch.uint8, device=device)


def layernorm_split_image(x, gamma, beta, stats, bound, image, frames, eps):
    """LayerNorm(x) written as the fp16-split image of the dense convolution; also fills stats (B, 2, ld) and bound (B)."""
    b, c, ld = x.shape
    if image.dtype != torch.uint8 or image.numel() < load_library().nbasr_split_image_bytes(b, c, ld):
        raise HipError('image buffer too small: allocate it with split_image(batch, channels, ld, device)')
    _check(load_library().nbasr_layernorm_split_image(_dev(x, 'x'), _dev(gamma, 'gamma'), _dev(beta, 'beta'), _dev(stats, 'stats'),
                                                      _dev(bound, 'bound'), image.data_ptr(), b, c, frames, ld, float(eps), _stream(x)),
           'nbasr_layernorm_split_image')
    return image


def dense_conv1d_fused_packed_f16_img(image, bound, batch, c_in, frames_in, ld_in, packed, c_out, kernel, bias, y, stride,
                                      row_tile=128, stats_part=None, frame_tile=256):
    """The fp16x2 convolution on the pre-split operand image (layernorm_split_image): LDS-DMA-only GEMM.  ``stats_part``: also emit
    the partial LayerNorm statistics of y (dense_stats_part_floats)."""
    _check_packed(packed, 'f16x2', c_out, c_in, kernel, row_tile)
    if image.numel() < load_library().nbasr_split_image_bytes(batch, c_in, ld_in):
        raise HipError('image buffer too small for (batch, c_in, ld_in)')
    return _dense_packed('f16x2', image.data_ptr(), True, bound, None, packed, bias, (None, None, None), y, _dev(y, 'y'), batch, c_in,
                         frames_in, ld_in, c_out, kernel, stride, row_tile, None, _stream(y), stats_part, frame_tile)


def input_range(x, frames, out):
    """out (B, 4) float32 <- per utterance (max finite |x|, quietest non-silent frame's max, non-finite flag, -); see nbasr.h."""
    b, c, ld = x.shape
    if out.numel() < 4 * b:
        raise HipError('input_range: out needs 4 floats per utterance')
    _check(load_library().nbasr_input_range(_dev(x, 'x'), _dev(out, 'range'), b, c, frames, ld, _stream(x)), 'nbasr_input_range')
    return out


def dense_conv1d_first_ranged(x, frames_in, x_range, packed_f16, packed_bf16x3, c_out, kernel, bias, y, stride, image=None, row_tile=128,
                              stats_part=None, frame_tile=256):
    """The model's first dense conv with per-utterance routing on the device: ordinary utterances on the 2-way fp16 split,
    extreme ones (non-finite samples, > 2^12 dynamic range between frames) on the 3-way bf16 split; same output tensor.
    ``image``: uint8 workspace of ``nbasr_split_image_bytes`` -- the fp16 leg then runs on the image path (one split pass over
    x, LDS-DMA-only GEMM; ``packed_f16`` packed for ``row_tile``); None: the input is split in the GEMM's prologue."""
    lib = load_library()
    b, c_in, ld_in = x.shape
    _check_packed(packed_f16, 'f16x2', c_out, c_in, kernel, 128 if image is None else row_tile)
    _check_packed(packed_bf16x3, 'bf16x3', c_out, c_in, kernel)
    none3, stream = (None, None, None), _stream(x)
    if image is not None:
        if image.dtype != torch.uint8 or image.numel() < lib.nbasr_split_image_bytes(b, c_in, ld_in):
            raise HipError('dense_conv1d_first_ranged: image workspace too small (nbasr_split_image_bytes)')
        _check(lib.nbasr_split_image_ranged(_dev(x, 'x'), _dev(x_range, 'x_range'), image.data_ptr(), b, c_in, frames_in, ld_in, stream),
               'nbasr_split_image_ranged')
    if image is not None:
        _dense_packed('f16x2', image.data_ptr(), True, None, x_range, packed_f16, bias, none3, y, _dev(y, 'y'), b, c_in, frames_in, ld_in,
                      c_out, kernel, stride, row_tile, None, stream, stats_part, frame_tile)
    else:
        _dense_packed('f16x2', _dev(x, 'x'), False, None, x_range, packed_f16, bias, none3, y, _dev(y, 'y'), b, c_in, frames_in, ld_in,
                      c_out, kernel, stride, 128, None, stream, stats_part)
    return _dense_packed('bf16x3', _dev(x, 'x'), False, None, x_range, packed_bf16x3, bias, none3, y, _dev(y, 'y'), b, c_in, frames_in,
                         ld_in, c_out, kernel, stride, 128, None, stream, stats_part)


def dense_conv1d_fused(x, frames_in, weight, bias, skips, y, stride, ln=None, ln_on_x=False, ln_on_skip0=False):
    b, c_in, ld_in = x.shape
    c_out, _, kernel = weight.shape if weight.dim() == 3 else (weight.shape[0], weight.shape[1], 1)
    ld_out = y.shape[2]
    s = list(skips) + [None] * (3 - len(skips))
    _check(load_library().nbasr_dense_conv1d_fused(
        _dev(x, 'x'), _dev(weight, 'weight'), _dev(bias, 'bias'), _opt(s[0], 'skip0'), _opt(s[1], 'skip1'),
        _opt(s[2], 'skip2'), _dev(y, 'y'), b, c_in, frames_in, ld_in, c_out, ld_out, kernel, stride, _ln(ln), int(ln_on_x),
        int(ln_on_skip0), _stream(x)), 'nbasr_dense_conv1d_fused')
    return y


DENSE_SCHEMES = {'bf16x3': 0, 'f16x2': 1, 'bf16': 2}        # NBASR_DENSE_* of nbasr_dense_conv1d_packed


def _scheme_code(scheme):
    if scheme not in DENSE_SCHEMES:
        raise HipError(f"unknown operand scheme {scheme!r} (expected 'bf16x3', 'f16x2' or 'bf16')")
    return DENSE_SCHEMES[scheme]


def packed_dense_weights_bytes(scheme, c_out, c_in, kernel, row_tile=128):
    return load_library().nbasr_packed_dense_weights_bytes(_scheme_code(scheme), c_out, c_in, kernel, row_tile)


def pack_dense_weights(weight, stride, scheme='bf16x3', row_tile=128):
    """(c_out, c_in, 8) fp32 weight -> opaque uint8 tensor holding its operand form (3 x bf16 split, 2 x fp16 split for
    ``scheme='f16x2'``, one bf16 term for 'bf16') in the LDS layout of the stride-`stride` kernel that will consume it.  ``row_tile``
    other than 128: the 64- / 160-row tiles of the image-path kernels (f16x2: 64, 160; bf16: 160)."""
    code = _scheme_code(scheme)
    lib = load_library()
    c_out, c_in, kernel = weight.shape
    nbytes = lib.nbasr_packed_dense_weights_bytes(code, c_out, c_in, kernel, row_tile)
    if nbytes == 0:
        raise HipError(f'packed dense path ({scheme}) does not cover weight shape {tuple(weight.shape)} with row_tile={row_tile}')
    if not weight.is_cuda:
        raise HipError('weight must be on a HIP device')
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    _check(lib.nbasr_pack_dense_weights(code, _dev(weight, 'weight'), packed.data_ptr(), c_out, c_in, kernel, stride, row_tile,
                                        _stream(weight)), 'nbasr_pack_dense_weights')
    packed.nbasr_row_tile = row_tile           # the packed sizes of the two tilings can coincide (c_out = 1200): remember which
    return packed


def _dense_packed(scheme, x_ptr, image, x_absmax, x_range, packed, bias, s, y, y_ptr, b, c_in, frames_in, ld_in, c_out, kernel, stride,
                  row_tile, ln, stream, stats_part=None, frame_tile=256):
    """The one C entry point of the packed k = 8 convolution (nbasr.h: nbasr_dense_conv1d_packed)."""
    if stats_part is not None and stats_part.numel() < dense_stats_part_floats(b, c_out, y.shape[2]):
        raise HipError('stats_part too small: ceil(c_out / 16) * batch * 2 * ld_out floats (dense_stats_part_floats)')
    _check(load_library().nbasr_dense_conv1d_packed(
        _scheme_code(scheme), x_ptr, int(image), _opt(x_absmax, 'x_absmax'), _opt(x_range, 'x_range'), packed.data_ptr(),
        _dev(bias, 'bias'), _opt(s[0], 'skip0'), _opt(s[1], 'skip1'), _opt(s[2], 'skip2'), y_ptr, b, c_in, frames_in, ld_in, c_out,
        y.shape[2], kernel, stride, row_tile, int(frame_tile), _ln(ln), _opt(stats_part, 'stats_part'), stream), 'nbasr_dense_conv1d_packed')
    return y


DENSE_STATS_UNIT = 16         # nbasr.h: NBASR_DENSE_STATS_UNIT


def dense_stats_part_floats(batch, c_out, ld_out):
    """Floats of the statistics partials a dense convolution emits with ``stats_part``: one (mean, M2) row pair per 16 output channels
    (merge: grouped_stats_finalize(part, stats, c_out, frames_out, c_out, eps, DENSE_STATS_UNIT))."""
    return -(-c_out // DENSE_STATS_UNIT) * batch * 2 * ld_out


def _check_packed(packed, scheme, c_out, c_in, kernel, row_tile=128):
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise HipError('packed weights must be the uint8 device tensor returned by pack_dense_weights')
    if packed.numel() != packed_dense_weights_bytes(scheme, c_out, c_in, kernel, row_tile) or getattr(packed, 'nbasr_row_tile', 128) != row_tile:
        raise HipError(f'packed weights have {packed.numel()} bytes: not a {scheme} image of a ({c_out}, {c_in}, {kernel}) weight for '
                       f'row_tile={row_tile}')


def dense_conv1d_fused_packed(x, frames_in, packed, c_out, kernel, bias, skips, y, stride, ln=None, scheme='bf16x3',
                              x_absmax=None, stats_part=None):
    """Dense k=8 conv on packed weights; ``scheme`` must be the one the weights were packed with.  'f16x2' needs
    ``x_absmax``: a (B,) float32 device tensor of upper bounds of max|x[b]| (see nbasr.h) and takes no deferred LayerNorm.
    ``stats_part`` (no skips): also emit the partial LayerNorm statistics of y (dense_stats_part_floats)."""
    b, c_in, ld_in = x.shape
    s = list(skips) + [None] * (3 - len(skips))
    _check_packed(packed, scheme, c_out, c_in, kernel)
    if scheme == 'f16x2':
        if ln is not None:
            raise HipError('the f16x2 scheme takes no deferred LayerNorm (it needs the range of the normalised tensor)')
        if x_absmax is None or x_absmax.numel() != b:
            raise HipError('the f16x2 scheme needs x_absmax: one float32 bound of max|x[b]| per utterance')
    elif ln is not None and any(t is not None for t in s):
        raise HipError('the packed path takes a deferred LayerNorm only without skip inputs')
    return _dense_packed(scheme, _dev(x, 'x'), False, x_absmax if scheme == 'f16x2' else None, None, packed, bias, s, y, _dev(y, 'y'),
                         b, c_in, frames_in, ld_in, c_out, kernel, stride, 128, ln, _stream(x), stats_part)


def lstm_forward(x, frames, w_ih, w_hh, b_ih, b_hh, gates_ws, cell_ws, h_out, ln=None):
    """nn.LSTM forward as its two library calls on one stream: the exact-fp32 input projection, then the per-frame recurrence."""
    lstm_input_projection(x, frames, w_ih, b_ih, b_hh, gates_ws, w_hh.shape[1], ln)
    return lstm_recurrence(gates_ws, w_hh, cell_ws, h_out)


def lstm_input_projection(x, frames, w_ih, b_ih, b_hh, gates_ws, hidden, ln=None):
    """gates_ws (frames, batch, 4*hidden) <- x . w_ih^T + b_ih + b_hh for all frames (first half of lstm_forward)."""
    b, c_in, ld = x.shape
    _check(load_library().nbasr_lstm_input_projection(
        _dev(x, 'x'), _dev(w_ih, 'w_ih'), _dev(b_ih, 'b_ih'), _dev(b_hh, 'b_hh'), _dev(gates_ws, 'gates_ws'),
        b, c_in, frames, ld, hidden, _ln(ln), _stream(x)), 'nbasr_lstm_input_projection')
    return gates_ws


def lstm_pack_whh(w_hh):
    """(4H, H) recurrent weight -> opaque uint8 tensor in the operand-fragment order of the step kernel."""
    hidden = w_hh.shape[1]
    packed = torch.empty(load_library().nbasr_lstm_packed_whh_bytes(hidden), dtype=torch.uint8, device=w_hh.device)
    _check(load_library().nbasr_lstm_pack_whh(_dev(w_hh, 'w_hh'), packed.data_ptr(), hidden, _stream(w_hh)), 'nbasr_lstm_pack_whh')
    return packed


def lstm_recurrence_packed(gates_ws, packed_whh, cell_ws, h_out):
    b, frames, hidden = h_out.shape
    if not packed_whh.is_cuda or packed_whh.dtype != torch.uint8 or packed_whh.numel() != load_library().nbasr_lstm_packed_whh_bytes(hidden):
        raise HipError('packed_whh must be the uint8 device tensor returned by lstm_pack_whh for this hidden size')
    _check(load_library().nbasr_lstm_recurrence_packed(_dev(gates_ws, 'gates_ws'), packed_whh.data_ptr(), _dev(cell_ws, 'cell_ws'),
                                                       _dev(h_out, 'h_out'), b, frames, hidden, _stream(h_out)),
           'nbasr_lstm_recurrence_packed')
    return h_out


def lstm_seq_workspace(batch, hidden, device):
    """Workspace of the one-launch recurrence (flags + the exchange images of h), or None where that form does not apply."""
    nbytes = lstm_seq_workspace_bytes(int(batch), int(hidden), device)
    return torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None


HIP_ERROR_COOPERATIVE_LAUNCH_TOO_LARGE = 720       # hipErrorCooperativeLaunchTooLarge: nbasr_lstm_recurrence_seq passes a refused grid's code through
LSTM_SEQ_INJECT_FAULT = 1       # NBASR_LSTM_SEQ_INJECT_FAULT of nbasr_lstm_recurrence_seq (tests)


def lstm_recurrence_seq(gates_ws, packed_whh, cell_ws, h_out, seq_ws, flags=0):
    """lstm_recurrence_packed with all frames in ONE (cooperative) launch: w_hh resident in registers, h exchanged in payload-tagged
    granules; same h_out, bit for bit.  The first int32 of ``seq_ws`` is the status word (non-zero: a step timed out, h_out holds NaN)."""
    b, frames, hidden = h_out.shape
    lib = load_library()
    if not packed_whh.is_cuda or packed_whh.dtype != torch.uint8 or packed_whh.numel() != lib.nbasr_lstm_packed_whh_bytes(hidden):
        raise HipError('packed_whh must be the uint8 device tensor returned by lstm_pack_whh for this hidden size')
    need = lstm_seq_workspace_bytes(b, hidden, h_out.device)
    if need == 0 or seq_ws is None or not seq_ws.is_cuda or seq_ws.dtype != torch.uint8 or seq_ws.numel() < need:
        raise HipError(f'lstm_recurrence_seq: batch={b} hidden={hidden} needs a uint8 device workspace of {need} bytes from lstm_seq_workspace '
                       '(0 = this form does not apply)')
    _check(lib.nbasr_lstm_recurrence_seq(_dev(gates_ws, 'gates_ws'), packed_whh.data_ptr(), _dev(cell_ws, 'cell_ws'), _dev(h_out, 'h_out'),
                                         seq_ws.data_ptr(), b, frames, hidden, int(flags), _stream(h_out)), 'nbasr_lstm_recurrence_seq')
    return h_out


def lstm_seq_workspace_bytes(batch, hidden, device):
    """Workspace of the one-launch recurrence on ``device`` (0: that form does not apply there).  The answer depends on how many
    workgroups THAT device holds at once, and the C query asks the current device: it runs with ``device`` current."""
    with torch.cuda.device(device):
        return load_library().nbasr_lstm_seq_workspace_bytes(batch, hidden)


def lstm_seq_status(seq_ws):
    """Synchronises the current stream and raises HipError if a step of the last one-launch recurrence timed out."""
    _check(load_library().nbasr_lstm_seq_status(seq_ws.data_ptr(), _stream(seq_ws)), 'nbasr_lstm_seq_status')


def lstm_pack_whh16(w_hh):
    """w_hh (4*hidden, hidden) fp32 -> the resident operand image of lstm_recurrence_xcd (two fp16 terms per weight; hidden <= 512)."""
    hidden = w_hh.shape[1]
    lib = load_library()
    nbytes = lib.nbasr_lstm_packed_whh16_bytes(hidden)
    if w_hh.shape[0] != 4 * hidden or nbytes == 0:
        raise HipError(f'lstm_pack_whh16: w_hh must be (4*hidden, hidden) with hidden <= 512, got {tuple(w_hh.shape)}')
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w_hh.device)
    _check(lib.nbasr_lstm_pack_whh16(_dev(w_hh, 'w_hh'), packed.data_ptr(), hidden, _stream(w_hh)), 'nbasr_lstm_pack_whh16')
    return packed


def lstm_xcd_workspace_bytes(batch, hidden):
    return load_library().nbasr_lstm_xcd_workspace_bytes(int(batch), int(hidden))


def lstm_xcd_workspace(batch, hidden, device):
    nbytes = lstm_xcd_workspace_bytes(batch, hidden)
    if nbytes == 0:
        raise HipError(f'lstm_recurrence_xcd does not apply to batch={batch}, hidden={hidden} (hidden <= 512, batch <= 4096)')
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def lstm_recurrence_xcd(gates_ws, packed_whh16, cell_ws, h_out, xcd_ws, flags=0):
    """The recurrence as ONE resident launch, a tile of 16 utterances per XCD, on the fp16 matrix cores (nbasr.h, ABI 6): h_out agrees
    with lstm_recurrence_packed to fp32 round-off.  The status word of ``xcd_ws`` is read by ``lstm_seq_status``."""
    b, frames, hidden = h_out.shape
    lib = load_library()
    if not packed_whh16.is_cuda or packed_whh16.dtype != torch.uint8 or packed_whh16.numel() != lib.nbasr_lstm_packed_whh16_bytes(hidden):
        raise HipError('packed_whh16 must be the uint8 device tensor returned by lstm_pack_whh16 for this hidden size')
    need = lstm_xcd_workspace_bytes(b, hidden)
    if need == 0 or not xcd_ws.is_cuda or xcd_ws.dtype != torch.uint8 or xcd_ws.numel() < need:
        raise HipError(f'lstm_recurrence_xcd: batch={b} hidden={hidden} needs a uint8 device workspace of {need} bytes from lstm_xcd_workspace '
                       f'(0 = the form does not apply)')
    _check(lib.nbasr_lstm_recurrence_xcd(_dev(gates_ws, 'gates_ws'), packed_whh16.data_ptr(), _dev(cell_ws, 'cell_ws'), _dev(h_out, 'h_out'),
                                         xcd_ws.data_ptr(), b, frames, hidden, int(flags), _stream(h_out)), 'nbasr_lstm_recurrence_xcd')
    return h_out


def lstm_recurrence_frames16(gates_ws, packed_whh16, cell_ws, h_out, xcd_ws):
    """lstm_recurrence_xcd's arithmetic as one launch per frame (bit-identical h_out): the form of a pipelined tail and of a demoted plan."""
    b, frames, hidden = h_out.shape
    lib = load_library()
    if not packed_whh16.is_cuda or packed_whh16.dtype != torch.uint8 or packed_whh16.numel() != lib.nbasr_lstm_packed_whh16_bytes(hidden):
        raise HipError('packed_whh16 must be the uint8 device tensor returned by lstm_pack_whh16 for this hidden size')
    need = lstm_xcd_workspace_bytes(b, hidden)
    if need == 0 or not xcd_ws.is_cuda or xcd_ws.dtype != torch.uint8 or xcd_ws.numel() < need:
        raise HipError(f'lstm_recurrence_frames16: batch={b} hidden={hidden} needs a uint8 device workspace of {need} bytes from lstm_xcd_workspace')
    _check(lib.nbasr_lstm_recurrence_frames16(_dev(gates_ws, 'gates_ws'), packed_whh16.data_ptr(), _dev(cell_ws, 'cell_ws'), _dev(h_out, 'h_out'),
                                              xcd_ws.data_ptr(), b, frames, hidden, _stream(h_out)), 'nbasr_lstm_recurrence_frames16')
    return h_out


def lstm_recurrence(gates_ws, w_hh, cell_ws, h_out):
    """h_out (batch, frames, hidden) from the projected gates (second half of lstm_forward), on the current stream; packs w_hh per call
    (a caller that keeps the weight packs it once: lstm_pack_whh + lstm_recurrence_packed)."""
    return lstm_recurrence_packed(gates_ws, lstm_pack_whh(w_hh), cell_ws, h_out)


def pack_pointwise_weights(weight):
    """(c_out, c_in) fp32 weight of a per-frame linear map -> opaque uint8 tensor (2 x fp16 split in LDS-image order + row scales)."""
    c_out, c_in = weight.shape
    nbytes = load_library().nbasr_pointwise_packed_weights_bytes(c_out, c_in)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    _check(load_library().nbasr_pack_pointwise_weights(_dev(weight, 'weight'), packed.data_ptr(), c_out, c_in, _stream(weight)),
           'nbasr_pack_pointwise_weights')
    return packed


def pointwise_workspace(batch, c_in, ld, device):
    """Scratch for the pre-split activation image of linear_fused_packed / lstm_input_projection_packed."""
    return torch.empty(max(load_library().nbasr_pointwise_workspace_bytes(batch, c_in, ld), 16), dtype=torch.uint8, device=device)


def _check_pointwise(packed, ws, c_out, c_in, batch, ld):
    lib = load_library()
    if not packed.is_cuda or packed.dtype != torch.uint8 or packed.numel() != lib.nbasr_pointwise_packed_weights_bytes(c_out, c_in):
        raise HipError(f'packed weights are not the pack_pointwise_weights image of a ({c_out}, {c_in}) weight')
    if not ws.is_cuda or ws.dtype != torch.uint8 or ws.numel() < lib.nbasr_pointwise_workspace_bytes(batch, c_in, ld):
        raise HipError('workspace too small: allocate it with pointwise_workspace(batch, c_in, ld, device)')


def linear_fused_packed(x, frames, packed, c_out, bias, skips, y, ws, ln=None, ln_on_x=False, ln_on_skip0=False):
    """The `linear` node op on the fp16 matrix cores: y = min(relu(W x + b), 20) + skips (see nbasr.h)."""
    b, c_in, ld = x.shape
    _check_pointwise(packed, ws, c_out, c_in, b, ld)
    s = list(skips) + [None] * (3 - len(skips))
    _check(load_library().nbasr_linear_fused_packed(
        _dev(x, 'x'), ws.data_ptr(), packed.data_ptr(), _dev(bias, 'bias'), _opt(s[0], 'skip0'), _opt(s[1], 'skip1'),
        _opt(s[2], 'skip2'), _dev(y, 'y'), b, c_in, frames, ld, c_out, _ln(ln), int(ln_on_x), int(ln_on_skip0), _stream(x)),
        'nbasr_linear_fused_packed')
    return y


def lstm_input_projection_packed(x, frames, packed_w_ih, b_ih, b_hh, gates_ws, hidden, ws, ln=None, batch_total=None, batch_offset=0):
    """``batch_total``: gates_ws is the (frames, batch_total, 4 hidden) gate tensor of SEVERAL forwards and this call fills utterances
    batch_offset .. batch_offset + batch - 1 of it (one recurrence over all of them follows: nbasr.h)."""
    b, c_in, ld = x.shape
    _check_pointwise(packed_w_ih, ws, 4 * hidden, c_in, b, ld)
    if batch_total is not None:
        if gates_ws.numel() < frames * batch_total * 4 * hidden:
            raise HipError(f'lstm_input_projection_packed: gates_ws holds {gates_ws.numel()} floats, ({frames}, {batch_total}, {4 * hidden}) needs more')
        _check(load_library().nbasr_lstm_input_projection_packed_into(
            _dev(x, 'x'), ws.data_ptr(), packed_w_ih.data_ptr(), _dev(b_ih, 'b_ih'), _dev(b_hh, 'b_hh'), _dev(gates_ws, 'gates_ws'),
            b, c_in, frames, ld, hidden, _ln(ln), batch_total, batch_offset, _stream(x)), 'nbasr_lstm_input_projection_packed_into')
        return gates_ws
    _check(load_library().nbasr_lstm_input_projection_packed(
        _dev(x, 'x'), ws.data_ptr(), packed_w_ih.data_ptr(), _dev(b_ih, 'b_ih'), _dev(b_hh, 'b_hh'), _dev(gates_ws, 'gates_ws'),
        b, c_in, frames, ld, hidden, _ln(ln), _stream(x)), 'nbasr_lstm_input_projection_packed')
    return gates_ws


def pack_pointwise_weights_bf16(weight):
    """(c_out, c_in) fp32 values of a bf16 weight -> opaque uint8 tensor (bf16, LDS-image order) for the bf16 per-frame maps."""
    c_out, c_in = weight.shape
    packed = torch.empty(load_library().nbasr_pointwise_bf16_weights_bytes(c_out, c_in), dtype=torch.uint8, device=weight.device)
    _check(load_library().nbasr_pack_pointwise_weights_bf16(_dev(weight, 'weight'), packed.data_ptr(), c_out, c_in, _stream(weight)),
           'nbasr_pack_pointwise_weights_bf16')
    return packed


def pointwise_bf16_workspace(batch, c_in, ld, device):
    return torch.empty(max(load_library().nbasr_pointwise_bf16_workspace_bytes(batch, c_in, ld), 16), dtype=torch.uint8, device=device)


def _check_pointwise_bf16(packed, ws, c_out, c_in, batch, ld):
    lib = load_library()
    if not packed.is_cuda or packed.dtype != torch.uint8 or packed.numel() != lib.nbasr_pointwise_bf16_weights_bytes(c_out, c_in):
        raise HipError(f'packed weights are not the pack_pointwise_weights_bf16 image of a ({c_out}, {c_in}) weight')
    if not ws.is_cuda or ws.dtype != torch.uint8 or ws.numel() < lib.nbasr_pointwise_bf16_workspace_bytes(batch, c_in, ld):
        raise HipError('workspace too small: allocate it with pointwise_bf16_workspace(batch, c_in, ld, device)')


def linear_fused_bf16(x, frames, packed, c_out, bias, skips, y, ws, ln=None, ln_on_x=False, ln_on_skip0=False):
    """The `linear` node op on bf16 rows, one bf16 MFMA per product: y = bf16(min(relu(W x + b), 20) + skips) (see nbasr.h)."""
    b, c_in, ld = x.shape
    bf = torch.bfloat16
    _check_pointwise_bf16(packed, ws, c_out, c_in, b, ld)
    s = list(skips) + [None] * (3 - len(skips))
    _check(load_library().nbasr_linear_fused_bf16(
        _act(x, 'x', bf), ws.data_ptr(), packed.data_ptr(), _dev(bias, 'bias'), _act_opt(s[0], 'skip0', bf), _act_opt(s[1], 'skip1', bf),
        _act_opt(s[2], 'skip2', bf), _act(y, 'y', bf), b, c_in, frames, ld, c_out, _ln(ln), int(ln_on_x), int(ln_on_skip0), _stream(x)),
        'nbasr_linear_fused_bf16')
    return y


def lstm_input_projection_bf16(x, frames, packed_w_ih, b_ih, b_hh, gates_ws, hidden, ws, ln=None):
    """gates (frames, batch, 4 hidden) fp32 from the bf16 encoder output x (batch, c_in, ld), its pending LayerNorm applied on the way."""
    b, c_in, ld = x.shape
    _check_pointwise_bf16(packed_w_ih, ws, 4 * hidden, c_in, b, ld)
    _check(load_library().nbasr_lstm_input_projection_bf16(
        _act(x, 'x', torch.bfloat16), ws.data_ptr(), packed_w_ih.data_ptr(), _dev(b_ih, 'b_ih'), _dev(b_hh, 'b_hh'), _dev(gates_ws, 'gates_ws'),
        b, c_in, frames, ld, hidden, _ln(ln), _stream(x)), 'nbasr_lstm_input_projection_bf16')
    return gates_ws


def _lengths_ptr(lengths, batch):
    if lengths is None:
        return None
    if not lengths.is_cuda or lengths.dtype != torch.int32 or not lengths.is_contiguous() or lengths.numel() != batch:
        raise HipError('lengths must be a contiguous int32 device tensor with one entry per utterance')
    return lengths.data_ptr()


def frame_signal(wave, lengths, frames, win, hop):
    """wave (B, L) -> frames (B, win, ld): centred, reflect-padded analysis frames, sample-in-window major."""
    b, ld_wave = wave.shape
    _check(load_library().nbasr_frame_signal(_dev(wave, 'wave'), _lengths_ptr(lengths, b), _dev(frames, 'frames'), b, ld_wave,
                                             ld_wave, win, hop, frames.shape[2], _stream(wave)), 'nbasr_frame_signal')
    return frames


def pointwise_linear(x, frames, weight, bias, y):
    """y (B, c_out, ld_out) = weight (c_out, c_in) . x (B, c_in, ld_in) + bias, no activation."""
    b, c_in, ld_in = x.shape
    _check(load_library().nbasr_pointwise_linear(_dev(x, 'x'), _dev(weight, 'weight'), _dev(bias, 'bias'), _dev(y, 'y'), b, c_in,
                                                 frames, ld_in, weight.shape[0], y.shape[2], _stream(x)), 'nbasr_pointwise_linear')
    return y


def power_spectrum(spec, bins, power):
    b, _, ld = spec.shape
    _check(load_library().nbasr_power_spectrum(_dev(spec, 'spec'), _dev(power, 'power'), b, bins, power.shape[1], ld, _stream(spec)),
           'nbasr_power_spectrum')
    return power


def log_normalize(mel, lengths, mean, inv_scale, feats, samples, hop):
    b, n_mels, ld = mel.shape
    _check(load_library().nbasr_log_normalize(_dev(mel, 'mel'), _lengths_ptr(lengths, b), _dev(mean, 'mean'),
                                              _dev(inv_scale, 'inv_scale'), _dev(feats, 'feats'), b, samples, hop, n_mels, ld,
                                              _stream(mel)), 'nbasr_log_normalize')
    return feats


def linear_head(h, weight, bias, logits):
    classes, features = weight.shape
    rows = h.numel() // features
    _check(load_library().nbasr_linear_head(_dev(h, 'h'), _dev(weight, 'weight'), _dev(bias, 'bias'),
                                            _dev(logits, 'logits'), rows, features, classes, _stream(h)),
           'nbasr_linear_head')
    return logits


def linear_head_bct(x, frames, weight, bias, logits, ln=None):
    b, features, ld = x.shape
    classes = weight.shape[0]
    _check(load_library().nbasr_linear_head_bct(_dev(x, 'x'), _dev(weight, 'weight'), _dev(bias, 'bias'),
                                                _dev(logits, 'logits'), b, features, frames, ld, classes, _ln(ln),
                                                _stream(x)), 'nbasr_linear_head_bct')
    return logits


def repitch(src, dst, frames):
    """src (..., ld_src) -> dst (..., ld_dst), float32 or bfloat16: copy `frames` columns, zero the rest of dst's pitch."""
    rows = src.numel() // src.shape[-1]
    _check(load_library().nbasr_repitch(_act(src, 'src'), _act(dst, 'dst', src.dtype), rows, frames, src.shape[-1], dst.shape[-1],
                                        dtype_code(src.dtype), _stream(src)), 'nbasr_repitch')
    return dst


def ctc_postprocess(logits, lengths=None, want_log_probs=True, want_tokens=True, blank=0):
    """logits (B, T', C) -> (log_probs or None, tokens (B, T') int32 padded with -1 or None, token_counts (B) or None)."""
    _dev(logits, 'logits')
    b, t, c = logits.shape
    if lengths is not None:
        if not lengths.is_cuda or lengths.dtype != torch.int32 or not lengths.is_contiguous() or lengths.numel() != b:
            raise HipError('lengths must be a contiguous int32 device tensor with one entry per utterance')
    log_probs = torch.empty_like(logits) if want_log_probs else None
    tokens = torch.empty(b, t, dtype=torch.int32, device=logits.device) if want_tokens else None
    counts = torch.empty(b, dtype=torch.int32, device=logits.device) if want_tokens else None
    _check(load_library().nbasr_ctc_postprocess(
        logits.data_ptr(), None if lengths is None else lengths.data_ptr(), None if log_probs is None else log_probs.data_ptr(),
        None if tokens is None else tokens.data_ptr(), None if counts is None else counts.data_ptr(), b, t, c, blank,
        _stream(logits)), 'nbasr_ctc_postprocess')
    return log_probs, tokens, counts


def _int_tensor(t, what, device, shape=None):
    if not t.is_cuda or t.device != device or t.dtype != torch.int32 or not t.is_contiguous():
        raise HipError(f'{what} must be a contiguous int32 tensor on {device}')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise HipError(f'{what} must have shape {tuple(shape)}, got {tuple(t.shape)}')
    return t


def ctc_beam_search(log_probs, lengths=None, beam_width=12, blank=0, cutoff_top_n=40):
    """log_probs (B, T', C) float32 log-probabilities -> (beams (B, W, T') int32 best first, scores (B, W) = -log P,
    beam_lens (B, W) int32).  ``lengths``: int32 device tensor (B) of valid output frames, or None."""
    _dev(log_probs, 'log_probs')
    b, t, c = log_probs.shape
    if lengths is not None:
        _int_tensor(lengths, 'lengths', log_probs.device, (b,))
    dev = log_probs.device
    beams = torch.empty(b, beam_width, t, dtype=torch.int32, device=dev)
    scores = torch.empty(b, beam_width, dtype=torch.float32, device=dev)
    lens = torch.empty(b, beam_width, dtype=torch.int32, device=dev)
    lib = load_library()
    ws = torch.empty(max(lib.nbasr_ctc_beam_workspace_bytes(b, t, c, beam_width), 8) // 8, dtype=torch.int64, device=dev)
    _check(lib.nbasr_ctc_beam_search(log_probs.data_ptr(), None if lengths is None else lengths.data_ptr(), ws.data_ptr(),
                                     beams.data_ptr(), scores.data_ptr(), lens.data_ptr(), b, t, c, beam_width, blank, cutoff_top_n,
                                     _stream(log_probs)), 'nbasr_ctc_beam_search')
    return beams, scores, lens


def token_error_counts(hyp, hyp_len, ref, ref_len, table=None, blank=0):
    """hyp (B, Lh), ref (B, Lr) int32 label matrices with their lengths (B) -> counts (B, 2) int32 = (Levenshtein distance,
    reference length) after mapping through ``table`` (int32, optional) and dropping ``blank``."""
    if hyp.dim() != 2 or ref.dim() != 2 or hyp.shape[0] != ref.shape[0]:
        raise HipError('hyp and ref must be (batch, length) matrices over the same batch')
    dev = hyp.device
    b = hyp.shape[0]
    for t, what in ((hyp, 'hyp'), (ref, 'ref')):
        _int_tensor(t, what, dev)
    _int_tensor(hyp_len, 'hyp_len', dev, (b,))
    _int_tensor(ref_len, 'ref_len', dev, (b,))
    if table is not None:
        _int_tensor(table, 'table', dev)
    counts = torch.empty(b, 2, dtype=torch.int32, device=dev)
    _check(load_library().nbasr_token_error_counts(
        hyp.data_ptr(), hyp_len.data_ptr(), hyp.shape[1], ref.data_ptr(), ref_len.data_ptr(), ref.shape[1],
        None if table is None else table.data_ptr(), 0 if table is None else table.numel(), blank, counts.data_ptr(), b,
        torch.cuda.current_stream(dev).cuda_stream), 'nbasr_token_error_counts')
    return counts


def ctc_loss(log_probs, lengths, targets, target_lengths, blank=0, divide_by_length=False):
    """log_probs (B, T', C) float32, lengths (B) int32, targets (B, L) int32, target_lengths (B) int32 (all on the device)
    -> per-utterance negative log-likelihood (B) float32 with zero_infinity semantics, optionally divided by ``lengths``."""
    _dev(log_probs, 'log_probs')
    b, t, c = log_probs.shape
    dev = log_probs.device
    _int_tensor(lengths, 'lengths', dev, (b,))
    _int_tensor(target_lengths, 'target_lengths', dev, (b,))
    _int_tensor(targets, 'targets', dev)
    if targets.dim() != 2 or targets.shape[0] != b:
        raise HipError(f'targets must be (batch, labels), got {tuple(targets.shape)}')
    losses = torch.empty(b, dtype=torch.float32, device=dev)
    _check(load_library().nbasr_ctc_loss(log_probs.data_ptr(), lengths.data_ptr(), targets.data_ptr(), target_lengths.data_ptr(),
                                         losses.data_ptr(), b, t, c, targets.shape[1], blank, 1 if divide_by_length else 0,
                                         _stream(log_probs)), 'nbasr_ctc_loss')
    return losses


def ctc_loss_grad(log_probs, lengths, targets, target_lengths, blank=0):
    """-> (per-utterance loss / length (B), d mean(loss / length) / d logits (B, T', C)) for log_probs = log_softmax(logits)."""
    _dev(log_probs, 'log_probs')
    b, t, c = log_probs.shape
    dev = log_probs.device
    _int_tensor(lengths, 'lengths', dev, (b,))
    _int_tensor(target_lengths, 'target_lengths', dev, (b,))
    _int_tensor(targets, 'targets', dev)
    if targets.dim() != 2 or targets.shape[0] != b:
        raise HipError(f'targets must be (batch, labels), got {tuple(targets.shape)}')
    lib = load_library()
    losses = torch.empty(b, dtype=torch.float32, device=dev)
    grad = torch.empty_like(log_probs)
    ws = torch.empty(max(lib.nbasr_ctc_grad_workspace_bytes(b, t, targets.shape[1]) // 4, 1), dtype=torch.float32, device=dev)
    _check(lib.nbasr_ctc_loss_grad(log_probs.data_ptr(), lengths.data_ptr(), targets.data_ptr(), target_lengths.data_ptr(),
                                   ws.data_ptr(), losses.data_ptr(), grad.data_ptr(), b, t, c, targets.shape[1], blank,
                                   _stream(log_probs)), 'nbasr_ctc_loss_grad')
    return losses, grad


# ---------------------------------------------------------------------------------------------
# storage-type generic entry points (float32 | bfloat16 activations; SURVEY 8 row g1 / BASELINE config 4)
# ---------------------------------------------------------------------------------------------
def pack_grouped_weights(weight, groups):
    """(C, C/groups, k) fp32 weight -> [group][ci][tap][co] copy for the GC_WPERM kernel variants."""
    c, cg, k = weight.shape
    packed = torch.empty_like(weight)
    _check(load_library().nbasr_pack_grouped_weights(_dev(weight, 'weight'), _dev(packed, 'packed'), c, groups, k, _stream(weight)),
           'nbasr_pack_grouped_weights')
    return packed


def grouped_conv1d_node(x, weight, bias, skips, y, frames, groups, kernel, dilation, ln=None, ln_on_x=False, ln_on_skip0=False,
                        stats_ws=None, variant=0):
    """The grouped-conv node op for float32 or bfloat16 activations; ``weight`` / ``bias`` are float32 (``weight`` in the
    layout the variant wants: torch's, or pack_grouped_weights' for GC_WPERM)."""
    b, c, ld = x.shape
    dt = x.dtype
    s = list(skips) + [None] * (3 - len(skips))
    _check(load_library().nbasr_grouped_conv1d_node(
        _act(x, 'x'), _dev(weight, 'weight'), _dev(bias, 'bias'), _act_opt(s[0], 'skip0', dt), _act_opt(s[1], 'skip1', dt),
        _act_opt(s[2], 'skip2', dt), _act(y, 'y', dt), b, c, frames, ld, groups, kernel, dilation, _ln(ln), int(ln_on_x),
        int(ln_on_skip0), _opt(stats_ws, 'stats_ws'), dtype_code(dt), variant, _stream(x)), 'nbasr_grouped_conv1d_node')
    return y


def convert(x, y):
    """Storage conversion float32 <-> bfloat16 of equally shaped contiguous tensors (numel % 8 == 0)."""
    if x.shape != y.shape:
        raise HipError('convert: shapes differ')
    _check(load_library().nbasr_convert(_act(x, 'x'), _act(y, 'y'), x.numel(), dtype_code(x.dtype), dtype_code(y.dtype), _stream(x)),
           'nbasr_convert')
    return y


def bf16_image_bytes(batch, channels, ld):
    return load_library().nbasr_bf16_image_bytes(batch, channels, ld)


def bf16_image(x, image, frames, norm=None, stats=None, eps=0.0):
    """Operand image of the bf16 dense conv from x (B, C, ld); ``norm`` = (gamma, beta): image of LayerNorm(x), ``stats`` filled."""
    b, c, ld = x.shape
    if image.dtype != torch.uint8 or image.numel() < bf16_image_bytes(b, c, ld):
        raise HipError('image buffer too small: needs bf16_image_bytes(batch, channels, ld) bytes of uint8')
    gamma, beta = norm if norm is not None else (None, None)
    _check(load_library().nbasr_bf16_image(_act(x, 'x'), _opt(gamma, 'gamma'), _opt(beta, 'beta'), _opt(stats, 'stats'),
                                           image.data_ptr(), b, c, frames, ld, float(eps), dtype_code(x.dtype), _stream(x)),
           'nbasr_bf16_image')
    return image


def pack_dense_weights_bf16(weight, stride, row_tile=128):
    """(c_out, c_in, 8) fp32 weight (the values of a bf16 parameter) -> packed one-term bf16 image of the stride-`stride` kernel."""
    return pack_dense_weights(weight, stride, 'bf16', row_tile)


def dense_conv1d_bf16_img(image, batch, c_in, frames_in, ld_in, packed, c_out, kernel, bias, y, stride, row_tile=128, frame_tile=256):
    _check_packed(packed, 'bf16', c_out, c_in, kernel, row_tile)
    if image.numel() < load_library().nbasr_bf16_image_bytes(batch, c_in, ld_in):
        raise HipError('image buffer too small for (batch, c_in, ld_in)')
    return _dense_packed('bf16', image.data_ptr(), True, None, None, packed, bias, (None, None, None), y, _act(y, 'y', torch.bfloat16), batch,
                         c_in, frames_in, ld_in, c_out, kernel, stride, row_tile, None, _stream(y), None, frame_tile)


# ---------------------------------------------------------------------------------------------
# backward building blocks
# ---------------------------------------------------------------------------------------------
def grouped_conv1d_backward(x, weight, z, dz, frames, groups, kernel, dilation, need_dx=True, need_dw=True):
    """Gradients of z = min(relu(grouped_conv(x, weight) + bias), 20) given dz: (dx or None, dw or None, db or None).
    x, z, dz: (B, C, ld) float32 pitched tensors (z = the op's output; dz's pitch columns zero)."""
    b, c, ld = z.shape
    lib = load_library()
    dx = torch.empty_like(z) if need_dx else None
    dw = torch.empty_like(weight) if need_dw else None
    db = torch.empty(c, dtype=torch.float32, device=z.device) if need_dw else None
    ws = None
    if need_dw:
        ws = torch.empty(max(lib.nbasr_grouped_conv1d_backward_workspace_bytes(max(b, 1), c, groups, kernel) // 4, 4), dtype=torch.float32,
                         device=z.device)
    _check(lib.nbasr_grouped_conv1d_backward(_opt(x, 'x'), _dev(weight, 'weight'), _dev(z, 'z'), _dev(dz, 'dz'), _opt(dx, 'dx'), _opt(dw, 'dw'),
                                             _opt(db, 'db'), _opt(ws, 'workspace'), b, c, frames, ld, groups, kernel, dilation, _stream(z)),
           'nbasr_grouped_conv1d_backward')
    return dx, dw, db


def layernorm_channels_backward(x, stats, gamma, dy, frames):
    """(dx, dgamma, dbeta) of y = LayerNorm_channels(x) given dy; stats = channel_stats(x)."""
    b, c, ld = x.shape
    lib = load_library()
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = torch.empty(max(lib.nbasr_layernorm_backward_workspace_bytes(max(b, 1), c, max(ld, 4)) // 4, 4), dtype=torch.float32, device=x.device)
    _check(lib.nbasr_layernorm_channels_backward(_dev(x, 'x'), _dev(stats, 'stats'), _dev(gamma, 'gamma'), _dev(dy, 'dy'), _dev(dx, 'dx'),
                                                 _dev(dgamma, 'dgamma'), _dev(dbeta, 'dbeta'), _dev(ws, 'workspace'), b, c, frames, ld, _stream(x)),
           'nbasr_layernorm_channels_backward')
    return dx, dgamma, dbeta


def dense_conv1d_backward(x, weight, y, dy, frames_in, stride, need_dx=True, need_dw=True, activation=True):
    """Backward of ``y = min(relu(conv1d(zero_pad(x), weight, bias, stride)), 20)`` for the dense k = 8 downsample convs (stride 1 | 2) and
    the per-frame ``linear`` op (k = 1): x (B, C_in, ld_in), y / dy (B, C_out, ld_out) pitched -> (dx, dw, db).  ``activation=False``: the map
    without ReLU / clamp (the CTC head).

    Correctness-first.  Weight and bias gradients: ONE (C_out, B * T') x (B * T', C_in * k + 1) GEMM on a materialised
    column matrix.  Input gradient of the k = 8 convs: ONE (C_in * 8, C_out) x (C_out, B * T') GEMM, then a fold of the 8 tap rows onto
    the input frames (nbasr_conv_fold).  Both GEMMs run on the fp16 matrix cores with the fp32-accurate two-term split;
    ``NBASR_DENSE_MODE=f32`` keeps every product on the exact-fp32 MFMA GEMMs of the forward (there the input gradient is a stride-1 conv
    of the zero-stuffed, masked output gradient with the flipped, channel-transposed kernel)."""
    lib = load_library()
    b, c_in, ld_in = x.shape
    c_out, _, kernel = weight.shape if weight.dim() == 3 else (weight.shape[0], weight.shape[1], 1)
    ld_out = y.shape[2]
    frames_out = (frames_in + stride - 1) // stride
    lpad = pad_amounts(kernel, 1, stride)[0]
    stream = _stream(x)
    if activation:
        dz = torch.empty_like(y)
        _check(lib.nbasr_relu_clamp_backward(_dev(y, 'y'), _dev(dy, 'dy'), _dev(dz, 'dz'), y.numel(), stream), 'nbasr_relu_clamp_backward')
    else:
        dz = dy                                             # a plain linear map (the CTC head): no mask
    dx = dw = db = None
    if need_dx:
        dx = torch.empty(b, c_in, ld_in, device=x.device, dtype=torch.float32)
        zero = torch.zeros(c_in, device=x.device, dtype=torch.float32)
        if kernel == 1:
            wt = weight.detach().reshape(c_out, c_in).t().contiguous()
            _check(lib.nbasr_pointwise_linear(_dev(dz, 'dz'), _dev(wt, 'wt'), _dev(zero, 'zero'), _dev(dx, 'dx'), b, c_out, frames_in, ld_out,
                                              c_in, ld_in, stream), 'nbasr_pointwise_linear')
        elif kernel == 8 and (c_in * kernel) % 16 == 0 and os.environ.get('NBASR_DENSE_MODE', 'auto') != 'f32':
            # ONE GEMM on the fp16 matrix cores (fp32-accurate two-term split, the kernel of the LSTM input projection): rows (ci, tap) of
            # w^T times the masked output gradient, stored time-major, every tap's contribution to dx in its own row; nbasr_conv_fold adds
            # the 8 (stride 1) or 4 (stride 2) rows that land on one input frame.  No zero-stuffing: half the products at stride 2.
            rows_w = c_in * kernel
            wt = weight.detach().permute(1, 2, 0).reshape(rows_w, c_out).contiguous()
            packed = pack_pointwise_weights(wt)
            ws = pointwise_workspace(b, c_out, ld_out, x.device)
            zero_r = torch.zeros(rows_w, device=x.device, dtype=torch.float32)
            taps = torch.empty(max(frames_out, 1), b, rows_w, device=x.device, dtype=torch.float32)
            lstm_input_projection_packed(dz, frames_out, packed, zero_r, zero_r, taps, rows_w // 4, ws)
            _check(lib.nbasr_conv_fold(_dev(taps, 'taps'), _dev(dx, 'dx'), b, c_in, frames_in, ld_in, frames_out, kernel, stride, lpad, stream),
                   'nbasr_conv_fold')
        else:
            up = torch.empty(b, c_out, ld_in, device=x.device, dtype=torch.float32)
            _check(lib.nbasr_zero_stuff(_dev(dz, 'dz'), _dev(up, 'up'), b * c_out, frames_out, ld_out, frames_in, ld_in, stride, 0, stream),
                   'nbasr_zero_stuff')
            wf = weight.detach().flip(2).permute(1, 0, 2).contiguous()                 # (C_in, C_out, k): flipped taps, channels swapped
            _check(lib.nbasr_dense_conv1d_linear(_dev(up, 'up'), _dev(wf, 'wf'), _dev(zero, 'zero'), _dev(dx, 'dx'), b, c_out, frames_in, ld_in,
                                                 c_in, ld_in, kernel, kernel - 1 - lpad, stream), 'nbasr_dense_conv1d_linear')
    if need_dw:
        t_pad = round_up4(max(frames_out, 1))
        ld_cols = round_up4(c_in * kernel + 1)
        cols = torch.empty(b * t_pad, ld_cols, device=x.device, dtype=torch.float32)
        _check(lib.nbasr_conv_cols(_dev(x, 'x'), _dev(cols, 'cols'), b, c_in, frames_in, ld_in, frames_out, t_pad, kernel, stride, lpad, ld_cols,
                                   stream), 'nbasr_conv_cols')
        rows = torch.empty(c_out, b * t_pad, device=x.device, dtype=torch.float32)
        _check(lib.nbasr_rows_of_channels(_dev(dz, 'dz'), _dev(rows, 'rows'), b, c_out, frames_out, ld_out, t_pad, stream), 'nbasr_rows_of_channels')
        zero_o = torch.zeros(c_out, device=x.device, dtype=torch.float32)
        if os.environ.get('NBASR_DENSE_MODE', 'auto') != 'f32':
            # the (C_out, B T') x (B T', C_in k + 1) product on the fp16 matrix cores, fp32-accurate two-term split (the GEMM of the LSTM
            # input projection: "weights" = the masked output gradient packed per call, "x" = the column matrix pre-split per column tile);
            # it stores time-major, i.e. the transpose (C_in k + 1, C_out).  3-5 x the exact-fp32 MFMA GEMM, which was half of a training step
            c_pad = (c_out + 15) & ~15                       # (that entry point takes 4 * hidden rows, hidden % 4 == 0)
            if c_pad != c_out:
                rows = torch.cat([rows, rows.new_zeros(c_pad - c_out, rows.shape[1])])
            zero_p = torch.zeros(c_pad, device=x.device, dtype=torch.float32)
            packed = pack_pointwise_weights(rows)
            ws = pointwise_workspace(1, b * t_pad, ld_cols, x.device)
            out_t = torch.empty(ld_cols, 1, c_pad, device=x.device, dtype=torch.float32)
            lstm_input_projection_packed(cols.view(1, b * t_pad, ld_cols), c_in * kernel + 1, packed, zero_p, zero_p, out_t, c_pad // 4, ws)
            out = out_t.view(ld_cols, c_pad).t()[:c_out]
        else:
            out3 = torch.empty(1, c_out, ld_cols, device=x.device, dtype=torch.float32)
            _check(lib.nbasr_pointwise_linear(_dev(cols, 'cols'), _dev(rows, 'rows'), _dev(zero_o, 'zero'), _dev(out3, 'out'), 1, b * t_pad,
                                              c_in * kernel + 1, ld_cols, c_out, ld_cols, stream), 'nbasr_pointwise_linear')
            out = out3[0]
        dw = out[:, : c_in * kernel].reshape(c_out, c_in, kernel).contiguous()
        if weight.dim() == 2:
            dw = dw.reshape(c_out, c_in)
        db = out[:, c_in * kernel].contiguous()
    return dx, dw, db


def lstm_backward(xp, frames, gates, h_out, w_ih, w_hh, dh_out):
    """BPTT of the single-layer LSTM (reference model.py:100,118-121): xp (B, C, ld) the layer input, gates (T, B, 4H) its saved input
    projection (both biases included), h_out (B, T, H) the saved output, dh_out (B, T, H) -> (dx (B, C, T), dw_ih, dw_hh, db).

    Correctness first: gate pre-activations of all frames are recomputed from the saved h by ONE GEMM, a serial
    scan restores the cell states, the reverse recurrence is T launches of a step kernel that forms w_hh^T . dpre of the next frame in place, and the weight / input gradients
    are batched GEMMs -- on the fp16 matrix cores with the fp32-accurate two-term split (``NBASR_DENSE_MODE=f32``: the exact-fp32 MFMA GEMM
    nbasr_pointwise_linear); tensor re-layouts are torch copies."""
    lib = load_library()
    b, c, _ = xp.shape
    t_n, hidden = frames, w_hh.shape[1]
    g4 = 4 * hidden
    dev, f32 = xp.device, torch.float32
    if hidden % 4:
        raise HipError('lstm_backward: hidden must be a multiple of 4')
    ldb = round_up4(b)
    n = t_n * ldb                                              # GEMM column count; (frame, utterance) pairs, utterance innermost
    stream = _stream(xp)

    def gemm(x_ptr, c_in, cols, ld_in, w, y):
        """y (1, c_out, ld_out) = w (c_out, c_in) . x (c_in rows of `cols` floats at pitch ld_in)"""
        zero = torch.zeros(w.shape[0], device=dev, dtype=f32)
        _check(lib.nbasr_pointwise_linear(x_ptr, _dev(w, 'w'), _dev(zero, 'zero'), _dev(y, 'y'), 1, c_in, cols, ld_in, w.shape[0], y.shape[2],
                                          stream), 'nbasr_pointwise_linear')
        return y

    split = os.environ.get('NBASR_DENSE_MODE', 'auto') != 'f32'

    def gemm_t(x3, cols, w):
        """(w (rows, K) . x3 (1, K, ld))^T -> (cols, rows): the same product on the fp16 matrix cores (fp32-accurate two-term split; the
        GEMM of the LSTM input projection, which stores time-major, i.e. transposed); w is packed per call."""
        rows, k = w.shape
        r_pad = (rows + 15) & ~15                               # (that entry point takes 4 * hidden rows, hidden % 4 == 0)
        if r_pad != rows:
            w = torch.cat([w, w.new_zeros(r_pad - rows, k)])
        zero = torch.zeros(r_pad, device=dev, dtype=f32)
        out_t = torch.empty(x3.shape[2], 1, r_pad, device=dev, dtype=f32)
        lstm_input_projection_packed(x3, cols, pack_pointwise_weights(w.contiguous()), zero, zero, out_t, r_pad // 4,
                                     pointwise_workspace(1, k, x3.shape[2], dev))
        return out_t.view(x3.shape[2], r_pad)[:cols, :rows]

    # h_(t-1) for every (t, b): rows of the (T * ldb, H) matrix, zero for t = 0 and for the pitch utterances
    hp = torch.zeros(t_n, ldb, hidden, device=dev, dtype=f32)
    if t_n > 1:
        hp[1:, :b] = h_out[:, : t_n - 1].permute(1, 0, 2)
    hp_t = hp.reshape(n, hidden).t().contiguous()              # (H, T * ldb)
    if split:
        pre = gemm_t(hp_t.view(1, hidden, n), n, w_hh.detach()).t().contiguous()
    else:
        pre = torch.empty(1, g4, n, device=dev, dtype=f32)
        gemm(hp_t.data_ptr(), hidden, n, n, w_hh.detach().contiguous(), pre)
    pre = pre.view(g4, t_n, ldb)
    pre[:, :, :b] += gates[:t_n].permute(2, 0, 1)             # + input projection and biases
    cells = torch.zeros(hidden, t_n, ldb, device=dev, dtype=f32)
    _check(lib.nbasr_lstm_gate_scan(_dev(pre, 'pre'), _dev(cells, 'cells'), hidden, t_n, b, ldb, stream), 'nbasr_lstm_gate_scan')
    acts = pre                                                 # overwritten in place by the scan
    dho = torch.zeros(hidden, t_n, ldb, device=dev, dtype=f32)
    dho[:, :, :b] = dh_out.detach().permute(2, 1, 0)
    dpre = torch.empty(g4, t_n, ldb, device=dev, dtype=f32)
    dc = torch.zeros(hidden, ldb, device=dev, dtype=f32)
    w_hh_t = w_hh.detach().t().contiguous()                    # (H, 4H)
    # (t = -1: the whole reverse recurrence, frames T-1 .. 0, in one call -- round 6; rounds 2-5 issued the frames from a python loop)
    _check(lib.nbasr_lstm_backward_step(_dev(dho, 'dho'), _dev(w_hh_t, 'w_hh_t'), _dev(dc, 'dc'), _dev(acts, 'acts'), _dev(cells, 'cells'),
                                        _dev(dpre, 'dpre'), hidden, t_n, b, ldb, -1, stream), 'nbasr_lstm_backward_step')
    d2 = dpre.view(g4, n)
    ldc = round_up4(c + 1)
    if split:
        # (dw_ih | db | dw_hh) (4H, C + 1 + H) = dpre (4H, n) . (x | 1 | h_prev) (n, .): ONE GEMM, dpre packed once
        xh = torch.zeros(t_n, ldb, ldc + hidden, device=dev, dtype=f32)
        xh[:, :b, :c] = xp[:, :, :t_n].permute(2, 0, 1)
        xh[:, :b, c] = 1.0
        xh[:, :, ldc:] = hp
        wb = gemm_t(xh.view(1, n, ldc + hidden), ldc + hidden, d2).t()
        dw_ih, db, dw_hh = wb[:, :c].contiguous(), wb[:, c].contiguous(), wb[:, ldc:].contiguous()
        # dx (C, n) = w_ih^T (C, 4H) . dpre (4H, n), delivered transposed: (n, C)
        dx = gemm_t(dpre.view(1, g4, n), n, w_ih.detach().t()).view(t_n, ldb, c)[:, :b].permute(1, 2, 0).contiguous()
        return dx, dw_ih, dw_hh, db
    # dw_hh (4H, H) = dpre (4H, n) . h_prev (n, H)
    dw_hh = gemm(hp.data_ptr(), n, hidden, hidden, d2, torch.empty(1, g4, hidden, device=dev, dtype=f32))[0]
    # (dw_ih | db) (4H, C + 1) = dpre . (x | 1)
    xt = torch.zeros(t_n, ldb, ldc, device=dev, dtype=f32)
    xt[:, :b, :c] = xp[:, :, :t_n].permute(2, 0, 1)
    xt[:, :b, c] = 1.0
    wb = gemm(xt.data_ptr(), n, c + 1, ldc, d2, torch.empty(1, g4, ldc, device=dev, dtype=f32))[0]
    dw_ih, db = wb[:, :c].contiguous(), wb[:, c].contiguous()
    # dx (C, n) = w_ih^T (C, 4H) . dpre (4H, n)
    dxc = gemm(dpre.data_ptr(), g4, n, n, w_ih.detach().t().contiguous(), torch.empty(1, c, n, device=dev, dtype=f32))[0]
    dx = dxc.view(c, t_n, ldb)[:, :, :b].permute(2, 0, 1).contiguous()
    return dx, dw_ih, dw_hh, db
