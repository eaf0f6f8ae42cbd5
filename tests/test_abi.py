"""The C-ABI library loads and exports exactly what include/nbasr.h declares (no GPU needed)."""
import ctypes
import pathlib
import re

import pytest

from nb_asr_amd import hip

HEADER = pathlib.Path(__file__).resolve().parent.parent / 'include' / 'nbasr.h'


def declared_symbols():
    text = re.sub(r'/\*.*?\*/', '', HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r'\b(nbasr_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    names = declared_symbols()
    assert len(names) >= 12
    assert sorted(hip.SIGNATURES) == names


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(str(built_library))
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert hip.load_library().nbasr_version() == hip.ABI_VERSION == 6


def test_loaded_library_was_built_from_these_sources(built_library):
    """The .so is git-ignored and travels to the GPU box as a binary: its compiled-in build id must be the hash of the
    sources in this checkout, or the numbers it produces cannot be tied to HEAD."""
    from nb_asr_amd import build
    assert hip.build_id() == build.source_hash()
    assert len(hip.build_id()) == 16 and hip.build_id() != 'unknown'


def test_signatures_have_no_torch_types():
    code = re.sub(r'/\*.*?\*/', '', HEADER.read_text(), flags=re.S)          # declarations only, comments stripped
    assert 'extern "C"' in code
    for forbidden in ('torch', 'Tensor', 'at::', 'c10::', 'std::'):
        assert forbidden not in code


def test_argument_errors_are_reported_without_a_gpu():
    lib = hip.load_library()
    # NULL pointers / bad sizes are rejected on the host before anything touches a device
    rc = lib.nbasr_grouped_conv1d_node(None, None, None, None, None, None, None, 1, 600, 16, 16, 100, 5, 1, None, 0, 0, None, 0, 0, None)
    assert rc == -3 and b'non-NULL' in lib.nbasr_last_error()
    rc = lib.nbasr_layernorm_channels(16, 16, 16, 16, None, 1, 600, 10, 10, 1e-3, 0, 0, None)      # ld not a multiple of 4
    assert rc == -2 and b'multiple of 4' in lib.nbasr_last_error()
    rc = lib.nbasr_grouped_conv1d_node(16, 16, 16, None, None, None, 16, 1, 700, 16, 16, 100, 5, 1, None, 0, 0, None, 0, 0, None)
    assert rc == -1 and b'channels/groups=7' in lib.nbasr_last_error()
    rc = lib.nbasr_dense_conv1d_fused(16, 16, 16, None, None, None, 16, 1, 80, 16, 16, 600, 16, 3, 1, None, 0, 0, None)
    assert rc == -1 and b'unsupported' in lib.nbasr_last_error()
    rc = lib.nbasr_linear_head(16, 16, 16, 16, 10, 500, 65, None)
    assert rc == -1
    # the packed / split entry points and the front-end validate the same way
    # nbasr_dense_conv1d_packed(scheme, x, x_is_image, x_absmax, x_range, packed_w, bias, skip0..2, y, batch, c_in, frames_in, ld_in, c_out, ld_out, kernel, stride, row_tile, frame_tile, ln, stats_part, stream)
    rc = lib.nbasr_dense_conv1d_packed(0, 16, 0, None, None, 16, 16, None, None, None, 16, 1, 600, 10, 10, 800, 12, 8, 1, 128, 256, None, None, None)   # ld_in % 4
    assert rc == -2 and b'multiple of 4' in lib.nbasr_last_error()
    rc = lib.nbasr_dense_conv1d_packed(1, 16, 0, None, None, 16, 16, None, None, None, 16, 1, 600, 16, 16, 800, 16, 8, 1, 128, 256, None, None, None)
    assert rc == -3 and b'x_absmax' in lib.nbasr_last_error()
    rc = lib.nbasr_pack_dense_weights(1, 16, 16, 128, 16, 5, 1, 128, None)
    assert rc == -1 and b'unsupported' in lib.nbasr_last_error()
    rc = lib.nbasr_dense_conv1d_packed(7, 16, 0, None, None, 16, 16, None, None, None, 16, 1, 600, 16, 16, 800, 16, 8, 1, 128, 256, None, None, None)
    assert rc == -1 and b'unknown scheme' in lib.nbasr_last_error()
    rc = lib.nbasr_dense_conv1d_packed(2, 16, 0, None, None, 16, 16, None, None, None, 16, 1, 600, 16, 16, 800, 16, 8, 1, 128, 256, None, None, None)
    assert rc == -1 and b'operand image' in lib.nbasr_last_error()            # the bf16 scheme reads images only
    rc = lib.nbasr_linear_fused_packed(16, None, 16, 16, None, None, None, 16, 1, 600, 16, 16, 600, None, 0, 0, None)
    assert rc == -3 and b'workspace' in lib.nbasr_last_error()
    rc = lib.nbasr_lstm_input_projection_packed(16, 16, 16, 16, 16, 16, 1, 1200, 16, 16, 501, None, None)
    assert rc == -2
    rc = lib.nbasr_frame_signal(16, None, 16, 1, 100, 100, 400, 160, 4, None)                 # too short for reflect padding
    assert rc == -1 and b'reflect' in lib.nbasr_last_error()
    rc = lib.nbasr_pointwise_linear(16, 16, 16, 16, 1, 401, 10, 12, 80, 12, None)               # c_in % 4
    assert rc == -2
    assert lib.nbasr_pointwise_workspace_bytes(64, 1200, 252) == 64 * 38 * 32768 + 64 * 4 + 64 * 38 * 4     # image, inverse scales, partial maxima
    assert lib.nbasr_pointwise_packed_weights_bytes(2000, 1200) == 16 * 38 * 16384 + 2 * 16 * 128 * 4
    assert lib.nbasr_packed_dense_weights_bytes(1, 800, 600, 8, 128) == 7 * 38 * 2 * 8 * 128 * 16 * 2 + 2 * 7 * 128 * 4
    assert lib.nbasr_packed_dense_weights_bytes(9, 800, 600, 8, 128) == 0 and lib.nbasr_lstm_seq_workspace_bytes(65, 500) == 0
    assert lib.nbasr_lstm_seq_workspace_bytes(64, 500) == 64 * 4 + 4 * 2 * 128 * 16 * 16       # status words + two images per utterance tile


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(hip.HipError, match='no CPU fallback'):
        hip.load_library(tmp_path / 'libnbasr_hip.so')
