"""The drop-in boundary as code: ``nb_asr_amd.register_with(nasbench_asr)`` makes this package the reference's
'hip' model backend with no edit to the reference (model/__init__.py:4,19-24; utils.py:150-153).  Runs where the reference checkout
exists (the build container); on the GPU box the file is skipped -- /root/reference does not travel."""
import pathlib
import sys

import pytest
import torch

import nb_asr_amd

REF = pathlib.Path('/root/reference')
pytestmark = pytest.mark.skipif(not REF.exists(), reason='reference checkout only exists in the build container')

ARCH = [[3, 1], [4, 1, 1], [2, 1, 1, 1]]


@pytest.fixture()
def nasbench_asr():
    sys.path.insert(0, str(REF))
    sys.dont_write_bytecode = True
    try:
        import nasbench_asr as ref
        saved = dict(ref.model._backends.backends), list(ref.model._backends.available_backends)
        yield ref
        ref.model._backends.backends.clear()
        ref.model._backends.backends.update(saved[0])
        ref.model._backends.available_backends[:] = saved[1]
    finally:
        sys.path.remove(str(REF))


def test_reference_get_model_returns_the_hip_model(nasbench_asr, capsys):
    assert nb_asr_amd.register_with(nasbench_asr) == 'hip'
    assert 'hip' in nasbench_asr.model.get_available_backends()
    model = nasbench_asr.get_model(ARCH, use_rnn=True, dropout_rate=0.0, backend='hip')      # the reference's own entry point
    assert isinstance(model, nb_asr_amd.model.ASRModel) and model.backend == 'hip' and model.training
    assert sum(p.numel() for p in model.parameters()) == 27032549
    # same state_dict keys and shapes as the reference's torch backend: checkpoints move between the two unchanged
    theirs = nasbench_asr.get_model(ARCH, use_rnn=True, dropout_rate=0.0, backend='torch')
    assert type(theirs).__module__.startswith('nasbench_asr.')
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == {k: tuple(v.shape) for k, v in theirs.state_dict().items()}
    model.load_state_dict(theirs.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), theirs.state_dict().values()))
    # print_model_summary dispatches on model.backend (model/__init__.py:23-24) back into this package
    nasbench_asr.model.print_model_summary(model)
    out = capsys.readouterr().out
    assert 'Trainable parameters: 27,032,549' in out
    # the torch backend is untouched and still the default
    assert nasbench_asr.model.get_backend_name('torch') == 'torch'


def test_register_as_default_backend(nasbench_asr):
    nb_asr_amd.register_with(nasbench_asr, default=True)
    model = nasbench_asr.get_model(ARCH, use_rnn=False, dropout_rate=0.0)                    # no backend= argument
    assert isinstance(model, nb_asr_amd.model.ASRModel)
    assert sum(p.numel() for p in model.parameters()) == 23662849 and model.use_rnn is False


def test_no_cpu_forward_through_the_reference_entry_point(nasbench_asr):
    """The registered backend keeps the package's contract: a CPU tensor raises, nothing falls back to torch."""
    nb_asr_amd.register_with(nasbench_asr)
    model = nasbench_asr.get_model(ARCH, use_rnn=True, dropout_rate=0.0, backend='hip').eval()
    with pytest.raises(nb_asr_amd.hip.HipError):
        model(torch.zeros(1, 80, 16))
