"""pytest configuration: the `gpu` marker, repo import path, golden-fixture loaders.

`-m "not gpu"` tests run in the build container (no GPU): oracle vs golden vectors, host logic,
C-ABI surface.  `-m gpu` tests are the parity tests proper and call the HIP kernels through the C ABI
on a real MI355X.  Nothing here reads /root/reference: fixtures are committed under tests/golden/.
"""
import json
import pathlib
import sys

import numpy as np
import pytest

REPO = pathlib.Path(__file__).resolve().parent.parent
GOLDEN = REPO / 'tests' / 'golden'
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (HIP device); run with -m gpu on the GPU box')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no HIP device visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def built_library():
    """Make sure libnbasr_hip.so exists (hipcc cross-compiles without a GPU); no-op when up to date."""
    from nb_asr_amd import build
    return build.build_library()


@pytest.fixture(scope='session')
def known():
    return json.loads((GOLDEN / 'host_known_answers.json').read_text())


@pytest.fixture(scope='session')
def op_fx():
    with np.load(GOLDEN / 'ops_fixtures.npz') as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def model_fx():
    with np.load(GOLDEN / 'model_fixtures.npz') as z:
        return {k: z[k] for k in z.files}
