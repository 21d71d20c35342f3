"""pytest configuration: ``gpu`` marker + shared helpers.

``-m "not gpu"`` runs everywhere (oracle vs golden fixtures, host logic, C-ABI
symbol check, gloo world_size-2 slab tests); ``-m gpu`` are the parity tests
proper and call the HIP engine through the C-ABI on a real MI355X.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "experiments: exercises a kernel that lost its A/B and is only in the library when "
                                       "it was built with `make -C lettuce_amd/csrc EXPERIMENTS=1`")


def experiments_built() -> bool:
    try:
        from lettuce_amd import _native
        return os.path.exists(_native.library_path()) and _native.experiments_built()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    no_gpu = pytest.mark.skip(reason="no GPU visible")
    no_experiments = pytest.mark.skip(reason="kernel of the experiments build (make EXPERIMENTS=1)")
    have_gpu, have_experiments = torch.cuda.is_available(), None
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(no_gpu)
        if "experiments" in item.keywords:
            if have_experiments is None:
                have_experiments = experiments_built()
            if not have_experiments:
                item.add_marker(no_experiments)


def golden(name):
    """Load tests/golden/<name>.npz as a dict of numpy arrays."""
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def unpack_nsm(g):
    shape = tuple(int(s) for s in g["no_streaming_mask_shape"])
    n = int(np.prod(shape))
    return np.unpackbits(g["no_streaming_mask"])[:n].reshape(shape).astype(np.uint8)


TORCH_DT = {"f64": torch.float64, "f32": torch.float32}


@pytest.fixture(scope="session")
def engine_library():
    """Path of the in-tree HIP engine library; built on demand (hipcc cross-compiles without a
    GPU) so that a fresh checkout, where the git-ignored .so does not exist yet, still passes."""
    from lettuce_amd import _native
    if not os.path.exists(_native.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _native.library_path()


@pytest.fixture(scope="session", autouse=True)
def _engine_library_for_gpu_runs(request):
    """On a GPU box make sure the in-tree library exists before any parity test runs (the tests
    build it; the product itself never does -- a missing library is an error there)."""
    if torch.cuda.is_available():
        request.getfixturevalue("engine_library")
