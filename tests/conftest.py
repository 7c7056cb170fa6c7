import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


# Order of the GPU suite (the driver runs it with -x): the BASELINE-config parity and bit-exact evidence first - whole-model forward parity
# (configs #2 S/8 and B/16, #3 incl. the bs-64 batch, pillar membership, greedy tokens), configs[0] on the demo tile, FFL (config #5),
# decode / assignment / post-processing (integer outputs) - then training-step and multi-rank tests, the HiSup heads, micro-op (incl. the planes kernels) and
# backward-kernel tests last.
_ORDER = ["test_model_gpu", "test_pillar_membership_gpu", "test_predict_demo_gpu", "test_ffl_gpu", "test_decode_layer_gpu", "test_assignment_gpu", "test_postprocess_gpu",
          "test_input_pipeline_gpu", "test_ffl_loss_gpu", "test_afm_gpu", "test_train_gpu", "test_syncbn_gpu", "test_rccl_single_rank_gpu",
          "test_hisup_gpu", "test_ops_gpu", "test_x3_gpu", "test_backward_gpu"]


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _ORDER.index(name) if name in _ORDER else len(_ORDER)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_rank)          # stable: the order inside a file stays
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_process_state(request):
    """Every GPU test starts from a clean product state: an optimizer a previous test never closed must not leak its arena views, the
    direct-gradient switch or the reducer hook into the next test (the suite's order is not the files' alphabetical order any more)."""
    yield
    if "gpu" in request.keywords:
        try:
            from pixelspointspolygons_amd import ops
            ops.reset_process_state()
        except Exception:
            pass
