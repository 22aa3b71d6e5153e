import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def weights():
    from infercam_onnx_amd import synth

    return synth.synthetic_weights()


def pytest_terminal_summary(terminalreporter):
    try:
        from helpers import EXCUSED
    except Exception:
        return
    terminalreporter.write_line("assert_dets_match: borderline excuse fired for %d detections in %d frames" %
                                (EXCUSED["detections"], EXCUSED["frames"]))
