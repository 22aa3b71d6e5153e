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
    from helpers import REFERENCE_PINS, SESSION_NOTES

    for k, v in sorted(SESSION_NOTES.items()):
        terminalreporter.write_line("note %s: %s" % (k, v))
    for k, v in sorted(REFERENCE_PINS.items()):
        # (the only results the reference's own tests pin need the zoo .onnx files, a run-time download: nn.rs:21-22,155-162)
        terminalreporter.write_line("reference pin %s: %s" % (k, v))
    if any(v.startswith("NOT CHECKED") for v in REFERENCE_PINS.values()):
        terminalreporter.write_line("WARNING: the reference's face counts (integration_tests.rs:20-29: 3,6,4,3,1,1,10,0) were NOT "
                                    "checked in this session -- parity with the reference itself stays unpinned")
