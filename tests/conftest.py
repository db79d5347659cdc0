import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    import parity
    terminalreporter.write_line(f"parity: {parity.COMPARED[0]} streams compared token-exactly, {len(parity.EXCUSED)} excused")
    for e in parity.EXCUSED:
        terminalreporter.write_line(f"  excused: {e}")
    if parity.NEAR_TIES:
        terminalreporter.write_line(f"parity: {len(parity.NEAR_TIES)} beam-search streams differ on a LOCALISED near-tie (what, stream, frame, oracle gap at that frame):")
        for e in parity.NEAR_TIES:
            terminalreporter.write_line(f"  near-tie: {e}")


def pytest_sessionfinish(session, exitstatus):
    # a tolerated near-tie divergence is never silent: it fails the session unless explicitly allowed
    import parity
    if parity.EXCUSED and os.environ.get("K2HIP_ALLOW_TIES") != "1" and session.exitstatus == 0:
        session.exitstatus = 1


def _has_gpu():
    try:
        from k2transducerasr_amd import load_library
        return load_library().k2hip_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip silently; but plain
    # `pytest tests` on the CPU container should not try GPU tests.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if not _has_gpu():
        skip = pytest.mark.skip(reason="no HIP device")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def tiny_model_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("models") / "tiny.k2w")
    write_synthetic_model(p, "zipformer2-tiny-test")
    return p


@pytest.fixture(scope="session")
def oracle_tiny(tiny_model_path):
    from oracle import Oracle
    return Oracle(tiny_model_path)


@pytest.fixture(scope="session")
def hip_tiny(tiny_model_path):
    from k2transducerasr_amd import Model
    return Model(tiny_model_path, 0)


@pytest.fixture(scope="session")
def utts():
    from k2transducerasr_amd.synth import synth_utterance
    return [synth_utterance(u, s) for u, s in enumerate([1.3, 0.9, 1.1, 1.3, 0.7])]


@pytest.fixture(scope="session")
def conformer_tiny_path(tmp_path_factory):
    from k2transducerasr_amd.synth import write_synthetic_model
    p = str(tmp_path_factory.mktemp("models") / "conformer_tiny.k2w")
    write_synthetic_model(p, "conformer-tiny-test")
    return p


@pytest.fixture(scope="session")
def oracle_conformer(conformer_tiny_path):
    from oracle import Oracle
    return Oracle(conformer_tiny_path)


@pytest.fixture(scope="session")
def hip_conformer(conformer_tiny_path):
    from k2transducerasr_amd import Model
    return Model(conformer_tiny_path, 0)
