import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O  # oracle/oracle.py -- the CPU checker (test infrastructure)

    O.build()
    return O


@pytest.fixture(scope="session")
def literals():
    with open(os.path.join(GOLDEN, "reference_literals.json")) as f:
        return json.load(f)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden_pcm():
    """data/s16_mono_22_5kHz.flac decoded exactly as ffmpeg does (s16 / 32768), Adler-32 0x5e01930b."""
    s16 = load_golden("s16_mono_22_5kHz.pcm_s16.npy")
    return (s16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


@pytest.fixture(scope="session")
def piano_pcm():
    return load_golden("librosa-decoded.npy")
