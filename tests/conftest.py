"""pytest configuration: registers the `gpu` marker and builds the test-infrastructure libraries
(oracle/libptoracle.so always; oracle/_ref/libptref.so and libptref_mipt.so only where /root/reference exists)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    targets = ["oracle", "ref"]
    if os.path.exists(os.path.join(ROOT, "pathtracer_amd", "libmipt.so")):
        targets.append("refmipt")     # the reference with the USE_MIPT switch (integration/use_mipt), linked against libmipt.so
    subprocess.run(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile")] + targets, check=True, cwd=ROOT)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
