import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def unit_scenes(tmp_path_factory):
    """Deterministically generated unit scenes: name -> path of the .json file."""
    from raytracer_amd import scenes
    d = tmp_path_factory.mktemp("unit_scenes")
    return scenes.write_unit_scenes(str(d))


@pytest.fixture(scope="session")
def torch_warm():
    """The first `import torch` on a fresh machine pages the library in for a minute or two.  Tests that
    start a child process which imports torch ask for this fixture: it pays that once, in a child of
    its own with a generous limit, so that the tests' own limits measure their work and a test that
    runs into one has failed (not: is skipped)."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.__version__)"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip()
