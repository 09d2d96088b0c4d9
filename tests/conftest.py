import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
# The library reads its AC_* tuning knobs ONCE per process; the knob-parity tests change them between builds of this process
# (monkeypatch.setenv): this variable, read when the library is first used, makes every build read them again.
os.environ.setdefault("AC_TUNING_FOLLOW_ENV", "1")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle (test infrastructure) is compiled on demand; prebuilt .so files travel to the GPU box."""
    so = ROOT / "oracle" / "liboracle.so"
    srcs = [ROOT / "oracle" / n for n in ("ac_oracle.cpp", "ac_oracle.hpp", "oracle_capi.cpp")]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL)
    yield
