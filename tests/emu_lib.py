"""Builds and locates the serial CPU emulation of the kernels (tests only; see csrc/Makefile `emu`)."""
import fcntl
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
EMU = ROOT / "tests" / "_emu" / "libautocycler_emu.so"


def emu_path():
    srcs = list((ROOT / "autocycler_amd" / "csrc").glob("*.[ch]*")) + list((ROOT / "autocycler_amd" / "csrc").glob("*.inc")) + \
        [ROOT / "include" / "autocycler_hip.h"]
    EMU.parent.mkdir(parents=True, exist_ok=True)
    with open(EMU.parent / ".build.lock", "w") as lock:      # pytest-xdist workers: one builds, the others wait (never load a half-written .so)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not EMU.exists() or any(s.stat().st_mtime > EMU.stat().st_mtime for s in srcs):
            subprocess.check_call(["make", "-j", "4", "-C", str(ROOT / "autocycler_amd" / "csrc"), "emu"], stdout=subprocess.DEVNULL)
    return EMU
