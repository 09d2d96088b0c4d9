#!/usr/bin/env python3
"""The digest `ac_source_hash()` returns for a library built from this tree (autocycler_amd/csrc/Makefile: SRC_HASH)."""
import hashlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def source_hash():
    csrc = ROOT / "autocycler_amd" / "csrc"
    files = sorted(p.name for ext in ("*.hip", "*.inc", "*.hpp", "*.cpp") for p in csrc.glob(ext))
    h = hashlib.sha256()
    for name in files:
        h.update((csrc / name).read_bytes())
    h.update((ROOT / "include" / "autocycler_hip.h").read_bytes())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
