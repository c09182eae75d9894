"""ORACLE (test infrastructure): build the C restatement of the SRU recurrence (oracle/csrc/sru_scan.c) with gcc into oracle/_build/libsru_scan.so.

    python -m oracle.build_c

Called by __graft_entry__.build() and, lazily, by oracle/sru_ref.py; without gcc the oracle keeps its Python loop (same arithmetic, slower)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "sru_scan.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libsru_scan.so")


def build(force: bool = False):
    """-> path of the shared library, or None when it cannot be built (no gcc)"""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = LIB + f".{os.getpid()}.tmp"
    r = subprocess.run([gcc, "-O3", "-fPIC", "-shared", "-fopenmp", "-fno-fast-math", SRC, "-o", tmp, "-lm"], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    os.replace(tmp, LIB)  # (atomic: several pytest workers may build at once)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
