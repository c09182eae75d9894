"""ORACLE (test infrastructure, not product code): how the fixture generators import the Python reference.

    prepare_path()   /root/reference goes to the FRONT of sys.path (the `src` package), the import stubs of oracle/stubs to the BACK:
                     a third-party package the environment really has (sru, timm, thop, pytorch_lightning, cv2, soundfile) is imported
                     from the environment, a stub only stands in for one that is absent.
    sru_source()     which SRU arithmetic the reference ran with: "package sru <version>" (the real third-party package, what
                     src/models/layers/rnn_layers.py:6 imports and setup/requirements.yaml:18,33 pins) or
                     "restatement oracle/sru_ref.py" (the stub: the published sru 2.6.0 algorithm restated; "parity unpinned").
                     Every fixture whose reference run went through an SRU records it under the key `sru_source`.

Closing the SRU pin on a box that has the package (`pip install sru==2.6.0`, any box with network access and /root/reference):

    python -m oracle.regenerate_all        # every generator; fixtures then carry sru_source = "package sru 2.6.0"
    python -m pytest tests/test_sru_ref.py::test_restatement_matches_the_package tests/test_oracle_golden.py -q

Until round 6 the stub directory was inserted at the front of sys.path and shadowed the package even where it was installed.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
STUBS = os.path.join(ROOT, "oracle", "stubs")
RESTATEMENT = "restatement oracle/sru_ref.py"


def prepare_path(ref=REF):
    """idempotent; returns the stub directory (last on sys.path)"""
    for p in (ROOT, ref):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    if STUBS in sys.path:
        sys.path.remove(STUBS)
    sys.path.append(STUBS)
    return STUBS


def sru_source():
    """-> "package sru <version>" | "restatement oracle/sru_ref.py", decided by where `import sru` resolves AFTER prepare_path()"""
    prepare_path()
    import sru

    where = os.path.realpath(getattr(sru, "__file__", "") or "")
    if where.startswith(os.path.realpath(STUBS) + os.sep):
        return RESTATEMENT
    version = getattr(sru, "__version__", None)
    if version is None:
        try:
            from importlib.metadata import version as _v

            version = _v("sru")
        except Exception:
            version = "unknown"
    return f"package sru {version}"


def import_reference():
    """-> the reference's AVNet class (src/models/__init__.py:8)"""
    prepare_path()
    from src.models import AVNet

    return AVNet
