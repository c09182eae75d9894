"""ORACLE (test infrastructure): re-run every fixture generator against /root/reference in one command.

    python -m oracle.regenerate_all [--stamp-only]

The generators import the reference with whatever third-party packages the environment has (oracle/ref_import.py): on a box where
`pip install sru==2.6.0` worked, this is the command that closes the "parity unpinned" gap of the SRU - the fixtures are then written by the
real package (each generator still refuses to write unless oracle/avnet_ref.py, which always runs the restatement oracle/sru_ref.py, agrees with
the reference), and `sru_source` in every fixture says "package sru 2.6.0".  About 40 minutes on 8 cores (the float64 gradient cases dominate).

--stamp-only: write the `sru_source` key into existing fixtures that predate it (rounds 1-5: generated with the restatement - there was no other
SRU in the container, and the stub shadowed everything) without re-running anything.
"""
from __future__ import annotations

import argparse
import glob
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
GENERATORS = ("gen_golden", "gen_golden_long", "gen_golden_grads", "gen_golden_loss", "gen_golden_lip", "gen_golden_roi", "gen_golden_manifest")
# fixtures whose reference run goes through an SRU (the VP-block gradients, loss head, lip encoder and ROI pipeline do not)
SRU_FIXTURES = ("tiny.npz", "rtfs*.npz", "long_*.npz", "scale_x.npz", "grads_*.npz")


def sru_fixture_files():
    return sorted(f for pat in SRU_FIXTURES for f in glob.glob(os.path.join(GOLDEN, pat)))


def stamp_existing(source):
    for path in sru_fixture_files():
        with np.load(path) as z:
            if "sru_source" in z.files:
                continue
            arrays = {k: z[k] for k in z.files}
        arrays["sru_source"] = np.array(source)
        np.savez_compressed(path, **arrays)
        print("stamped", os.path.basename(path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stamp-only", action="store_true")
    args = ap.parse_args()
    if args.stamp_only:
        from oracle.ref_import import RESTATEMENT

        stamp_existing(RESTATEMENT)
        return
    for g in GENERATORS:
        print(f"== python -m oracle.{g}", flush=True)
        subprocess.run([sys.executable, "-m", f"oracle.{g}"], cwd=ROOT, check=True)


if __name__ == "__main__":
    main()
