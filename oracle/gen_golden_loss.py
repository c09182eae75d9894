"""Golden vectors of the loss / PIT head (f1), produced by RUNNING THE REFERENCE (src/losses) in the authoring container:

    python -m oracle.gen_golden_loss      ->  tests/golden/loss.npz

Data only: inputs, the reference's pairwise matrices for snr / sisdr / sdsdr, PIT mean losses and d(loss)/d(est) from its autograd.
Also checks oracle/loss_ref.py against the reference and refuses to write if they differ by more than 1e-6.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.ref_import import prepare_path

    prepare_path()  # /root/reference first, the import stubs last: packages the environment has are used as they are
    from src.losses.matrix import PairwiseNegSDR  # the reference
    from src.losses.pit_wrapper import PITLossWrapper

    from oracle.loss_ref import pairwise_neg_sdr, pit_pw_mtx

    g = torch.Generator().manual_seed(20240301)
    out = {}
    for tag, (B, n, T) in {"b3n1": (3, 1, 4000), "b2n2": (2, 2, 1500), "b2n3": (2, 3, 700)}.items():
        tgt = torch.randn(B, n, T, generator=g) * 0.3 + 0.05
        est = (tgt.flip(1) if n > 1 else tgt) * 0.8 + 0.25 * torch.randn(B, n, T, generator=g) + 0.02
        out[f"{tag}.est"], out[f"{tag}.tgt"] = est.numpy(), tgt.numpy()
        for kind in ("snr", "sisdr", "sdsdr"):
            e = est.clone().requires_grad_(True)
            ref_fn = PairwiseNegSDR(kind)
            pw = ref_fn(e, tgt)
            loss = PITLossWrapper(ref_fn, pit_from="pw_mtx")(e, tgt)
            loss.backward()
            mine = pairwise_neg_sdr(est, tgt, kind)
            lm, _ = pit_pw_mtx(mine)
            d1, d2 = float((mine - pw.detach()).abs().max()), abs(float(lm - loss.detach()))
            assert d1 < 1e-6 * max(1.0, float(pw.abs().max())) * 10 and d2 < 1e-5, (tag, kind, d1, d2)
            out[f"{tag}.{kind}.pw"], out[f"{tag}.{kind}.loss"], out[f"{tag}.{kind}.grad"] = pw.detach().numpy(), loss.detach().numpy(), e.grad.numpy()
            print(tag, kind, "oracle vs reference: pairwise", d1, "pit", d2)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss.npz"), **out)
    print("wrote tests/golden/loss.npz")


if __name__ == "__main__":
    main()
