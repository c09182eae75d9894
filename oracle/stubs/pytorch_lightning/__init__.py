import torch as _torch

__version__ = "stub-2.1.3"


class LightningModule(_torch.nn.Module):
    """Import-time placeholder (the autoencoder video model subclasses it at import; it is never instantiated here)."""
