"""Host-side mirror of the reference's data formats either side of the path: the `mix.json / s1.json / s2.json`
manifests and `AVSpeechDataset` (src/datas/avspeech_dataset.py:18-200); the mouth-ROI arithmetic lives on the GPU
(`rtfs_net_amd.models.videomodels.MouthROI`)."""
from .avspeech_dataset import AVSpeechDataset, normalize_tensor_wav, read_manifests, read_wav

__all__ = ["AVSpeechDataset", "normalize_tensor_wav", "read_manifests", "read_wav"]
