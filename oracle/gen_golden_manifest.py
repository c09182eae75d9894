"""Generate tests/golden/manifest.json by RUNNING THE REFERENCE's AVSpeechDataset.__init__ (src/datas/avspeech_dataset.py:18-110) on
synthetic manifests: pins the index construction (drop rule, ordering, len()).  `python -m oracle.gen_golden_manifest`"""
import contextlib
import io
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synthetic_manifests(d, n=7):
    lens = [40000, 15000, 32000, 31999, 64000, 8000, 33000][:n]
    mix = [[f"/data/mix/utt{i}_utt{i + 100}.wav", L] for i, L in enumerate(lens)]
    s = [[[f"/data/s{k}/utt{i}.wav", f"/data/mouths/spk{k}_{i}.npz", L] for i, L in enumerate(lens)] for k in (1, 2)]
    for name, obj in (("mix", mix), ("s1", s[0]), ("s2", s[1])):
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(obj, f, indent=4)


def main():
    sys.path.insert(0, ROOT)
    from oracle.ref_import import prepare_path

    prepare_path()
    import importlib.util
    import types

    pkg = types.ModuleType("refdatas")
    pkg.__path__ = ["/root/reference/src/datas"]
    sys.modules["refdatas"] = pkg
    for name in ("transform", "avspeech_dataset"):
        spec = importlib.util.spec_from_file_location(f"refdatas.{name}", f"/root/reference/src/datas/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refdatas.{name}"] = mod
        spec.loader.exec_module(mod)
    Ref = sys.modules["refdatas.avspeech_dataset"].AVSpeechDataset
    out = {}
    with tempfile.TemporaryDirectory() as d:
        synthetic_manifests(d)
        for n_src in (1, 2):
            for segment in (None, 2.0, 2.5):
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    ds = Ref(json_dir=d, n_src=n_src, sample_rate=16000, segment=segment)
                out[f"n{n_src}_seg{segment}"] = {"mix": ds.mix, "sources": ds.sources, "len": len(ds), "printed": buf.getvalue()}
                print(n_src, segment, len(ds), len(ds.mix), buf.getvalue().strip())
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tests/golden/manifest.json")


if __name__ == "__main__":
    main()
