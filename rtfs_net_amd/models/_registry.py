"""Name -> class lookup shared by `rtfs_net_amd.models` and `rtfs_net_amd.models.videomodels`.

Behaviour of the reference's two package registries (src/models/__init__.py:15-42, src/models/videomodels/__init__.py:22-50): lookup is
case-insensitive over the package namespace, unknown or non-string identifiers raise `ValueError("Could not interpret model name :
...")`, registering a name that already exists (in either case) raises `ValueError`."""
from __future__ import annotations


def make_registry(namespace: dict):
    """-> (register_model, get) bound to a package's globals()."""

    def register_model(custom_model):
        name = custom_model.__name__
        if name in namespace or name.lower() in namespace:
            raise ValueError(f"Model {name} already exists. Choose another name.")
        namespace[name] = custom_model

    def get(identifier):
        found = None
        if isinstance(identifier, str):
            wanted = identifier.lower()
            for key, value in namespace.items():
                if key.lower() == wanted:
                    found = value
        if found is None:
            raise ValueError(f"Could not interpret model name : {str(identifier)}")
        return found

    return register_model, get
