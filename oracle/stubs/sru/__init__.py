from oracle.sru_ref import SRU, SRUCell  # noqa: F401
