"""Import stub (oracle/gen_golden_manifest.py only): lets the reference's dataset module import; nothing here is ever called."""
