"""ORACLE-side name of the deterministic synthetic weights / inputs: the generators themselves live in
rtfs_net_amd/synthetic.py (bench.py must start without the oracle package); re-exported here so that the oracle's
fixture scripts and the tests keep one import path."""
from rtfs_net_amd.synthetic import *  # noqa: F401,F403
from rtfs_net_amd.synthetic import INPUT_SEED, TINY_AUDIONET, _gen, lip_inputs, rtfs_audionet, synth_inputs, synth_state_dict  # noqa: F401
