"""Tools only: ASDR_TOOLS_LIB=<path of a variant build> makes this process load that build instead of the in-tree library (ONE library per
process: every library has its own stream pool, and several of them in one process share hardware queues -- profiles/README.md).
Import before anything creates a batch."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import audiosdr_amd.binding as _binding  # noqa: E402

if os.environ.get("ASDR_TOOLS_LIB"):
    _binding.library_path = lambda _p=os.path.abspath(os.environ["ASDR_TOOLS_LIB"]): _p
