"""oracle -- CPU restatement of the reference's naive attention `Network`.

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never by metal_flash_attention_amd/.

Follows /root/reference/Tests/FlashAttentionTests/Utilities/Network.swift:70-402
(see network.c for per-function line citations).  PARITY STATUS: parity
unpinned -- the reference holds no golden vectors and no Swift toolchain exists
here; the restatement is cross-checked against an fp64 numpy twin, finite
differences and torch autograd (tests/test_oracle.py).
"""
from .network import (  # noqa: F401
    Network,
    NetworkDescriptor,
    build,
    max_threads,
    round_trip,
)
