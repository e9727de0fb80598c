"""ORACLE — test infrastructure only.

A CPU (torch fp32 / numpy fp64) restatement of the UnCLTMO tone-mapping hot path, pinned against golden
vectors captured from the upstream reference (tests/golden/).  Allowed importers: tests/,
__graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product package `uncltmo_amd` must never
import it.
"""
