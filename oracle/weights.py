"""Synthetic checkpoint / input generators used by the oracle-side tests.

The generator itself is plain data synthesis (numpy PCG64) and lives in the product package
(emotivoice_amd/synthetic.py) because bench.py needs it without touching the oracle; the oracle
re-exports it so tests can keep importing everything checker-related from ``oracle``.
"""
from emotivoice_amd.config import EVShapes  # noqa: F401
from emotivoice_amd.synthetic import synth_inputs, synth_state_dict  # noqa: F401
