"""CPU oracle for the EmotiVoice inference hot path (JETSGenerator.forward).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.

The oracle is a plain fp32 restatement (torch CPU functional ops, B=1 per
utterance -- the only batch size any reference call site uses) of

    /root/reference/models/prompt_tts_modified/jets.py:50-71
    /root/reference/models/prompt_tts_modified/model_open_source.py:102-173
    /root/reference/models/prompt_tts_modified/modules/encoder.py
    /root/reference/models/prompt_tts_modified/modules/variance.py
    /root/reference/models/prompt_tts_modified/modules/alignment.py:175-211
    /root/reference/models/hifigan/models.py:26-63,90-131

Pinning: the reference ships no golden vectors or tests for this path
(SURVEY.md section 4), so the oracle is pinned against the reference itself
executed in the build container: ``tests/golden/make_golden.py`` loads the
synthetic weights of ``oracle/weights.py`` into the *reference's own*
``JETSGenerator`` (strict ``load_state_dict``), runs it, and commits the
inputs/outputs/stage taps under ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks this restatement against those files.
"""
from .jets_oracle import (  # noqa: F401
    jets_forward,
    am_forward,
    hifigan_forward,
    fold_weight_norm,
    ORACLE_TAPS,
)
from .weights import synth_state_dict, synth_inputs, EVShapes  # noqa: F401
