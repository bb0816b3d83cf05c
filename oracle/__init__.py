"""CPU oracle for the BayesOD inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This package restates, in NumPy (float64 / float32 / bf16-rounding emulation), the
algorithm of asharakeh/bayes-od-rc's inference path.
Every function cites the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker.  The product package (``bayes-od-rc_amd/``) never imports it and
fails loudly when its HIP library is missing.

Pinning status:
  * NumPy half of the reference (``bayes_od_clustering``, ``*_np`` box utils,
    ``map_dataset_classes``, entropy helpers, writers): PINNED against golden vectors
    captured by importing the reference itself (``tests/golden/make_golden.py``).
  * TensorFlow / TFP half (network forward, ``bayes_od_inference``): **parity unpinned** --
    TensorFlow cannot be installed or run here or on the GPU box and the reference has no
    tests or fixtures.  The restatement follows the reference source plus the documented op
    semantics and is cross-checked by a second, independent restatement
    (``oracle/torch_ref.py``: PyTorch-CPU fp32 with explicit padding) and by hand-derivable
    known answers.  ``bayes_od.py`` is additionally checked against the reference's own
    ``bayes_od_inference`` SOURCE executed under a NumPy stand-in for TensorFlow
    (tests/test_reference_transcription.py): a transcription check of formulas and branches,
    which leaves the op semantics -- and therefore the "unpinned" label -- where they were.
"""
