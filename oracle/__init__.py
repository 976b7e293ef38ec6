"""CPU oracle for the FLUX denoise-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product package
(``fastdm_amd``) never imports this package and has no CPU fallback.

What it is: a plain torch-CPU / numpy restatement of the reference's *torch
backend* (``/root/reference/fastdm/kernel/torch/*.py``), its layers
(``fastdm/layer/*.py``), the FLUX core (``fastdm/model/flux.py``) and TeaCache
(``fastdm/caching/xcaching.py``).  Every function cites the reference
file:line it follows.

Pinning: the reference is Python and importable in the build container (with
two stub modules, see ``oracle/make_golden.py``).  ``make_golden.py`` runs the
reference itself on seeded inputs and writes the input/output vectors under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors (bit-exact where the arithmetic is deterministic).
Third-party attention arithmetic (torch SDPA / SageAttention) is pinned only
to the reference's own test tolerance (atol 1.8e-2 vs an fp32 softmax
reference, ``tests/test_attention.py:23-63,94``).
"""
