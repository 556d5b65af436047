"""CPU oracle for the CleverRec BPR hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  Nothing under ``cleverrec_amd/`` does; the product
path fails loudly when its HIP library is missing instead of falling back here.

Parity status (see oracle/crec_oracle.c header and DESIGN.md):
  * RNG stream, samplers, ranking metrics: pinned by golden vectors generated
    from the reference's own utils/sampler.py / utils/metrics.py
    (tools/make_golden.py -> tests/golden/).
  * TF1 graph arithmetic (BPR loss/grad, Adam): parity unpinned (TensorFlow is
    an absent third-party dependency, reference README.md:57); restated from the
    graph definition and cross-checked by float64 autograd.
"""
from .oracle import *  # noqa: F401,F403
