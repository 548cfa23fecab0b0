"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's GAN-training hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported baseline.  The product path
(``ndivplanning_amd``) never imports this package and has no CPU fallback.

Pinning: the reference (goodmattg/ndivplanning) ships no tests and no golden
vectors, so parity is pinned by outputs of the reference
modules themselves, generated in the build container by
``tests/golden/make_golden.py`` (which imports ``/root/reference`` with stub
modules for its *unused* third-party imports) and committed under
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks this
restatement against those vectors.
"""
