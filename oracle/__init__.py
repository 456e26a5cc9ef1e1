"""CPU oracle for the deComP iterative-update hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement of the reference's algorithms for the hot path
(NMF multiplicative update, batched LASSO inner solves, online dictionary
learning).  Every function cites the reference file:line it follows.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and there only as the checker / reported baseline.  The product
package ``decomp_amd`` never imports anything from here: its compute path is
the HIP library and it fails loudly when that library is missing.

Parity pinning: the restatement is checked against the real reference
(imported from /root/reference in the build container by
``oracle/make_golden.py``) through the fixtures committed under
``tests/golden/``; see ``tests/test_oracle_golden.py``.
"""

from . import common, nmf, nmf_minibatch, lasso, dictionary_learning  # noqa: F401
