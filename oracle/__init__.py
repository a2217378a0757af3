"""CPU oracle for the outlier-suppression fake-quant / observer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``outlier_suppression_amd``) never imports this package and raises if its HIP
library is missing.

What it is: a NumPy restatement (fp32 arithmetic written op by op) of the
algorithms in the reference's ``quant_transformer/quantization`` package and
``solver/gamma_migration.py``; every function cites the reference file:line it
follows.  Third-party arithmetic the reference relies on and that is not in its
tree (``torch.quantile``'s linear interpolation, ``scipy.optimize``'s bounded
Brent) is restated from the published algorithm and pinned against the
installed torch 2.10 / scipy 1.15.3 in ``tests/test_oracle_pinning.py``.

Parity pinning: the reference ships no tests or golden vectors.  The oracle is
pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (imports ``/root/reference``) and
committed as ``tests/golden/*.npz``.
"""
from . import fake_quant_oracle, observer_oracle, gamma_oracle, brent  # noqa: F401
