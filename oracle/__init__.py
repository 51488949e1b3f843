"""CPU oracle for the baryon_painter CVAE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``baryon_painter_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may.  It is the checker, never the product path.

* ``oracle.ops``          -- NumPy restatement of every operator the path uses
                             (conv / transposed conv / batch-norm / activations,
                             forward and hand-derived backward).
* ``oracle.cvae_oracle``  -- NumPy restatement of ``baryon_painter/models/cvae.py``
                             and ``models/utils.py`` (arch-dict compiler, Q, prior,
                             P, KL, log-likelihood, ELBO, sample_P) + backward.
* ``oracle.torch_ref``    -- the same module graph assembled from stock
                             ``torch.nn.functional`` calls on the CPU (what the
                             reference executes); used for the timed CPU baseline
                             and for full-size checks.

Parity pin: both are checked against ``tests/golden/*.npz``, which were produced
by importing the real reference (``/root/reference``) in the build container with
``tests/golden/make_goldens.py``.
"""
