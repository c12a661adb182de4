"""CPU oracle for the SMRT VAE train-and-generate hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``smrt_amd/`` may import this package:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the CPU baseline, never as the product.

Parity status (see DESIGN.md "Oracle"):

* arithmetic (everything the reference hands to TensorFlow): **parity unpinned**
  against TensorFlow itself.  TensorFlow (>=1.0, requirements.txt:4) is a
  third-party dependency that is neither vendored under /root/reference nor
  installed here, and the reference's own tests pin no numeric value of the VAE
  path (smrt/autoencode/tests/test_vae.py:39,48-50 are commented out).  The
  restatement follows the reference's call sites file:line and TF-1.x public
  semantics, and is cross-checked against torch CPU autograd (tests/).
* host logic (validation, per-class orchestration, numpy RNG stream order):
  pinned by fixtures generated from the reference's own Python
  (tests/golden/make_host_goldens.py).
"""
