"""CPU fp32 oracle for the Multi-HMR batched-inference path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``multi_hmr_amd``) never does.

PARITY STATUS: **unpinned by the reference** -- naver/multi-hmr ships no tests, golden vectors or
fixtures, and its three arithmetic dependencies (facebookresearch/dinov2 via
torch.hub, ``smplx``, ``roma``; all unpinned, none vendored under /root/reference) are absent from this
environment.  What pins this oracle instead:

* every reference-authored function on the path (model.py, blocks/*, utils/*) is *executed verbatim*
  from /root/reference by ``oracle/ref_shim.py`` in the build container, with only the third-party
  modules replaced by the restatements in this package; its outputs on seeded inputs are committed under
  ``tests/golden/`` by ``tests/golden/make_golden.py``; ``oracle/multihmr_ref.py`` (the portable
  restatement used on the GPU box, where /root/reference does not exist) is checked against them;
* the third-party restatements are cross-checked against independent implementations that *are*
  installed: HF ``transformers`` Dinov2 (block math), ``scipy.spatial.transform.Rotation`` (roma), and
  analytic known-answer tests (LBS) -- see tests/test_oracle_*.py.
"""
