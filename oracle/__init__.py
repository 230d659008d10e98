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

**The smplx arithmetic is unpinned.**  ``oracle/smplx_ref.py`` restates ``smplx.SMPLX.forward`` / ``lbs`` from the published
algorithm (Pavlakos et al. 2019; Loper et al. 2015) because the package and its source are absent here.  Unlike DINOv2 (checked
against an independently written implementation, HF ``transformers``) and roma (checked against scipy), the LBS restatement has NO
independent implementation to be checked against in this environment: its cross-check (``tests/test_oracle_thirdparty.py``) is a
second implementation written differently on purpose (fp64 numpy, scipy's Rodrigues, a recursive 4x4 chain, an explicit sparse skinning
sum) -- but by the same author from the same reading of the papers -- plus analytic cases (zero pose = template + blend shapes, a
single-joint rotation, a rigid root rotation).  A misreading shared by both would not be caught.  The first thing to do when the real
``smplx`` package is available: run ``smplx.create(..., use_pca=False, flat_hand_mean=True)`` on the seeded parameters of
``tests/test_oracle_thirdparty.py`` and compare vertices and the 127 joints.
"""
