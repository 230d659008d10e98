"""The parity tolerances of the -m gpu tests, in one place.

north_star (BASELINE.json): "Outputs (person scores, SMPL-X params, 3D vertices) match the reference CPU path on the same inputs
within 1e-3 relative fp tolerance".  Metric: per-tensor relative L2 over all persons of the case,  |got - ref|_2 / |ref|_2.

  TOL = 1e-3      EVERY tensor of the forward's output dict, for f16 MFMA operands (the product precision, what bench.py reports):
                  person scores, offset, loc, dist, shape, expression, rotmat / rotvec (compared through the rotation they encode: the
                  axis-angle map is discontinuous at pi), transl, transl_pelvis, v3d, j3d, j2d, v2d -- and the joint SMPL-X parameter
                  vector [rotation matrices | shape | expression] of every person.  No per-key exceptions.
                  How the two constant-free linear read-outs (`expression`, `offset`: their relative error equals that of the features
                  they read, up to 2x after the decoder) get there: a single f16 pass left `expression` at 1.2e-3 on one of the four
                  BASELINE-size goldens -- 68 % of the error variance is the one-time rounding of the ViT weights to f16 -- so the V and
                  attention-output projections of blocks 0..11 also run the LOW halves of their weights through the matrix pipe
                  (multi_hmr_amd/vit.py DEFAULT_WLO; round 6: the output projections of blocks 0-11 alone, worst key 5.9e-4
                  over the five ViT-B / ViT-L goldens).
  bf16 operands (8-bit significand) miss the contract by 3-8x and are held to 2e-2; they are measured, not the product default.
  MAXTOL          the same keys in the max norm (|err|_inf / |ref|_inf) at BASELINE.json's sizes and on the hostile-weight goldens."""
import numpy as np
import torch

TOL = {"f16": 1e-3, "bf16": 2e-2}
CHECKED = ["scores", "offset", "loc", "dist", "dist_postprocessed", "shape", "expression", "rotmat", "transl", "transl_pelvis",
           "v3d", "j3d", "j2d", "v2d"]


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def maxrel(a, b):
    """Max-norm relative error  |got - ref|_inf / |ref|_inf : what a single wrong element cannot hide in (the L2 form averages it away)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def tolerance(key, precision):
    return TOL[precision]


#: max-norm gate of the full-size parity tests (tests/test_gpu_parity_fullsize.py).  Measured with f16 operands on the round-4 build
#: (profiles/r04_parity_fullsize.json): every key of every BASELINE-size case is below 1e-3 EXCEPT `rotmat` -- the worst of 3816 ... 9540
#: rotation-matrix entries (|ref|_inf = 1) is 1.2e-3 / 0.9e-3 / 1.6e-3 at 896^2 / 672^2 / 1288^2 (0.9e-3 / 1.0e-3 / 1.8e-3 before the attention kernel
#: changed its MFMA shape: the value moves with every change of rounding order): a ~3.5-sigma draw of an error whose
#: rms is 3e-4 -- and, on the hostile-mean golden, `v2d` (1.7e-3).  The gate is therefore 2e-3: the max-norm form of the 1e-3 contract
#: is NOT met on those two keys, and this constant says by how much.
MAXTOL = {"f16": 2e-3, "bf16": 4e-2}
#: Round 6: the gate is PER KEY (maxtol() below).  Why `rotmat` cannot meet 1e-3 in the max norm while its relative L2 error is 2e-4
#: (tools/rotmat_amplification.py, profiles/r06_rotmat_amplification.txt; fp64, on the goldens' own 6D read-outs): the reference's
#: 6D -> rotation decode (utils/humans.py:12-22) is a Gram-Schmidt that divides by |a1| and |a2 - (b1.a2) b1|; the seeded heads produce
#: joints whose second column is nearly parallel to the first (|a2_perp| down to 0.10-0.22 against a median of 1.0), where the decode
#: amplifies a perturbation 10-19x.  A WHITE error of relative L2 2e-4 on the 6D read-out -- what the f16 backbone leaves -- comes out as
#: rotmat rel-L2 2.3e-4 (x1.1) and max norm 1.2-2.0e-3 (mean over draws; 2.1-3.0e-3 the worst of 32), at every BASELINE size; even the
#: f16x3 mode (rel-L2 1.0e-4) lands at 0.6-1.1e-3.  At 1288^2 (20 persons, amplification up to 19x) a white 3e-4 comes out at 3.1e-3 in the
#: mean; measured on the round-6 build: 0.7 / 1.5 / 2.4e-3 at 896^2 / 672^2 / 1288^2 with rel-L2 2.2-2.8e-4 (the statistic is the maximum of
#: 4-10 thousand heavy-tailed entries: it moves by 2x with any change of rounding order).  So: 3e-3 for `rotmat`, 1.5e-3 for the two PIXEL
#: keys downstream of the amplified joints (`v2d` 1.1-1.3e-3 on the hostile-mean golden), 1e-3 -- the contract's own number -- for every
#: other key (rounds 4-5: 2e-3 for all of them).
MAXTOL_F16 = {"rotmat": 3e-3, "v2d": 1.5e-3, "j2d": 1.5e-3}


def maxtol(key, precision):
    """max-norm gate of `key` (tests/test_gpu_parity_fullsize.py)"""
    if precision == "f16":
        return MAXTOL_F16.get(key, 1e-3)
    return MAXTOL[precision]


def smplx_param_vector(rotmat, shape, expression):
    """[P, 53*9 + nb + 10]: what the body model consumes, per person."""
    t = lambda v: torch.as_tensor(np.asarray(v)).float()
    rotmat, shape, expression = t(rotmat), t(shape), t(expression)
    return torch.cat([rotmat.reshape(rotmat.shape[0], -1), shape, expression], dim=1).numpy()


def assert_within(errs: dict, precision: str, where=""):
    for k, v in errs.items():
        assert v < tolerance(k, precision), (where, k, v, tolerance(k, precision))
