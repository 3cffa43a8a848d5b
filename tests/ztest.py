"""The reference's statistical render test (src/render/tests/test_renders.py:146-236) re-hosted: per-pixel Z-test of a rendered mean
against a reference mean + per-sample variance, Sidak-corrected significance, accepted when >= 99.75 % of the pixels pass.
The reference compares against stored EXR references (not in the tree); here the reference is rendered by the oracle in independent
batches, so its own noise enters the statistic (two-sample form: var / n_test + var / n_ref)."""
import math

import numpy as np


def reference_moments(render_batch, batches):
    """render_batch(seed) -> H x W x 3 image of `spp_b` samples per pixel (box filter: pixels are independent).
    Returns (mean, per-sample variance estimate / spp_b factor applied by the caller, n_batches)."""
    imgs = np.stack([render_batch(s).astype(np.float64) for s in range(batches)])
    return imgs.mean(axis=0), imgs.var(axis=0, ddof=1)


def z_test(mean, n_test, ref_mean, ref_var_per_sample, n_ref):
    var = np.maximum(ref_var_per_sample, 1e-4)                           # test_renders.py:160
    z = np.abs(mean - ref_mean) / np.sqrt(var / n_test + var / n_ref)
    cdf = 0.5 * (1.0 + np.vectorize(math.erf)(z / math.sqrt(2.0)))
    return 2.0 * (1.0 - cdf)


def accept(mean, n_test, ref_mean, ref_var_per_sample, n_ref, significance=0.01):
    p = z_test(np.asarray(mean, np.float64), n_test, ref_mean, ref_var_per_sample, n_ref)
    pixel_count = p.shape[0] * p.shape[1]
    alpha = 1.0 - (1.0 - significance) ** (1.0 / pixel_count)           # Sidak correction (test_renders.py:212-215)
    success = p > alpha
    return (np.count_nonzero(success) / 3) >= 0.9975 * pixel_count, float(p.min()), alpha


def oracle_reference(osc, sensor, spp_b, batches, max_depth, seed0=1000, threads=0, **kw):
    """mean image and per-sample variance from `batches` independent oracle renders of spp_b samples each"""
    imgs = []
    for b in range(batches):
        img, _ = osc.render_path(sensor, seed=seed0 + b, spp=spp_b, max_depth=max_depth, threads=threads, **kw)
        imgs.append(img.astype(np.float64))
    imgs = np.stack(imgs)
    return imgs.mean(axis=0), imgs.var(axis=0, ddof=1) * spp_b, spp_b * batches
