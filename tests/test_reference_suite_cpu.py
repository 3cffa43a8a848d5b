"""Runs the REFERENCE'S OWN test functions (read from /root/reference at test time; nothing is copied) against the
CPU oracle and against the product's HAR_HD code compiled for the host, through the `mitsuba` / `drjit` stand-ins of
tests/ref_shim.  These are the golden vectors that pin the microfacet model.  Skipped when the
reference tree is not mounted (e.g. on the GPU box)."""
import ctypes as C
import os

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")

MICROFACET_TESTS = ["test02_eval_pdf_beckmann", "test03_smith_g1_beckmann", "test04_sample_beckmann", "test03_smith_g1_ggx", "test05_sample_ggx"]


@pytest.fixture(scope="module")
def harness():
    H = C.CDLL(os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"))
    H.hh_fresnel_conductor.restype = C.c_float
    return H


@pytest.mark.parametrize("backend", ["oracle", "product"])
def test_reference_microfacet_golden_vectors(O, harness, backend):
    """src/render/tests/test_microfacet.py:15-285 (Mitsuba 0.6 golden data for eval / pdf / smith_g1 / sample)"""
    import numpy as np
    from tests.ref_shim import make_modules, run_reference_tests
    mi, dr = make_modules(backend, O=O, H=harness)
    ran = run_reference_tests(os.path.join(REF, "src/render/tests/test_microfacet.py"), MICROFACET_TESTS, mi, dr, extra={"np": np})
    assert ran == MICROFACET_TESTS
