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


FRESNEL_TESTS = ["test01_fresnel", "test02_fresnel_polarized_vec", "test03_fresnel_conductor", "test04_snell"]


@pytest.mark.parametrize("backend", ["oracle", "product"])
def test_reference_fresnel_known_answers(O, harness, backend):
    """src/render/tests/test_fresnel.py:6-82 -- the unpolarised functions the dielectric / conductor / plastic BSDFs call (fresnel.h:38-116): normal incidence
    (4 %), the critical angle, the hyperphysics spot checks at 45 and 10 degrees, index-matched interfaces, conductor == dielectric for a real index, Snell's law.
    (`test02_fresnel_polarized_vec` is the unpolarised `mi.fresnel` on 20 angles despite its name; the polarised tests are outside an rgb variant.)"""
    import numpy as np
    from tests.ref_shim import make_modules, run_reference_tests
    mi, dr = make_modules(backend, O=O, H=harness)
    ran = run_reference_tests(os.path.join(REF, "src/render/tests/test_fresnel.py"), FRESNEL_TESTS, mi, dr, extra={"np": np})
    assert ran == FRESNEL_TESTS


def test_reference_discrete_distribution_known_answers(O, harness):
    """src/core/tests/test_distr_1d.py:46-86,107-113 -- DiscreteDistribution::sample / sample_pmf / sample_reuse / sample_reuse_pmf on the hand-computed
    [1, 3, 2] table (values below 0 and above 1, either side of a bucket boundary, the re-used sample) and on a table with leading and trailing zeros:
    the face choice of the mesh area lights (Mesh::sample_position, src/render/mesh.cpp:1662-1712) in the oracle"""
    import numpy as np
    from tests.ref_shim import make_modules, run_reference_tests
    mi, dr = make_modules("oracle", O=O, H=harness)
    names = ["test05_discr_sample", "test07_discr_leading_trailing_zeros"]
    assert run_reference_tests(os.path.join(REF, "src/core/tests/test_distr_1d.py"), names, mi, dr, extra={"np": np}) == names
