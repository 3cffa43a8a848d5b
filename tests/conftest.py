import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    _gpu_preflight(config)


def _gpu_preflight(config):
    """`-m gpu` sessions only, before this process touches the HIP runtime: a PyTorch-only host-to-device copy + reduction in a CHILD process.
    Twice in ~70 leases of round 2 every process started on a box died with "Memory access fault by GPU" inside the runtime's copy of its first
    upload -- with builds that are clean on every other box.  If the child dies that way, this session takes the
    runtime's other copy path (shader blits instead of the SDMA engines; the variable is read when the runtime initialises, i.e. after this
    point) and says so; the tests themselves are unchanged.  No GPU / no torch: nothing happens here, the tests report that themselves."""
    expr = (config.getoption("-m", default="") or "").strip()
    if "gpu" not in expr or "not gpu" in expr or os.environ.get("HAR_TEST_PREFLIGHT", "1") == "0":
        return
    import subprocess
    code = ("import sys, torch\n"
            "if not torch.cuda.is_available(): sys.exit(3)\n"
            "x = torch.ones(1 << 22, dtype=torch.float32).to('cuda'); assert float(x.sum().item()) == float(1 << 22)\n")
    try:
        p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    except Exception:
        return
    if p.returncode < 0 or p.returncode in (134, 139):
        os.environ["HSA_ENABLE_SDMA"] = "0"
        sys.stderr.write("[conftest] the PyTorch-only GPU preflight died (return code %d): %s\n[conftest] continuing with HSA_ENABLE_SDMA=0\n"
                         % (p.returncode, p.stderr.decode(errors="replace").strip().splitlines()[-1:] or ""))


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Make sure the oracle, the host harness and the HIP library exist (hipcc cross-compiles on CPU)."""
    import __graft_entry__ as g
    need = [os.path.join(ROOT, "oracle", "libmi_oracle.so"),
            os.path.join(ROOT, "tests", "host_harness", "libhost_harness.so"),
            os.path.join(ROOT, "mitsuba3_amd", "libhip_ad_rgb.so")]
    if not all(os.path.exists(p) for p in need):
        g.build()


@pytest.fixture(scope="session")
def mi():
    import mitsuba3_amd as mi
    mi.set_variant("hip_ad_rgb")
    return mi


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    return oracle
