import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _oracle_models_the_products_pll_bound(request):
    """The oracle's default is the reference's behaviour.  Tests marked `gpu` compare with the HIP product, whose SAM PLL bounds the
    reference's unbounded phase-wrap loops (AudioSDR.cpp:735-736): their oracles model that bound."""
    from oracle import asdr_oracle
    asdr_oracle.PRODUCT_PLL_BOUND = request.node.get_closest_marker("gpu") is not None
    yield
    asdr_oracle.PRODUCT_PLL_BOUND = False


@pytest.fixture(scope="session")
def ao():
    """The CPU oracle binding (built on demand with gcc)."""
    from oracle import asdr_oracle
    asdr_oracle.build()
    asdr_oracle.lib()
    return asdr_oracle


@pytest.fixture(scope="session")
def A():
    """The product binding; the HIP library is built on demand (hipcc cross-compiles without a GPU).
    PyTorch (the device buffers of the full-size tests) is imported BEFORE the library is loaded: this image's torch wheel bundles its
    own libamdhip64, and torch finds no device when /opt/rocm's copy -- the one libasdr_hip.so names -- was mapped first (imported
    first, torch's copy serves both)."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from audiosdr_amd import build as b
    b.build()
    import audiosdr_amd
    audiosdr_amd.load_library()
    return audiosdr_amd


@pytest.fixture(scope="session")
def gpu(A):
    """Skips (loudly named) ONLY when there is no HIP device; any other asdr_create failure (allocation, table upload,
    bad ordinal) fails the suite instead of skipping it green.  GPU tests call through the C ABI only.
    """
    try:
        b = A.AudioSDRBatch(1, device=0)
    except A.AsdrError as e:
        if "no HIP device" in str(e):
            pytest.skip("no HIP device: %s" % e)
        raise
    b.close()
    return A
