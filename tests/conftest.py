"""pytest configuration: `gpu` marker, library fixtures.

`-m "not gpu"` (CPU container): oracle vs golden vectors, host logic, C-ABI symbol checks.
`-m gpu` (MI355X box): the HIP path against the oracle, through the C ABI.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

import qz_bind as B  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    # build products are git-ignored; (re)build when missing.  hipcc cross-compiles without a GPU.
    if not os.path.isfile(B.ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    if not os.path.isfile(B.PLUGIN_SO):
        subprocess.check_call(["make", "-C", B.PKG_DIR, "ZSTDLIB=" + B.find_libzstd()], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def zstd():
    return B.Zstd()


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    return B.Oracle()


@pytest.fixture(scope="session")
def plugin():
    _ensure_built()
    return B.Plugin()


@pytest.fixture(scope="session")
def gpu_plugin(plugin):
    """The product library on a box with a GPU; fails loudly (no CPU fallback) otherwise."""
    n = plugin.lib.qzstd_hip_device_count()
    assert n > 0, "no HIP device visible: " + plugin.err()
    return plugin
