"""qat-zstd-plugin_amd — MI355X-native ZSTD block-level sequence producer.

The product is the C library in ``lib/`` (``libqatseqprod.so`` / ``.a``; public headers in
``../include``): plain-C host code (``host/qatseqprod.c``) over a thin HIP C ABI
(``csrc/qzstd_kernels.hip``).  This Python package is only a loader for tests / bench /
the driver's build check; it adds no behaviour of its own and has no CPU fallback.
"""
import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "lib", "libqatseqprod.so")

REQUIRED_SYMBOLS = (
    "QZSTD_version", "QZSTD_startQatDevice", "QZSTD_stopQatDevice", "QZSTD_createSeqProdState",
    "QZSTD_freeSeqProdState", "qatSequenceProducer", "QZSTD_hintSource", "QZSTD_hintSourceEx", "qzstd_hip_find_sequences",
)


def load() -> ctypes.CDLL:
    """dlopen the product library; raises loudly when it has not been built."""
    if not os.path.isfile(LIB_PATH):
        raise OSError("%s not built: run `make -C %s` or __graft_entry__.build()" % (LIB_PATH, PKG_DIR))
    lib = ctypes.CDLL(LIB_PATH)
    for s in REQUIRED_SYMBOLS:
        if not hasattr(lib, s):
            raise OSError("libqatseqprod.so lacks symbol " + s)
    lib.QZSTD_version.restype = ctypes.c_char_p
    return lib
