"""Executable fuzzing (reference: test/fuzzing/README.md:9-28 drives upstream zstd's round-trip targets through the
FUZZ_* adapter; those need a zstd source tree, which this image lacks).  tests/fuzz/fuzz_roundtrip.c is an in-repo
randomised round-trip driver over the same surface (incl. the five FUZZ_* symbols):

* CPU: host/qatseqprod.c + profile + the adapter + the mock device layer (oracle as the kernel), everything compiled with
  -fsanitize=address,undefined, so heap overflows / use-after-free / UB in the host logic abort the run;
* GPU box (-m gpu): the same driver against lib/libqatseqprod.so — the real kernels under random sizes, levels,
  block-size limits, streaming, announcements and rewritten buffers.

Both compare frames with the oracle's: the driver compresses the same case a second time with qzo_sequence_producer registered and
memcmp's the two frames (every iteration on the GPU, every third under ASan).
"""
import os
import subprocess

import pytest

import qz_bind as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "fuzz", "fuzz_roundtrip.c")
ADAPTER = os.path.join(B.PKG_DIR, "test", "fuzzing", "qatseqprodfuzzer.c")
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle")]


def test_fuzz_host_logic_under_asan_ubsan_over_the_mock(tmp_path):
    zlib = B.find_libzstd()
    exe = str(tmp_path / "fuzz_mock")
    srcs = [SRC, ADAPTER, os.path.join(B.PKG_DIR, "host", "qatseqprod.c"), os.path.join(B.PKG_DIR, "csrc", "qzstd_profile.c"),
            os.path.join(ROOT, "tests", "mock", "mock_hip.c"), os.path.join(ROOT, "oracle", "qzstd_oracle.c")]
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-pthread"] + INC + ["-o", exe] + srcs +
                          [zlib, "-Wl,-rpath," + os.path.dirname(zlib)])
    # bounded for the CPU suite (the oracle under ASan walks ~1 MB/s at the chain levels): buffers up to 384 KiB
    # (the mock's service runs the oracle once per work item, each over the block up to the item's end: the default 32 items per block
    # only in the first run, coarser items in the others)
    for seed, iters, env in ((1, 16, {}), (2, 30, {"QZSTD_HIP_SERVICE_ITEM": "32768"}),
                             (3, 20, {"QZSTD_HIP_COALESCE": "0", "QZSTD_HIP_SERVICE_ITEM": "65536"}), (5, 16, {"QZSTD_MOCK_PROGRESSIVE": "1", "QZSTD_HIP_SERVICE_ITEM": "16384"}),
                             (4, 20, {"QZSTD_MOCK_DEVICES": "3", "QZSTD_HIP_EXT_REPCODES": "1", "QZSTD_HIP_SERVICE_ITEM": "32768"})):
        out = subprocess.run([exe, str(seed), str(iters), "384", "3"], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", **env))
        assert out.returncode == 0 and "fuzz ok" in out.stdout, (env, (out.stdout + out.stderr)[-1500:])


@pytest.mark.gpu
def test_fuzz_real_kernels(tmp_path, gpu_plugin):
    zlib = B.find_libzstd()
    exe = str(tmp_path / "fuzz_gpu")
    subprocess.check_call(["gcc", "-O2", "-g", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-pthread"] + INC + ["-o", exe, SRC, ADAPTER,
                           os.path.join(ROOT, "oracle", "qzstd_oracle.c"),  # the checker, linked into the TEST binary only
                           "-L" + os.path.join(B.PKG_DIR, "lib"), "-lqatseqprod", zlib, "-Wl,-rpath," + os.path.join(B.PKG_DIR, "lib"),
                           "-Wl,-rpath," + os.path.dirname(zlib)])
    for seed, iters, env in ((11, 150, {}), (12, 100, {"QZSTD_HIP_SERVICE": "0"}), (13, 60, {"QZSTD_HIP_COALESCE": "0"}),
                             (14, 60, {"QZSTD_HIP_EXT_REPCODES": "1"})):
        out = subprocess.run([exe, str(seed), str(iters), "3072", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert out.returncode == 0 and "fuzz ok" in out.stdout, (env, (out.stdout + out.stderr)[-1500:])
        # every iteration's frame was also produced through the oracle's producer and compared byte for byte (f3: parity, not a property)
        # (an iteration whose callbacks were served as joined blocks of a finer announced grid is checked list by list instead)
        import re
        m = re.search(r"(\d+) frames identical to the oracle's", out.stdout)
        assert m and int(m.group(1)) >= iters - 10, out.stdout[-400:]
