"""GPU box, GPU HIDDEN: BASELINE config 1 ("plugin registered but no QAT present -> libzstd software fallback") on the box that has the
hardware (round-5 verdict, "Next round" 8).  HIP_VISIBLE_DEVICES=-1 makes the runtime report no device; the plugin must then behave as
on a machine without a GPU — start fails (/root/reference/src/qatseqprod.c:798-845: QZSTD_FAIL when no instance is found), every
callback returns ZSTD_SEQUENCE_PRODUCER_ERROR, libzstd's ZSTD_c_enableSeqProducerFallback produces the software frame, and WITHOUT the
fallback the call fails loudly: no CPU path inside the plugin.  Child processes: the variable is read when HIP starts."""
import os
import subprocess
import sys

import pytest

import qz_bind as B
import qz_corpus as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "tools"))
import qz_bind as B, qz_corpus as K
plug = B.Plugin(); L = plug.lib; z = B.Zstd()
assert L.qzstd_hip_device_count() <= 0, "the GPU is not hidden"
assert L.QZSTD_startQatDevice() != 0          # QZSTD_FAIL: nothing to start
data = K.text(1, 131072)                       # config 1: ONE 128 KiB block
st = L.QZSTD_createSeqProdState()
for level in (1, 6, 12):
    zc = z.cctx(level, producer=plug.producer_addr, state=st, fallback=True)
    frame = z.compress2(zc, data); z.free(zc)
    zc = z.cctx(level); sw = z.compress2(zc, data); z.free(zc)
    assert frame == sw, "level %%d: the fallback frame is not the software frame" %% level
    assert z.decompress(frame, len(data)) == data
    zc = z.cctx(level, producer=plug.producer_addr, state=st, fallback=False)
    try:
        z.compress2(zc, data)
        raise SystemExit("level %%d: no error without the fallback — a CPU path inside the plugin?" %% level)
    except RuntimeError as e:
        assert "sequence producer" in str(e).lower(), e
    z.free(zc)
L.QZSTD_freeSeqProdState(st)
L.QZSTD_stopQatDevice()
print("HIDDEN-GPU OK")
'''


def test_config1_fallback_with_the_gpu_hidden(gpu_plugin):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0 and "HIDDEN-GPU OK" in out.stdout, (out.stdout + out.stderr)[-3000:]


def test_reference_shaped_test_program_with_the_gpu_hidden(gpu_plugin, tmp_path):
    """qat-zstd-plugin_amd/test/test (the counterpart of the reference's test/test.c: start, register, fallback on, one ZSTD_compress2,
    decompress, compare) exits 0 with the GPU hidden, exactly as on a machine without one"""
    tdir = os.path.join(B.PKG_DIR, "test")
    subprocess.check_call(["make", "-C", tdir, "ZSTDLIB=" + B.find_libzstd()], stdout=subprocess.DEVNULL)
    f = tmp_path / "dickens_like.bin"
    f.write_bytes(K.text(1, 131072))
    out = subprocess.run([os.path.join(tdir, "test"), str(f)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, HIP_VISIBLE_DEVICES="-1"))
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr
