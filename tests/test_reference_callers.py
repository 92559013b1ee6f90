"""The reference's own callers build UNCHANGED against this repo's headers and library (drop-in check).

/root/reference/test/test.c and test/benchmark.c include <zstd.h>, <zstd_errors.h> and "qatseqprod.h" and link
-lqatseqprod (reference test/Makefile:37-43).  When the reference tree is present (the build container; it does not exist
on the GPU box) they are compiled as they are — never copied — against include/ + include/compat/ and
lib/libqatseqprod.so, and run: without a GPU the plugin reports the device as down, libzstd's fallback takes over
(test.c enables ZSTD_c_enableSeqProducerFallback; benchmark.c does not, so it runs with -m0 here — its plugin calls are
still linked and its software path exercised), and both must report success."""
import os
import subprocess

import pytest

import qz_bind as B
import qz_corpus as K

REF = "/root/reference/test"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "test.c")), reason="reference tree not present")
@pytest.mark.parametrize("prog,args", [("test", []), ("benchmark", ["-m0", "-t2", "-l2", "-c64K", "-L3"])])
def test_reference_caller_builds_unchanged_and_runs(tmp_path, plugin, prog, args):
    zlib = B.find_libzstd()
    exe = str(tmp_path / prog)
    cmd = ["gcc", "-O2", "-o", exe, os.path.join(REF, prog + ".c"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "include", "compat"), "-L" + os.path.join(B.PKG_DIR, "lib"), "-lqatseqprod", zlib,
           "-Wl,-rpath," + os.path.join(B.PKG_DIR, "lib"), "-Wl,-rpath," + os.path.dirname(zlib), "-lpthread", "-lm"]
    subprocess.check_call(cmd)
    f = tmp_path / "in.bin"
    f.write_bytes(K.text(1, 300000))
    out = subprocess.run([exe] + args + [str(f)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout + out.stderr)[-800:]
    text = out.stdout + out.stderr
    assert "FAIL" not in text and "failed" not in text.lower() and ("PASS" in text or "successful" in text), text[-800:]


def test_install_tree_is_self_contained(tmp_path, plugin):
    """`make install DESTDIR=...` lays down the reference's artefacts under the reference's names (src/Makefile:89-103) and a
    program builds against nothing but that tree (+ libzstd); `make uninstall` removes them again."""
    dest = tmp_path / "root"
    subprocess.check_call(["make", "-C", B.PKG_DIR, "install", "DESTDIR=" + str(dest)], stdout=subprocess.DEVNULL)
    lib, inc = dest / "usr/local/lib", dest / "usr/local/include"
    for f in (lib / "libqatseqprod.so", lib / "libqatseqprod.a", inc / "qatseqprod.h"):
        assert f.is_file(), f
    src = tmp_path / "use.c"
    src.write_text('#include <stdio.h>\n#include "qatseqprod.h"\n'
                   'int main(void) { int rc = QZSTD_startQatDevice(); void *s = QZSTD_createSeqProdState();\n'
                   '  printf("version %s start %d state %s\\n", QZSTD_VERSION, rc, s ? "ok" : "null");\n'
                   '  QZSTD_freeSeqProdState(s); QZSTD_stopQatDevice(); return s ? 0 : 1; }\n')
    zlib = B.find_libzstd()
    exe = str(tmp_path / "use")
    subprocess.check_call(["gcc", "-O1", "-o", exe, str(src), "-I" + str(inc), "-L" + str(lib), "-lqatseqprod", zlib,
                           "-Wl,-rpath," + str(lib), "-Wl,-rpath," + os.path.dirname(zlib), "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "state ok" in out.stdout, out.stdout + out.stderr
    subprocess.check_call(["make", "-C", B.PKG_DIR, "uninstall", "DESTDIR=" + str(dest)], stdout=subprocess.DEVNULL)
    assert not (lib / "libqatseqprod.so").exists() and not (inc / "qatseqprod.h").exists()
