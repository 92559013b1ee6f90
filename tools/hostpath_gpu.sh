#!/bin/bash
# GPU box: tests/stress/hostpath_bench.c against the REAL library — announcements two ahead, every block taken by qatSequenceProducer, no
# libzstd behind the callbacks: what one GPU + these host cores sustain through the announcement path when the entropy stage is not the
# limit (round-4 verdict, weak 7: "nothing shows the front-end above ~25 GB/s per GPU").   usage: tools/hostpath_gpu.sh
python - <<PY
import sys; sys.path.insert(0, "tools"); import qz_corpus as K
open("/tmp/hp.bin","wb").write(K.system_corpus(512 << 20)[0])
PY
L=$PWD/qat-zstd-plugin_amd/lib
gcc -O2 -g -std=c11 -pthread -Iinclude -o /tmp/hostpath_gpu tests/stress/hostpath_bench.c -L$L -lqatseqprod -Wl,-rpath,$L || exit 1
for lv in ${LEVELS:-1}; do for seg in ${SEGS:-2 4 8}; do for t in ${THREADS:-4 8 16 32}; do
  /tmp/hostpath_gpu /tmp/hp.bin $t 6 $seg $lv 2>&1 | sed 's/; the mock.*host path alone [0-9]* MB\/s//' | cut -c1-220
done; done; done
