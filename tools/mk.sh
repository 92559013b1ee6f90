#!/bin/bash
# build the product library (and the front-end against the callers' libzstd: tools/zstdshim when it works), optionally A/B variants:
#   tools/mk.sh                      the product
#   tools/mk.sh debug                the profiling build
#   tools/mk.sh NAME "XFLAGS"...     variants (pairs), in parallel
R=$(cd "$(dirname "$0")/.." && pwd)
Z=$(python3 $R/tools/qz_bind.py --libzstd)
if [ $# -eq 0 ]; then make -C $R/qat-zstd-plugin_amd ZSTDLIB=$Z 2>&1 | grep -E "error|warning:|Error" ; exit 0; fi
if [ "$1" = debug ]; then make -C $R/qat-zstd-plugin_amd debug 2>&1 | grep -E "error|warning:|Error"; exit 0; fi
make -C $R/qat-zstd-plugin_amd build/qatseqprod.o > /dev/null
while [ $# -ge 2 ]; do
  (make -C $R/qat-zstd-plugin_amd variant NAME=$1 XFLAGS="$2" > /tmp/var_$1.log 2>&1 || { echo "FAIL $1"; grep -E "error" /tmp/var_$1.log | head -5; }) &
  shift 2
done
wait
