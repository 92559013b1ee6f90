/*
 * compat/zstd.h — forwarding header for build environments that have a libzstd >= 1.5.4 shared object but no matching
 * header (this ROCm image: only /opt/conda/include/zstd.h 1.4.9).  Callers written against the reference —
 * /root/reference/test/test.c:45-51, test/benchmark.c:48-52 include <zstd.h>, <zstd_errors.h> and "qatseqprod.h" —
 * compile unchanged with  -I<repo>/include -I<repo>/include/compat.  With a real zstd.h >= 1.5.4 on the include path,
 * leave this directory out.
 */
#ifndef QZSTD_COMPAT_ZSTD_H
#define QZSTD_COMPAT_ZSTD_H
#include "../qzstd_zstd_abi.h"
#endif
