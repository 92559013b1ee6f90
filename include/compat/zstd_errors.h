/* compat/zstd_errors.h — see compat/zstd.h.  The reference's callers include it but use nothing from it beyond what
 * qzstd_zstd_abi.h declares (ZSTD_isError / ZSTD_getErrorName). */
#ifndef QZSTD_COMPAT_ZSTD_ERRORS_H
#define QZSTD_COMPAT_ZSTD_ERRORS_H
#endif
