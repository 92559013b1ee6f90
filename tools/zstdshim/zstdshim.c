/* zstdshim — a well-built libzstd 1.5.7 for the MEASUREMENT legs, found inside the image (round 4).
 *
 * Not part of the product: libqatseqprod.so does not know or care which libzstd calls its producer.  This is about the caller.
 *
 * The sequence-producer API needs libzstd >= 1.5.4.  The only shared library of that age on the ROCm image is the copy bundled with
 * Pillow (pillow.libs/libzstd-*.so.1.5.7), and that build is about four times slower than a normal one (level 1, 128 KiB blocks, one core:
 * 169 MB/s against 690 MB/s for the distribution's libzstd 1.4.8) — so every end-to-end figure of rounds 1-3 measured that library's
 * entropy stage, not the producer.  A normal build of zstd 1.5.7 IS on the image: pyarrow's libarrow.so links one statically (684 MB/s on
 * the same input, byte-identical output to the Pillow copy).  Its ZSTD_* functions are not exported, but libarrow.so ships its full symbol
 * table, so they can be found by name: this library dlopens libarrow.so, reads .symtab from the file, and forwards each public ZSTD_* entry
 * point (zstd_names.h) to load address + st_value through one `jmp *slot(%rip)` trampoline.  No zstd code is compiled or copied here.
 *
 * Used by bench.py's end-to-end legs and tests/test_zstdshim.py (tools/qz_bind.py: fast_libzstd()); everything falls back to the Pillow copy
 * when this library cannot be built or cannot find libarrow.so (zstdshim_ok() == 0), and the bench line says which libzstd it ran
 * (e2e.libzstd).  $QZ_ZSTD_ARROW_SO names the file to read instead of the search below.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <elf.h>
#include <fcntl.h>
#include <glob.h>
#include <link.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define X(name) #name,
static const char *const kNames[] = {
#include "zstd_names.h"
};
#undef X
#define N_NAMES (sizeof(kNames) / sizeof(kNames[0]))

static void zstdshim_missing(void)
{
    fprintf(stderr, "zstdshim: a ZSTD_* function was called that could not be resolved from libarrow.so (zstdshim_ok() == 0?)\n");
    abort();
}

/* one slot per name, in the order of zstd_names.h; the trampolines below jump through them */
__attribute__((visibility("hidden"))) void *zstdshim_slots[N_NAMES];

#define TRAMP_(i, name)                                                                                                         \
    __asm__(".text\n.globl " #name "\n.type " #name ",@function\n" #name ":\n\tjmp *zstdshim_slots+8*" #i "(%rip)\n.size " #name ", .-" #name "\n");
#define TRAMP(i, name) TRAMP_(i, name)
#define X(name) TRAMP(__COUNTER__, name)
#include "zstd_names.h"
#undef X
_Static_assert(__COUNTER__ == N_NAMES, "the trampolines' slot numbers start at 0 and follow zstd_names.h");

static int gOk;
static unsigned gResolved;
static char gSource[1024];

static const char *const kSearch[] = {
    "/usr/local/lib/python3*/dist-packages/pyarrow/libarrow.so.*",
    "/usr/lib/python3*/dist-packages/pyarrow/libarrow.so.*",
    "/usr/lib/python3*/site-packages/pyarrow/libarrow.so.*",
    "/opt/conda/lib/python3*/site-packages/pyarrow/libarrow.so.*",
};

static int find_arrow(char *out, size_t cap)
{
    const char *env = getenv("QZ_ZSTD_ARROW_SO");
    size_t i;
    if (env && *env) {
        snprintf(out, cap, "%s", env);
        return access(out, R_OK) == 0;
    }
    for (i = 0; i < sizeof(kSearch) / sizeof(kSearch[0]); i++) {
        glob_t g;
        int found = 0;
        if (glob(kSearch[i], 0, NULL, &g) == 0 && g.gl_pathc > 0) {
            snprintf(out, cap, "%s", g.gl_pathv[g.gl_pathc - 1]);
            found = 1;
        }
        globfree(&g);
        if (found) return 1;
    }
    return 0;
}

/* st_value of every wanted FUNC symbol of the file's .symtab (local ones included: that is the point) */
static unsigned read_symtab(const char *path, Elf64_Addr *values)
{
    struct stat st;
    unsigned found = 0;
    int fd = open(path, O_RDONLY);
    const unsigned char *m;
    if (fd < 0) return 0;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(Elf64_Ehdr)) { close(fd); return 0; }
    m = (const unsigned char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 0;
    {
        const Elf64_Ehdr *eh = (const Elf64_Ehdr *)m;
        if (memcmp(eh->e_ident, ELFMAG, SELFMAG) == 0 && eh->e_ident[EI_CLASS] == ELFCLASS64 && eh->e_shoff != 0 &&
            eh->e_shoff + (size_t)eh->e_shnum * sizeof(Elf64_Shdr) <= (size_t)st.st_size) {
            const Elf64_Shdr *sh = (const Elf64_Shdr *)(m + eh->e_shoff);
            unsigned s;
            for (s = 0; s < eh->e_shnum; s++) {
                const Elf64_Sym *sym;
                const char *str;
                size_t n, k, strSize;
                if (sh[s].sh_type != SHT_SYMTAB || sh[s].sh_link >= eh->e_shnum) continue;
                if (sh[s].sh_offset + sh[s].sh_size > (size_t)st.st_size) continue;
                sym = (const Elf64_Sym *)(m + sh[s].sh_offset);
                n = sh[s].sh_size / sizeof(Elf64_Sym);
                str = (const char *)(m + sh[sh[s].sh_link].sh_offset);
                strSize = sh[sh[s].sh_link].sh_size;
                if (sh[sh[s].sh_link].sh_offset + strSize > (size_t)st.st_size) continue;
                for (k = 0; k < n; k++) {
                    const char *nm;
                    size_t i;
                    if (ELF64_ST_TYPE(sym[k].st_info) != STT_FUNC || sym[k].st_shndx == SHN_UNDEF || sym[k].st_name >= strSize) continue;
                    nm = str + sym[k].st_name;
                    if (nm[0] != 'Z' || nm[1] != 'S' || nm[2] != 'T' || nm[3] != 'D' || nm[4] != '_') continue;
                    for (i = 0; i < N_NAMES; i++)
                        if (values[i] == 0 && strcmp(nm, kNames[i]) == 0) {
                            values[i] = sym[k].st_value;
                            found++;
                            break;
                        }
                }
            }
        }
    }
    munmap((void *)m, (size_t)st.st_size);
    return found;
}

__attribute__((constructor)) static void zstdshim_init(void)
{
    static Elf64_Addr values[N_NAMES];
    struct link_map *lm = NULL;
    void *h;
    size_t i;
    for (i = 0; i < N_NAMES; i++) zstdshim_slots[i] = (void *)zstdshim_missing;
    if (!find_arrow(gSource, sizeof(gSource))) {
        snprintf(gSource, sizeof(gSource), "(no libarrow.so found)");
        return;
    }
    h = dlopen(gSource, RTLD_NOW | RTLD_LOCAL); /* kept open for the life of the process */
    if (!h || dlinfo(h, RTLD_DI_LINKMAP, &lm) != 0 || !lm) return;
    gResolved = read_symtab(gSource, values);
    for (i = 0; i < N_NAMES; i++)
        if (values[i]) zstdshim_slots[i] = (void *)(lm->l_addr + values[i]);
    /* usable when the whole public API resolved and the copy is new enough for the sequence-producer API */
    if (gResolved == N_NAMES) {
        unsigned (*ver)(void) = NULL;
        for (i = 0; i < N_NAMES; i++)
            if (strcmp(kNames[i], "ZSTD_versionNumber") == 0) ver = (unsigned (*)(void))zstdshim_slots[i];
        gOk = ver && ver() >= 10504;
    }
}

/* 1 when every name of zstd_names.h was found and the copy is >= 1.5.4 */
int zstdshim_ok(void) { return gOk; }
unsigned zstdshim_resolved(void) { return gResolved; }
unsigned zstdshim_wanted(void) { return (unsigned)N_NAMES; }
/* the file the functions were taken from */
const char *zstdshim_source(void) { return gSource; }
