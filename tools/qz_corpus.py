"""Deterministic corpora for tests and bench.py (no network, no Silesia/enwik in the image).

Two families:

* ``system_corpus`` — a Silesia-like mix assembled from files that ship with the ROCm
  image (python sources, license prose, XML, a collation table, ELF binaries, JSON).
  The GPU box runs the same image, so the bytes are reproducible there; every part is
  optional and the mix falls back to the seeded generators below when a file is absent.
* seeded generators (``text``, ``binary_struct``, ``weblog``, ``mixed_entropy``,
  ``mix``) — the shapes SURVEY.md §8(d) names for BASELINE configs 1-5.

Nothing here reads /root/reference.
"""
from __future__ import annotations

import glob
import os

import numpy as np

KiB = 1024
MiB = 1024 * 1024


# --------------------------------------------------------------------------- generators
def _vocab(rng: np.random.Generator, n_words: int) -> list[bytes]:
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    probs = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4,
                      2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
    probs = probs / probs.sum()
    lens = np.clip(rng.poisson(4.2, n_words) + 1, 1, 14)
    words = []
    for ln in lens:
        words.append(bytes(rng.choice(letters, size=int(ln), p=probs)))
    return words


def text(seed: int, size: int) -> bytes:
    """Zipf-distributed pseudo-English with punctuation, capitals and paragraphs."""
    rng = np.random.default_rng(seed)
    vocab = _vocab(rng, 20000)
    out = bytearray()
    ranks = np.arange(1, len(vocab) + 1, dtype=np.float64)
    p = 1.0 / ranks ** 1.05
    p /= p.sum()
    while len(out) < size:
        n = 4096
        idx = rng.choice(len(vocab), size=n, p=p)
        punct = rng.random(n)
        cap_next = True
        for k in range(n):
            w = vocab[idx[k]]
            if cap_next:
                w = w[:1].upper() + w[1:]
                cap_next = False
            out += w
            r = punct[k]
            if r < 0.08:
                out += b". "
                cap_next = True
                if r < 0.01:
                    out += b"\n\n"
            elif r < 0.16:
                out += b", "
            else:
                out += b" "
    return bytes(out[:size])


def binary_struct(seed: int, size: int) -> bytes:
    """Fixed-width little-endian records: counters, slowly varying floats, flags, ids."""
    rng = np.random.default_rng(seed)
    n = size // 32 + 1
    rec = np.zeros(n, dtype=[("id", "<u4"), ("ts", "<u8"), ("x", "<f4"), ("y", "<f4"),
                             ("flags", "<u2"), ("kind", "u1"), ("pad", "u1"), ("ref", "<u4"),
                             ("crc", "<u4")])
    rec["id"] = np.arange(n, dtype=np.uint32) + 100000
    rec["ts"] = 1_700_000_000_000 + np.cumsum(rng.integers(1, 50, n)).astype(np.uint64)
    rec["x"] = np.cumsum(rng.normal(0, 0.01, n)).astype(np.float32)
    rec["y"] = np.round(rng.normal(10, 2, n), 1).astype(np.float32)
    rec["flags"] = rng.choice([0, 1, 2, 4, 0x10, 0x8000], n).astype(np.uint16)
    rec["kind"] = rng.choice(8, n, p=[.5, .2, .1, .08, .05, .04, .02, .01]).astype(np.uint8)
    rec["ref"] = rng.zipf(1.3, n).astype(np.uint32)
    rec["crc"] = rng.integers(0, 2 ** 32, n, dtype=np.uint32)
    return rec.tobytes()[:size]


def weblog(seed: int, size: int) -> bytes:
    """Templated access-log lines with Zipf-distributed fields (BASELINE config 4)."""
    rng = np.random.default_rng(seed)
    paths = [b"/", b"/index.html", b"/api/v1/items", b"/api/v1/users", b"/static/app.js",
             b"/static/site.css", b"/img/logo.png", b"/search", b"/login", b"/cart"]
    paths += [b"/product/%d" % i for i in range(400)]
    agents = [b"Mozilla/5.0 (X11; Linux x86_64) AppleWebKit/537.36 (KHTML, like Gecko) Chrome/120.0 Safari/537.36",
              b"Mozilla/5.0 (Macintosh; Intel Mac OS X 10_15_7) AppleWebKit/605.1.15 (KHTML, like Gecko) Version/17.1 Safari/605.1.15",
              b"Mozilla/5.0 (Windows NT 10.0; Win64; x64; rv:121.0) Gecko/20100101 Firefox/121.0",
              b"curl/8.4.0", b"Googlebot/2.1 (+http://www.google.com/bot.html)"]
    status = [b"200", b"200", b"200", b"200", b"304", b"404", b"302", b"500"]
    out = bytearray()
    t = 1_700_000_000
    while len(out) < size:
        n = 2048
        ip = rng.zipf(1.2, (n, 2)) % 250
        pi = np.minimum(rng.zipf(1.3, n) - 1, len(paths) - 1)
        ai = np.minimum(rng.zipf(1.6, n) - 1, len(agents) - 1)
        si = rng.integers(0, len(status), n)
        sz = rng.integers(200, 60000, n)
        dt = rng.integers(0, 3, n)
        for k in range(n):
            t += int(dt[k])
            out += b"10.%d.%d.%d - - [%d] \"GET %s HTTP/1.1\" %s %d \"-\" \"%s\"\n" % (
                ip[k, 0], ip[k, 1], (ip[k, 0] * 7 + ip[k, 1]) % 250, t, paths[pi[k]],
                status[si[k]], sz[k], agents[ai[k]])
    return bytes(out[:size])


def mixed_entropy(seed: int, size: int, seg: int = 64 * KiB) -> bytes:
    """Segments cycling 0 / 2 / 4 / 6 / 8 bits-per-byte of entropy (BASELINE config 5)."""
    rng = np.random.default_rng(seed)
    out = bytearray()
    k = 0
    while len(out) < size:
        bits = (0, 2, 4, 6, 8)[k % 5]
        if bits == 0:
            out += bytes([k & 0xFF]) * seg
        else:
            out += rng.integers(0, 1 << bits, seg, dtype=np.uint8).tobytes()
        k += 1
    return bytes(out[:size])


def incompressible(seed: int, size: int) -> bytes:
    return np.random.default_rng(seed).integers(0, 256, size, dtype=np.uint8).tobytes()


def mix(seed: int, size: int) -> bytes:
    """text / binary-struct / web-log / low-entropy / incompressible segments (config 2 fallback)."""
    parts = []
    seg = max(size // 16, 64 * KiB)
    gens = [text, binary_struct, weblog, text, mixed_entropy, text, binary_struct, incompressible]
    k = 0
    total = 0
    while total < size:
        g = gens[k % len(gens)]
        parts.append(g(seed * 1000 + k, seg))
        total += seg
        k += 1
    return b"".join(parts)[:size]


# ------------------------------------------------------------------------ system corpus
_SYSTEM_PARTS = [
    # (label, glob pattern(s), max bytes taken)  — sorted file order, concatenated
    ("py_source", ["/usr/lib/python3.10/*.py", "/usr/lib/python3.10/*/*.py"], 24 * MiB),
    ("prose_html", ["/usr/local/lib/python3.10/dist-packages/kaleido/executable/CREDITS.html"], 16 * MiB),
    ("xml", ["/usr/share/mime/packages/freedesktop.org.xml",
             "/usr/local/lib/python3.10/dist-packages/torch/share/rccl/msccl-algorithms/allreduce-allpairs-8n-simple.xml"], 8 * MiB),
    ("table_txt", ["/usr/share/perl/5.34.0/Unicode/Collate/allkeys.txt"], 8 * MiB),
    ("elf", ["/usr/bin/python3.10", "/usr/lib/x86_64-linux-gnu/libc.so.6",
             "/usr/lib/x86_64-linux-gnu/libstdc++.so.6", "/usr/bin/perl"], 16 * MiB),
    ("json", ["/usr/local/lib/python3.10/dist-packages/plotly/validators/_validators.json",
              "/usr/local/lib/python3.10/dist-packages/dash_svg/metadata.json"], 8 * MiB),
    ("doc_text", ["/usr/share/doc/*/copyright", "/usr/share/common-licenses/*"], 8 * MiB),
]


def system_corpus_parts(max_total: int | None = None) -> list[tuple[str, bytes]]:
    """Labelled parts of the Silesia-like system corpus (missing files are skipped)."""
    parts: list[tuple[str, bytes]] = []
    total = 0
    for label, pats, cap in _SYSTEM_PARTS:
        buf = bytearray()
        for pat in pats:
            for fn in sorted(glob.glob(pat)):
                if len(buf) >= cap:
                    break
                try:
                    if os.path.isfile(fn):
                        with open(fn, "rb") as f:
                            buf += f.read(cap - len(buf))
                except OSError:
                    continue
        if buf:
            parts.append((label, bytes(buf)))
            total += len(buf)
            if max_total and total >= max_total:
                break
    return parts


def system_corpus(size: int, min_real: int = 4 * MiB) -> tuple[bytes, str]:
    """`size` bytes of the system corpus, repeated/cut like "Silesia repeated to 1 GiB".

    Returns (data, provenance).  Falls back to ``mix(seed=2)`` when fewer than
    ``min_real`` bytes of system files are readable.
    """
    parts = system_corpus_parts()
    real = b"".join(p for _, p in parts)
    if len(real) < min_real:
        return mix(2, size), "synthetic:mix(seed=2)"
    prov = "system:" + "+".join("%s(%d)" % (l, len(p)) for l, p in parts)
    reps = -(-size // len(real))
    return (real * reps)[:size], prov


def by_name(name: str, size: int, seed: int = 1) -> bytes:
    if name == "system":
        return system_corpus(size)[0]
    return {"text": text, "binary": binary_struct, "weblog": weblog, "mixed_entropy": mixed_entropy,
            "mix": mix, "random": incompressible}[name](seed, size)
