"""GPU box: close the evidence chain golden fixtures -> oracle -> GPU without a live oracle in between.

* every row of tests/golden/sw_sizes.json: the sequences the HIP path produces hash (FNV-1a over offset, litLength,
  matchLength of every sequence, the same function as qzo_seq_stats) to the COMMITTED oracle_fnv, block by block;
* the freshly compiled oracle is compared with the same fixtures on this box too (the CPU suite is deselected by
  `-m gpu`, so without this the oracle that judges the GPU here would never meet its own fixtures here);
* the north-star ratio bound (within 2 % of libzstd's own match-finder at the same level) at every BASELINE level,
  through the real callback path: level 1 / 3 (config 2, 5), level 6 on 128 KiB text blocks (config 3), level 12 on
  32 KiB web-log blocks (config 4).
"""
import json
import os

import numpy as np
import pytest

import qz_bind as B
import qz_corpus as K
import test_oracle_golden as TG

pytestmark = pytest.mark.gpu

ROWS = TG.load_rows()["rows"]


def fnv1a_of_sequences(a: np.ndarray) -> int:
    """FNV-1a (64 bit) over the little-endian bytes of (offset, litLength, matchLength) of every sequence"""
    h = 1469598103934665603
    for byte in np.ascontiguousarray(a[:, :3].astype("<u4")).tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("row", ROWS, ids=lambda r: "%s-s%d-b%d-L%x" % (r["gen"], r["seed"], r["block"], r["level"]))
def test_gpu_sequences_hash_to_the_committed_golden(gpu_plugin, row):
    data = K.by_name(row["gen"], row["block"] * row["blocks"], row["seed"])
    blocks = [data[o:o + row["block"]] for o in range(0, len(data), row["block"])]
    counts, seqs, stride = gpu_plugin.find_batch(blocks, row["level"])
    a = np.frombuffer(seqs, dtype=np.uint32).reshape(-1, 4)
    nseq = summ = 0
    for i in range(len(blocks)):
        n = counts[i]
        assert n != B.NSEQ_ERROR and n >= 1
        s = a[i * stride:i * stride + n]
        assert int(s[:, 1].sum() + s[:, 2].sum()) == len(blocks[i])
        assert "%016x" % fnv1a_of_sequences(s) == row["oracle_fnv"][i], "block %d of %s" % (i, row["gen"])
        nseq += n
        summ += int(s[:, 2].sum())
    assert (nseq, summ) == (row["oracle_nseq"], row["oracle_sum_match"])


@pytest.mark.parametrize("row", ROWS, ids=lambda r: "%s-s%d-b%d-L%x" % (r["gen"], r["seed"], r["block"], r["level"]))
def test_oracle_built_on_this_box_matches_golden(oracle, row):
    TG.test_oracle_matches_golden_stats(oracle, row)


def test_golden_ratio_rows_on_this_box():
    TG.test_level1_ratio_within_2pct_of_software_on_fixtures()
    TG.test_repcode_aware_rows_beat_software_with_external_repcode_search()
    rows = {(r["gen"], r["block"], r["level"]): r for r in ROWS}
    for key in (("weblog", 32768, 12), ("text", 131072, 6)):  # BASELINE configs 4 and 3 at fixture scale
        assert rows[key]["oracle_size"] <= 1.02 * rows[key]["sw_size"], key
