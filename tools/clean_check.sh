#!/bin/bash
# A clean checkout builds and passes the CPU suite (round-4 verdict, weak 3: "331/344 green" had never been run from one — a stale,
# git-ignored binary hid a Makefile whose default goal built nothing).  git archive HEAD -> a temporary directory ->
# __graft_entry__.build() -> plain `make` in test/ -> pytest -m "not gpu".  Run it before a round ends:  tools/clean_check.sh [pytest args]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/qz_clean.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive HEAD | tar -x -C "$TMP"
cd "$TMP"
if ! git -C "$ROOT" diff --quiet HEAD; then echo "clean_check: NOTE the working tree has uncommitted changes; checking HEAD"; fi
python -c "import __graft_entry__ as g; g.build()" > build.log 2>&1 || { tail -30 build.log; echo "clean_check: build() FAILED"; exit 1; }
make -C qat-zstd-plugin_amd/test > make_test.log 2>&1 || { tail -30 make_test.log; echo "clean_check: make -C test FAILED"; exit 1; }
for t in test benchmark frontbench replaybench; do [ -x qat-zstd-plugin_amd/test/$t ] || { echo "clean_check: plain make did not build test/$t"; exit 1; }; done
python -m pytest tests/ -q -m "not gpu" -p no:cacheprovider "${@:--n4}" 2>&1 | tail -5
echo "clean_check: done ($(git -C "$ROOT" rev-parse --short HEAD))"
