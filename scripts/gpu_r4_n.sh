#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r4_n"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
timeout 120 python scripts/prof_stft.py > "$OUT/prof_stft.log" 2>&1; tail -n 1 "$OUT/prof_stft.log"
timeout 900 python -m pytest tests -m gpu -x -q -k "stft or mel or audio" -p no:cacheprovider > "$OUT/pytest_a.log" 2>&1
tail -n 3 "$OUT/pytest_a.log"
