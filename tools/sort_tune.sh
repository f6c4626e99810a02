#!/bin/bash
# Block shape of the index's radix passes by cloud size (flood_index.hip: flooder_index_sort_zeroed, option "sort_shape"):
# whole index build, median of 15, for 0.1 - 16 M points; and the sort's own tests.  Run on the GPU box from the repo root.
set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
python tools/time_index.py by_size 0 2>/dev/null
python tools/time_index.py 512x8 1 2>/dev/null
python tools/time_index.py 1024x16 2 2>/dev/null
python tools/time_index.py 1024x8 3 2>/dev/null
python -m pytest tests -m gpu -x -q -k "index_sort" 2>&1 | tail -2
