set -u
cd ${GRAFT_REPO_ROOT:-$PWD}
python tools/time_index.py auto 0 2>/dev/null
python tools/time_index.py 512x8 1 2>/dev/null
python tools/time_index.py 1024x16 2 2>/dev/null
python tools/time_index.py 1024x8 3 2>/dev/null
python tools/time_index.py auto 0 2>/dev/null
python -m pytest tests -m gpu -x -q -k "index_sort or index_of or bvh_build" 2>&1 | tail -2
