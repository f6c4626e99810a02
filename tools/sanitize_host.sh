#!/bin/bash
# The host-parallel C++ of round 6 (csrc/delaunay_nd.cpp, cell_faces.cpp) under AddressSanitizer + UBSan and under
# ThreadSanitizer (CPU only: GPU sanitizers are not available on this pool).   tools/sanitize_host.sh
cd "$(dirname "$0")/.."
for san in address,undefined thread; do
  g++ -O1 -g -std=c++17 -pthread -fsanitize=$san -fno-omit-frame-pointer -o /tmp/flooder_san tools/sanitize_host_main.cpp \
      flooder_amd/csrc/delaunay_nd.cpp flooder_amd/csrc/cell_faces.cpp || exit 1
  echo "== -fsanitize=$san"
  for cfg in "4 300 4" "6 120 8" "2 500 3" "8 40 4" "3 400 1"; do
    /tmp/flooder_san $cfg 2>&1 | grep -i "cells,\|sanitizer\|ERROR\|WARNING" | cut -c1-200
  done
done
