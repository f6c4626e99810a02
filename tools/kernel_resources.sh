#!/bin/bash
# Register / LDS / spill figures of the kernels of one source: tools/kernel_resources.sh flood_wit.hip [name filter] [extra hipcc flags]
SRC=flooder_amd/csrc/$1; FLT=${2:-}; shift; [ $# -gt 0 ] && shift
T=$(mktemp -d)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DFLOODER_BUILD "$@" --cuda-device-only -c $SRC -o $T/dev.bundle 2>/dev/null || { echo "compile failed"; exit 1; }
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if flt in name:
        print(f\"{name[:100]:100s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>5} vspill {g('vgpr_spill_count'):>3} sspill {g('sgpr_spill_count'):>3}\")
" "$FLT"
[ -n "${KEEP_CO:-}" ] && cp $T/dev.co $KEEP_CO
rm -rf $T
