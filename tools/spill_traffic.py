"""How much of a kernel's instruction stream is SGPR-spill traffic, loop by loop: v_readlane_b32 / v_writelane_b32 (what a
spilled SGPR costs on gfx950 - reload / save through a lane of a VGPR) inside every backward branch's body.
usage: KEEP_CO=/tmp/k.co bash tools/kernel_resources.sh flood_wit.hip >/dev/null; python tools/spill_traffic.py /tmp/k.co wit_sweep_kernelILi3"""
import re, subprocess, sys
co, pat = sys.argv[1], sys.argv[2]
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
for f in re.split(r'\n(?=[0-9a-f]+ <)', txt):
    name = f.split('\n', 1)[0]
    if pat not in name:
        continue
    lines = f.split('\n')
    base = int(re.match(r'([0-9a-f]+) <', name).group(1), 16)
    rl = [i for i, l in enumerate(lines) if 'v_readlane_b32' in l]
    wl = [i for i, l in enumerate(lines) if 'v_writelane_b32' in l]
    sc = [i for i, l in enumerate(lines) if 'scratch_' in l]
    addr = {}
    for i, l in enumerate(lines):
        m = re.search(r'//\s*([0-9A-F]{12}):', l)
        if m:
            addr[int(m.group(1), 16)] = i
    print(name[:100]); print(f"  {len(lines)} instructions, {len(rl)} v_readlane, {len(wl)} v_writelane, {len(sc)} scratch accesses")
    out = []
    for i, l in enumerate(lines):
        m = re.search(r'(s_cbranch_\w+|s_branch)\s.*<[^>]*\+0x([0-9a-fA-F]+)>', l)
        m2 = re.search(r'//\s*([0-9A-F]{12}):', l)
        if m and m2 and base + int(m.group(2), 16) < int(m2.group(1), 16):
            lo = addr.get(base + int(m.group(2), 16))
            if lo is not None and i - lo >= 40:
                out.append((i - lo, sum(lo <= j <= i for j in rl), sum(lo <= j <= i for j in wl), sum(lo <= j <= i for j in sc)))
    for n, r, w, s in sorted(out):
        print(f"    loop of {n:5d} instructions: {r:4d} readlane ({100 * r / n:4.1f} %), {w:3d} writelane, {s:3d} scratch")
