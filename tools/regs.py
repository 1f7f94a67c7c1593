#!/usr/bin/env python3
"""VGPR / spill / LDS of every kernel of the library (hipcc -S per source, no GPU needed).  Run after every kernel edit: the
occupancy boundaries (waves per SIMD = floor(512 / VGPRs rounded up to 8)) are one register wide — DESIGN.md §5.3."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fmhr_amd", "csrc")
STRICT = {"raster.hip", "antialias.hip"}          # -ffp-contract=off, as in build.sh
want = sys.argv[1:]                               # optional substrings of kernel names
for f in ("raster.hip", "antialias.hip", "shade.hip", "mesh.hip", "ncc.hip", "stage1.hip"):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math",
               "-ffp-contract=" + ("off" if f in STRICT else "on"), *os.environ.get("FMHR_EXTRA_FLAGS", "").split(), "-x", "hip", "-S",
               "--cuda-device-only", os.path.join(SRC, f),
               "-o", tmp.name]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        s = open(tmp.name).read()
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", s, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        short = re.sub(r"^_ZN4fmhr\d+", "", name)
        short = re.sub(r"E(Pv|PK|NS|P\d|v).*$", "", short)[:28]
        if want and not any(w in name for w in want):
            continue
        v = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
        sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
        waves = min(8, 512 // (((v + 7) // 8) * 8)) if v else 8
        print(f"{f:14s} {short:28s} vgpr {v:4d}  spills {sp:3d}  lds {lds:6d}  waves/SIMD {waves}")
