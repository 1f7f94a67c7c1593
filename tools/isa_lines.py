"""Static instruction counts of one kernel per SOURCE LINE, from an assembly listing with line tables:

    cd fmhr_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off -gline-tables-only -x hip \
        --cuda-device-only -S raster.hip -o /tmp/raster.s
    python tools/isa_lines.py /tmp/raster.s k_raster_trisILb1ELb1ELb1E [--top 40]

Counts are STATIC (an instruction inside a loop counts once): they say where the code is, not how often it runs — read them next to the
issue counters (tools/pmc_issue.py: VALU instructions per wave).  Columns: VALU / SALU / LDS / VMEM / other instructions of the line."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    files, cur, inside = {}, None, False
    counts = collections.defaultdict(lambda: [0, 0, 0, 0, 0])
    order = []
    for ln in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if not inside:
            if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', ln):
                inside = True
            continue
        if ln.startswith(".Lfunc_end"):
            break
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r'\s+([a-z_0-9]+)', ln)
        if not m or cur is None:
            continue
        op = m.group(1)
        cls = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3 if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else 4
        if op.startswith(("s_waitcnt", "s_nop")):
            cls = 4
        if cur not in counts:
            order.append(cur)
        counts[cur][cls] += 1
    tot = [sum(c[i] for c in counts.values()) for i in range(5)]
    print("total  VALU %d  SALU %d  LDS %d  VMEM %d  other %d" % tuple(tot))
    rows = sorted(counts.items(), key=lambda kv: -kv[1][0])[:top]
    for (f, l), c in rows:
        print("%-18s %5d   VALU %4d  SALU %4d  LDS %3d  VMEM %3d" % (f, l, c[0], c[1], c[2], c[3]))
    if "--ranges" in sys.argv:                                        # raster.hip line ranges: lo-hi,lo-hi,...
        for r in sys.argv[sys.argv.index("--ranges") + 1].split(","):
            f, lo, hi = r.split(":")
            s = [sum(c[i] for (ff, l), c in counts.items() if ff == f and int(lo) <= l <= int(hi)) for i in range(4)]
            print("%s:%s-%s  VALU %d  SALU %d  LDS %d  VMEM %d" % (f, lo, hi, *s))


if __name__ == "__main__":
    main()
