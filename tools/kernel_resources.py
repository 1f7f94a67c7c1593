"""tools/kernel_resources.py [pattern] — rebuild libfmhr_ham.so with the compiler's resource-usage remarks and print one line per kernel:
VGPRs, spills, SGPRs, scratch, LDS, occupancy.  hipcc cross-compiles: no GPU needed."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
env = dict(os.environ, FMHR_FORCE_BUILD="1", FMHR_EXTRA_FLAGS=os.environ.get("FMHR_EXTRA_FLAGS", "") + " -Rpass-analysis=kernel-resource-usage")
out = subprocess.run([os.path.join(ROOT, "fmhr_amd", "csrc", "build.sh")], env=env, capture_output=True, text=True)
cur, rows = None, {}
for line in (out.stdout + out.stderr).splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
for n, d in zip(names, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "").replace("fmhr::", "")
    if pat and pat not in d:
        continue
    r = rows[n]
    g = lambda k: r.get(k, "?")
    print(f"{d:44s} VGPR {g('VGPRs'):>4} spill {g('VGPRs Spill'):>3} SGPR {g('SGPRs'):>4} sspill {g('SGPRs Spill'):>3} scratch {g('ScratchSize [bytes/lane]'):>5} "
          f"LDS {g('LDS Size [bytes/block]'):>6} occ {g('Occupancy [waves/SIMD]')}")
