#!/usr/bin/env python3
"""Issue profile of the step's kernels: what each wave spends its instructions on, how full its vector instructions are and
what the memory pipeline sees.  rocprofv3 --pmc passes with --kernel-trace only (one group of counters per pass), each over
`bench.py --steps 10 --warmup 2`; per kernel the mean over its dispatches of ONE pass (the passes are normalised separately).

    gpurun --timeout 900 -- 'python3 tools/pmc_issue.py r02_v62'        -> gpurun_out/r02_v62_issue.txt

Derived columns (per launch):
  valu/wave    vector-ALU instructions per wave            (SQ_INSTS_VALU / SQ_WAVES)
  lanes        mean active lanes per vector instruction    (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 4: both count quad-cycles)
  valu_us      SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / clock: the time the vector ALUs alone need at one wave64 instruction per
               4 cycles (tools/micro/valu_rate.hip: 4.3 nominal cycles per simple op); compare with the kernel's duration
  ldsconf_us   SQ_LDS_BANK_CONFLICT (extra LDS-array cycles, summed over the chip) / 256 CUs / clock: the time ONE CU's LDS array
               spends on bank conflicts (MI355X_MICROARCH.md §LDS); ldsact_us the same for SQ_ACTIVE_INST_LDS (quad-cycles x 4)
  conf/ldsi    conflict cycles per LDS instruction
(TA_* / TCP_* counters are not collected: a pass with them hung rocprofv3 on this pool for its whole limit.)
"""
import csv
import glob
import os
import signal
import subprocess
import sys
from collections import defaultdict

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "issue"
BENCH_ARGS = os.environ.get("BENCH_ARGS", "").split()
PASSES = [
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE",
    "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU",
    "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT",
    "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum",
]
if os.environ.get("PASSES"):
    PASSES = [PASSES[int(i)] for i in os.environ["PASSES"].split(",")]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.chdir("/tmp")
os.environ["TMPDIR"] = "/tmp"
vals = defaultdict(dict)            # kernel -> counter -> mean per dispatch
dur = defaultdict(list)
for p, counters in enumerate(PASSES):
    if p == 0 and "SQ_WAVES" not in counters:
        raise SystemExit("the first pass must hold SQ_WAVES / SQ_INSTS_VALU")
    out = os.path.join(ROOT, "gpurun_out", f"{TAG}_issue{p}")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters.split(), "-d", out, "-o", "pmc", "--output-format", "csv", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-stage1", "--no-run-length", *BENCH_ARGS]
    # own process group: on a timeout the profiler AND the bench child that holds the GPU are killed together
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        stdout, _ = proc.communicate(timeout=int(os.environ.get("PASS_TIMEOUT", "200")))
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        stdout, _ = proc.communicate()
        open(out + ".log", "w").write(stdout or "")
        print(f"pass {p} timed out and was killed; no further GPU pass is started", flush=True)
        sys.exit(3)
    open(out + ".log", "w").write(stdout)
    if proc.returncode != 0:
        print(f"pass {p} failed (rc {proc.returncode}):", stdout[-1500:])
        continue
    f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(f"pass {p}: no counter_collection.csv")
        continue
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "fmhr::" not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
    for k in acc:
        for c, v in acc[k].items():
            vals[k][c] = v / max(1, len(disp[k]))
    kt = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
    if kt and p == 0:
        for row in csv.DictReader(open(kt[0])):
            if "fmhr::" in row["Kernel_Name"]:
                dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    print(f"pass {p} done", flush=True)

CLK = 2.4e9
lines = []
order = sorted(vals, key=lambda k: -vals[k].get("SQ_INSTS_VALU", 0))
hdr = ("kernel", "us", "waves", "valu/wave", "salu/wave", "lds/wave", "vmrd/wave", "vmwr/wave", "lanes", "valu_us", "int32%", "int64%", "fma%", "trans%",
       "ldsconf_us", "ldsact_us", "conf/ldsi", "l2hit%", "l2req", "ea_atom")
lines.append(" ".join(h.rjust(9) if i else h.ljust(34) for i, h in enumerate(hdr)))
for k in order:
    v = vals[k]
    w = max(v.get("SQ_WAVES", 0.0), 1.0)
    iv = v.get("SQ_INSTS_VALU", 0.0)
    d = sorted(dur.get(k, [0.0]))
    us = d[len(d) // 2]
    lanes = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(v.get("SQ_ACTIVE_INST_VALU", 0.0), 1.0) / 4.0
    hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    row = (k.replace("fmhr::", "").replace("void ", "")[:34], us, w, iv / w, v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w,
           v.get("SQ_INSTS_VMEM_RD", 0) / w, v.get("SQ_INSTS_VMEM_WR", 0) / w, lanes, iv * 4 / 1024 / CLK * 1e6,
           100 * v.get("SQ_INSTS_VALU_INT32", 0) / max(iv, 1), 100 * v.get("SQ_INSTS_VALU_INT64", 0) / max(iv, 1),
           100 * v.get("SQ_INSTS_VALU_FMA_F32", 0) / max(iv, 1), 100 * v.get("SQ_INSTS_VALU_TRANS_F32", 0) / max(iv, 1),
           v.get("SQ_LDS_BANK_CONFLICT", 0) / 256 / CLK * 1e6, v.get("SQ_ACTIVE_INST_LDS", 0) * 4 / 256 / CLK * 1e6,
           v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_INSTS_LDS", 0), 1),
           100 * hit / max(hit + miss, 1), hit + miss, v.get("TCC_EA0_ATOMIC_sum", 0))
    lines.append(" ".join((("%.4g" % x).rjust(9) if i else x.ljust(34)) for i, x in enumerate(row)))
lines.append("")
lines.append("raw counters (mean per dispatch):")
for k in order:
    lines.append(k[:60])
    lines.append("   " + "  ".join(f"{c}={vals[k][c]:.4g}" for c in sorted(vals[k])))
txt = "\n".join(lines)
open(os.path.join(ROOT, "gpurun_out", f"{TAG}_issue.txt"), "w").write(txt + "\n")
print("\n".join(lines[:14]))
