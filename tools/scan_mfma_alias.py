#!/usr/bin/env python3
"""Developer / CI tool: compile every csrc/*.hip to gfx950 ISA and list the MFMAs whose destination registers overlap their A or B operand
(without being the in-place C accumulate).  hipcc 7.2 emits such an MFMA when an operand fragment is dead behind it and C is not tied to the
destination (e.g. the constant 0); v_mfma_f32_16x16x32_{bf16,f16} then corrupts its own result (round 4: igemm_xgeglu_kernel<K = 512>, caught by
tests/test_gpu_ops.py::test_geglu_projection_kernel).  Exit status 1 when any is found in a kernel not listed in ALLOW.
usage: python3 tools/scan_mfma_alias.py [file.hip ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "diffusion-classifier_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))
pat = re.compile(r"\s*(v_mfma\S+) (v\[(\d+):(\d+)\]), (v\[(\d+):(\d+)\]|\S+), (v\[(\d+):(\d+)\]|\S+), (\S+)")
bad = 0
with tempfile.TemporaryDirectory() as tmp:
    for f in files:
        out = os.path.join(tmp, f + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{SRC}", "-S", "--cuda-device-only",
                        "-Wno-unused-command-line-argument", "-o", out, os.path.join(SRC, f)], check=True, stderr=subprocess.DEVNULL)
        cur, n, hits = "?", 0, {}
        for line in open(out):
            if line.startswith(".amdhsa_kernel") or (line[:1] == "_" and line.rstrip().endswith(":")) or (line[:2] == "_Z"):
                if line.rstrip().endswith(":"):
                    cur = line.strip()[:-1]
            m = pat.match(line)
            if not m:
                continue
            n += 1
            d0, d1 = int(m.group(3)), int(m.group(4))
            for k in (5, 8):
                if m.group(k + 1) is None:
                    continue
                s0, s1 = int(m.group(k + 1)), int(m.group(k + 2))
                if not (s1 < d0 or s0 > d1):
                    hits.setdefault(cur, []).append(line.strip())
        print(f"{f}: {n} MFMAs, {sum(len(v) for v in hits.values())} with a destination overlapping A or B")
        for k, v in hits.items():
            demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print(f"   {demangled[:110]}: {len(v)}   e.g. {v[0]}")
            bad += len(v)
sys.exit(1 if bad else 0)
