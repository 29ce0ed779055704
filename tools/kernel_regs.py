"""Developer tool: VGPR / SGPR / scratch (spill) figures of every kernel in a hipcc --save-temps ISA listing.
usage: python3 tools/kernel_regs.py FILE.s [name filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        dem = name
    if flt and flt not in dem:
        continue
    g = lambda k: (re.search(k + r'\s+(\d+)', body) or [None, "?"])[1]
    print(f"{dem[:120]:120s} vgpr {g(r'.amdhsa_next_free_vgpr')} sgpr {g(r'.amdhsa_next_free_sgpr')} acc_off {g(r'.amdhsa_accum_offset')} scratch {g(r'.amdhsa_private_segment_fixed_size')}")
