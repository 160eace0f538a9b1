#!/usr/bin/env python3
"""Host: registers, spills and LDS of every kernel in a `hipcc -S --cuda-device-only` file: python tools/regs.py file.s [name-fragment]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
frag = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.index("amdhsa.kernels"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = re.sub(r"\(.*", "", name)
    if frag not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {blk.splitlines()[0].strip():>4s} spill {g('vgpr_spill_count'):>4s} "
          f"sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
