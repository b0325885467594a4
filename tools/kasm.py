#!/usr/bin/env python3
"""Development tool: print the assembly of one kernel from rust-ida_amd/csrc/*.s (after `make asm`).
usage: tools/kasm.py <substring of the mangled name> [file.s] > kernel.s"""
import sys, re
pat = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "rust-ida_amd/csrc/idahip.s"
out, on = [], False
for line in open(path):
    if not on and re.match(r"^_Z\w*:", line) and pat in line:
        on = True
    if on:
        out.append(line)
        if line.strip().startswith(".end_amdhsa_kernel") or line.strip() == "s_endpgm" and False:
            pass
        if line.startswith(".Lfunc_end"):
            break
sys.stdout.write("".join(out))
