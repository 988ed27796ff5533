#!/usr/bin/env python3
"""print VGPR / SGPR / spill / LDS of every kernel in a hipcc -save-temps gfx950 .s file (the metadata YAML at its end)"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if pat and not re.search(pat, name):
        continue
    print("%-90s vgpr %4s agpr %3s sgpr %3s spill %3s scratch %4s lds %6s" % (
        name.replace("_ZN3pgo3dev", "")[:90], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
