#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel.

usage: hipcc ... -Rpass-analysis=kernel-resource-usage --cuda-device-only -c x.hip -o /dev/null 2> log; kernel_resources.py log [filter]
"""
import re
import sys

text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
keys = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "SGPRs Spill", "VGPRs Spill",
        "LDS Size \\[bytes/block\\]"]
short = ["vgpr", "agpr", "sgpr", "scratch", "occ", "sspill", "vspill", "lds"]
for blk in re.split(r"remark: Function Name: ", text)[1:]:
    name = blk.split(" ")[0]
    if pat and not re.search(pat, name):
        continue
    vals = []
    for k, s in zip(keys, short):
        m = re.search(r"remark:\s+" + k + r": (\d+)", blk)
        vals.append(f"{s}={m.group(1) if m else '?'}")
    print(name, " ".join(vals))
