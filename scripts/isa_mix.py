#!/usr/bin/env python3
"""Static instruction mix of the kernels in a `hipcc -S --cuda-device-only` listing: per kernel (name filter = regex on the mangled
name) the counts of MFMA, other VALU, LDS, global-memory and scalar instructions, whole function and per loop (a loop = the
lines between a label and the last backward branch to it; nested loops are reported separately).
usage: isa_mix.py listing.s regex"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2])


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem" if not op.startswith("scratch_") else "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


i = 0
while i < len(txt):
    m = re.match(r"^(_Z\S+):\s", txt[i])
    if not m or not pat.search(m.group(1)):
        i += 1
        continue
    name, body, j = m.group(1), [], i + 1
    while j < len(txt) and not txt[j].startswith("\t.section") and not re.match(r"^\s*s_endpgm", txt[j]) or (j < len(txt) and "s_endpgm" in txt[j] and False):
        body.append(txt[j])
        j += 1
        if ".Lfunc_end" in txt[j - 1]:
            break
    labels, ins = {}, []
    for ln in body:
        lm = re.match(r"^(\.LBB\S+):", ln)
        if lm:
            labels[lm.group(1)] = len(ins)
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".")):
            continue
        ins.append(s.split(";")[0].strip())
    tot = {}
    for s in ins:
        tot[classify(s)] = tot.get(classify(s), 0) + 1
    print(name[:110])
    print("  whole:", " ".join(f"{k}={v}" for k, v in sorted(tot.items())))
    loops = []
    for k, s in enumerate(ins):
        bm = re.match(r"^s_cbranch\S*\s+(\.LBB\S+)|^s_branch\s+(\.LBB\S+)", s)
        if bm:
            lab = bm.group(1) or bm.group(2)
            if lab in labels and labels[lab] <= k:
                loops.append((labels[lab], k))
    for a, b in sorted(set(loops)):
        c = {}
        for s in ins[a:b + 1]:
            c[classify(s)] = c.get(classify(s), 0) + 1
        if b - a > 40:
            print(f"  loop [{a}, {b}] ({b - a + 1} instr):", " ".join(f"{k}={v}" for k, v in sorted(c.items())))
    i = j
