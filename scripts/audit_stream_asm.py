#!/usr/bin/env python3
"""Audit of the hand-placed ticket atomic in pa_nd_hex_stream.hip (inline assembly the compiler does not track).

For every nd_hex_stream_kernel instantiation in the generated gfx950 assembly: between the `global_atomic_add ... sc0`
that draws the next batch (marker PA_TICKET_ISSUE) and the explicit `s_waitcnt vmcnt(0)` in front of its first use
(marker PA_TICKET_WAIT) no instruction may read or write the destination VGPR, the wait must name the same register, and
the kernel must not spill.  usage: audit_stream_asm.py file.s  (hipcc --cuda-device-only -S pa_nd_hex_stream.hip)"""
import re
import sys


def audit(text):
    problems, kernels = [], 0
    for name, body in re.findall(r"^(_ZN2pa20nd_hex_stream_kernel\w+):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
        kernels += 1
        lines = body.split("\n")
        issue = [i for i, l in enumerate(lines) if "PA_TICKET_ISSUE" in l]
        wait = [i for i, l in enumerate(lines) if "PA_TICKET_WAIT" in l]
        if len(issue) != 1 or len(wait) != 1:
            problems.append(f"{name}: expected one issue / one wait, found {len(issue)} / {len(wait)}")
            continue
        m = re.search(r"global_atomic_add\s+(v\d+),", lines[issue[0]])
        reg = m.group(1)
        if not re.search(r"PA_TICKET_WAIT\s+" + reg + r"\b", lines[wait[0]]):
            problems.append(f"{name}: wait names {lines[wait[0]].strip()} but the atomic returns into {reg}")
        if wait[0] < issue[0]:
            problems.append(f"{name}: wait precedes the issue")
        num = int(reg[1:])
        for l in lines[issue[0] + 1:wait[0]]:
            code = l.split(";")[0]
            hit = re.search(r"\b" + reg + r"\b", code)
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", code):
                hit = hit or int(a) <= num <= int(b)
            if hit:
                problems.append(f"{name}: {reg} touched while the atomic is in flight: {l.strip()}")
        for key in (".vgpr_spill_count", ".sgpr_spill_count"):
            pass
    for name, n in re.findall(r"\.name:\s+(_ZN2pa20nd_hex_stream_kernel\w+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
        if int(n):
            problems.append(f"{name}: {n} spilled VGPRs")
    return kernels, problems


if __name__ == "__main__":
    k, p = audit(open(sys.argv[1]).read())
    print(f"{k} nd_hex_stream_kernel instantiations audited, {len(p)} problems")
    for x in p:
        print("  " + x)
    sys.exit(1 if p or k == 0 else 0)
