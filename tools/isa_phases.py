#!/usr/bin/env python3
"""Static instruction counts of a kernel between the `; SB_PHASE_END k` markers of its assembly listing (hipcc -save-temps):
VALU / f64 / transcendental / SALU / LDS / VMEM / scratch / barriers per phase.
usage: python tools/isa_phases.py <listing.s> <kernel name substring>"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(_Z\w*%s\w*):\s*(;.*)?$" % re.escape(name), s, re.M)
a = m.start()
b = s.index(".end_amdhsa_kernel", a)
cur = dict(valu=0, f64=0, trans=0, salu=0, lds=0, vmem=0, scratch=0, bar=0, cyc=0)
rows = []
label = "start"
for line in s[a:b].split("\n"):
    t = line.strip()
    mm = re.match(r"; SB_PHASE_END (\d+ ?F?\d?)", t)
    if mm:
        rows.append((label, cur)); cur = dict.fromkeys(cur, 0); label = "after %s" % mm.group(1)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur["valu"] += 1
        c = 4
        if "f64" in op:
            cur["f64"] += 1; c = 8 if ("add" in op or "cvt" in op or "mov" in op) else 16
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
            cur["trans"] += 1; c = 16
        cur["cyc"] += c
    elif op.startswith("s_barrier"):
        cur["bar"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith("scratch_"):
        cur["scratch"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        cur["vmem"] += 1
rows.append((label, cur))
print("%-10s %6s %5s %5s %6s %5s %5s %7s %4s %8s" % ("phase", "VALU", "f64", "trans", "SALU", "LDS", "VMEM", "scratch", "bar", "~cycles"))
for lab, c in rows:
    print("%-10s %6d %5d %5d %6d %5d %5d %7d %4d %8d" % (lab, c["valu"], c["f64"], c["trans"], c["salu"], c["lds"], c["vmem"], c["scratch"], c["bar"], c["cyc"]))
