#!/usr/bin/env python3
"""Static look at the FAST segment path of the stage-B kernels in a -save-temps listing: VALU instructions, global memory operations
and the VALU instructions that only do address / bookkeeping work.   usage: python tools/isa_addr.py <listing.s>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
for name in ("ILi1E", "ILi2E", "ILi0E"):
    a = s.index("_ZN3fmx13stageb_kernel%sEEvNS_10StageBArgsE:" % name)
    b = s.index(".end_amdhsa_kernel", a)
    body = s[a:b].split("\n")
    idx = [i for i, l in enumerate(body) if "SB_PHASE_END" in l]
    f1 = [i for i in idx if " F1" in body[i]]
    lo = max(i for i in idx if i < f1[0]); hi = f1[-1]
    c = collections.Counter(); g = 0
    for t in body[lo:hi]:
        t = t.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            c[re.sub(r"_e(32|64)$", "", op)] += 1
        if op.startswith(("global_", "buffer_")):
            g += 1
    keys = ["v_lshl_add_u64", "v_ashrrev_i32", "v_mad_u64_u32", "v_lshlrev_b64", "v_add_co_u32", "v_addc_co_u32", "v_mov_b32", "v_readlane_b32",
            "v_writelane_b32", "v_cndmask_b32"]
    print(name, "VALU", sum(c.values()), "global ops", g, " ".join("%s=%d" % (k.replace("v_", ""), c[k]) for k in keys))
