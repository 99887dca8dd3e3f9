#!/usr/bin/env python3
"""Static check of a hipcc -S listing: after every inline-asm block of plain (non-sc1) global loads in a kernel, no
register copy or spill (v_mov / v_accvgpr_write / scratch_store) may read the loaded registers before the next
`s_waitcnt vmcnt(0)` (linear order, wrapping once at the loop back edge) -- the signature of hipcc merging an old and a new
value of an asm-defined variable.  Guards the asm-prefetch idiom of gru_persist.hip (tests/test_host_api.py runs it).
usage: check_inflight.py listing.s [kernel-name-substring]"""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "gru_persist_bwd_kernel"
bad = 0
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = body.split("\n")

    def regs(l):
        out = set()
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l):
            out.update(range(int(a), int(b) + 1))
        out.update(int(x) for x in re.findall(r"\bv(\d+)\b", l))
        return out

    loops = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
    i = 0
    blocks = []
    while i < len(lines):
        if "#ASMSTART" in lines[i]:
            j = i + 1
            dst = set()
            plain = False
            while "#ASMEND" not in lines[j]:
                l = lines[j].strip()
                if l.startswith("global_load") and "sc1" not in l:
                    plain = True
                    dst |= regs(l.split(",")[0])
                j += 1
            if plain:
                blocks.append((j, dst))
            i = j
        i += 1
    for end, dst in blocks:
        k = end + 1
        wrapped = False
        while True:
            if k >= len(lines):
                if wrapped or not loops or end < loops[0]:
                    break
                k, wrapped = loops[0], True
                continue
            l = lines[k].strip()
            if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                break
            is_copy = l.startswith(("v_mov_b", "v_accvgpr_write", "scratch_store", "v_pk_mov"))
            if is_copy and "," in l and regs(l.split(",", 1)[1]) & dst:       # a compiler copy / spill READING them
                print("%s: line %d touches in-flight v%s: %s" % (name[:60], k + 1, sorted(regs(l) & dst), l))
                bad += 1
            k += 1
    print("%-90s %d asm load blocks checked" % (name[:90], len(blocks)))
sys.exit(1 if bad else 0)
