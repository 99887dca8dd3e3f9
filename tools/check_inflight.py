#!/usr/bin/env python3
"""Static check of a hipcc -S listing: after every inline-asm block of plain (non-sc1) global loads in a kernel, no
register copy or spill (v_mov / v_pk_mov / v_accvgpr_write / scratch_store) may read the loaded registers on any control
flow path before an `s_waitcnt vmcnt(0)` -- the signature of hipcc merging an old and a new value of an asm-defined
variable.  Guards the asm-prefetch idiom of gru_persist.hip (tests/test_host_api.py runs it).
usage: check_inflight.py listing.s [kernel-name-substring]"""
import re
import sys

text = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "gru_persist_"
bad = 0


def regs(l):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", l):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", l))
    return out


for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = [l.strip() for l in body.split("\n")]
    label_at = {l[:-1].split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\w+:", l)}
    blocks = []                                   # (index of #ASMEND, loaded registers)
    i = 0
    while i < len(lines):
        if "#ASMSTART" in lines[i]:
            j, dst, plain = i + 1, set(), False
            while "#ASMEND" not in lines[j]:
                if lines[j].startswith("global_load") and "sc1" not in lines[j]:
                    plain = True
                    dst |= regs(lines[j].split(",")[0])
                j += 1
            if plain:
                blocks.append((j, dst))
            i = j
        i += 1
    for end, dst in blocks:
        seen, stack = set(), [end + 1]
        while stack:                              # every path from the asm block to its first vmcnt(0)
            k = stack.pop()
            while k < len(lines) and k not in seen:
                seen.add(k)
                l = lines[k]
                if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                    break
                if l.startswith(("v_mov_b", "v_pk_mov", "v_accvgpr_write", "scratch_store")) and "," in l and \
                        regs(l.split(",", 1)[1]) & dst:
                    print("%s: line %d copies in-flight v%s: %s" % (name[:60], k + 1, sorted(regs(l) & dst), l))
                    bad += 1
                tgt = re.match(r"^s_(c?branch\w*)\s+(\.LBB\w+)", l)
                if tgt:
                    if tgt.group(2) in label_at:
                        stack.append(label_at[tgt.group(2)])
                    if tgt.group(1) == "branch":
                        break                     # unconditional: no fall-through
                k += 1
    print("%-90s %d asm load blocks checked" % (name[:90], len(blocks)))
sys.exit(1 if bad else 0)
