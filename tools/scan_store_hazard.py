#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --offload-device-only -S) for the store-data hazard of round 2: a store
with more than 64 bits of data (dwordx3 / dwordx4) followed within fewer than 2 wait states by a VALU instruction that
writes one of its data registers.  The compiler's hazard recognizer keeps its own instructions apart, but not inline-asm
moves (sor.hip: moved()); sor.hip therefore puts store_data_guard() behind every solver store.  Exit code 1 on any hit.

usage: scan_store_hazard.py file.s [file.s ...]"""
import re, sys
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m: return {int(m.group(1))}
    return set()
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    kern = "?"
    ins = []
    for ln in lines:
        t = ln.strip()
        if t.endswith(":") and not t.startswith(".") and not t.startswith(";"):
            kern = t[:-1]
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ins.append((kern, t.split(";")[0].strip()))
    hits = {}
    for i, (k, t) in enumerate(ins):
        m = re.match(r"(buffer|global|flat|scratch)_store_(dwordx3|dwordx4)\s+(.*)", t)
        if not m: continue
        ops = [o.strip() for o in m.group(3).split(",")]
        data = regs(ops[0]) if m.group(1) == "buffer" else regs(ops[1])
        ws = 0
        for j in range(i + 1, min(i + 4, len(ins))):
            k2, t2 = ins[j]
            if k2 != k: break
            if ws >= 2: break
            op = t2.split()[0]
            if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readfirstlane"):
                dst = regs(t2.split()[1].rstrip(","))
                if dst & data:
                    hits.setdefault(k, []).append((ws, t, t2))
                    break
            m2 = re.match(r"s_nop\s+(\d+)", t2)
            ws += (int(m2.group(1)) + 1) if m2 else 1
    total = globals().get("total", 0) + len(hits)
    globals()["total"] = total
    print(path, "kernels with a VALU write of store data within < 2 wait states:", len(hits))
    for k, v in hits.items():
        print("  ", k[:90], len(v), "e.g. wait states", v[0][0], "|", v[0][1][:60], "|", v[0][2][:50])


# second pass: a DPP instruction reading a VGPR that an INLINE-ASM move wrote fewer than 2 wait states earlier (the hazard
# recognizer looks for VALU writers; an inline-asm statement is not one to it)
def any_regs(tok):
    return regs(tok.strip().rstrip(","))
for path in sys.argv[1:]:
    ins, in_asm = [], False
    for ln in open(path).read().split("\n"):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ins.append((t.split(";")[0].strip(), in_asm))
    dpp_hits = 0
    for i, (t, a) in enumerate(ins):
        if not a or not t.startswith("v_"):
            continue
        dst = any_regs(t.split()[1])
        ws = 0
        for j in range(i + 1, min(i + 4, len(ins))):
            t2 = ins[j][0]
            if ws >= 2:
                break
            if "dpp" in t2:
                srcs = set()
                for tok in t2.split()[2:]:
                    srcs |= any_regs(tok)
                if srcs & dst:
                    dpp_hits += 1
            m2 = re.match(r"s_nop\s+(\d+)", t2)
            ws += (int(m2.group(1)) + 1) if m2 else 1
    print(path, "DPP reads of an inline-asm result within < 2 wait states:", dpp_hits)
    globals()["total"] = globals().get("total", 0) + dpp_hits

sys.exit(1 if globals().get('total', 0) else 0)
