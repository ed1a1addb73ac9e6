#!/usr/bin/env python3
"""Scan gfx950 assembly (the device .s of a translation unit: `hipcc -save-temps=obj`, or `--offload-device-only -S`) for
data hazards the compiler does NOT guard in this code base.  Exit code 1 on any hit.  Run by csrc/Makefile on every
build of every translation unit, and by tests/test_codegen_hazards.py.

Why a scanner: gfx940+ needs software wait states between certain pairs of instructions.  The compiler's hazard
recognizer inserts them for pairs of ITS OWN instructions, with two holes that matter here:
  * an inline-asm statement is opaque to it: it neither sees what the statement writes nor what it reads -- sor.hip's
    software pipeline uses inline-asm `v_mov_b64` (moved()) and EXEC-narrowed LDS writes;
  * it skips the store-data hazard for buffer stores that carry their step offset in an SGPR soffset -- exactly the form
    of every solver store (sor.hip) -- so there even compiler-generated VALU writes are unguarded.  Round 2 found that
    hazard to be real on MI355X (wrong quads of lanes whenever a second wave shared the SIMD): every
    solver store is followed by store_data_guard() since, and pass A below checks ALL VALU writers behind ALL wide stores.

Pass A (every instruction): a store with more than 64 bits of data (dwordx3 / dwordx4) followed within fewer than 2 wait
        states by a VALU instruction that writes one of its data registers.
Pass B (pairs with at least one side inside ;;#ASMSTART ... ;;#ASMEND), the gfx940 rules of the ISA guide's "manually
        inserted wait states" table that VALU / SALU / memory code without MFMA can meet:
          VALU writes VGPR            -> DPP instruction reads it                      2 wait states
          VALU writes VGPR            -> v_readlane / v_readfirstlane reads it         1
          VALU writes VGPR            -> v_permlane*_swap reads it (gfx950)            2
          VALU writes SGPR / VCC      -> vector-memory instruction reads that SGPR     5
          VALU writes SGPR / VCC      -> v_readlane / v_writelane lane select          4
          VALU writes SGPR            -> VALU reads that SGPR                          2
          VALU writes VCC             -> v_div_fmas                                    4
          VALU writes EXEC (v_cmpx)   -> DPP instruction                               5
          VALU writes EXEC (v_cmpx)   -> v_readlane / v_readfirstlane / v_writelane    4
          trans op (exp/log/rcp/rsq/sqrt/sin/cos) -> non-trans VALU reads its result   1
        (store-data write-after-read is pass A.)  An instruction in between counts one wait state, `s_nop N` N + 1.

usage: scan_store_hazard.py file.s [file.s ...]"""
import re
import sys

REG = re.compile(r"\b([vs])(\d+)\b|\b([vs])\[(\d+):(\d+)\]|\b(vcc|exec)(?:_lo|_hi)?\b")
TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|legacy_f32|bf16)")
VMEM = re.compile(r"^(buffer|global|flat|scratch|tbuffer)_")
TWO_DST = re.compile(r"^v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|div_scale|mad_u64_u32|mad_i64_i32)")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            out |= {(m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1)}
        else:
            out.add((m.group(6), 0))
    return out


class Ins(object):
    __slots__ = ("text", "op", "dst", "src", "in_asm", "kern", "ws")

    def __init__(self, text, in_asm, kern):
        self.text, self.in_asm, self.kern = text, in_asm, kern
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        m = re.match(r"s_nop\s+(\d+)", text)
        self.ws = int(m.group(1)) + 1 if m else 1
        self.dst, self.src = set(), set()
        if self.op.startswith("v_") or self.op.startswith("s_") or self.op.startswith("ds_"):
            ndst = 0
            if self.op.startswith("v_"):
                ndst = 2 if TWO_DST.match(self.op) else 1
                if self.op.startswith("v_cmpx"):
                    self.dst.add(("exec", 0))
                    ndst = 1 if ops and not ops[0].startswith("v") and REG.match(ops[0]) else 0
                if self.op.startswith("v_nop"):
                    ndst = 0
            elif self.op.startswith("s_") and not re.match(r"s_(nop|waitcnt|sleep|endpgm|barrier|branch|cbranch|setprio|sethalt|"
                                                            r"cmp|bitcmp|setreg|sendmsg|trap|icache|dcache|clause|code_end)", self.op):
                ndst = 1
            for i, o in enumerate(ops):
                (self.dst if i < ndst else self.src).update(regs_of(o))
            if self.op.startswith("ds_"):  # ds_write*: no register results; ds_read*: first operand
                self.src |= self.dst
                self.dst = regs_of(ops[0]) if ops and self.op.startswith("ds_read") else set()
                self.src -= self.dst
        elif VMEM.match(self.op):
            for o in ops:
                self.src.update(regs_of(o))
            if "_load" in self.op or "_atomic" in self.op:  # results: the first operand
                self.dst = regs_of(ops[0]) if ops else set()


def parse(path):
    ins, in_asm, kern = [], False, "?"
    for ln in open(path).read().split("\n"):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.endswith(":") and not t.startswith(".") and not t.startswith(";"):
            kern = t[:-1]
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if t:
            ins.append(Ins(t, in_asm, kern))
    return ins


def is_valu(i):
    return i.op.startswith("v_") and not i.op.startswith("v_nop")


def is_dpp(i):
    return "dpp" in i.text and i.op.startswith("v_")


def lane_op(i):
    return i.op.startswith(("v_readlane", "v_readfirstlane", "v_writelane"))


def hazard(p, c):
    """wait states required between producer p and consumer c (0: none), and the rule's name"""
    need, why = 0, ""

    def rule(n, name):
        nonlocal need, why
        if n > need:
            need, why = n, name
    if not is_valu(p):
        return 0, ""
    vdst = {r for r in p.dst if r[0] == "v"}
    sdst = {r for r in p.dst if r[0] in ("s", "vcc")}
    if vdst & c.src:
        if is_dpp(c):
            rule(2, "VALU write -> DPP read")
        if c.op.startswith(("v_readlane", "v_readfirstlane")):
            rule(1, "VALU write -> v_readlane read")
        if c.op.startswith("v_permlane") and "swap" in c.op:
            rule(2, "VALU write -> v_permlane swap read")
        if TRANS.match(p.op) and is_valu(c) and not TRANS.match(c.op):
            rule(1, "trans result -> non-trans VALU")
    if sdst & c.src:
        if VMEM.match(c.op):
            rule(5, "VALU writes SGPR -> VMEM reads it")
        if c.op.startswith(("v_readlane", "v_writelane")):
            rule(4, "VALU writes SGPR -> lane select")
        if is_valu(c):
            rule(2, "VALU writes SGPR -> VALU reads it")
        if c.op.startswith("v_div_fmas") and ("vcc", 0) in sdst:
            rule(4, "VALU writes VCC -> v_div_fmas")
    if ("vcc", 0) in sdst and c.op.startswith("v_div_fmas"):
        rule(4, "VALU writes VCC -> v_div_fmas")
    if ("exec", 0) in p.dst:
        if is_dpp(c):
            rule(5, "VALU writes EXEC -> DPP")
        if lane_op(c):
            rule(4, "VALU writes EXEC -> lane op")
    return need, why


def scan(path):
    ins = parse(path)
    total = 0
    # ---- pass A: wide store -> VALU write of its data registers within < 2 wait states (all writers)
    hits = {}
    for i, a in enumerate(ins):
        m = re.match(r"(buffer|global|flat|scratch)_store_(dwordx3|dwordx4)\s+(.*)", a.text)
        if not m:
            continue
        ops = [o.strip() for o in m.group(3).split(",")]
        data = {r for r in regs_of(ops[0] if m.group(1) == "buffer" else ops[1]) if r[0] == "v"}
        ws = 0
        for b in ins[i + 1:i + 4]:
            if b.kern != a.kern or ws >= 2:
                break
            if is_valu(b) and not b.op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")) and (b.dst & data):
                hits.setdefault(a.kern, []).append((ws, a.text, b.text))
                break
            ws += b.ws
    print(path, "kernels with a VALU write of store data within < 2 wait states:", len(hits))
    for k, v in hits.items():
        print("  ", k[:90], len(v), "e.g. wait states", v[0][0], "|", v[0][1][:60], "|", v[0][2][:50])
    total += len(hits)
    # ---- pass B: producer / consumer pairs with an inline-asm side
    pairs = {}
    for i, p in enumerate(ins):
        if not is_valu(p):
            continue
        ws = 0
        for c in ins[i + 1:i + 7]:
            if c.kern != p.kern or ws >= 5:
                break
            if p.in_asm or c.in_asm:
                need, why = hazard(p, c)
                if need > ws:
                    pairs.setdefault(why, []).append((p.kern, ws, p.text, c.text))
            ws += c.ws
    n_pairs = sum(len(v) for v in pairs.values())
    print(path, "unguarded hazard pairs with an inline-asm side:", n_pairs)
    for why, v in pairs.items():
        print("  ", why, len(v), "e.g. after", v[0][1], "wait states |", v[0][2][:50], "|", v[0][3][:60], "|", v[0][0][:60])
    return total + n_pairs


if __name__ == "__main__":
    bad = sum(scan(p) for p in sys.argv[1:])
    sys.exit(1 if bad else 0)
