#!/usr/bin/env python3
"""Static check of the compiler's -save-temps assembly: inline asm that reads a VGPR too soon after an MFMA wrote it.

hipcc's hazard recognizer pads the MFMA -> VALU-read wait states (19 for the 16-pass 32x32x64 f8f6f4, 11 for the 8-pass
16x16x128) in front of its OWN instructions only; the operands of an asm statement (`;;#ASMSTART ... ;;#ASMEND`) get none, and
gfx950 does not interlock this dependency: the asm reads the old register value.  This walks every kernel in textual order
(forward branches are ignored = the shortest path is the fall-through; a backward branch is followed for `window` instructions
from its target with the state at the branch) and reports asm instructions that read a register whose MFMA is fewer wait
states back than it needs, counting one wait state per instruction plus N + 1 per `s_nop N`.  A later MFMA of the same or
larger pass count issued in between also proves the older one has drained (the matrix pipe is in order), minus its own issue:
conservative, we do not use it.

  python tools/asm_hazards.py <file.s> [...]   -> one line per finding, exit status 1 if any
"""
import re
import sys

NEED = {"32x32x64": 19, "16x16x128": 11, "32x32x16": 19, "16x16x32": 11, "32x32x8": 19, "16x16x16": 11}
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path):
    """-> {kernel: [(line_no, kind, text)]}, kind in 'inst', 'asm', 'label'."""
    kernels, cur, in_asm = {}, None, False
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            line = raw.split(";;#")[0] if not raw.lstrip().startswith(";;#") else raw
            s = raw.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if re.match(r"^_Z\w+:", s) or re.match(r"^\w+:\s*(;.*)?$", s) and not s.startswith("."):
                cur = kernels.setdefault(s.split(":")[0], [])
                continue
            if cur is None or not s or s.startswith(";") or s.startswith(".") and not s.startswith(".LBB"):
                if s.startswith(".Lfunc_end"):
                    cur = None
                continue
            if s.startswith(".LBB"):
                cur.append((no, "label", s.split(":")[0]))
                continue
            body = s.split(";")[0].strip()
            if body:
                cur.append((no, "asm" if in_asm else "inst", body))
    return kernels


def walk(insts, start, state, clock, findings, name, limit=None):
    """state: reg -> (ready_clock, mfma line)."""
    n = 0
    for i in range(start, len(insts)):
        no, kind, text = insts[i]
        if kind == "label":
            continue
        if limit is not None:
            n += 1
            if n > limit:
                return
        op = text.split()[0]
        if op == "s_nop":
            clock += int(text.split()[1], 0) + 1
            continue
        operands = text[len(op):]
        parts = operands.split(",")
        if kind == "asm" and op.startswith("v_") and not op.startswith("v_mfma"):
            for r in regs(",".join(parts[1:])):
                if r in state and state[r][0] > clock:
                    findings.append((name, no, text, r, state[r][1], state[r][0] - clock))
        if op.startswith("v_mfma"):
            shape = next((k for k in NEED if k in op), None)
            need = NEED.get(shape, 19)
            for r in regs(parts[0]):
                state[r] = (clock + 1 + need, no)
        elif op.startswith("v_") or op.startswith("ds_read") or op.startswith("global_load") or op.startswith("buffer_load"):
            for r in regs(parts[0]):   # overwritten by something else: the MFMA result is gone
                state.pop(r, None)
        clock += 1
        if limit is None and op.startswith("s_cbranch") or op == "s_branch":
            target = text.split()[-1]
            idx = next((j for j, x in enumerate(insts) if x[1] == "label" and x[2] == target), None)
            if limit is None and idx is not None and idx < i:
                walk(insts, idx, dict(state), clock, findings, name, limit=60)
    return


def check(path):
    findings = []
    for name, insts in parse(path).items():
        walk(insts, 0, {}, 0, findings, name)
    seen, out = set(), []
    for f in findings:
        if (f[0], f[1]) not in seen:
            seen.add((f[0], f[1]))
            out.append(f)
    return out


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for name, no, text, r, mf, short in check(p):
            bad += 1
            print(f"{p}:{no}: {name[:60]}: asm `{text}` reads v{r} {short} wait states too early after the MFMA at line {mf}")
    sys.exit(1 if bad else 0)
