#!/usr/bin/env python3
"""Static check of the 4-wave kernel's assembly (qattn_attn_w4.hip): the MFMAs there are inline asm, so the compiler neither pads the
wait states behind them nor knows that their results land late.  Walks every w4 kernel in textual order and reports any instruction
that reads or writes a register of an MFMA result (VGPR or AccVGPR) fewer wait states behind that MFMA than the hardware needs
(19 for the 16-pass 32x32x64, 11 for the 8-pass 16x16x128; one wait state per instruction, N + 1 per `s_nop N`, and a later MFMA holds
the wave until the matrix pipe takes it: 16 / 8).  The accumulate chain of an MFMA on its own result (dst = C of the same registers) is
exempt, and so are A / B operands.  Branches: a forward branch carries the state to its label (merged), the fall-through continues; a
backward branch is followed for `window` instructions from its target.

  python tools/asm_hazards_w4.py <file.s>   -> one line per finding, exit status 1 if any"""
import re
import sys

PASSES = {"32x32x64": 16, "16x16x128": 8}
NEED = {"32x32x64": 19, "16x16x128": 11}
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def kernels(path):
    out, cur = {}, None
    for no, raw in enumerate(open(path, errors="replace"), 1):
        s = raw.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
        if cur is None or not s or s.startswith(";"):
            continue
        if s.startswith(".LBB"):
            cur.append((no, "label", s.split(":")[0]))
        elif not s.startswith("."):
            body = s.split(";")[0].strip()
            if body:
                cur.append((no, "inst", body))
    return out


def step(state, clock, text, no, findings, name):
    """state: reg -> ready clock.  Returns the new clock."""
    op = text.split()[0]
    if op == "s_nop":
        return clock + int(text.split()[1], 0) + 1
    operands = text[len(op):]
    m = re.match(r"v_mfma\w*?_(\d+x\d+x\d+)", op)
    if m:
        shape = m.group(1)
        parts = [x.strip() for x in operands.split(",")]
        dst, a, b, c = parts[0], parts[1], parts[2], parts[3].split()[0]
        # srcC of in-flight results: only the chain on the same registers is free
        for r in regs(c):
            if r in state and state[r] > clock and regs(c) != regs(dst):
                findings.append(f"{name}: line {no}: MFMA reads C {c} {state[r] - clock} wait states early: {text}")
                break
        for r in regs(a) | regs(b):
            if r in state and state[r] > clock:
                findings.append(f"{name}: line {no}: MFMA reads A/B of an MFMA result in flight: {text}")
                break
        clock += PASSES.get(shape, 16)   # the wave waits for the pipe (in order)
        for r in regs(dst):
            state[r] = clock - PASSES.get(shape, 16) + NEED.get(shape, 19) + PASSES.get(shape, 16)   # conservative: issue may have waited a whole product
        return clock
    touched = regs(operands)
    for r in touched:
        if r in state and state[r] > clock:
            findings.append(f"{name}: line {no}: {state[r] - clock} wait states short on {r[0]}{r[1]}: {text}")
            break
    return clock + 1


def check(path, only="attn_fwd_kernel_w4", window=400):
    findings = []
    for name, insts in kernels(path).items():
        if only not in name:
            continue
        labels = {t: i for i, (_, k, t) in enumerate(insts) if k == "label"}
        pending = {}   # label -> (state, clock) carried by forward branches
        state, clock = {}, 0
        for i, (no, kind, text) in enumerate(insts):
            if kind == "label":
                if text in pending:
                    st2, ck2 = pending.pop(text)
                    for r, v in st2.items():   # merge: the later ready time relative to each clock
                        state[r] = max(state.get(r, 0) - clock, v - ck2) + clock
                continue
            clock = step(state, clock, text, no, findings, name)
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", text)
            if m and m.group(1) in labels:
                tgt = labels[m.group(1)]
                if tgt > i:
                    if m.group(1) in pending:
                        st2, ck2 = pending[m.group(1)]
                        merged = {r: max(st2.get(r, 0) - ck2, state.get(r, 0) - clock) + clock for r in set(st2) | set(state)}
                        pending[m.group(1)] = (merged, clock)
                    else:
                        pending[m.group(1)] = (dict(state), clock)
                else:   # back edge: replay the head of the loop with the state at the branch
                    st2, ck2 = dict(state), clock
                    n = 0
                    for no2, kind2, text2 in insts[tgt:]:
                        if kind2 == "label":
                            continue
                        n += 1
                        if n > window:
                            break
                        ck2 = step(st2, ck2, text2, no2, findings, name + " (back edge)")
    return sorted(set(findings))


if __name__ == "__main__":
    bad = []
    for f in sys.argv[1:]:
        bad += check(f)
    print("\n".join(bad) if bad else "no findings")
    sys.exit(1 if bad else 0)
