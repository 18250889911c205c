#!/usr/bin/env python3
"""Gate for the hand-counted waits of hint_sub.hpp (run by `make -C hint_amd/csrc asm` and tests/test_asm_gate.py).

The subtree phase prefetches the next node's W2 tiles (and a2 sign bytes) with INLINE-ASM global loads the compiler's
wait-count bookkeeping does not see (`sub_load`, `sub_load_byte`) and waits for them with hand-placed
`s_waitcnt vmcnt(N)` (`sub_wait<N>`), N = the vector-memory operations known to be younger than the loads.  The
hardware has no interlock: an instruction that touches a destination register of such a load before the load has
retired - a spill store, a register copy the allocator inserts, a use that got scheduled in front of the wait -
reads or clobbers stale data, silently.  This script proves, on the generated device assembly, for every path
of the control-flow graph from every such load:

  * no instruction reads or writes a destination register of the load until a `s_waitcnt` whose `vmcnt(K)` operand
    guarantees the load has retired - on gfx9 vector-memory operations retire in order, so the load has retired once
    the counter is waited down to K <= (vector-memory instructions issued after it on that path);
  * the kernel does not end with such a load in flight.

Branches that test the same source-level condition are correlated through markers the source leaves as assembly
comments (`; hint-path <name>` / `; hint-path !<name>`, hint_sub.hpp `SUB_PATH`): a path that has passed `<name>` is
not followed through `!<name>` and vice versa (the two `if (train)` of a subtree node).

    python tools/check_untracked_loads.py hint_amd/lib/hint_fwd.s hint_amd/lib/hint_bwd.s [...]
exit code 0: every file clean; 1: violations (printed with file:line).
"""
import re
import sys

VMEM = re.compile(r"^(global_|buffer_|scratch_|flat_)")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
CAP = 8           # younger counts saturate here (no wait in the sources asks for more)


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
    return out


def parse(path):
    """-> list of functions: (name, [instr]) with instr = dict(line, op, text, inline, labels_before)"""
    funcs, cur, name = [], None, None
    inline = False
    pending_labels = []
    for ln, raw in enumerate(open(path), 1):
        s = raw.strip()
        if not s:
            continue
        if s.startswith(";;#ASMSTART"):
            inline = True
            continue
        if s.startswith(";;#ASMEND"):
            inline = False
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m and not inline:
            lab = m.group(1)
            if not lab.startswith(".L"):
                name = lab
                cur = []
                funcs.append((name, cur))
                pending_labels = []
            else:
                pending_labels.append(lab)
            continue
        if cur is None or s.startswith("."):
            continue
        if s.startswith(";"):
            mk = re.match(r"^;\s*hint-path\s+(!?)(\w+)", s)
            if mk and inline:
                cur.append(dict(line=ln, op="#marker", text=s, inline=True, labels=pending_labels, marker=(mk.group(2), mk.group(1) == "")))
                pending_labels = []
            continue
        body = s.split(";")[0].strip()
        if not body:
            continue
        op = body.split()[0]
        cur.append(dict(line=ln, op=op, text=body, inline=inline, labels=pending_labels, regs=frozenset(regs_of(body)),
                        vmem=bool(VMEM.match(op))))
        pending_labels = []
    return [(n, f) for n, f in funcs if n.startswith("_Z") or n.startswith("hint_")]


def check_function(fname, name, ins):
    label_at = {}
    for i, it in enumerate(ins):
        for lab in it["labels"]:
            label_at[lab] = i
    errors = []
    starts = [i for i, it in enumerate(ins) if it["inline"] and it.get("vmem") and "load" in it["op"]]
    if not starts:
        return 0, errors
    seen = set()
    # state: (index, pending = tuple of (start line, frozenset regs, younger)), facts = frozenset of (name, value)
    stack = []
    for i in starts:
        stack.append((i, (), frozenset()))
    n_paths = 0
    while stack:
        i, pending, facts = stack.pop()
        while True:
            if i >= len(ins):
                break
            key = (i, pending, facts)
            if key in seen:
                break
            seen.add(key)
            it = ins[i]
            op, text = it["op"], it["text"]
            if op == "#marker":
                nm, val = it["marker"]
                if (nm, not val) in facts:
                    break                      # infeasible: the other side of the same condition was taken before
                facts = facts | {(nm, val)}
                i += 1
                continue
            is_untracked = it["inline"] and it["vmem"] and "load" in op
            touched = it["regs"]
            if is_untracked:
                dst = regs_of(text.split(",")[0])
                src = regs_of(",".join(text.split(",")[1:]))
                for (ln, regs, y) in pending:
                    if regs & (dst | src):
                        errors.append(f"{fname}:{it['line']}: `{text}` touches v-registers of the untracked load at line {ln} still in flight ({name})")
                if any(ln == it["line"] for (ln, _, _) in pending):
                    errors.append(f"{fname}:{it['line']}: `{text}` is issued again while its previous issue is still in flight ({name})")
                    break
                pending = tuple((ln, regs, min(CAP, y + 1)) for (ln, regs, y) in pending) + ((it["line"], frozenset(dst), 0),)
                i += 1
                continue
            if not pending:
                break                          # every load of this path has retired
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", text)
                if m:
                    k = int(m.group(1))
                    pending = tuple(p for p in pending if p[2] < k)
                elif re.fullmatch(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)", text):      # raw immediate: decode vmcnt (gfx9: bits 3:0 and 15:14)
                    imm = int(text.split()[1], 0)
                    k = (imm & 0xF) | (((imm >> 14) & 0x3) << 4)
                    pending = tuple(p for p in pending if p[2] < k)
                i += 1
                continue
            for (ln, regs, y) in pending:
                hit = regs & touched
                if hit:
                    r = sorted(hit)[0]
                    errors.append(f"{fname}:{it['line']}: `{text}` touches {r[0]}{r[1]} of the untracked load at line {ln} "
                                  f"before a wait that covers it ({y} younger vector-memory operations so far; {name})")
                    pending = tuple(p for p in pending if p[0] != ln)
            if it["vmem"]:
                pending = tuple((ln, regs, min(CAP, y + 1)) for (ln, regs, y) in pending)
            if op == "s_endpgm":
                for (ln, regs, y) in pending:
                    errors.append(f"{fname}:{it['line']}: kernel ends with the untracked load of line {ln} in flight ({name})")
                break
            if op == "s_branch":
                tgt = text.split()[1]
                if tgt not in label_at:
                    break
                i = label_at[tgt]
                continue
            if op.startswith("s_cbranch"):
                tgt = text.split()[1]
                if tgt in label_at:
                    stack.append((label_at[tgt], pending, facts))
                    n_paths += 1
                i += 1
                continue
            if op in ("s_setpc_b64", "s_swappc_b64"):
                errors.append(f"{fname}:{it['line']}: indirect branch with an untracked load in flight ({name})")
                break
            i += 1
    return len(starts), sorted(set(errors))


def main(paths):
    bad = 0
    for p in paths:
        total = 0
        for name, ins in parse(p):
            n, errs = check_function(p, name, ins)
            total += n
            for e in errs:
                print(e)
            bad += len(errs)
        print(f"{p}: {total} untracked loads checked")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
