#!/usr/bin/env python3
"""Static check of the hand-scheduled loads of k_conv_tile4 (run by occuseg_amd/build.py).

The kernel issues its buffer loads from inline asm and waits for them with inline-asm s_waitcnt vmcnt(N),
so hipcc does not know that a destination register is not valid yet.  This script compiles scn_conv.hip to
assembly, walks every k_conv_tile4 instantiation in layout order (the pass loop twice, to cover the back edge)
with a FIFO model of the outstanding asm loads, and fails if ANY instruction outside the asm blocks reads or
writes a register that an in-flight load is still going to write.
usage: python tools/check_asm_loads.py [--tuning] [file.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "occuseg_amd", "csrc", "scn_conv.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def regs(tok):
    """registers named by an operand like v12, v[12:15], a3 or a[8:11]: VGPR n -> n, AGPR n -> 1000 + n (loads may
    target accumulation registers too)"""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {int(m.group(2)) + (1000 if m.group(1) == "a" else 0)}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        base = 1000 if m.group(1) == "a" else 0
        return set(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    return set()


def sregs(tok):
    """scalar registers named by an operand like s12 or s[12:13]"""
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def operands(line):
    line = line.split(";")[0]
    parts = line.strip().split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    return parts[0], [t.strip() for t in re.split(r",\s*", parts[1])]


def check_kernel(name, lines):
    # locate the pass loop: the loop that contains asm buffer loads
    in_asm = False
    fifo = []      # list of (dest register set, line number)
    errors = []
    seq = []
    for no, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        seq.append((no, in_asm, s))
    # find loop extent: first and last asm load after the first label that is a loop header with asm loads
    headers = [i for i, (no, a, s) in enumerate(seq)]
    def run(seq_part):
        for no, is_asm, s in seq_part:
            op, ops = operands(s)
            if is_asm:
                if op.startswith("buffer_load"):
                    fifo.append((regs(ops[0]), no))
                elif op == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", s)
                    if m:
                        keep = int(m.group(1))
                        while len(fifo) > keep:
                            fifo.pop(0)
                continue
            touched = set()
            for t in ops:
                touched |= regs(t.split(" ")[0])
            for dst, lno in fifo:
                if touched & dst:
                    errors.append("%s: line %d `%s` touches v%s, target of the asm load at line %d still in flight"
                                  % (name, no + 1, s, sorted(touched & dst), lno + 1))
    # the pass loop: the loop (header label .. last branch back to it) that contains asm buffer loads
    loop = back = None
    for h, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):.*Loop Header", ln)
        if not m:
            continue
        label = m.group(1)
        br = [i for i, l2 in enumerate(lines) if i > h and re.search(r"s_c?branch\S*\s+" + re.escape(label) + r"\b", l2)]
        if not br:
            continue
        inside = lines[h:max(br) + 1]
        if any("#ASMSTART" in l2 for l2 in inside) and any("buffer_load" in l2 for l2 in inside):
            loop, back = h, max(br)
            break
    if loop is None:
        return ["%s: pass loop not found" % name]
    pre = [x for x in seq if x[0] < loop]
    body = [x for x in seq if loop <= x[0] <= back]
    post = [x for x in seq if x[0] > back]
    run(pre)
    run(body)
    run(body)
    run(post)
    if fifo:
        errors.append("%s: %d asm loads never waited for" % (name, len(fifo)))
    return errors


def check_kernel_cfg(name, lines):
    """Path-exact variant (k_conv_tile6: nested pass loops, forward branches around the MFMAs of an absent group): walks the
    control-flow graph of the emitted kernel from its entry, EVERY path, with the FIFO of outstanding vector-memory loads as
    part of the state (vmcnt counts loads and stores in issue order); a state is (instruction, FIFO) and is visited once."""
    code = []       # (line number, text, inside an asm block)
    label_at = {}
    in_asm = False
    for no, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        s = ln.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.L\w+):", s)
        if m:
            label_at[m.group(1)] = len(code)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        code.append((no, s, in_asm))
    errors = {}
    seen = set()
    work = [(0, (), frozenset())]
    steps = 0
    while work:
        pc, fifo, spend = work.pop()  # spend: scalar registers that an inline-asm s_load is still going to write
        while True:
            if pc >= len(code) or (pc, fifo, spend) in seen:
                break
            seen.add((pc, fifo, spend))
            steps += 1
            if steps > 4000000:
                return ["%s: state space too large" % name]
            no, s, is_asm = code[pc]
            op, ops = operands(s)
            is_load = re.match(r"(buffer|global|flat|scratch)_load", op) is not None
            is_store = re.match(r"(buffer|global|flat|scratch)_(store|atomic)", op) is not None
            if not (is_asm and (is_load or op == "s_waitcnt")):
                touched = set()
                for t in ops:
                    touched |= regs(t.split(" ")[0])
                for dst, lno in fifo:
                    if touched & dst:
                        errors.setdefault((no, lno), "%s: line %d `%s` touches v%s, target of the load at line %d still in flight"
                                          % (name, no + 1, s, sorted(touched & dst), (lno if lno >= 0 else -2 - lno) + 1))
            # scalar loads issued from inline asm (k_conv_tile4's pass-table entry: requested in one asm statement, picked up
            # ~430 instructions later in another): hipcc does not know the destination is not valid yet either.  Scalar loads
            # may return out of order: only lgkmcnt(0) makes them safe.
            if spend and not (is_asm and op.startswith("s_load")):
                st = set()
                for t in ops:
                    st |= sregs(t.split(" ")[0])
                if st & spend:
                    errors.setdefault((no, "s"), "%s: line %d `%s` touches s%s, target of an inline-asm scalar load still in flight"
                                      % (name, no + 1, s, sorted(st & spend)))
            if is_asm and op.startswith("s_load"):
                spend = spend | frozenset(sregs(ops[0]))
            if op == "s_waitcnt":
                if spend and re.search(r"lgkmcnt\(0\)", s):
                    spend = frozenset()
                m = re.search(r"vmcnt\((\d+)\)", s)
                if m and len(fifo) > int(m.group(1)):
                    fifo = fifo[len(fifo) - int(m.group(1)):]
            elif is_load:
                # (line number negated - 2 for hipcc's own loads, e.g. spill reloads: hipcc waits for those itself)
                fifo = fifo + ((frozenset(regs(ops[0])), no if is_asm else -2 - no),)
            elif is_store:
                fifo = fifo + ((frozenset(), -1),)  # (no destination: where it was issued does not matter -- states merge)
            while len(fifo) > 62 and (not fifo[0][0] or fifo[0][1] < -1):
                fifo = fifo[1:]  # (the hardware counter saturates: issue stalls until the oldest store -- or load of hipcc's own,
                                 #  whose wait hipcc places itself -- has completed)
            if len(fifo) > 62:
                return ["%s: more than 62 vector-memory operations outstanding at line %d" % (name, no + 1)]
            if op == "s_endpgm":
                if spend:
                    errors.setdefault(("end-s", pc), "%s: inline-asm scalar loads never waited for" % name)
                if any(d for d, _ in fifo):
                    errors.setdefault(("end", pc), "%s: %d loads never waited for" % (name, sum(1 for d, _ in fifo if d)))
                break
            if op == "s_branch":
                pc = label_at[ops[0]]
                continue
            if op.startswith("s_cbranch"):
                work.append((label_at[ops[0]], fifo, spend))
            pc += 1
    return list(errors.values())


def check_copies_under_mask(name, lines):
    """ASM = 1 instantiations of k_conv_tile4 (the pass loop is one asm block that clobbers v60..v255): hipcc moves every vector
    value that must survive the block into a low register, and round 6 found such a copy at the TOP of the block that joins a
    divergent region -- in front of the `s_or_b64 exec, exec, ...` that re-enables the masked lanes, so a whole wave kept stale
    register contents.  The kernel now keeps no per-lane value across the block; this check refuses any build in which register copies or
    spill code -- and nothing else -- stand between a block label and the EXEC restore of that block."""
    errors = []
    i = 0
    while i < len(lines):
        if re.match(r"^\.LBB\d+_\d+:", lines[i]):
            j = i + 1
            pending = []
            while j < len(lines):
                t = lines[j].split(";")[0].strip()
                if not t:
                    j += 1
                    continue
                if re.match(r"^\.L\w+:", t) or t.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_barrier")) or "#ASM" in lines[j]:
                    break
                if re.match(r"s_or_b64 exec, exec,", t):
                    for no, v in pending:
                        errors.append("%s: line %d `%s` runs before the EXEC restore of its block (line %d)" % (name, no + 1, v, j + 1))
                    break
                if t.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr", "scratch_load", "scratch_store")):
                    pending.append((j, t))  # (register copies / spill code: what the register allocator inserts at a block's top)
                elif t.startswith(("v_", "ds_", "buffer_", "global_", "flat_")) and not t.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                    break               # (real work of a divergent region: its values belong to the active lanes)
                j += 1
        i += 1
    return errors


def main():
    args = [a for a in sys.argv[1:] if a != "--tuning"]
    if args:
        text = open(args[0]).read()
    else:
        extra = ["-DSCN_TUNING"] if "--tuning" in sys.argv else []  # the tuning build holds more instantiations
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "conv.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-S", "--cuda-device-only", SRC, "-o", out],
                                  stderr=subprocess.DEVNULL)
            text = open(out).read()
    lines = text.split("\n")
    # (the timing-only instantiations of k_conv_tile6, DBG != 0, drop loads and waits on purpose: not checked)
    # (... and the timing-only ablations of k_conv_tile4's assembly loop, ASM >= 2, tuning library only)
    starts = [i for i, ln in enumerate(lines) if re.match(r"^_ZN3scn12k_conv_tile[456]\S*:", ln)
              and not re.match(r"^_ZN3scn12k_conv_tile6ILi\d+ELb[01]ELi[1-9]", ln)
              and not re.match(r"^_ZN3scn12k_conv_tile4ILi\d+ELin?\d+ELi\d+ELi\d+ELi\d+ELi[2-9]E", ln)]
    errors = []
    n = 0
    for st in starts:
        name = lines[st].split(":")[0]
        # (a kernel can hold several s_endpgm: it ends at its .Lfunc_end label; a bare snippet at its last line)
        end = next((i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines)) - 1
        errors += check_kernel_cfg(name, lines[st:end + 1])  # every path of the control-flow graph
        stream_k = re.search(r"k_conv_tile4ILi\d+ELin?\d+ELi\d+ELi1E", name) is not None  # (set-up branches: the linear walk
        # would see the loads of both arms of an if / else in flight at once; the path-exact walk above covers them)
        # (ASM = 1 instantiations: the pass loop is ONE asm block -- tools/gen_tile4_loop.py checks every path through it on its
        #  own instruction list -- and the path-exact walk above has covered it again as emitted)
        asm_loop = re.search(r"k_conv_tile4ILi\d+ELin?\d+ELi\d+ELi\d+ELi\d+ELi1E", name) is not None
        if asm_loop:
            errors += check_copies_under_mask(name, lines[st:end + 1])
        if "k_conv_tile6" not in name and not stream_k and not asm_loop and not re.search(r"k_conv_tile4ILi\d*[13579]ELi", name):
            errors += check_kernel(name, lines[st:end + 1])  # and the older linear walk of the loop (twice), where it applies
                                                             # (the "no tile update" timing builds have another loop shape)
        n += 1
    for e in errors[:40]:
        print("check_asm_loads:", e)
    print("check_asm_loads: %d kernels, %d violations" % (n, len(errors)))
    return 1 if errors or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
