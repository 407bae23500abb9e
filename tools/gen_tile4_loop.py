#!/usr/bin/env python3
"""Generator of k_conv_tile4's hand-written pass loop (gfx950 assembly) -- writes occuseg_amd/csrc/tile4_loop_asm.inc.

hipcc's version of the loop (scn_conv.hip, `body`) executes ~26 vector-ALU instructions per pass of 64 MFMAs: 8 packed adds,
4 address mads, the weight-set hand-over at a phase change (19 v_mov_b64), 4 unconditional v_mov_b64 that reconcile register
assignments at the joins, 2 v_min -- and beside v_mfma_f32_16x16x4_f32 every vector-ALU instruction costs 8-12 cycles of
matrix-pipe time.  The loop written here executes 12: the 8 packed adds of the output-tile update and
the 4 address mads.
  * The pass body exists six times: three rotations of the row register sets x two roles of the weight register sets.  A pass
    ends with a scalar branch to the copy for (next rotation, role swapped if the next pass starts a phase): nothing is copied.
  * Lanes past the end of a pair list are masked out of the index and row LOADS by an EXEC mask taken from the pass table (no
    v_min clamp, no fetch for them at all) and out of the output-tile write as before.
  * Same pipeline as the C++ loop: rows two passes ahead (three sets), weights one pass ahead and only at a phase change,
    indices three passes ahead, table entry four ahead (scalar load); the same loads in the same order every pass (8 weight
    when the next pass starts a phase, 2 index, 8 row), so every s_waitcnt is a constant.
  * Same arithmetic in the same order: bit-identical to the C++ loop and to k_conv_tile2 (tests/test_gpu_ops.py).
The generator simulates every path through the six copies with a FIFO model of the outstanding loads and fails if any
instruction touches a register with a load in flight, if an MFMA result is read by a non-MFMA instruction less than 10 issue
slots after the MFMA, or if an LDS / scalar-memory result is used before its wait (the checks tools/check_asm_loads.py makes on
hipcc's output, here on the instruction list itself).

Pass table entry (16 bytes, written by the kernel's set-up): x = byte offset of the pass's first pair in `rules`,
y = pairs left in the list (8 bits) | chunk << 8 | offset k << 12 | first pass of its (offset, chunk) phase << 20,
z = byte offset of the phase's weight block [sp = 0][q = 0] of wave 0 in the packed weights, w = lane masks of the pass's two
16-pair groups (low / high half).  Eight zero entries follow the last pass.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "occuseg_amd", "csrc", "tile4_loop_asm.inc")
OUT_TUNING = os.path.join(ROOT, "occuseg_amd", "csrc", "tile4_loop_asm_tuning.inc")  # the timing-only ablations (-DSCN_TUNING)

DEBUG_SKIP = False
VB = 60   # first vector register owned by the loop (v60 .. v255)
SB = 52   # first scalar register owned by the loop (s52 .. s95)


# ---------------------------------------------------------------- registers
def vr(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def sr(base, n=1):
    return "s%d" % base if n == 1 else "s[%d:%d]" % (base, base + n - 1)


def RS(s, g, c):      # rows: set s, group g, 16-byte piece c  (4 registers)
    return VB + ((s * 2 + g) * 4 + c) * 4


def WS(w, q, sp):     # weights: role w, 16-plane block q, 16-k slice sp (4 registers)
    return VB + 96 + ((w * 2 + q) * 4 + sp) * 4


def ACC(g, q):
    return VB + 160 + (g * 2 + q) * 4


def OLD(q):
    return VB + 176 + q * 4


def PR(g):            # pair (x, y) of the lane, group g
    return VB + 184 + g * 2


def OFF(g):
    return VB + 188 + g


def OAD(s, g):        # output-tile byte address of the lane's pair: pass set s, group g
    return VB + 190 + s * 2 + g


V_END = VB + 196
assert V_END == 256

E = [SB + 4 * i for i in range(5)]   # table entries p .. p+4 (4 registers each)
TOFF = SB + 20
MP1 = SB + 21
WSO = [SB + 22 + i for i in range(4)]
RSO = SB + 26
T = [SB + 27 + i for i in range(6)]
M0 = SB + 33
S_END = SB + 34


class Ins(object):
    __slots__ = ("text", "kind", "vdef", "vuse", "sdef", "suse", "n", "label", "target")

    def __init__(self, text, kind, vdef=(), vuse=(), sdef=(), suse=(), n=0, label=None, target=None):
        self.text, self.kind = text, kind
        self.vdef, self.vuse, self.sdef, self.suse = set(vdef), set(vuse), set(sdef), set(suse)
        self.n, self.label, self.target = n, label, target


def rng(b, n):
    return range(b, b + n)


# timing-only ablations (tuning library, tools/bench_conv.py --asm): what each part of a pass costs.  Wrong results by design.
ABLATIONS = {2: ("nowait",), 3: ("notile",), 4: ("norows",), 5: ("norows", "noidx", "now", "nowait"),
             6: ("norows", "noidx", "now", "nowait", "notile"),
             9: ("norows", "noidx", "now", "nowait", "notile", "nopieces")}


class Gen(object):
    def __init__(self, incol, flags=()):
        self.incol = incol
        self.flags = set(flags)
        self.ins = []

    def emit(self, *a, **k):
        self.ins.append(Ins(*a, **k))

    # ---- scalar helpers (operands: int = scalar register, str = literal / input operand)
    def s(self, op, dst, *src):
        txt = "%s %s" % (op, ", ".join([sr(dst)] + [sr(x) if isinstance(x, int) else x for x in src]))
        self.emit(txt, "salu", sdef=[dst], suse=[x for x in src if isinstance(x, int)])

    def s_raw(self, txt, sdef=(), suse=()):
        self.emit(txt, "salu", sdef=sdef, suse=suse)

    def label(self, name):
        self.emit(name + ":", "label", label=name)

    def branch(self, op, target):
        self.emit("%s %s" % (op, target), "branch", target=target)

    def exec_from(self, m16, tmp):
        """EXEC = the 16-bit lane mask in scalar m16 (an expression already in a register), replicated to all four quarters"""
        self.s("s_mul_i32", tmp, m16, "0x10001")
        self.s_raw("s_mov_b32 exec_lo, %s" % sr(tmp), suse=[tmp])
        self.s_raw("s_mov_b32 exec_hi, %s" % sr(tmp), suse=[tmp])

    def exec_all(self):
        self.s_raw("s_mov_b64 exec, -1")

    def masks_of(self, e, g, tmp):
        """tmp = 16-bit mask of group g of entry e"""
        if g == 0:
            self.s("s_and_b32", tmp, e + 3, "0xffff")
        else:
            self.s("s_lshr_b32", tmp, e + 3, "16")

    # ---- vector / memory
    def mfma(self, acc, a, b, zero):
        if "nomfma" in self.flags:
            return
        c = "0" if zero else vr(acc, 4)
        self.emit("v_mfma_f32_16x16x4_f32 %s, %s, %s, %s" % (vr(acc, 4), vr(a), vr(b), c), "mfma",
                  vdef=rng(acc, 4), vuse=[a, b] + ([] if zero else list(rng(acc, 4))))

    def mads(self, g, oad):
        src, dst = (PR(g) + 1, PR(g)) if self.incol else (PR(g), PR(g) + 1)
        self.emit("v_mad_u32_u24 %s, %s, %%[row_bytes], %%[lq16]" % (vr(OFF(g)), vr(src)), "valu", vdef=[OFF(g)], vuse=[src])
        self.emit("v_mad_u32_u24 %s, %s, %%[os_u], %%[tile0]" % (vr(oad), vr(dst)), "valu", vdef=[oad], vuse=[dst])

    def load_idx(self, g, e):
        if "noidx" in self.flags:
            return
        self.emit("buffer_load_dwordx2 %s, %%[li8_%d], %%[ru], %s offen" % (vr(PR(g), 2), g, sr(e)), "vmem",
                  vdef=rng(PR(g), 2), suse=[e], n=2)

    def load_row(self, dst, g, c):
        if "norows" in self.flags:
            return
        off = "" if c == 0 else " offset:%d" % (64 * c)
        self.emit("buffer_load_dwordx4 %s, %s, %%[in], %s offen%s" % (vr(dst, 4), vr(OFF(g)), sr(RSO), off), "vmem",
                  vdef=rng(dst, 4), vuse=[OFF(g)], suse=[RSO], n=4)

    def load_w(self, dst, sp, q):
        if "now" in self.flags:
            return
        off = "" if q == 0 else " offset:1024"
        self.emit("buffer_load_dwordx4 %s, %%[wlane16], %%[wp], %s offen%s" % (vr(dst, 4), sr(WSO[sp]), off), "vmem",
                  vdef=rng(dst, 4), suse=[WSO[sp]], n=4)

    def tile_read(self, addr):
        if "notile" in self.flags:
            return
        for q in range(2):
            off = "" if q == 0 else " offset:64"
            self.emit("ds_read_b128 %s, %s%s" % (vr(OLD(q), 4), vr(addr), off), "ds_read", vdef=rng(OLD(q), 4), vuse=[addr])

    def tile_add_write(self, g, addr, m16, tmp):
        """OLD += ACC[g]; masked write back"""
        if "notile" in self.flags:
            return
        for q in range(2):
            for h in range(2):
                o, a = OLD(q) + 2 * h, ACC(g, q) + 2 * h
                self.emit("v_pk_add_f32 %s, %s, %s" % (vr(o, 2), vr(o, 2), vr(a, 2)), "valu", vdef=rng(o, 2),
                          vuse=list(rng(o, 2)) + list(rng(a, 2)))
        self.exec_from(m16, tmp)   # (three scalar instructions between the adds and the stores that read their results)
        for q in range(2):
            off = "" if q == 0 else " offset:64"
            self.emit("ds_write_b128 %s, %s%s" % (vr(addr), vr(OLD(q), 4), off), "ds_write", vuse=[addr] + list(rng(OLD(q), 4)))
        self.exec_all()

    def waitcnt(self, vm=None, lgkm=None):
        if "nowait" in self.flags and vm == 8 and lgkm is None:
            return
        parts = []
        if vm is not None:
            parts.append("vmcnt(%d)" % vm)
        if lgkm is not None:
            parts.append("lgkmcnt(%d)" % lgkm)
        self.emit("s_waitcnt " + " ".join(parts), "wait", n=(vm, lgkm))

    def nop(self, n):
        self.emit("s_nop %d" % n, "nop", n=n + 1)

    # ---- pieces of a pass
    def weight_scalars(self, e):
        self.s("s_add_u32", WSO[0], e + 2, "%[wbase]")
        for sp in range(1, 4):
            self.s("s_add_u32", WSO[sp], WSO[sp - 1], "%[wstep]")

    def row_soffset(self, e):
        self.s("s_bfe_u32", RSO, e + 1, "0x40008")
        self.s("s_lshl_b32", RSO, RSO, "8")

    def rows_masked(self, rset, e, g, cs):
        self.masks_of(e, g, T[0])
        self.exec_from(T[0], T[1])
        for c in cs:
            self.load_row(RS(rset, g, c), g, c)
        self.exec_all()

    def idx_masked(self, e):
        for g in range(2):
            self.masks_of(e, g, T[0])
            self.exec_from(T[0], T[1])
            self.load_idx(g, e)
        self.exec_all()

    def rotate_entries(self):
        for i in range(4):
            for h in (0, 2):
                self.s_raw("s_mov_b64 %s, %s" % (sr(E[i] + h, 2), sr(E[i + 1] + h, 2)), sdef=[E[i] + h, E[i] + h + 1],
                           suse=[E[i + 1] + h, E[i + 1] + h + 1])

    def prologue(self, L):
        if DEBUG_SKIP:  # (debugging aid, tuning builds: leave the loop at once when the operand says so)
            self.s_raw("s_mov_b32 %s, %%[skip]" % sr(T[0]), sdef=[T[0]])
            self.s_raw("s_cmp_lg_u32 %s, 0" % sr(T[0]), suse=[T[0]])
            self.branch("s_cbranch_scc1", L("END"))
        for i in range(4):
            self.emit("s_load_dwordx4 %s, %%[gt], %s" % (sr(E[i], 4), "0x%x" % (16 * i)), "smem", sdef=rng(E[i], 4))
        self.s_raw("s_movk_i32 %s, 64" % sr(TOFF), sdef=[TOFF])
        self.s_raw("s_mov_b32 %s, 0" % sr(MP1), sdef=[MP1])
        # lanes that no masked load ever reaches keep what these registers hold: a valid pair (0, 0) and a valid, aligned tile
        # address, not whatever the registers held before the loop (their rows are never loaded, their tile rows never written)
        for g in range(2):
            for h in range(2):
                self.emit("v_mov_b32 %s, 0" % vr(PR(g) + h), "valu", vdef=[PR(g) + h])
        self.emit("v_mov_b32 %s, %%[tile0]" % vr(OAD(2, 1)), "valu", vdef=[OAD(2, 1)])
        self.waitcnt(lgkm=0)
        self.s("s_and_b32", T[0], E[0] + 1, "0xff")
        self.branch("s_cbranch_scc0", L("END"))
        # pass 0: indices, then weights + rows
        self.idx_masked(E[0])
        self.waitcnt(vm=0)
        for g in range(2):
            self.mads(g, OAD(0, g))
        self.weight_scalars(E[0])
        for sp in range(4):
            for q in range(2):
                self.load_w(WS(0, q, sp), sp, q)
        self.row_soffset(E[0])
        for g in range(2):
            self.rows_masked(0, E[0], g, range(4))
        # pass 1: indices, rows
        self.idx_masked(E[1])
        self.waitcnt(vm=0)
        for g in range(2):
            self.mads(g, OAD(1, g))
        self.row_soffset(E[1])
        for g in range(2):
            self.rows_masked(1, E[1], g, range(4))
        # indices of pass 2
        self.idx_masked(E[2])
        self.waitcnt(vm=0)

    def body(self, r, w, L):
        rc, rn = r, (r + 2) % 3
        oc, on = r, (r + 2) % 3   # `on` still holds the previous pass' addresses (group 1 pending until piece 6)
        self.label(L("B%d%d" % (r, w)))

        def piece(m):
            if "nopieces" in self.flags:   # (MFMAs + loop control only; the table entry still has to come in)
                if m == 8:
                    self.emit("s_load_dwordx4 %s, %%[gt], %s" % (sr(E[4], 4), sr(TOFF)), "smem", sdef=rng(E[4], 4), suse=[TOFF])
                    self.s("s_add_u32", TOFF, TOFF, "16")
                elif m == 15:
                    self.waitcnt(lgkm=0)
                return
            if m == 0:      # pending group 1 of the previous pass: read its tile rows; addresses of (p+2, group 0)
                self.tile_read(OAD(on, 1))
                self.mads(0, OAD(on, 0))
            elif m == 1:    # weights of the next pass only when it starts a phase
                self.s_raw("s_bitcmp1_b32 %s, 20" % sr(E[1] + 1), suse=[E[1] + 1])
                self.branch("s_cbranch_scc0", L("B%d%d_w1" % (r, w)))
                self.weight_scalars(E[1])
                self.label(L("B%d%d_w1" % (r, w)))
            elif 2 <= m <= 5:
                sp = m - 2
                self.s_raw("s_bitcmp1_b32 %s, 20" % sr(E[1] + 1), suse=[E[1] + 1])
                self.branch("s_cbranch_scc0", L("B%d%d_w%d" % (r, w, m)))
                for q in range(2):
                    self.load_w(WS(1 - w, q, sp), sp, q)
                self.label(L("B%d%d_w%d" % (r, w, m)))
            elif m == 6:    # ... add, write (only where there was a second group)
                self.waitcnt(lgkm=0)
                self.s_raw("s_cmp_eq_u32 %s, 0" % sr(MP1), suse=[MP1])
                self.branch("s_cbranch_scc1", L("B%d%d_p" % (r, w)))
                self.tile_add_write(1, OAD(on, 1), MP1, T[1])
                self.label(L("B%d%d_p" % (r, w)))
            elif m == 7:    # addresses of (p+2, group 1) -- its slot is free now
                self.mads(1, OAD(on, 1))
                self.row_soffset(E[2])
            elif m == 8:    # indices of pass p+3 (the pair registers were consumed by pieces 0 and 7); table entry p+4
                self.idx_masked(E[3])
                self.emit("s_load_dwordx4 %s, %%[gt], %s" % (sr(E[4], 4), sr(TOFF)), "smem", sdef=rng(E[4], 4), suse=[TOFF])
                self.s("s_add_u32", TOFF, TOFF, "16")
            elif 9 <= m <= 12:  # rows of pass p+2
                g, half = (m - 9) >> 1, (m - 9) & 1
                self.rows_masked(rn, E[2], g, (2, 3) if half else (0, 1))
            elif m == 14:   # group 0 of this pass: read its tile rows
                self.tile_read(OAD(oc, 0))
            elif m == 15:
                self.waitcnt(lgkm=0)

        def group(g, hook=None):
            for i in range(32):
                sp, h = i // 8, i % 8
                q, j = h & 1, h >> 1
                self.mfma(ACC(g, q), WS(w, q, sp) + j, RS(rc, g, sp) + j, sp == 0 and h < 2)
                if hook:
                    hook(i)

        group(0, lambda i: piece(i) if i < 16 else None)
        # lane masks of this pass' groups (before the entries rotate)
        self.s("s_and_b32", M0, E[0] + 3, "0xffff")
        self.s("s_and_b32", T[2], E[0] + 1, "0xff")
        self.s_raw("s_cmp_gt_u32 %s, 16" % sr(T[2]), suse=[T[2]])
        self.branch("s_cbranch_scc0", L("B%d%d_one" % (r, w)))

        def tail(single):
            self.waitcnt(vm=8)
            self.s("s_lshr_b32", MP1, E[0] + 3, "16")
            self.rotate_entries()
            if single:  # group 0's result: >= 10 issue slots behind its last MFMA (the wait, the shift and eight moves)
                self.tile_add_write(0, OAD(oc, 0), M0, T[1])
            self.s("s_and_b32", T[2], E[0] + 1, "0xff")
            self.branch("s_cbranch_scc0", L("X%d" % ((r + 1) % 3)))
            self.s_raw("s_bitcmp1_b32 %s, 20" % sr(E[0] + 1), suse=[E[0] + 1])
            self.branch("s_cbranch_scc1", L("B%d%d" % ((r + 1) % 3, 1 - w)))
            self.branch("s_branch", L("B%d%d" % ((r + 1) % 3, w)))

        # two groups: group 0 goes to the tile behind the first eight MFMAs of group 1
        def hook1(i):
            if i == 7:   # (12 issue slots behind group 0's last MFMA)
                self.tile_add_write(0, OAD(oc, 0), M0, T[1])
        group(1, hook1)
        tail(False)
        self.label(L("B%d%d_one" % (r, w)))
        tail(True)

    def exit_stub(self, rnext, L):
        prev = (rnext + 2) % 3
        self.label(L("X%d" % rnext))
        self.tile_read(OAD(prev, 1))
        self.waitcnt(vm=0, lgkm=0)
        self.s_raw("s_cmp_eq_u32 %s, 0" % sr(MP1), suse=[MP1])
        self.branch("s_cbranch_scc1", L("END"))
        self.nop(3)
        self.tile_add_write(1, OAD(prev, 1), MP1, T[1])
        self.branch("s_branch", L("END"))

    def build(self):
        L = lambda name: ".LT4A%d_%%=_%s" % (self.incol, name)  # noqa: E731
        self.prologue(L)
        self.branch("s_branch", L("B00"))
        for w in range(2):
            for r in range(3):
                self.body(r, w, L)
        for r in range(3):
            self.exit_stub(r, L)
        self.label(L("END"))
        self.waitcnt(vm=0, lgkm=0)
        self.exec_all()
        return self


# ---------------------------------------------------------------- the checker
class State(object):
    def __init__(self):
        self.vm = []        # FIFO of destination register sets of outstanding vector-memory loads
        self.lds = []       # FIFO of destination sets of outstanding LDS reads (in order)
        self.smem = []      # outstanding scalar loads (any order)
        self.lgkm_other = 0  # LDS writes outstanding (counted, nothing to protect)
        self.mfma_age = {}  # vector register -> issue slots since the MFMA that wrote it
        self.valu_fresh = set()  # registers written by the previous vector-ALU instruction (store-data hazard)

    def key(self):
        return (tuple(tuple(sorted(d)) for d in self.vm), len(self.lds), len(self.smem))

    def copy(self):
        s = State()
        s.vm = [set(d) for d in self.vm]
        s.lds = [set(d) for d in self.lds]
        s.smem = [set(d) for d in self.smem]
        s.lgkm_other = self.lgkm_other
        s.mfma_age = dict(self.mfma_age)
        s.valu_fresh = set(self.valu_fresh)
        return s


def check(gen):
    ins = gen.ins
    labels = {i.label: k for k, i in enumerate(ins) if i.kind == "label"}
    errors = []
    seen = set()
    work = [(0, State())]
    steps = 0
    while work:
        pc, st = work.pop()
        while pc < len(ins):
            i = ins[pc]
            steps += 1
            if len(errors) > 200:
                break
            if i.kind == "label":
                k = (pc, st.key())
                if k in seen:
                    break
                seen.add(k)
                pc += 1
                continue
            touched = i.vdef | i.vuse
            for d in st.vm:
                if d & touched:
                    errors.append("%d: `%s` touches v%s with a vector-memory load in flight" % (pc, i.text, sorted(d & touched)))
            for d in st.lds:
                if d & touched:
                    errors.append("%d: `%s` touches v%s with an LDS read in flight" % (pc, i.text, sorted(d & touched)))
            for d in st.smem:
                if d & (i.sdef | i.suse):
                    errors.append("%d: `%s` touches s%s with a scalar load in flight" % (pc, i.text, sorted(d & (i.sdef | i.suse))))
            if i.kind != "mfma":
                for v in i.vuse | i.vdef:
                    if st.mfma_age.get(v, 99) < 10:
                        errors.append("%d: `%s` touches v%d %d slots after the MFMA that writes it" % (pc, i.text, v, st.mfma_age[v]))
            else:
                for v in i.vuse - set(i.vdef):  # A / B operands that are fresh MFMA results (never here)
                    if st.mfma_age.get(v, 99) < 10:
                        errors.append("%d: MFMA operand v%d is a fresh MFMA result" % (pc, v))
            if i.kind == "ds_write" and (st.valu_fresh & i.vuse):
                errors.append("%d: `%s` stores v%s right behind the vector-ALU instruction that writes it" % (pc, i.text, sorted(st.valu_fresh & i.vuse)))
            slots = i.n if i.kind == "nop" else 1
            for v in list(st.mfma_age):
                st.mfma_age[v] += slots
            st.valu_fresh = set(i.vdef) if i.kind == "valu" else set()
            if i.kind == "mfma":
                for v in i.vdef:
                    st.mfma_age[v] = 0
            elif i.kind == "vmem":
                st.vm.append(set(i.vdef))
                if len(st.vm) > 60:
                    errors.append("%d: more than 60 vector-memory operations in flight" % pc)
            elif i.kind == "ds_read":
                st.lds.append(set(i.vdef))
            elif i.kind == "ds_write":
                st.lgkm_other += 1
            elif i.kind == "smem":
                st.smem.append(set(i.sdef))
            elif i.kind == "wait":
                vm, lgkm = i.n
                if vm is not None:
                    while len(st.vm) > vm:
                        st.vm.pop(0)
                if lgkm is not None:
                    if lgkm != 0 and st.smem:
                        errors.append("%d: lgkmcnt(%d) with a scalar load outstanding (they return out of order)" % (pc, lgkm))
                    st.lds, st.smem, st.lgkm_other = [], [], 0
            if i.kind == "branch":
                tgt = labels[i.target]
                if i.text.startswith("s_branch"):
                    pc = tgt
                    continue
                work.append((tgt, st.copy()))
            pc += 1
        else:
            if st.vm or st.lds or st.smem:
                errors.append("end of the loop with operations in flight")
        if steps > 20_000_000:
            errors.append("checker did not converge")
            break
        if len(errors) > 200:  # (a broken program: every trip round the loop adds states; the first errors say it all)
            break
    # every label reached?
    reached = {pc for pc, _ in seen}
    for name, k in labels.items():
        if k not in reached:
            errors.append("label %s is never reached" % name)
    return errors, len(seen)


def count_valu(gen, r=0, w=0):
    """vector-ALU instructions of one two-group pass (copy r, w): everything between its label and the first tail"""
    names = [i.label for i in gen.ins]
    a = next(k for k, n in enumerate(names) if n and n.endswith("_B%d%d" % (r, w)))
    b = next(k for k, n in enumerate(names) if n and n.endswith("_B%d%d_one" % (r, w)))
    kinds = {}
    for i in gen.ins[a:b]:
        kinds[i.kind] = kinds.get(i.kind, 0) + 1
    return kinds


def render(gen):
    return "\n".join('  "%s\\n\\t"' % i.text for i in gen.ins)


def main():
    out = ["// GENERATED by tools/gen_tile4_loop.py -- do not edit; `python tools/gen_tile4_loop.py` rewrites it.",
           "// k_conv_tile4's pass loop in gfx950 assembly: see the generator for the design and for the checks it passed.", ""]
    total_err = []
    for incol in (0, 1):
        g = Gen(incol).build()
        errs, nstates = check(g)
        total_err += errs
        kinds = count_valu(g)
        print("incol %d: %d instructions, %d (label, state) pairs checked, one two-group pass: %s" % (incol, len(g.ins), nstates, kinds))
        out.append("#define SCN_T4_ASM_INCOL%d \\" % incol)
        out.append(" \\\n".join('  "%s\\n\\t"' % i.text for i in g.ins))
        out.append("")
    tun = ["// GENERATED by tools/gen_tile4_loop.py -- do not edit.  Timing-only ablations of k_conv_tile4's assembly pass loop (wrong",
           "// results by design): only libscn_hip_tuning.so (-DSCN_TUNING) holds them; tools/bench_conv.py --asm 2,3,4,5,6,9.", ""]
    for n, flags in sorted(ABLATIONS.items()):
        g = Gen(0, flags).build()
        tun.append("// ablation %d: %s" % (n, " ".join(flags)))
        tun.append("#define SCN_T4_ASM_ABL%d \\" % n)
        tun.append(" \\\n".join('  "%s\\n\\t"' % i.text for i in g.ins))
        tun.append("")
    clob = ['"v%d"' % v for v in range(VB, V_END)] + ['"s%d"' % s for s in range(SB, S_END)] + ['"vcc"', '"scc"', '"memory"']
    out.append("#define SCN_T4_ASM_CLOBBERS " + ", ".join(clob))
    out.append("")
    if total_err:
        for e in total_err[:40]:
            print("ERROR", e)
        print("%d errors" % len(total_err))
        return 1
    if "--check" in sys.argv:
        for path, text in ((OUT, out), (OUT_TUNING, tun)):
            old = open(path).read() if os.path.exists(path) else ""
            if old != "\n".join(text):
                print("%s is not what the generator writes: run python tools/gen_tile4_loop.py" % os.path.basename(path))
                return 1
        return 0
    open(OUT, "w").write("\n".join(out))
    open(OUT_TUNING, "w").write("\n".join(tun))
    print("wrote", OUT, "and", OUT_TUNING)
    return 0


if __name__ == "__main__":
    sys.exit(main())
