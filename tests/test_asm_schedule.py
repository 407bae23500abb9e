"""k_conv_tile4 issues its loads from inline asm and waits with hand-counted s_waitcnt; hipcc does not know that a
destination register is invalid until the wait.  tools/check_asm_loads.py proves on the emitted assembly that nothing
touches a register with a load in flight.  Here: the checker finds a planted violation, and the current source is clean."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SNIPPET = """
_ZN3scn12k_conv_tile4ILi0EEEvPKf:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\t;;#ASMSTART
\tbuffer_load_dwordx4 v[8:11], v2, s[12:15], s3 offen
\t;;#ASMEND
\tv_mfma_f32_16x16x4_f32 a[0:3], v20, v21, a[0:3]
%s
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_add_f32_e32 v30, v8, v9
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
"""


def _run(text, tmp_path):
    import check_asm_loads
    f = tmp_path / "k.s"
    f.write_text(text)
    argv = sys.argv
    try:
        sys.argv = ["check_asm_loads.py", str(f)]
        return check_asm_loads.main()
    finally:
        sys.argv = argv


def test_checker_detects_a_touched_in_flight_register(tmp_path, capsys):
    assert _run(SNIPPET % "\tv_mov_b32_e32 v40, v41", tmp_path) == 0          # unrelated registers: fine
    assert _run(SNIPPET % "\tv_mov_b32_e32 v40, v9", tmp_path) == 1           # reads a destination before the wait
    assert "still in flight" in capsys.readouterr().out
    assert _run(SNIPPET % "\tv_mov_b32_e32 v10, v41", tmp_path) == 1          # overwrites one


BRANCHY = """
_ZN3scn12k_conv_tile6ILi128ELb0ELi0EEEvPKf:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:
	;;#ASMSTART
	buffer_load_dwordx4 v[8:11], v2, s[12:15], s3 offen
	;;#ASMEND
	s_cbranch_scc1 .LBB0_3
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
.LBB0_3:
%s
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_cbranch_vccnz .LBB0_1
	s_endpgm
.Lfunc_end0:
"""


def test_checker_follows_branches(tmp_path, capsys):
    """the wait is skipped on ONE path only: a linear walk of the listing sees it, the path walk does not let it count"""
    assert _run(BRANCHY % "\tv_mov_b32_e32 v40, v41", tmp_path) == 0
    assert _run(BRANCHY % "\tv_mov_b32_e32 v40, v9", tmp_path) == 1
    assert "still in flight" in capsys.readouterr().out


SCALAR = """
_ZN3scn12k_conv_tile6ILi128ELb0ELi0EEEvPKf:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:
	;;#ASMSTART
	s_load_dwordx2 s[60:61], s[40:41], s7
	;;#ASMEND
	v_mfma_f32_16x16x4_f32 a[0:3], v20, v21, a[0:3]
%s
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	s_mov_b64 s[62:63], s[60:61]
	;;#ASMEND
	s_lshr_b32 s8, s63, 12
	s_cbranch_vccnz .LBB0_1
	s_endpgm
.Lfunc_end0:
"""


def test_checker_tracks_scalar_loads_issued_from_inline_asm(tmp_path, capsys):
    """round-3 advisor: k_conv_tile4 requests its pass-table entry with an s_load in one asm statement and picks it up ~430
    instructions later in another; nothing proved that hipcc leaves the destination pair alone in between"""
    assert _run(SCALAR % "\ts_add_u32 s9, s9, 1", tmp_path) == 0
    assert _run(SCALAR % "\ts_mov_b32 s9, s61", tmp_path) == 1            # reads the pair before the wait
    assert "scalar load still in flight" in capsys.readouterr().out
    assert _run(SCALAR % "\ts_mov_b32 s60, 0", tmp_path) == 1             # re-homes something into it


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_current_kernel_source_is_clean():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py")], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " 0 violations" in out.stdout


# ------------------------------------------------------------------ round 6: the generated assembly pass loop
def test_generated_pass_loop_is_current_and_passes_its_own_checks():
    """csrc/tile4_loop_asm.inc is what tools/gen_tile4_loop.py writes (the build refuses a stale file too), and the generator's
    path-exact checks -- loads in flight, MFMA result hazards, LDS / scalar-load waits, over every path through the six copies of
    the pass body -- pass for both rulebook columns."""
    import gen_tile4_loop as G
    for incol in (0, 1):
        g = G.Gen(incol).build()
        errs, nstates = G.check(g)
        assert errs == [] and nstates > 300
        kinds = G.count_valu(g)
        assert kinds["mfma"] == 64 and kinds["valu"] == 12 and kinds["vmem"] == 18  # a two-group pass: 12 vector-ALU instructions
    rc = subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_tile4_loop.py"), "--check"], stdout=subprocess.DEVNULL)
    assert rc == 0, "occuseg_amd/csrc/tile4_loop_asm.inc is stale: python tools/gen_tile4_loop.py"


def test_pass_loop_checker_finds_planted_faults():
    """the generator's checker is not vacuous: a wait that leaves one load too many in flight, a missing LDS wait and a tile
    write placed right behind the MFMA that produces its operand are all found"""
    import gen_tile4_loop as G

    def errors_of(mutate):
        g = G.Gen(0).build()
        mutate(g)
        return G.check(g)[0]

    def loosen_vmcnt(g):
        for i in g.ins:
            if i.kind == "wait" and i.n == (8, None):
                i.n, i.text = (9, None), "s_waitcnt vmcnt(9)"
    assert any("vector-memory load in flight" in e for e in errors_of(loosen_vmcnt))

    def drop_lgkm_waits(g):
        g.ins = [i for i in g.ins if not (i.kind == "wait" and i.n == (None, 0))]
    assert any("LDS read in flight" in e or "scalar load in flight" in e for e in errors_of(drop_lgkm_waits))

    def early_tile_add(g):  # move the first packed add of a tile update right behind the last MFMA that writes its operand
        adds = [k for k, i in enumerate(g.ins) if i.kind == "valu" and i.text.startswith("v_pk_add_f32")]
        at = next(k for k in adds if any(G.ACC(0, 0) <= v < G.ACC(1, 0) for v in g.ins[k].vuse))  # a group-0 result
        add = g.ins[at]
        src = next(v for v in sorted(add.vuse) if G.ACC(0, 0) <= v < G.ACC(1, 0))
        k = max(j for j in range(at) if g.ins[j].kind == "mfma" and src in g.ins[j].vdef)
        g.ins.insert(k + 1, G.Ins(add.text, "valu", vdef=add.vdef, vuse=add.vuse))
    assert any("slots after the MFMA" in e for e in errors_of(early_tile_add))


JOIN_COPY = """
_ZN3scn12k_conv_tile4ILi0ELi0ELi1ELi1ELi0ELi1EEEvPKf:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	v_and_b32_e32 v101, 28, v2
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
	v_add_u32_e32 v5, 1, v5
.LBB0_2:
%s
	s_or_b64 exec, exec, s[0:1]
	v_lshlrev_b32_e32 v3, 2, v27
	s_endpgm
.Lfunc_end0:
"""


def test_checker_refuses_register_copies_in_front_of_an_exec_restore(tmp_path, capsys):
    """Round 6, found with rocgdb: hipcc put `v_mov_b32 v27, v101` -- a value every lane needs behind the assembly block -- at the
    top of the block that joins the `tid < 64` region, IN FRONT of the `s_or_b64 exec` that re-enables wave 1.  The build now
    refuses that shape in the ASM = 1 instantiations of k_conv_tile4."""
    assert _run(JOIN_COPY % "\tv_mov_b32_e32 v27, v101", tmp_path) == 1
    assert "before the EXEC restore" in capsys.readouterr().out
    assert _run(JOIN_COPY % "\ts_nop 0", tmp_path) == 0
