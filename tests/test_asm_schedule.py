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
