"""tools/isa_lint.py: the build-time recount of the hand-counted `s_waitcnt vmcnt(N)` in front of the LDS-DMA landing slots
(kernel_regtile64p.h).  Runs on a synthetic assembly listing — no compiler, no GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_lint)


def listing(first, steady, extra_store_in_else=True, tags=True):
    """Prologue: 1 DMA + 3 loads.  Loop: guard(s), DMA, then 2 stores + 2 loads on the hot path (+ 1 store on a cold branch)."""
    cold = "\tbuffer_store_dwordx4 v[0:3], v9, s[4:7], 0 offen\n" if extra_store_in_else else ""
    t1, t2 = (" ; lint: first", " ; lint: steady") if tags else ("", "")
    return f"""
_Z6kernelv:                             ; @_Z6kernelv
\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds
\tbuffer_load_dwordx4 v[0:3], v9, s[4:7], 0 offen
\tbuffer_load_dwordx4 v[4:7], v9, s[4:7], 0 offen
\tglobal_load_dwordx2 v[8:9], v[10:11], off
\ts_branch .LBB0_2
.LBB0_1:
\ts_cbranch_scc1 .LBB0_9
.LBB0_2:                                ; =>This Inner Loop Header: Depth=1
\ts_cmp_eq_u32 s1, 0
\ts_cbranch_scc0 .LBB0_4
\t;;#ASMSTART
\ts_waitcnt vmcnt({first}){t1}
\t;;#ASMEND
\ts_branch .LBB0_5
.LBB0_4:
\t;;#ASMSTART
\ts_waitcnt vmcnt({steady}){t2}
\t;;#ASMEND
.LBB0_5:
\tds_read_b128 v[0:3], v20
\ts_waitcnt vmcnt(0)
\tv_add_f32_e32 v0, v1, v2
\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds
\tbuffer_store_dwordx4 v[0:3], v9, s[4:7], 0 offen
\tbuffer_load_dwordx4 v[0:3], v9, s[4:7], 0 offen
\ts_cbranch_vccnz .LBB0_7
{cold}.LBB0_7:
\tbuffer_store_dwordx4 v[4:7], v9, s[4:7], 0 offen
\tbuffer_load_dwordx4 v[4:7], v9, s[4:7], 0 offen
\ts_branch .LBB0_1
.LBB0_9:
\ts_endpgm
.Lfunc_end0:
"""


def run(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    kernels = isa_lint.parse_kernels(str(p))
    assert list(kernels) == ["_Z6kernelv"]
    return isa_lint.lint_kernel("_Z6kernelv", kernels["_Z6kernelv"])


def test_correct_counts_pass(tmp_path):
    errors, notes = run(tmp_path, listing(first=3, steady=4))
    assert not errors, errors
    assert any("vmcnt(4) [steady state]" in n and "min 4, max 5" in n for n in notes), notes      # the cold branch adds a store: stricter, not wrong
    assert any("vmcnt(3) [first tile]" in n and "min 3, max 3" in n for n in notes), notes


@pytest.mark.parametrize("first,steady", [(3, 5), (4, 4)])
def test_too_loose_guard_fails(tmp_path, first, steady):
    errors, _ = run(tmp_path, listing(first=first, steady=steady))
    assert errors and "TOO LOOSE" in errors[0]


def test_missing_guard_is_an_error(tmp_path):
    text = listing(3, 4).replace(";;#ASMSTART", "; x").replace(";;#ASMEND", "; y")   # the waits no longer come from inline asm
    errors, _ = run(tmp_path, text)
    assert errors and "no inline-asm" in errors[0]


def test_main_exit_codes(tmp_path):
    good, bad = tmp_path / "good.s", tmp_path / "bad.s"
    good.write_text(listing(3, 4))
    bad.write_text(listing(3, 6))
    assert isa_lint.main([str(good), "--kernel", "kernel", "--quiet"]) == 0
    assert isa_lint.main([str(bad), "--kernel", "kernel", "--quiet"]) == 1


def test_untagged_guards_fall_back_to_their_counts(tmp_path):
    errors, notes = run(tmp_path, listing(3, 4, tags=False))
    assert not errors and len(notes) == 2
    errors, _ = run(tmp_path, listing(4, 4, tags=False))              # same count twice and no tags: refuse to guess
    assert errors and "untagged" in errors[0]
