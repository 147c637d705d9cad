"""fft_amd/isa_lint.py: the build-time recount of the hand-counted `s_waitcnt vmcnt(N)` in front of the LDS-DMA landing slots
(kernel_regtile64p.h).  Runs on a synthetic assembly listing — no compiler, no GPU."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "fft_amd", "isa_lint.py"))
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


# ---- --check lds: inline-asm ds_read_b32 (kernel_regtile_mixedp.h) must be consumed behind an s_waitcnt lgkmcnt(0) --------------------
def lds_listing(use_before_wait=False, wait="s_waitcnt lgkmcnt(0)", clobber_addr=False):
    early = "\tv_add_f32_e32 v3, v11, v4\n" if use_before_wait else ""
    dst2 = "v20" if clobber_addr else "v12"
    return f"""
_Z6kernelv:                             ; @_Z6kernelv
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB0_2
\t;;#ASMSTART
\tds_read_b32 v10, v20 offset:0
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b32 v11, v20 offset:32
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b32 {dst2}, v20 offset:64
\t;;#ASMEND
.LBB0_2:
\ts_or_b64 exec, exec, s[0:1]
{early}\tv_mul_f32_e32 v5, v6, v7
\t;;#ASMSTART
\t{wait}
\ts_barrier
\t;;#ASMEND
\tv_add_f32_e32 v3, v10, v11
\tv_add_f32_e32 v3, v3, v12
\ts_endpgm
.Lfunc_end0:
"""


def run_lds(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    kernels = isa_lint.parse_kernels(str(p))
    return isa_lint.lint_lds_reads("_Z6kernelv", kernels["_Z6kernelv"])


def test_lds_reads_behind_the_barrier_pass(tmp_path):
    errors, notes = run_lds(tmp_path, lds_listing())
    assert not errors, errors
    assert "3 untracked ds_read_b32" in notes[0]


def test_lds_read_used_before_the_wait_fails(tmp_path):
    errors, _ = run_lds(tmp_path, lds_listing(use_before_wait=True))
    assert errors and "touches v11" in errors[0]


def test_lds_read_behind_a_partial_wait_fails(tmp_path):
    errors, _ = run_lds(tmp_path, lds_listing(wait="s_waitcnt lgkmcnt(1)"))
    assert errors                                     # lgkmcnt(1) leaves the last read in flight: its first use is flagged


def test_lds_read_into_its_own_address_register_fails(tmp_path):
    errors, _ = run_lds(tmp_path, lds_listing(clobber_addr=True))
    assert errors and "address register" in errors[0]


def test_lds_check_without_reads_is_an_error(tmp_path):
    errors, _ = run_lds(tmp_path, lds_listing().replace("ds_read_b32", "ds_read_b64"))
    assert errors and "no inline-asm ds_read_b32" in errors[0]


# ---- --check addtid: ds_write_addtid_b32 takes its base from M0, set inside the same asm statement ------------------------------------
def addtid_listing(own_m0=True, compiler_m0=False, compiler_reads_m0=False):
    mov = "\ts_mov_b32 m0, s4\n\ts_nop 0\n" if own_m0 else ""
    other = "\ts_mov_b32 m0, s9\n\tbuffer_load_dword v1, s[0:3], 0 offen lds\n" if compiler_m0 else ""
    if compiler_reads_m0:
        other += "\ts_add_i32 m0, m0, s7\n"
    return f"""
_Z6kernelv:                             ; @_Z6kernelv
{other}\t;;#ASMSTART
{mov}\tds_write_addtid_b32 v3 offset:1024
\t;;#ASMEND
\t;;#ASMSTART
\ts_mov_b32 m0, s5
\ts_nop 0
\tds_write_addtid_b32 v4 offset:2048
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:
"""


def run_addtid(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    kernels = isa_lint.parse_kernels(str(p))
    return isa_lint.lint_addtid("_Z6kernelv", kernels["_Z6kernelv"])


def test_addtid_writes_with_their_own_m0_pass(tmp_path):
    errors, notes = run_addtid(tmp_path, addtid_listing())
    assert not errors, errors
    assert "2 ds_write_addtid_b32" in notes[0]


def test_addtid_write_without_m0_fails(tmp_path):
    errors, _ = run_addtid(tmp_path, addtid_listing(own_m0=False))
    assert errors and "not preceded by s_mov_b32 m0" in errors[0]


def test_compiler_setting_m0_for_an_lds_dma_next_to_addtid_passes(tmp_path):
    """Round 4: the persistent mixed-radix kernel stages row blocks by LDS-DMA; hipcc sets M0 in front of every such request.  The asm
    statements declare M0 clobbered and set it themselves, so a compiler-generated WRITE is harmless."""
    errors, notes = run_addtid(tmp_path, addtid_listing(compiler_m0=True))
    assert not errors, errors


def test_compiler_reading_m0_next_to_addtid_fails(tmp_path):
    errors, _ = run_addtid(tmp_path, addtid_listing(compiler_reads_m0=True))
    assert errors and "uses M0" in errors[0]


# ---- serialised global loads (round 6: replaces the timing asserts of tests/test_perf_sanity_gpu.py) ----------------------------------
def _serial_listing(n_pairs, name="_ZN4sfft9some_kernILi1EEEvNS_11RegtileArgsE", lds_dma=False):
    body = []
    for i in range(n_pairs):
        body.append(f"\tglobal_load_dword v{i}, v[100:101], off" + (" lds" if lds_dma else ""))
        body.append("\ts_waitcnt vmcnt(0)")
        body.append(f"\tv_lshlrev_b32_e32 v{i}, 16, v{i}")
    return f"{name}:                     ; @{name}\n" + "\n".join(body) + "\n\ts_endpgm\n.Lfunc_end0:\n"


def test_serialised_loads_fail_and_overlapped_loads_pass(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text(_serial_listing(9))
    assert isa_lint.serial_load_runs(str(bad)) == {"_ZN4sfft9some_kernILi1EEEvNS_11RegtileArgsE": 9}
    errors, _ = isa_lint.lint_serial_loads(str(bad))
    assert errors and "one request in flight" in errors[0]
    ok = tmp_path / "ok.s"
    ok.write_text(_serial_listing(7))                                     # below the threshold
    assert isa_lint.lint_serial_loads(str(ok)) == ([], [])
    # sixteen loads in flight, ONE wait: not a run
    name = "_ZN4sfft9some_kernILi2EEEvNS_11RegtileArgsE"
    flight = f"{name}:                     ; @{name}\n" + "\n".join(f"\tglobal_load_dword v{i}, v[100:101], off" for i in range(16)) + "\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n.Lfunc_end1:\n"
    fl = tmp_path / "flight.s"
    fl.write_text(flight)
    assert isa_lint.serial_load_runs(str(fl))[name] == 1
    dma = tmp_path / "dma.s"
    dma.write_text(_serial_listing(12, lds_dma=True))                     # LDS-DMA requests are not loads into registers
    assert isa_lint.lint_serial_loads(str(dma)) == ([], [])


def test_known_serial_runs_are_accepted_up_to_their_length(tmp_path):
    known = next(iter(isa_lint.KNOWN_SERIAL_RUNS.items()))
    name = "_ZN4sfft22" + known[0] + "EEvNS_11RegtileArgsE"
    f = tmp_path / "k.s"
    f.write_text(_serial_listing(known[1], name=name))
    errors, notes = isa_lint.lint_serial_loads(str(f))
    assert not errors and notes
    f.write_text(_serial_listing(known[1] + 1, name=name))
    errors, _ = isa_lint.lint_serial_loads(str(f))
    assert errors and "known run" in errors[0]


def test_lint_listing_runs_the_serial_check_on_every_kernel(tmp_path):
    f = tmp_path / "unit.s"
    f.write_text(_serial_listing(20))
    errors, _ = isa_lint.lint_listing(str(f))
    assert any("one request in flight" in e for e in errors)
