"""Wait-state rules of gfx950 that the compiler applies in its own code and cannot apply inside asm statements.  Round 3 hit
the first — escape_second_kernel<double> restored a spilled output pointer with v_readlane right in front of
store_packed's asm store, which then went out with the register's old upper half (a GPU memory fault).  Round 4 turned the
lint into a TABLE of rules (tools/scan_asm_hazards.py: writer class, reader class, wait states) that follows branches into
loop heads; here: one synthetic listing per rule (the pattern is seen, the fixed form is clean), and zero hits on the ISA the
library is built from (hipcc cross-compiles without a GPU)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("scan_asm_hazards", os.path.join(ROOT, "tools", "scan_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


T = _tool()


def rules_hit(text, only_asm=True):
    return sorted({h[0] for h in T.scan(text.split("\n"), only_asm)})


def asm(*body):
    return "\n".join(["\t;;#ASMSTART"] + ["\t" + b for b in body] + ["\t;;#ASMEND"])


def test_scanner_sees_round_3s_fault_and_its_fix():
    listing = "\tv_readlane_b32 s14, v93, 5\n\tv_readlane_b32 s15, v93, 6\n" + asm("global_store_short v0, v7, s[14:15]")
    assert rules_hit(listing) == ["sgpr->vmem"] and len(T.scan(listing.split("\n"))) == 2
    fixed = "\tv_readlane_b32 s14, v93, 5\n\tv_readlane_b32 s15, v93, 6\n" + asm("s_nop 4", "global_store_short v0, v7, s[14:15]")
    assert rules_hit(fixed) == []
    # a scalar-unit write needs no wait states; a pair wholly in compiler code is the compiler's business, not the lint's
    assert rules_hit("\ts_mov_b32 s14, s2\n\ts_mov_b32 s15, s3\n" + asm("global_store_short v0, v7, s[14:15]")) == []
    plain = "\tv_readlane_b32 s14, v93, 5\n\tglobal_store_short v0, v7, s[14:15]"
    assert rules_hit(plain) == [] and rules_hit(plain, only_asm=False) == ["sgpr->vmem"]


CASES = {
    # rule: (hazard, fixed)
    "sgpr->vmem (v_cmp carry-out as an offset)": (
        "\tv_add_co_u32_e64 v1, s[6:7], v2, v3\n" + asm("global_load_dword v4, v5, s[6:7]"),
        "\tv_add_co_u32_e64 v1, s[6:7], v2, v3\n" + asm("s_nop 4", "global_load_dword v4, v5, s[6:7]"), "sgpr->vmem"),
    "sgpr->vmem (v_cmp with a scalar destination)": (
        asm("v_cmp_lt_f64 s[8:9], v[0:1], v[2:3]", "s_nop 3", "buffer_store_dword v1, v2, s[8:11], 0 offen"),
        asm("v_cmp_lt_f64 s[8:9], v[0:1], v[2:3]", "s_nop 4", "buffer_store_dword v1, v2, s[8:11], 0 offen"), "sgpr->vmem"),
    "sgpr->smem": (
        "\tv_readfirstlane_b32 s4, v0\n\tv_readfirstlane_b32 s5, v1\n" + asm("s_load_dwordx2 s[8:9], s[4:5], 0x10", "s_waitcnt lgkmcnt(0)"),
        "\tv_readfirstlane_b32 s4, v0\n\tv_readfirstlane_b32 s5, v1\n" + asm("s_nop 4", "s_load_dwordx2 s[8:9], s[4:5], 0x10", "s_waitcnt lgkmcnt(0)"),
        "sgpr->smem"),
    "sgpr->lanesel (readlane)": (
        asm("v_readfirstlane_b32 s3, v9", "s_nop 2", "v_readlane_b32 s7, v4, s3"),
        asm("v_readfirstlane_b32 s3, v9", "s_nop 3", "v_readlane_b32 s7, v4, s3"), "sgpr->lanesel"),
    "sgpr->lanesel (writelane, vcc written by a compare)": (
        asm("v_cmp_eq_u32 vcc, v1, v2", "s_nop 2", "v_writelane_b32 v4, s9, vcc_lo"),
        asm("v_cmp_eq_u32 vcc, v1, v2", "s_nop 3", "v_writelane_b32 v4, s9, vcc_lo"), "sgpr->lanesel"),
    "sgpr->valu (v_cndmask reads the vcc a compare wrote)": (
        asm("v_cmp_gt_f64 vcc, s[0:1], v[2:3]", "s_nop 0", "v_cndmask_b32 v4, 0, v4, vcc"),
        asm("v_cmp_gt_f64 vcc, s[0:1], v[2:3]", "s_nop 1", "v_cndmask_b32 v4, 0, v4, vcc"), "sgpr->valu"),
    "sgpr->valu (a scalar source that readlane wrote)": (
        asm("v_readlane_b32 s20, v8, 3") + "\n\tv_add_f32_e32 v1, s20, v1",
        asm("v_readlane_b32 s20, v8, 3", "s_nop 1") + "\n\tv_add_f32_e32 v1, s20, v1", "sgpr->valu"),
    "exec->lane": (
        asm("v_cmpx_nlt_f64 vcc, v[0:1], v[2:3]") + "\n\ts_mov_b32 s1, 0\n\tv_readfirstlane_b32 s5, v3",
        asm("v_cmpx_nlt_f64 vcc, v[0:1], v[2:3]", "s_nop 3") + "\n\tv_readfirstlane_b32 s5, v3", "exec->lane"),
    "exec->dpp": (
        asm("v_cmpx_lt_f32 vcc, v0, v1", "s_nop 3") + "\n\tv_mov_b32_dpp v2, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
        asm("v_cmpx_lt_f32 vcc, v0, v1", "s_nop 4") + "\n\tv_mov_b32_dpp v2, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "exec->dpp"),
    "vgpr->dpp": (
        asm("v_add_f32 v3, v1, v2", "s_nop 0", "v_add_f32_dpp v5, v3, v3 row_shr:1 row_mask:0xf bank_mask:0xf"),
        asm("v_add_f32 v3, v1, v2", "s_nop 1", "v_add_f32_dpp v5, v3, v3 row_shr:1 row_mask:0xf bank_mask:0xf"), "vgpr->dpp"),
    "vgpr->readlane": (
        asm("v_mov_b32 v7, v1", "v_readlane_b32 s4, v7, 0"),
        asm("v_mov_b32 v7, v1", "s_nop 0", "v_readlane_b32 s4, v7, 0"), "vgpr->readlane"),
    "trans->valu": (
        asm("v_log_f32 v1, v0", "v_mul_f32 v2, 0x3e800000, v1"),
        asm("v_log_f32 v1, v0", "s_nop 0", "v_mul_f32 v2, 0x3e800000, v1"), "trans->valu"),
    "vcc->div_fmas": (
        asm("v_div_scale_f64 v[0:1], vcc, v[2:3], v[4:5], v[2:3]", "s_nop 2", "v_div_fmas_f64 v[6:7], v[0:1], v[8:9], v[10:11]"),
        asm("v_div_scale_f64 v[0:1], vcc, v[2:3], v[4:5], v[2:3]", "s_nop 3", "v_div_fmas_f64 v[6:7], v[0:1], v[8:9], v[10:11]"),
        "vcc->div_fmas"),
    "vccz/execz as data": (
        asm("v_cmpx_lt_f32 vcc, v0, v1", "s_nop 3", "v_mov_b32 v2, execz"),
        asm("v_cmpx_lt_f32 vcc, v0, v1", "s_nop 4", "v_mov_b32 v2, execz"), "vccz/execz"),
    "m0->lds": (
        asm("s_mov_b32 m0, s4", "buffer_load_dword v1, s[8:11], 0 offen lds"),
        asm("s_mov_b32 m0, s4", "s_nop 0", "buffer_load_dword v1, s[8:11], 0 offen lds"), "m0->lds"),
    "smem-in-asm (used before the wait)": (
        asm("s_load_dwordx2 s[8:9], s[0:1], 0x10", "v_mov_b32 v1, s8", "s_waitcnt lgkmcnt(0)"),
        asm("s_load_dwordx2 s[8:9], s[0:1], 0x10", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v1, s8"), "smem-in-asm"),
    "smem-in-asm (the statement ends with the load outstanding)": (
        asm("s_load_dwordx2 s[8:9], s[0:1], 0x10", "v_mov_b32 v1, v2"),
        asm("s_load_dwordx2 s[8:9], s[0:1], 0x10", "v_mov_b32 v1, v2", "s_waitcnt vmcnt(0) lgkmcnt(0)"), "smem-in-asm"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_each_rule_sees_its_pattern_and_accepts_the_fix(name):
    hazard, fixed, rule = CASES[name]
    assert rule in rules_hit(hazard), (name, T.scan(hazard.split("\n")))
    assert rule not in rules_hit(fixed), (name, T.scan(fixed.split("\n")))


def test_every_rule_of_the_table_has_a_case():
    assert {r[0] for r in T.RULES} | {"smem-in-asm"} == {c[2] for c in CASES.values()}


def test_the_look_back_follows_a_back_edge_into_the_loop_head():
    """The reader is the FIRST instruction of a loop, the writer its last: linear look-back (round 3's lint) sees only the
    harmless code in front of the loop."""
    loop = "\n".join(["\ts_mov_b32 s6, s2", "\ts_mov_b32 s7, s3", "\ts_nop 4", ".LBB0_1:", asm("global_store_dword v0, v1, s[6:7]"),
                      "\tv_add_u32_e32 v0, 4, v0", "\ts_cmp_lt_u32 s9, s10", "\tv_readlane_b32 s6, v9, 0", "\ts_cbranch_scc1 .LBB0_1",
                      "\ts_endpgm"])
    assert rules_hit(loop) == ["sgpr->vmem"]
    assert rules_hit(loop.replace("\ts_cbranch_scc1 .LBB0_1", "\ts_nop 3\n\ts_cbranch_scc1 .LBB0_1")) == []
    # an unconditional branch in front of a label: nothing falls through, only the branch sites count
    skip = "\n".join(["\tv_readlane_b32 s6, v9, 0", "\ts_branch .LBB0_3", ".LBB0_2:", asm("global_store_dword v0, v1, s[6:7]"), "\ts_endpgm",
                      ".LBB0_3:", "\ts_nop 4", "\ts_branch .LBB0_2"])
    assert rules_hit(skip) == []


def test_a_register_overwritten_on_the_way_is_not_a_hazard():
    text = "\tv_readlane_b32 s6, v9, 0\n\ts_mov_b32 s6, s20\n" + asm("global_store_dword v0, v1, s[6:7]")
    assert rules_hit(text) == []


def test_the_lint_uses_the_build_s_own_flags():
    flags, hipcc = T.build_flags()
    assert "--offload-arch=gfx950" in flags and "-ffp-contract=off" in flags and "-O3" in flags and "-shared" not in flags
    assert os.path.exists(hipcc)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_the_library_s_isa_holds_none_of_the_patterns():
    assert T.main() == 0
