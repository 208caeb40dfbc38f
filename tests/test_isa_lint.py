"""A wait-state rule of gfx950 that the compiler applies in its own code and cannot apply inside asm statements: a memory
instruction reading a scalar register which a vector instruction wrote fewer than five wait states before.  Round 3 hit
it — escape_second_kernel<double> restored a spilled output pointer with v_readlane right in front of store_packed's
asm store, which then went out with the register's old upper half (a GPU memory fault) — so the library's ISA is
scanned for the pattern on every CPU run (tools/scan_asm_hazards.py; hipcc cross-compiles without a GPU)."""
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


def test_scanner_sees_the_pattern():
    t = _tool()
    listing = """
	v_readlane_b32 s14, v93, 5
	v_readlane_b32 s15, v93, 6
	;;#ASMSTART
	global_store_short v0, v7, s[14:15]
	;;#ASMEND
""".split("\n")
    assert len(t.scan(listing)) == 2
    fixed = [ln for ln in listing]
    fixed.insert(4, "\ts_nop 4")
    assert t.scan(fixed) == []
    # a scalar-unit write needs no wait states
    assert t.scan(["\ts_mov_b32 s14, s2", "\ts_mov_b32 s15, s3", "\tglobal_store_short v0, v7, s[14:15]"]) == []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_kernels_hold_no_vector_written_scalar_operand_in_front_of_an_asm_memory_instruction():
    assert _tool().main() == 0
