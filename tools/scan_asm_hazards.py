#!/usr/bin/env python3
"""Static check of the library's gfx950 ISA for the wait-state rules the compiler cannot apply for us.

The compiler's hazard recogniser inserts the s_nop / s_waitcnt these rules need in ITS OWN code and does not look inside
asm statements.  The library has three dozen asm statements (the orbit loops, the tile path of the first pass, the packed
stores), so every pair (writer, reader) of the table below of which AT LEAST ONE instruction sits inside an asm statement is
this lint's business: the compiler knows nothing about that half.  (A pair entirely in compiler code is the compiler's.)

RULES (gfx940-family / gfx950; wait states = instructions issued in between, s_nop N counting N + 1).  Sources: the CDNA3
ISA guide's "manually inserted wait states" table and LLVM's GCNHazardRecognizer for this target (hasVDecCoExecHazard,
hasTransForwardingHazard); round 3's GPU fault was the first of them.

  sgpr->vmem     VALU writes an SGPR / VCC (v_readlane, v_readfirstlane, v_cmp* with a scalar destination, carry-out of
                 v_add_co / v_sub_co / v_addc_co / v_mad_u64, v_div_scale)  ->  VMEM (global_ / buffer_ / flat_ / scratch_)
                 reads it as an address or offset ............................................................... 5
  sgpr->smem     the same writers -> s_load / s_buffer_load reading it (conservative: required on older targets only) . 5
  sgpr->lanesel  VALU writes an SGPR / VCC -> v_readlane / v_writelane uses it as the LANE SELECT ................... 4
  sgpr->valu     VALU writes an SGPR / VCC -> VALU reads it as an operand (v_cndmask, v_addc, a scalar source) ...... 2
  exec->lane     VALU writes EXEC (v_cmpx*, a VALU with an exec destination) -> v_readlane / v_readfirstlane /
                 v_writelane ....................................................................................... 4
  exec->dpp      VALU writes EXEC -> a VALU DPP operation ........................................................... 5
  vgpr->dpp      VALU writes a VGPR -> a VALU DPP operation reads it ................................................ 2
  vgpr->readlane VALU writes a VGPR -> v_readlane reads it .......................................................... 1
  trans->valu    a transcendental (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) writes a VGPR -> a
                 non-transcendental VALU reads it ................................................................... 1
  vcc->div_fmas  VALU writes VCC -> v_div_fmas ...................................................................... 4
  vccz/execz     VALU writes VCC / EXEC -> a VALU uses vccz / execz as a data source ................................ 5
  m0->lds        SALU writes M0 -> LDS add-TID / GDS / s_sendmsg / a load-to-LDS / v_interp / s_movrel .............. 1
  smem-in-asm    an s_load inside an asm statement whose result is read, or whose statement ends, before an
                 s_waitcnt lgkmcnt(0) inside the same statement (the compiler's waitcnt insertion does not know the
                 load exists) ........................................................................... (not a count)

The look-back follows the control flow: at a label it continues both through the fall-through predecessor and through every
branch that targets the label (so a reader at a loop head sees the writer at the loop's end); it stops at unconditional
transfers.  A register overwritten on the way by another instruction is dropped from the search.  A lint, not a proof.

Usage: python tools/scan_asm_hazards.py [listing.s]    (no argument: compiles fr_kernels.hip with the build's flags)
Exit code 1 if a pattern is found."""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fractal-renderer_amd", "csrc")

TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)(_legacy|_iflag|_clamp)?_(f16|f32|f64)")
VMEM = ("global_", "buffer_", "flat_", "scratch_")
DPP_WORDS = ("quad_perm", "row_shl", "row_shr", "row_ror", "wave_shl", "wave_shr", "wave_rol", "wave_ror", "row_mirror",
             "row_half_mirror", "row_bcast", "row_newbcast", " dpp", "_dpp")
CARRY_OUT = re.compile(r"^v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|mad_u64_u32|mad_i64_i32|div_scale)")
UNCOND = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_rfe_b64", "s_trap")


def regs_of(tok):
    """'s[4:5]' -> {s4, s5}; 'vcc_lo' -> {vcc}; 'v[2:3]' -> {v2, v3}; 'exec' -> {exec}; anything else -> {}."""
    tok = tok.strip().lstrip("-|").rstrip("|")
    tok = re.sub(r"^(neg|abs|sext)\((.*)\)$", r"\2", tok)
    m = re.match(r"^([sva])\[(\d+):(\d+)\]$", tok)
    if m:
        return {"%s%d" % (m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([sva])(\d+)$", tok)
    if m:
        return {tok}
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    if tok in ("exec", "exec_lo", "exec_hi"):
        return {"exec"}
    if tok == "m0":
        return {"m0"}
    if tok in ("vccz", "execz", "scc"):
        return {tok}
    return set()


class Ins:
    __slots__ = ("line", "text", "op", "ops", "in_asm", "dst", "src")

    def __init__(self, line, text, in_asm):
        self.line, self.text, self.in_asm = line, text, in_asm
        body = text.split(";", 1)[0].split("//", 1)[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0] if parts else ""
        rest = parts[1] if len(parts) > 1 else ""
        # operands end where the modifiers begin (offset:.., glc, row_shr:.. are not registers: regs_of ignores them)
        self.ops = [o.strip() for o in rest.split(",")] if rest else []
        ndst = 1
        op = self.op
        if op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_nop", "s_waitcnt", "s_endpgm", "s_barrier", "s_sendmsg", "s_setprio",
                          "s_sleep", "s_bitcmp", "s_setreg", "s_dcache", "s_icache", "s_trap", "s_setpc")):
            ndst = 0
        elif op.startswith(VMEM) or op.startswith("ds_"):
            # loads: first operand is the destination; stores / atomics without return: none (their first operand is an address)
            ndst = 1 if ("load" in op or "_rtn" in op or "read" in op or "permute" in op or "swizzle" in op) else 0
        elif CARRY_OUT.match(op):
            ndst = 2
        elif op.startswith("v_cmpx"):
            ndst = 1
        self.dst = set()
        for o in self.ops[:ndst]:
            self.dst |= regs_of(o.split()[0] if o else o)
        if op.startswith("v_cmpx"):
            self.dst.add("exec")  # with or without a scalar destination printed
        self.src = set()
        for o in self.ops[ndst:]:
            for tok in o.split():
                self.src |= regs_of(tok)

    def is_valu(self):
        return self.op.startswith("v_")

    def is_dpp(self):
        return self.is_valu() and any(w in self.text for w in DPP_WORDS)

    def cost(self):
        if self.op == "s_nop":
            try:
                return int(self.ops[0], 0) + 1
            except (ValueError, IndexError):
                return 1
        return 1


def parse(lines):
    """-> (instructions, label -> index of the next instruction, asm blocks as (first, last+1) index ranges)."""
    ins, labels, blocks = [], {}, []
    in_asm, start = False, 0
    pending = []
    for n, raw in enumerate(lines, 1):
        s = raw.strip()
        if not s:
            continue
        if "#ASMSTART" in s:
            in_asm, start = True, len(ins)
            continue
        if "#ASMEND" in s:
            in_asm = False
            blocks.append((start, len(ins)))
            continue
        if s.startswith((";", "//")):
            continue
        m = re.match(r"^([.\w$@]+):", s)
        if m:
            pending.append(m.group(1))
            s = s[m.end():].strip()
            if not s:
                continue
        if s.startswith("."):
            continue  # a directive
        for lab in pending:
            labels[lab] = len(ins)
        pending = []
        ins.append(Ins(n, s, in_asm))
    return ins, labels, blocks


def build_preds(ins, labels):
    targets = {}
    for i, x in enumerate(ins):
        if x.op.startswith(("s_branch", "s_cbranch")) and x.ops:
            t = labels.get(x.ops[0])
            if t is not None:
                targets.setdefault(t, []).append(i)
    return targets


# (name, wait states, writer predicate -> set of written hazard registers, reader predicate -> set of registers it is sensitive to)
def w_valu_sgpr(x):
    return {r for r in x.dst if r[0] == "s" or r == "vcc"} if x.is_valu() else set()


def w_valu_exec(x):
    return {"exec"} if x.is_valu() and "exec" in x.dst else set()


def w_valu_vgpr(x):
    return {r for r in x.dst if r[0] == "v" and r != "vcc"} if x.is_valu() else set()


def w_trans_vgpr(x):
    return {r for r in x.dst if r[0] == "v" and r != "vcc"} if TRANS.match(x.op) else set()


def w_valu_vcc_exec(x):
    return {r for r in x.dst if r in ("vcc", "exec")} if x.is_valu() else set()


def w_salu_m0(x):
    return {"m0"} if x.op.startswith("s_") and "m0" in x.dst else set()


def scalar(regs):
    return {r for r in regs if r[0] == "s" or r == "vcc"}


def r_vmem(x):
    return scalar(x.src) if x.op.startswith(VMEM) else set()


def r_smem(x):
    return scalar(x.src) if x.op.startswith(("s_load", "s_buffer_load", "s_scratch_load", "s_store", "s_buffer_store")) else set()


def r_lanesel(x):
    if x.op.startswith(("v_readlane", "v_writelane")) and len(x.ops) >= 3:
        return scalar(regs_of(x.ops[2]))
    return set()


def r_valu_scalar(x):
    return scalar(x.src) if x.is_valu() else set()


def r_rwlane(x):
    return {"exec"} if x.op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) else set()


def r_dpp_exec(x):
    return {"exec"} if x.is_dpp() else set()


def r_dpp_vgpr(x):
    return {r for r in x.src if r[0] == "v" and r != "vcc"} if x.is_dpp() else set()


def r_readlane_vgpr(x):
    return {r for r in x.src if r[0] == "v" and r != "vcc"} if x.op.startswith("v_readlane") else set()


def r_nontrans_valu(x):
    return {r for r in x.src if r[0] == "v" and r != "vcc"} if x.is_valu() and not TRANS.match(x.op) else set()


def r_div_fmas(x):
    return {"vcc"} if x.op.startswith("v_div_fmas") else set()


def r_vccz_execz(x):
    out = set()
    if x.is_valu():
        if "vccz" in x.src:
            out.add("vcc")
        if "execz" in x.src:
            out.add("exec")
    return out


def r_m0_user(x):
    op = x.op
    if (op.startswith(("s_sendmsg", "s_movrel", "v_movrel", "v_interp", "ds_gws", "ds_ordered", "global_load_lds", "scratch_load_lds")) or "addtid" in op
            or "gds" in x.text.split() or (op.startswith("buffer_load") and " lds" in " " + x.text.replace(",", " "))):
        return {"m0"}
    return set()


RULES = [
    ("sgpr->vmem", 5, w_valu_sgpr, r_vmem),
    ("sgpr->smem", 5, w_valu_sgpr, r_smem),
    ("sgpr->lanesel", 4, w_valu_sgpr, r_lanesel),
    ("sgpr->valu", 2, w_valu_sgpr, r_valu_scalar),
    ("exec->lane", 4, w_valu_exec, r_rwlane),
    ("exec->dpp", 5, w_valu_exec, r_dpp_exec),
    ("vgpr->dpp", 2, w_valu_vgpr, r_dpp_vgpr),
    ("vgpr->readlane", 1, w_valu_vgpr, r_readlane_vgpr),
    ("trans->valu", 1, w_trans_vgpr, r_nontrans_valu),
    ("vcc->div_fmas", 4, w_valu_vcc_exec, r_div_fmas),
    ("vccz/execz", 5, w_valu_vcc_exec, r_vccz_execz),
    ("m0->lds", 1, w_salu_m0, r_m0_user),
]


def scan_pairs(ins, labels, only_asm=True):
    targets = build_preds(ins, labels)
    label_at = {}
    for lab, idx in labels.items():
        label_at.setdefault(idx, []).append(lab)
    bad = []
    for i, rd in enumerate(ins):
        for name, waits, wfn, rfn in RULES:
            want = rfn(rd)
            if not want:
                continue
            # depth-first over predecessor paths: (index whose predecessors to visit, wait states in between so far, live regs)
            stack = [(i, 0, frozenset(want))]
            seen = set()
            while stack:
                idx, between, live = stack.pop()
                preds = []
                if idx > 0 and not ins[idx - 1].op.startswith(UNCOND):
                    preds.append(idx - 1)
                preds += targets.get(idx, [])  # branches that target a label sitting right before `idx`
                for p in preds:
                    key = (p, between, live)
                    if key in seen:
                        continue
                    seen.add(key)
                    w = ins[p]
                    hit = wfn(w) & live
                    if hit and (rd.in_asm or w.in_asm or not only_asm):
                        bad.append((name, rd.line, rd.text, w.line, w.text, between, waits, sorted(hit)))
                    nlive = live - w.dst  # whatever `w` wrote, the reader sees w's value, not an older one
                    nb = between + w.cost()
                    if nlive and nb < waits:
                        stack.append((p, nb, frozenset(nlive)))
    return bad


def scan_smem_in_asm(ins, blocks):
    bad = []
    for a, b in blocks:
        pending = {}
        for k in range(a, b):
            x = ins[k]
            if x.op == "s_waitcnt" and ("lgkmcnt(0)" in x.text or re.fullmatch(r"s_waitcnt\s+(0x)?0+", x.text.strip())):
                pending = {}
                continue
            used = (x.src | x.dst) & set(pending)
            for r in sorted(used):
                bad.append(("smem-in-asm", x.line, x.text, pending[r].line, pending[r].text, 0, 0, [r]))
                pending.pop(r, None)
            if x.op.startswith(("s_load", "s_buffer_load")):
                for r in x.dst:
                    pending[r] = x
        for r, ld in sorted(pending.items()):
            bad.append(("smem-in-asm", ins[b - 1].line if b > a else ld.line, "(end of the asm statement)", ld.line, ld.text, 0, 0, [r]))
    return bad


def scan(lines, only_asm=True):
    ins, labels, blocks = parse(lines)
    out = scan_pairs(ins, labels, only_asm) + scan_smem_in_asm(ins, blocks)
    uniq, seen = [], set()
    for h in out:
        key = (h[0], h[1], h[3])
        if key not in seen:
            seen.add(key)
            uniq.append(h)
    return uniq


def build_flags():
    """The flags the library is built with (fractal-renderer_amd/build.py), so that the ISA linted is the ISA that ships."""
    spec = importlib.util.spec_from_file_location("_fr_build", os.path.join(ROOT, "fractal-renderer_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return [f for f in mod.HIPCC_FLAGS if f not in ("-shared", "-fPIC", "-pthread")], mod.find_hipcc()


def listing():
    flags, hipcc = build_flags()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, "fr_kernels.hip")], check=True, cwd=CSRC,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def main(argv=()):
    lines = open(argv[0]).read().split("\n") if len(argv) > 0 else listing()
    bad = scan(lines)
    for name, rl, rt, wl, wt, between, waits, regs in bad:
        print("[%s] line %d: %s   <-   line %d: %s   (%s; %d wait states in between, %d needed)" % (name, rl, rt, wl, wt, ",".join(regs), between, waits))
    ins, _, blocks = parse(lines)
    print("%d instructions, %d of them in %d asm statements, %d rules: %d hazard patterns" % (
        len(ins), sum(1 for x in ins if x.in_asm), len(blocks), len(RULES) + 1, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
