#!/usr/bin/env python3
"""profiles/r03_c4_first_pass_classes.txt: where the first pass of the two-pass render spends its instructions on C4.

Static side: the gfx950 ISA of escape_first_kernel<T, 4, 7, 4> (hipcc --cuda-device-only -S, the build's flags); its
all-asm tile path is split at its labels (prologue / first block / block loop / finish entry / finishing loop) and the
compiler's colour block behind it is taken from the listing; every instruction is classed by issue cost:
    f32-rate   v_{add,sub,mul,fma,mov,cndmask,cvt_pk_u8,floor,max,min}_f32/b32             (2 cycles nominal, 2.4 measured)
    f64/int    v_*_f64, v_mov_b64, 32-bit integer and logic, 64-bit address arithmetic, and EVERY compare — v_cmp / v_cmpx
               _f32 issue at the f64 rate too (profiles/r03_valu_rates.txt: 4.4 cycles)    (4 nominal, 4.4 measured)
    quarter    v_log_f32, v_rcp_*, v_cvt_f32_f64 / f64 conversions                        (8)
    lane       v_readlane / v_writelane / v_readfirstlane (a vector-issue slot each)
    salu       s_* except waits and nops (one per ~4 cycles per SIMD, shared by its waves)
    lds / vmem / smem   ds_* / global_* / s_load_*
Dynamic side: tools/sim/first_pass_dynamics.py (40 000 random tiles of C4 through the first pass's schedule on the CPU).
The product is compared with the counters of the same build (profiles/r03_c4_{f32,f64}_rocprofv3.txt); what the asm path
and the colour block do not explain is the compiler-generated rest (tile loop head, dispatch on the asm path's status,
store, and the general path of the tiles that leave the asm path: later episodes, hand-over, general colour).
Usage (build container, repo root): python tools/first_pass_classes.py > profiles/r03_c4_first_pass_classes.txt"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fractal-renderer_amd", "csrc")


def classify(ins):
    m = ins.split()[0]
    if m.startswith("s_load"):
        return "smem"
    if m in ("s_waitcnt", "s_nop", "s_endpgm"):
        return None
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith("global_") or m.startswith("buffer_") or m.startswith("flat_"):
        return "vmem"
    if m.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if m.startswith(("v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_")) or "cvt_f32_f64" in m or "cvt_f64_f32" in m:
        return "quarter"
    if m.startswith("v_cmp"):
        return "f64/int"
    if "_f64" in m or m.startswith(("v_mov_b64", "v_lshl_add_u64", "v_mad_u64", "v_lshlrev_b64", "v_pk_")):
        return "f64/int"
    if re.search(r"_(u32|i32|u16|i16|b16)(_|$)", m) and not m.startswith(("v_mov_b32", "v_cndmask", "v_cvt_f32_u32", "v_cvt_pk_u8")):
        return "f64/int"
    if m.startswith(("v_and_", "v_or_", "v_xor_", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_perm", "v_mbcnt", "v_mul_lo", "v_mul_hi", "v_mad_u", "v_add3", "v_add_lshl", "v_lshl_or", "v_and_or", "v_bitop")):
        return "f64/int"
    if m.startswith("v_"):
        return "f32-rate"
    return None


def count(lines):
    c = {}
    for ln in lines:
        ln = ln.strip()
        if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":"):
            continue
        k = classify(ln)
        if k:
            c[k] = c.get(k, 0) + 1
    return c


def kernel_listing(asm, tchar):
    name = "_ZN12_GLOBAL__N_119escape_first_kernelI%sLi4ELi7ELi4EEEv10fr_kparams7fr_kout:" % tchar
    out, on = [], False
    for ln in asm:
        if ln.startswith(name):
            on = True
        if on:
            out.append(ln.rstrip("\n"))
            if "s_endpgm" in ln:
                break
    return out


def regions(listing):
    """the all-asm tile path split at its labels, and the colour block the compiler put behind it"""
    start = next(i for i, ln in enumerate(listing) if "s_load_dwordx2" in ln and listing[i - 1].strip().startswith(";;#ASMSTART"))
    end = next(i for i in range(start, len(listing)) if listing[i].strip().startswith(";;#ASMEND"))
    body = listing[start:end]
    idx = {}
    for i, ln in enumerate(body):
        for tag in (".Ltgo_", ".Ltb_", ".Ltbd_", ".Ltnone_", ".Ltf_", ".Ltfd_", ".Ltout_"):
            if ln.startswith(tag) and ln.rstrip().endswith(":"):
                idx[tag] = i
    fin_end = next(i for i in range(idx[".Ltf_"], len(body)) if "s_cbranch_scc0 .Ltf_" in body[i]) + 1
    reg = {
        "asm prologue (3 scalar loads of the filter's constants, X, A, t, the two start tests)": body[:idx[".Ltgo_"]],
        "first block (reads Y0 / B0 in place; count set)": body[idx[".Ltgo_"]:idx[".Ltb_"]],
        "block loop, per further block": body[idx[".Ltb_"]:idx[".Ltbd_"]],
        "after the blocks: masks, status, finish entry": body[idx[".Ltbd_"]:idx[".Ltf_"]],
        "finishing loop, per iteration": body[idx[".Ltf_"]:fin_end],
        "asm epilogue": body[fin_end:],
    }
    # colour block: from the first v_log_f32 after the asm to the v_cmp_ne that closes the filter
    lo = next(i for i in range(end, len(listing)) if "v_log_f32" in listing[i])
    hi = next(i for i in range(lo, len(listing)) if "v_cmp_ne_u32" in listing[i]) + 1
    # include the d32 / range tests just before the first log (back to the asm's end)
    reg["colour: filter's f32 stage (compiler)"] = [ln for ln in listing[end:hi] if "LBB" not in ln]
    return reg


def main():
    # the flags the library is BUILT with (fractal-renderer_amd/build.py via the lint's helper): the ISA classed is the ISA that ships
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import scan_asm_hazards
    flags = scan_asm_hazards.build_flags()[0] + ["--cuda-device-only", "-S"]
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", s, os.path.join(CSRC, "fr_kernels.hip")], check=True, cwd=CSRC,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(s).readlines()
    dyn = json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sim", "first_pass_dynamics.py"), "40000"],
                                    check=True, capture_output=True, text=True).stdout)
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
    tiles = (16384 // 8) ** 2
    print(__doc__.split("Usage")[0].strip())
    print("\nDynamic weights (tools/sim/first_pass_dynamics.py):")
    for k, v in dyn.items():
        print("  %-48s %s" % (k, round(v, 4) if isinstance(v, float) else v))
    classes = ["f32-rate", "f64/int", "quarter", "lane", "salu", "lds", "vmem", "smem"]
    for tchar, tname, key in (("f", "float", "16384x16384_i4096_f32_julia"), ("d", "double", "16384x16384_i4096_f64_julia")):
        reg = regions(kernel_listing(asm, tchar))
        in_asm = dyn["tiles_finished_inside_the_asm_path"]
        w = {
            "asm prologue (3 scalar loads of the filter's constants, X, A, t, the two start tests)": 1.0,
            "first block (reads Y0 / B0 in place; count set)": 1.0,
            "block loop, per further block": dyn["blocks_per_tile_first_episode"] - 1.0,
            "after the blocks: masks, status, finish entry": 1.0,
            "finishing loop, per iteration": dyn["finishing_iterations_per_tile"] * in_asm,  # (tiles that leave the path finish in the general loop)
            "asm epilogue": 1.0,
            "colour: filter's f32 stage (compiler)": in_asm,
        }
        print("\n==== escape_first_kernel<%s, 4, 7, 4>: instructions per 8x8 tile on C4 (static count x executions per tile) ====" % tname)
        print("%-92s %6s | %s" % ("region", "x/tile", "  ".join("%8s" % c for c in classes)))
        tot = {c: 0.0 for c in classes}
        for name, lines in reg.items():
            c = count(lines)
            print("%-92s %6.2f | %s" % (name, w[name], "  ".join("%8s" % (("%d" % c[k]) if k in c else "-") for k in classes)))
            for k in classes:
                tot[k] += w[name] * c.get(k, 0)
        print("%-92s %6s | %s" % ("asm path + colour, per tile", "", "  ".join("%8.1f" % tot[k] for k in classes)))
        valu_model = tot["f32-rate"] + tot["f64/int"] + tot["quarter"] + tot["lane"]
        rec = pmc.get(key)

        def first_pass_counter(fname, counter, kernel="escape_first_kernel"):
            try:
                for ln in open(os.path.join(ROOT, "profiles", fname)):
                    if kernel in ln and ", %s, " % counter in ln:
                        return float(ln.rsplit(",", 1)[1])
            except OSError:
                pass
            return float("nan")

        fname = "r03_c4_%s_rocprofv3.txt" % ("f32" if tchar == "f" else "f64")
        fv, fs = first_pass_counter(fname, "SQ_INSTS_VALU") / tiles, first_pass_counter(fname, "SQ_INSTS_SALU") / tiles
        print("measured, the first pass alone (profiles/%s): %.1f vector / %.1f scalar instructions per tile" % (fname, fv, fs))
        if tchar == "f":
            v1v = first_pass_counter("r03_c4_f32_v1_rocprofv3.txt", "SQ_INSTS_VALU", "escape_first_v1_kernel") / tiles
            v1s = first_pass_counter("r03_c4_f32_v1_rocprofv3.txt", "SQ_INSTS_SALU", "escape_first_v1_kernel") / tiles
            print("          round 2's first pass, same box, same session (profiles/r03_c4_f32_v1_rocprofv3.txt): %.1f vector / %.1f scalar" % (v1v, v1s))
        if rec:
            first = [k for k in rec["kernels"] if "first" in k]
            print("measured (profiles/%s, build %s): the two kernels together %.1f vector / %.1f scalar instructions per tile;" % (
                os.path.basename(rec["source"]).replace("pmc_", "") + "_rocprofv3.txt", rec["build_id"], rec["sq_insts_valu"] / tiles,
                rec["sq_insts_salu"] / tiles))
        later = dyn["blocks_per_tile_later_episodes"]
        per_block = count(reg["block loop, per further block"])
        print("the first pass alone: see the SQ_INSTS_VALU / SQ_INSTS_SALU lines of escape_first_kernel in that file; of its vector instructions per tile,")
        print("  %.1f are explained above, ~%.0f more are the %.2f blocks per tile of LATER episodes (general path, same block body),"
              % (valu_model, later * sum(per_block.get(k, 0) for k in ("f32-rate", "f64/int")), later))
        print("  the rest is compiler-generated: tile-loop head (one ds_bpermute, a few integer ops), store (0 vector: scalar base + lane offset),")
        print("  hand-over (16 % of tiles: an atomic, 3-4 stores, ~20 vector), general colour path (2 % of tiles), block prologue (coordinates: 2 f64 divisions per 28 tiles).")


if __name__ == "__main__":
    main()
