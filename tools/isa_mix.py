#!/usr/bin/env python3
"""Instruction-mix audit of a compiled kernel's loop (development tool).

    bash tools/spec_asm.sh c3 <the scene's -D options>          # leaves /tmp/k/c3.s
    python tools/isa_mix.py /tmp/k/c3.s [--cost profiles/ubench_valu_r03.txt] [--split 0x17800000]

Takes the INNERMOST loop of the kernel (for the area-light kernels: one shadow sample of intensity_at), classifies every
instruction of its body by issue class, prices the VALU instructions with the per-class cost measured by tools/ubench.hip
on a loaded MI355X (SIMD cycles per wave-instruction at 4 waves per SIMD), and prints the histogram: instructions and
cycles per class, the mean cost of a VALU instruction of THIS mix -- the mix-aware roof -- next to the 2-cycle roof the
headline `roofline.frac` is quoted against.  --split <literal>: report the part of the loop body before the first
instruction that mentions the literal separately (C3: the margin-guarded fast decision before, the exact path after --
`s_mov_b32 s1, 0x17800000` is normalize_exact's range check).
The static count weights every basic block of the part once: a per-sample upper bound for straight-line parts (the fast
decision is straight-line per object by construction), not a profile.
"""
import argparse
import collections
import re

# cost classes: SIMD cycles per wave-instruction under load (tools/ubench.hip, waves/SIMD = 4, nominal 2.4 GHz)
DEFAULT_COST = {"full": 2.35, "half": 4.1, "trans": 8.2, "dp": 6.3}
FULL_RATE = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_lshrrev_b32", "v_lshlrev_b32", "v_xor_b32", "v_and_b32", "v_or_b32",
             "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_ashrrev_i32", "v_not_b32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32"}


def classify(line):
    """-> (unit, class, why) for one instruction line."""
    toks = line.replace(",", " ").split()
    op = toks[0]
    if op.startswith(("s_", "S_")):
        if op.startswith(("s_cbranch", "s_branch")):
            return "branch", "branch", op
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem", "smem", op
        if op.startswith("s_waitcnt"):
            return "wait", "wait", op
        return "salu", "salu", op
    if op.startswith(("ds_",)):
        return "lds", "lds", op
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", "vmem", op
    if not op.startswith("v_"):
        return "other", "other", op
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    srcs = toks[2:] if len(toks) > 2 else []
    sgpr_src = any(re.fullmatch(r"-?\|?s\d+\|?|-?\|?s\[\d+:\d+\]\|?|vcc|exec", t) for t in srcs)
    if base in TRANS:
        return "valu", "trans", base
    if base.endswith("_f64") or base.startswith("v_pk_"):
        return "valu", "dp", base
    if base in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        return "valu", "half", base + " (lane access)"
    # (measured, profiles/ubench_valu_r03.txt: a VOP3 encoding with |abs| / -neg modifiers keeps v_add / v_sub / v_mul at full
    # rate -- 2.5 cycles --, an SDWA encoding or an SGPR source does not -- 4.2)
    if base in FULL_RATE and not sgpr_src and not op.endswith(("_sdwa", "_dpp")):
        return "valu", "full", base + (" (VOP3 modifiers)" if op.endswith("_e64") else "")
    why = base + (" +sgpr operand" if sgpr_src and base in FULL_RATE else "") + (" (sdwa encoding)" if base in FULL_RATE and op.endswith(("_sdwa", "_dpp")) else "")
    return "valu", "half", why


def innermost_loop(lines):
    """(first, last) line index of the deepest loop's body: from its header label to its last back-edge branch."""
    headers = [(i, int(m.group(1))) for i, ln in enumerate(lines) for m in [re.search(r"Loop Header: Depth=(\d+)", ln)] if m]
    if not headers:
        raise SystemExit("no loop found")
    depth = max(d for _, d in headers)
    hi = [i for i, d in headers if d == depth][0]
    # the label is the closest preceding ".LBB" line
    li = hi
    while li > 0 and not lines[li].startswith(".LBB"):
        li -= 1
    label = lines[li].split(":")[0]
    last = max(i for i, ln in enumerate(lines) if re.search(r"s_cbranch\w*\s+%s\b|s_branch\s+%s\b" % (re.escape(label), re.escape(label)), ln))
    return li, last, label, depth


def report(title, body, cost):
    units = collections.Counter()
    classes = collections.Counter()
    detail = collections.Counter()
    for ln in body:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        unit, cls, why = classify(t)
        units[unit] += 1
        if unit == "valu":
            classes[cls] += 1
            detail[(cls, why)] += 1
    n_valu = sum(classes.values())
    cyc = sum(cost[c] * n for c, n in classes.items())
    print("== %s" % title)
    print("   instructions: " + ", ".join("%s %d" % kv for kv in sorted(units.items(), key=lambda kv: -kv[1])))
    if not n_valu:
        return
    for c in ("full", "half", "trans", "dp"):
        if classes[c]:
            print("   VALU %-5s %4d instr x %.2f cyc = %7.1f cyc  (%4.1f %% of the VALU cycles)" % (c, classes[c], cost[c], cost[c] * classes[c], 100.0 * cost[c] * classes[c] / cyc))
    print("   VALU total %4d instr, %.1f SIMD-cycles per wave; mean %.2f cycles per VALU instruction (2.00 at the 78.6 Tlane-op/s peak:"
          " this mix's roof is %.3f of it)" % (n_valu, cyc, cyc / n_valu, 2.0 * n_valu / cyc))
    print("   by opcode (class, count):")
    for (cls, why), n in sorted(detail.items(), key=lambda kv: (-cost[kv[0][0]] * kv[1], kv[0])):
        print("      %-5s %3d  %s" % (cls, n, why))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--cost", help="ubench output to take the class costs from (full: v_mul 2src, half: v_cmp, trans: v_sqrt_f32, dp: v_fma_f64)")
    ap.add_argument("--split", help="report the body before / after the first instruction mentioning this literal separately")
    args = ap.parse_args()
    cost = dict(DEFAULT_COST)
    if args.cost:
        rows = {}
        for ln in open(args.cost):
            m = re.match(r"(.+?)\s+waves/SIMD=4\s+[\d.]+ ms\s+nominal ([\d.]+) cyc/inst", ln)
            if m:
                rows[m.group(1).strip()] = float(m.group(2))
        for cls, key in (("full", "v_mul 2src"), ("half", "v_cmp"), ("trans", "v_sqrt_f32"), ("dp", "v_fma_f64")):
            if key in rows:
                cost[cls] = rows[key]
    lines = open(args.asm).read().splitlines()
    first, last, label, depth = innermost_loop(lines)
    body = lines[first:last + 1]
    print("%s: innermost loop %s (depth %d), lines %d..%d; class costs %s" % (args.asm, label, depth, first + 1, last + 1, cost))
    if args.split:
        cut = next((i for i, ln in enumerate(body) if args.split in ln), None)
        if cut is None:
            raise SystemExit("--split literal not found in the loop")
        report("before %s" % args.split, body[:cut], cost)
        report("from %s on" % args.split, body[cut:], cost)
    report("whole loop body", body, cost)


if __name__ == "__main__":
    main()
