"""Static instruction breakdown of the megakernel's phases from its ISA listing (profiles/r02_megakernel_isa.s, made with
hipcc -S --cuda-device-only of tracy_hip.hip, function k_megakernel<false, true>).
    python tools/isa_breakdown.py profiles/r02_megakernel_isa.s            # the kernel round 2 started from
    python tools/isa_breakdown.py profiles/r02_final_megakernel_isa.s      # the kernel round 2 ends with
Phases are cut at the basic-block labels that start them (found by reading the listing: the first s_setprio 3 opens the
traversal loop, the three blocks holding twelve v_fma_mix_f32 are the unrolled inner steps, ...; see PHASES). VALU instructions are
split into 'algorithmic' (the arithmetic SURVEY 8d counts: fma / mul / add / sub / min / max / cmp on floats, rcp, sqrt,
sin, cos, cvt of the RNG) and 'overhead' (moves, selects, permutes, integer address and bookkeeping arithmetic)."""
import re
import sys

ALGO = re.compile(r"^v_(fma|fmac|mul|add|sub|min|max|min3|max3|med3|cmp_\w*_f32|rcp|rsq|sqrt|sin|cos|div_|pk_fma|pk_mul|pk_add|fract|floor|trunc|cvt_f32|mad|ldexp|frexp|exp|log)\w*f32|^v_fma_mix")


def classify(op):
    if op.startswith("v_"):
        return "valu_algo" if ALGO.match(op) else "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path):
    lines = open(path).read().split("\n")
    label_at, instr = {}, []          # label -> line index; (line index, opcode)
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            label_at[m.group(1)] = i
        m = re.match(r"^\t([a-z][a-z0-9_]+)", ln)
        if m and not ln.startswith("\t."):
            instr.append((i, m.group(1)))
    return lines, instr, label_at


# Phase boundaries, read off each listing's block structure (labels differ between compiles, so each archived listing
# has its own table): (phase, first label, label that ends it [+ line offset]).
PHASES = {
    "r02_megakernel_isa.s": [           # round-2 starting point (round 1's kernel)
        ("kernel prologue, tile fetch and accumulator clear", 0, "17"),
        ("shade: attribute interpolation, textures, Scatter, sky, roulette, LDS accumulate", "18", "83"),
        ("refill: ballot/mbcnt item hand-out + generate (RNG, camera ray, trav_begin)", "83", "91"),
        ("traversal phase entry (nP census, setprio)", "91", "93"),
        ("vote: census of inner / leaf lanes, exit test, majority rule (per iteration)", "93", ("103", 3)),
        ("triangle step: record fetch, Moller-Trumbore, 1/det, candidate test, leaf advance / pop", ("103", 3), "122"),
        ("inner step 1 of 3 (record fetch, 6 v_perm, 12 v_fma_mix, slab min/max, near/far, push/pop)", "122", "125"),
        ("inner step 2 of 3", "125", "128"),
        ("inner step 3 of 3", "128", "131"),
        ("loop back-edge blocks", "131", "134"),
        ("accept_candidate: cross-mesh tie rule (rare branch of the triangle step, IEEE 1/d)", "134", "141"),
        ("traversal phase exit, tile-done test", "141", "144"),
        ("tile flush (LDS copies -> accum), tile cost, epilogue counters", "144", None),
    ],
    "r02_final_megakernel_isa.s": [     # round-2 final kernel (trimmed vote, bitwise near/far swap, register diet, 96 VGPRs)
        ("kernel prologue, tile fetch and accumulator clear", 0, "17"),
        ("shade: attribute interpolation, textures, Scatter, sky, roulette, LDS accumulate", "18", "81"),
        ("refill: ballot/mbcnt item hand-out + generate (RNG, camera ray, trav_begin)", "81", "89"),
        ("traversal phase entry (nP census, wait limit, setprio)", "89", "91"),
        ("loop back-edge blocks (exec restores, exit branch)", "91", "94"),
        ("vote: census of inner / leaf lanes, exit test, majority rule (per iteration)", "94", ("94", 22)),
        ("inner-step dispatch (exec masks for the three unrolled steps)", ("94", 22), "99"),
        ("inner step 3 of 3 (record fetch, 6 v_perm, 12 v_fma_mix, slab min/max, near/far, push/pop)", "99", "100"),
        ("step join", "100", ("101", 3)),
        ("triangle step: record fetch, Moller-Trumbore, 1/det, candidate test, leaf advance / pop", ("101", 3), "119"),
        ("inner step 1 of 3", "119", "120"),
        ("inner step 2 of 3", "120", "121"),
        ("accept_candidate: cross-mesh tie rule (rare branch of the triangle step, IEEE 1/d)", "121", "128"),
        ("traversal phase exit, tile-done test", "128", "131"),
        ("tile flush (LDS copies -> accum), tile cost, epilogue counters", "131", None),
    ],
}


def main(path):
    import os
    lines, instr, L = parse(path)
    prefix = next(iter(L)).rsplit("_", 1)[0] + "_"          # ".LBB15_"

    def at(b):
        if b is None:
            return len(lines)
        if b == 0:
            return 0
        if isinstance(b, tuple):
            return L[prefix + b[0]] + b[1]
        return L[prefix + b]

    keys = ["valu_algo", "valu_other", "salu", "branch", "wait_nop", "lds", "vmem", "smem"]
    print("| phase | lines | VALU algorithmic | VALU overhead | SALU | branches | s_waitcnt / s_nop | LDS | VMEM | SMEM | total |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    tot = {k: 0 for k in keys}
    for name, lo, hi in PHASES[os.path.basename(path)]:
        lo, hi = at(lo), at(hi)
        c = {}
        for i, op in instr:
            if lo <= i < hi:
                c[classify(op)] = c.get(classify(op), 0) + 1
        row = [c.get(k, 0) for k in keys]
        for k, v in zip(keys, row):
            tot[k] += v
        print(f"| {name} | {lo + 1}-{hi} | " + " | ".join(str(v) for v in row) + f" | {sum(row)} |")
    print("| **whole kernel** | | " + " | ".join(str(tot[k]) for k in keys) + f" | {sum(tot.values())} |")


if __name__ == "__main__":
    main(sys.argv[1])
