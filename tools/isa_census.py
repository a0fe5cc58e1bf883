#!/usr/bin/env python3
"""Per-basic-block instruction census of a kernel from hipcc's gfx950 assembly (CPU only: hipcc cross-compiles).

  python tools/isa_census.py vg_attention.hip attn2_fwd_kernel [--min-mfma 8] [-DFLAG ...]

For every basic block of the kernel that holds at least --min-mfma MFMAs (the tile bodies of the hot loops) it prints the
MFMA count and the vector / scalar / LDS / memory instructions next to it, by class -- the "VALU per MFMA" figure of the
SQ counters broken down by what the instructions are."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vae-gslm_amd", "csrc")

CLASSES = [
    ("mfma", r"v_mfma"),
    ("exp/log/rcp", r"v_(exp|log|rcp|rsq|sqrt)_"),
    ("max/min", r"v_(max|min|max3|min3|med3)_"),
    ("pk_max/min", r"v_pk_(max|min)"),
    ("add/sub f32", r"v_(add|sub|subrev)_f32"),
    ("mul f32", r"v_mul_f32"),
    ("fma/mac f32", r"v_(fma|fmac|mac|mad)_f32"),
    ("pk f32", r"v_pk_(add|mul|fma)_f32"),
    ("cvt_pk bf16", r"v_cvt_pk_bf16"),
    ("cvt other", r"v_cvt_"),
    ("dot2", r"v_dot2"),
    ("permlane/dpp", r"v_permlane|_dpp|v_readlane|v_readfirstlane|v_writelane|ds_bpermute|ds_swizzle"),
    ("accvgpr mov", r"v_accvgpr"),
    ("v_mov", r"v_mov_b"),
    ("cmp/cndmask", r"v_cmp|v_cndmask"),
    ("perm/bfi/and/or/shift", r"v_(perm|bfi|and|or|xor|lshl|lshr|ashr|alignbit|bfe|lshlrev|lshrrev|ashrrev|and_or|or3|lshl_or|lshl_add)_"),
    ("int add/mul", r"v_(add|sub|mul|mad|add3)_(u|i|nc_u|co_u|lo_u|hi_u)|v_mad_u|v_mad_i|v_add_u|v_sub_u|v_mul_lo|v_mul_hi|v_mul_u|v_mul_i"),
    ("ds_read", r"ds_read|ds_load"),
    ("ds_write", r"ds_write|ds_store"),
    ("buffer/global load", r"(buffer|global|flat)_load"),
    ("buffer/global store", r"(buffer|global|flat)_(store|atomic)"),
    ("s_waitcnt", r"s_waitcnt"),
    ("s_barrier", r"s_barrier"),
    ("s_nop", r"s_nop"),
    ("s_load", r"s_load|s_buffer_load"),
    ("branch", r"s_cbranch|s_branch"),
    ("salu", r"s_"),
    ("other valu", r"v_"),
]
VALU = {"exp/log/rcp", "max/min", "pk_max/min", "add/sub f32", "mul f32", "fma/mac f32", "pk f32", "cvt_pk bf16",
        "cvt other", "dot2", "permlane/dpp", "accvgpr mov", "v_mov", "cmp/cndmask", "perm/bfi/and/or/shift", "int add/mul",
        "other valu"}


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return "other"


def kernel_text(asm, kernel):
    lines = asm.splitlines()
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w*" + re.escape(kernel) + r"\w*):", l)
        if m:
            start = i
            name = m.group(1)
            break
    if start is None:
        raise SystemExit(f"kernel {kernel} not found")
    out = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.strip().startswith(".end_amdhsa_kernel"):
            break
        out.append(l)
    return name, out


def main():
    args = sys.argv[1:]
    if len(args) < 2:
        raise SystemExit(__doc__)
    src, kernel = args[0], args[1]
    min_mfma = 8
    flags = []
    i = 2
    while i < len(args):
        if args[i] == "--min-mfma":
            min_mfma = int(args[i + 1]); i += 2
        else:
            flags.append(args[i]); i += 1
    out_s = f"/tmp/isa_census_{os.getpid()}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                    "-w", *flags, "-o", out_s, os.path.join(CSRC, src)], check=True, stderr=subprocess.DEVNULL)
    asm = open(out_s).read()
    os.unlink(out_s)
    name, body = kernel_text(asm, kernel)
    blocks, cur, label, nsplit = [], collections.Counter(), "entry", 0
    ops = collections.defaultdict(collections.Counter)
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                blocks.append((label, cur)); cur = collections.Counter(); label = s.rstrip(":")
            continue
        op = s.split()[0]
        c = classify(op)
        cur[c] += 1
        ops[label][op] += 1
        if op.startswith("s_cbranch") or op == "s_branch":      # a basic block ends at a branch as well as at a label
            blocks.append((label, cur)); cur = collections.Counter(); nsplit += 1; label = f"{label.split('+')[0]}+{nsplit}"
    blocks.append((label, cur))
    print(f"kernel {name}: {len(blocks)} basic blocks, {sum(sum(c.values()) for _, c in blocks)} instructions")
    for label, c in blocks:
        if c["mfma"] < min_mfma:
            continue
        valu = sum(v for k, v in c.items() if k in VALU)
        print(f"\n== block {label}: {c['mfma']} MFMA, {valu} VALU ({valu / max(c['mfma'], 1):.1f} per MFMA), "
              f"{sum(c.values())} instructions")
        for k, _ in CLASSES:
            if c[k]:
                print(f"   {k:24s} {c[k]:5d}")
        detail = sorted(ops[label].items(), key=lambda kv: -kv[1])
        print("   by opcode: " + ", ".join(f"{o} {n}" for o, n in detail if not o.startswith("v_mfma"))[:1500])


if __name__ == "__main__":
    main()
