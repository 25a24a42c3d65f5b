#!/usr/bin/env python3
"""Instruction-class histogram of gfx950 kernels from hipcc's assembly (--save-temps `.s`).

    python tools/isa_hist.py <file.s> [--kernel SUBSTR] [--json]

Per kernel: totals, and for every loop (a backward branch to a label) the instruction
classes inside it. Used to back DESIGN.md section 4's "VALU instructions per (tile, Gaussian)"
and bench.py's `valu_issue` with the ISA instead of an estimate, and by
tools/ubench/run_valu_rate.sh to count the microbenchmark's loop bodies.
"""
from __future__ import annotations

import argparse
import json
import re
import sys
from collections import Counter
from pathlib import Path

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
CROSS = ("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "v_mov_b32_dpp", "v_swap")


def classify(op: str, line: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "valu_trans"
        if op.startswith("v_pk_"):
            return "valu_packed"
        if op.startswith(CROSS) or "dpp" in line or "row_" in line or "quad_perm" in line:
            return "valu_cross_lane"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "valu_cmp"
        if op.startswith("v_cndmask"):
            return "valu_cndmask"
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_nop") or op.startswith("s_sleep"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return "vmem_atomic"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "vmem"
    return "other"


def kernels(text: str):
    """Yield (name, body lines, trailer lines) for every function in the assembly; the trailer
    (after .Lfunc_end) holds the `; NumVgprs: ...` resource comments."""
    name, body, tail, in_body = None, [], [], False
    for ln in text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", ln)
        if m and not m.group(1).startswith(".L") and not in_body:
            if name is not None:
                yield name, body, tail
            name, body, tail, in_body = m.group(1), [], [], True
            continue
        if in_body:
            if ln.strip().startswith(".Lfunc_end"):
                in_body = False
                continue
            body.append(ln)
        elif name is not None:
            tail.append(ln)
    if name is not None:
        yield name, body, tail


def analyse(lines, tail=()):
    insts = []          # (index, op, line)
    labels = {}         # label -> instruction index
    for ln in lines:
        s = ln.split(";")[0].strip()
        if not s or s.startswith("."):
            m = re.match(r"^(\.L\w+):", s)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.L\w+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        op = s.split()[0]
        insts.append((op, s))
    total = Counter(classify(op, s) for op, s in insts)
    loops = []
    for i, (op, s) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                a = labels[tgt]
                loops.append({"label": tgt, "start": a, "end": i, "n_insts": i - a + 1,
                              "classes": dict(Counter(classify(o, t) for o, t in insts[a:i + 1]))})
    # innermost = loops that contain no other loop
    for lp in loops:
        lp["innermost"] = not any(o is not lp and lp["start"] <= o["start"] and o["end"] <= lp["end"]
                                  for o in loops)
        c = lp["classes"]
        lp["valu_total"] = sum(v for k, v in c.items() if k.startswith("valu"))
    meta = {}
    for ln in tail:
        m = re.search(r"; (NumVgprs|NumSgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumVgprs): (\d+)", ln)
        if m:
            meta[m.group(1)] = int(m.group(2))
    return {"n_insts": len(insts), "classes": dict(total),
            "valu_total": sum(v for k, v in total.items() if k.startswith("valu")),
            "loops": loops, "meta": meta}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default=None, help="only kernels whose (mangled) name contains this")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    text = Path(args.asm).read_text()
    out = {}
    for name, lines, tail in kernels(text):
        if args.kernel and args.kernel not in name:
            continue
        res = analyse(lines, tail)
        if res["n_insts"] < 8:
            continue
        out[name] = res
    if args.json:
        json.dump(out, sys.stdout, indent=1)
        print()
        return
    for name, r in out.items():
        print(f"== {name}: {r['n_insts']} instructions, VALU {r['valu_total']}, meta {r['meta']}")
        print("   ", {k: v for k, v in sorted(r["classes"].items())})
        for lp in r["loops"]:
            tag = "innermost" if lp["innermost"] else "outer"
            print(f"    loop {lp['label']:<12} [{lp['start']:5d}..{lp['end']:5d}] {tag:9s} "
                  f"{lp['n_insts']:4d} insts, VALU {lp['valu_total']:4d}: "
                  f"{ {k: v for k, v in sorted(lp['classes'].items())} }")


if __name__ == "__main__":
    main()
