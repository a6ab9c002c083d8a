#!/usr/bin/env python3
"""tools/isa_stats.py [file.s] [name-filter ...] — instruction mix of the step kernels from the gfx950 assembly.

Without a file it compiles the library's two translation units to assembly first (hipcc -S --cuda-device-only, ~1 min).  For every
kernel whose mangled name contains one of the filters (default: the default template instances of the multi-step
kernels) it prints registers / LDS / spills from the metadata and the instruction classes of (a) the whole kernel
and (b) its loops — the blocks the compiler's comments assign to each loop header — ordered by arithmetic content: the row
loops of the two sweep directions, in their general (start-up) and steady forms.
Used for the VALU-diet bookkeeping in DESIGN.md (v_cndmask / v_mov per loop body)."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["step4ILb1ELi0", "step4pILb1ELi0", "step3ILb1ELi0ELb1ELi1", "step3pILb1ELi0", "step2ILb1ELi2", "stepILi4ELb1ELi2", "multiILi32"]


def classify(op):
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith(("v_mov", "v_pk_mov", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane")):
        return "v_mov" if not op.endswith("_dpp") else "dpp"
    if "_dpp" in op:
        return "dpp"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq")):
        return "v_trans"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def kernels(text):
    """yields (name, [instruction lines]) for every function in the assembly"""
    cur, body = None, []
    for ln in text.split("\n"):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        if ln.startswith(".Lfunc_end"):
            yield cur, body
            cur, body = None, []
            continue
        body.append(ln)
    if cur:
        yield cur, body


def loops(body):
    """(all instructions, {loop header label: [instructions of the blocks the compiler marks as part of that loop]}) — from the
    block comments of the assembly ('=>This Inner Loop Header', 'in Loop: Header=BBn_m'): blocks of a loop need not be
    contiguous in the layout, so a label-to-backward-branch span would miss or mix them"""
    ins, per, cur = [], {}, None
    for ln in body:
        m = re.match(r"^\.(LBB\w+):\s*(;.*)?$", ln)
        if m:
            c = m.group(2) or ""
            if "Loop Header" in c:
                cur = m.group(1)
            else:
                h = re.search(r"in Loop: Header=(BB\w+)", c)
                cur = "L" + h.group(1) if h else None
            continue
        h = re.match(r"^; %bb\.\d+:\s*;\s*in Loop: Header=(BB\w+)", ln)
        if h:
            cur = "L" + h.group(1)
            continue
        if re.match(r"^; %bb\.\d+:", ln):
            cur = None
            continue
        s = ln.strip()
        if ln.startswith("\t") and s and not s.startswith((".", ";")):
            ins.append(s)
            if cur:
                per.setdefault(cur, []).append(s)
    return ins, per


def main():
    args = sys.argv[1:]
    path = args[0] if args and args[0].endswith(".s") else None
    filters = [a for a in args if not a.endswith(".s")] or DEFAULT
    if path is None:
        # the library's two translation units with the flags the Makefile gives them (the deep window kernels: max-ILP scheduling)
        text = ""
        for src, extra in (("lbm_hip.cpp", []), ("lbm_deep.cpp", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"])):
            path = os.path.join(tempfile.gettempdir(), src.replace(".cpp", "_gfx950.s"))
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I/opt/rocm/include", "-w"] + extra +
                           ["-S", "--cuda-device-only", "-o", path, os.path.join(ROOT, "opencl-lattice-boltzmann_amd", "csrc", src)], check=True)
            text += open(path).read() + "\n"
    else:
        text = open(path).read()
    meta = {}
    for b in text.split("  - .agpr_count")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", b).group(1)
        meta[nm] = tuple(int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1)) for k in
                         ("vgpr_count", "vgpr_spill_count", "group_segment_fixed_size", "sgpr_count"))
    for name, body in kernels(text):
        if not any(f in name for f in filters):
            continue
        ins, per = loops(body)
        v, sp, lds, sg = meta.get(name, (0, 0, 0, 0))
        print("%s\n  vgpr %d  spilled %d  lds %d B  sgpr %d" % (name, v, sp, lds, sg))
        # the kernel, then its loops by arithmetic content (the row loops of the two sweep directions, general and steady form)
        rows = [("kernel", ins)] + [("loop " + h, seq) for h, seq in sorted(per.items(), key=lambda kv: -sum(1 for x in kv[1] if x.startswith("v_pk")))[:int(os.environ.get("ISA_LOOPS","4"))]]
        for label, seq in rows:
            c = Counter(classify(x.split()[0]) for x in seq)
            valu = sum(n for k, n in c.items() if k.startswith("v_") or k == "dpp")
            print("  %-14s %5d instructions, VALU %5d: " % (label, len(seq), valu) +
                  "  ".join("%s %d" % (k, c[k]) for k in ("v_pk", "v_other", "v_trans", "v_cndmask", "v_mov", "dpp", "lds", "vmem", "salu", "s_waitcnt", "other") if c[k]))


if __name__ == "__main__":
    main()
