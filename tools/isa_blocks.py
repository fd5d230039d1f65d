"""Per-basic-block instruction counts of a gfx950 kernel, from the assembly hipcc emits (cross-compiles, no GPU).

    python tools/isa_blocks.py dim.hip dim_fwd_lanes_kernelILi10      # file under csrc/, substring of the mangled name

Prints, per label, the number of VALU / SALU / LDS / global-memory instructions and LLVM's loop annotations, plus the
totals.  Multiplying the blocks on a wave's path by their trip counts gives the executed-instruction estimate the
issue model in profiles/r01/dim_isa_model.txt uses (a CU issues one wave64 VALU and one SALU instruction per cycle)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transferattack_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]


def kernel_asm(source, needle):
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(CSRC, source), "--save-temps=obj", "-o",
                                                          os.path.join(tmp, "k.o")], check=True, capture_output=True)
        text = open(glob.glob(os.path.join(tmp, "*gfx950.s"))[0]).read()
    found = re.findall(r"^(_Z\w*%s\w*):" % re.escape(needle), text, re.M)
    if not found:
        raise SystemExit("no kernel matching %r" % needle)
    name = found[0]
    body = re.search(r"^%s:.*?^\.Lfunc_end" % re.escape(name), text, re.S | re.M).group(0)
    return name, body.split("\n")


def classify(line):
    m = re.match(r"\s+([a-z_0-9]+)", line)
    if not m:
        return None
    op = m.group(1)
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


def main():
    source, needle = sys.argv[1], sys.argv[2]
    name, lines = kernel_asm(source, needle)
    blocks, current = [], dict(label="(entry)", note="", valu=0, salu=0, lds=0, vmem=0)
    for line in lines[1:]:
        label = re.match(r"^(\.LBB\w+):\s*(;.*)?$", line)
        if label:
            blocks.append(current)
            current = dict(label=label.group(1), note=(label.group(2) or "").strip("; "), valu=0, salu=0, lds=0, vmem=0)
            continue
        loop = re.match(r"^\s*;\s*(=>.*Loop Header.*|\s*in Loop:.*|\s*Parent Loop.*|\s*Child Loop.*)$", line)
        if loop and len(current["note"]) < 70:
            current["note"] += " " + loop.group(1).strip()
        kind = classify(line)
        if kind:
            current[kind] += 1
    blocks.append(current)
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(demangled)
    print("%-12s %5s %5s %4s %5s  %s" % ("block", "VALU", "SALU", "LDS", "VMEM", "LLVM note"))
    for b in blocks:
        if b["valu"] + b["salu"] + b["lds"] + b["vmem"]:
            print("%-12s %5d %5d %4d %5d  %s" % (b["label"], b["valu"], b["salu"], b["lds"], b["vmem"], b["note"][:90]))
    print("%-12s %5d %5d %4d %5d" % ("total", *(sum(b[k] for b in blocks) for k in ("valu", "salu", "lds", "vmem"))))


if __name__ == "__main__":
    main()
