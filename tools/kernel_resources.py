"""Static resource table of every gfx950 kernel in libta_hip.so: VGPRs / SGPRs / LDS / scratch / spills as the
compiler reports them in the code-object metadata, and the wave occupancy they allow (512 VGPRs per SIMD lane,
160 KiB LDS per CU, 8 waves per SIMD max on gfx950).  Cross-compiles; needs no GPU.
Usage: python tools/kernel_resources.py > profiles/r01/kernel_resources.txt"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "transferattack_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]


def demangle(name):
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    out = re.sub(r"^void ", "", out)
    return re.sub(r"\(.*", "", out)


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            stem = os.path.splitext(os.path.basename(src))[0]
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "--save-temps=obj", "-o",
                                                              os.path.join(tmp, stem + ".o")],
                           check=True, capture_output=True)
            asm = glob.glob(os.path.join(tmp, stem + "*gfx950.s"))
            if not asm:
                continue
            text = open(asm[0]).read()
            for block in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
                def field(key):
                    m = re.search(r"\.%s:\s+(\S+)" % key, block.group(0))
                    return m.group(1) if m else "?"
                vgpr, lds, wg = int(field("vgpr_count")), int(field("group_segment_fixed_size")), int(
                    field("max_flat_workgroup_size"))
                waves_vgpr = min(8, 512 // max(8 * ((vgpr + 7) // 8), 8))          # per SIMD, 8-register granules
                waves_per_wg = (wg + 63) // 64
                wgs_lds = (160 * 1024) // lds if lds else 99
                waves_lds = wgs_lds * waves_per_wg / 4.0                          # per SIMD (4 SIMDs per CU)
                rows.append((stem, demangle(field("name")), vgpr, int(field("sgpr_count")), lds,
                             int(field("private_segment_fixed_size")), int(field("vgpr_spill_count")), wg,
                             min(waves_vgpr, 8 if not lds else int(min(8, waves_lds)))))
    print("%-13s %-78s %5s %5s %7s %7s %6s %5s %s" % ("file", "kernel", "vgpr", "sgpr", "lds_B", "scratch", "spills", "wg",
                                                     "waves/SIMD (static LDS only)"))
    for r in rows:
        print("%-13s %-78s %5d %5d %7d %7d %6d %5d %d" % (r[0], r[1][:78], *r[2:]))


if __name__ == "__main__":
    main()
