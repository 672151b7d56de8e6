"""Register / spill / scratch / LDS report of the gfx950 kernels in a BUILT object (what really ships), no recompilation:
    python tools/kernel_regs_obj.py [gemini-seal_amd/build/ntt.hip.o] [--filter substr] [--spills-only]
Unbundles the gfx950 code object from .hip_fatbin and reads its AMDGPU metadata notes (llvm-readelf --notes).
tools/kernel_regs.py does the same from a hipcc -S assembly file."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
        "agpr_count")


def kernels(path):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "co")
        # (with an output file: without one objcopy rewrites its input in place, and make then relinks the library)
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(tmp, "copy.o")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    recs = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        rec = {}
        for k in KEYS:
            m = re.search(r"\.%s:\s*(\S+)" % k, blk)
            if m:
                rec[k] = m.group(1)
        if "name" in rec and "vgpr_count" in rec:
            recs.append(rec)
    return recs


def main():
    args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith("--") and sys.argv[i] != "--filter"]
    path = args[0] if args else os.path.join(os.path.dirname(__file__), "..", "gemini-seal_amd", "build", "ntt.hip.o")
    flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
    recs = kernels(path)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in recs), capture_output=True,
                           text=True).stdout.splitlines()
    rows = []
    for r, n in zip(recs, names):
        n = re.sub(r"^void sealhip::\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*$", "", n)
        spill = int(r.get("vgpr_spill_count", 0))
        if (flt and flt not in n) or ("--spills-only" in sys.argv and spill == 0):
            continue
        rows.append((n, int(r["vgpr_count"]), int(r["sgpr_count"]), spill, int(r.get("private_segment_fixed_size", 0)),
                     int(r.get("group_segment_fixed_size", 0))))
    print(f"{'kernel':72s} vgpr sgpr spill scratchB  ldsB")
    for row in sorted(rows):
        print(f"{row[0][:72]:72s} {row[1]:4d} {row[2]:4d} {row[3]:5d} {row[4]:8d} {row[5]:5d}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r[3])} with spills")


if __name__ == "__main__":
    main()
