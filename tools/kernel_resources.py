"""Developer probe: registers / LDS / spills of every kernel in the built library (llvm-readelf notes of the gfx950 code objects).
    python tools/kernel_resources.py [filter substring ...]"""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conan_amd", "libconan_hip.so")
with tempfile.TemporaryDirectory() as d:
    lib = shutil.copy(LIB, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True, cwd=d)
    rows = []
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip() or name.group(1)
            rows.append((dem.split("(")[0], int(blk.split()[0]), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
flt = sys.argv[1:]
print("%-78s %5s %5s %5s %7s %7s %6s %6s" % ("kernel", "agpr", "vgpr", "sgpr", "lds", "scratch", "vspill", "sspill"))
for r in sorted(rows):
    if not flt or any(x in r[0] for x in flt):
        print("%-78s %5d %5d %5d %7d %7d %6d %6d" % ((r[0][-78:],) + r[1:]))
