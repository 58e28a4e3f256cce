#!/usr/bin/env python3
"""Register / LDS budget of every kernel of the HIP translation units, with the residency the hardware
really admits (MI355X_MICROARCH.md, "Residency": the SGPR file holds 800 registers per SIMD and a wave
takes ceil(sgpr/16)*16 + 16 of them -- a kernel with more than 80 SGPRs does not reach 8 waves per SIMD
whatever the occupancy the compiler prints).

usage: python tools/kres.py [k_tokens k_match ...]      (default: every .hip under lz77_amd/csrc)
"""
import glob, os, re, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lz77_amd", "csrc")


def field(block, key):
    m = re.search(re.escape(key) + r": (\d+)", block)
    return int(m.group(1)) if m else 0


def main():
    names = sys.argv[1:] or sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "*.hip")))
    for f in names:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, f + ".hip"), "-o", "/dev/null"]
                           + os.environ.get("KRES_FLAGS", "").split(),
                           capture_output=True, text=True)
        blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
        print("==", f)
        for b in blocks:
            name = b.split()[0]
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem)
            sg, vg, ag = field(b, "TotalSGPRs"), field(b, "VGPRs"), field(b, "AGPRs")
            lds, scr, occ = field(b, "LDS Size [bytes/block]"), field(b, "ScratchSize [bytes/lane]"), field(b, "Occupancy [waves/SIMD]")
            by_sgpr = min(8, 800 // (((sg + 15) // 16) * 16 + 16))
            by_vgpr = min(8, 512 // max(8, ((vg + ag + 7) // 8) * 8))
            flag = "  <-- SGPR-limited" if by_sgpr < min(occ, by_vgpr) else ""
            print(f"  {dem[:46]:46s} sgpr {sg:4d} vgpr {vg:4d} scratch {scr:4d} lds(static) {lds:6d} occ: compiler {occ} sgpr {by_sgpr} vgpr {by_vgpr}{flag}")


if __name__ == "__main__":
    main()
