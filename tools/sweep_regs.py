"""Register count of bcd_sweep_tiled_kernel<K, KC, false> per (K, KC): compiles csrc/bcd_sweep_inst.cpp for one K with
-DFDX_KC_OVERRIDE=KC (device code only, no GPU needed) and reads .vgpr_count / spills from the assembly.  168 registers are
the limit for three waves per SIMD; sweep_chunk() in bcd_sweep_inst.cpp is tuned from this table.

usage: python tools/sweep_regs.py [--obj] K_LO K_HI KC [KC ...]"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "flashdeconv_amd", "csrc")


def regs(K, KC):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I.", "-I../../include", "-mllvm", "-pragma-unroll-threshold=4000000",
               "-mllvm", "-unroll-threshold=4000000", "-DFDX_PART=9", f"-DFDX_K_LO={K}", f"-DFDX_K_HI={K}", f"-DFDX_KC_OVERRIDE={KC}",
               "--cuda-device-only", "-S", "bcd_sweep_inst.cpp", "-o", out]
        subprocess.run(cmd, cwd=SRC, check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    m = re.search(r"\.name:\s+_ZN3fdx22bcd_sweep_tiled_kernelILi%dELi%dELb%dE.*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)" % (K, min(K, KC), OBJ), txt, re.S)
    return K, KC, int(m.group(1)), int(m.group(2))


OBJ = 0

if __name__ == "__main__":
    if "--obj" in sys.argv:                      # the objective variant of the kernel
        OBJ = 1
        sys.argv.remove("--obj")
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    kcs = [int(a) for a in sys.argv[3:]]
    jobs = [(K, KC) for K in range(lo, hi + 1) for KC in kcs]
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda a: regs(*a), jobs))
    print("K  " + "  ".join(f"KC={kc:<3d}" for kc in kcs))
    for K in range(lo, hi + 1):
        print(f"{K:<3d}" + "  ".join(f"{r[2]:4d}/{r[3]:<4d}" for r in res if r[0] == K))
