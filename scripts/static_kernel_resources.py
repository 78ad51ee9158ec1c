"""Static register / LDS / scratch usage of every kernel, from the compiler (no GPU needed):
    python scripts/static_kernel_resources.py > profiles/roundN/static_kernel_resources.md
Compiles the device side of each .hip file with the product's flags and -Rpass-analysis=kernel-resource-usage."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "warpx_amd", "csrc")
FILES = [("fields", ["-ffp-contract=off"]), ("deposit_tile", []), ("gather_tile", []), ("particles", [])]


def main():
    rows = []
    for name, extra in FILES:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", *extra, "-c",
               "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
               os.path.join(CSRC, name + ".hip")]
        txt = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
        for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
            sym = b.split()[0]

            def g(k):
                m = re.search(k + r": (\d+)", b)
                return int(m.group(1)) if m else None
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("wxa::", "")
            rows.append((name, dem, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                         g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    seen, out = set(), []
    for r in rows:
        if r[:2] not in seen:
            seen.add(r[:2])
            out.append(r)
    w = sys.stdout.write
    w("# Static resource usage of every kernel\n\n`hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage` "
      "(the product's flags; scripts/static_kernel_resources.py), taken in the build\ncontainer: compiler output, not a "
      "measurement.  Scratch = spilled bytes per lane, occupancy = waves per SIMD the register and LDS budgets allow.\n"
      "Template arguments: gather `<order, galerkin, pusher, move, part>`, deposit `<order, algo, ...>`.\n\n"
      "| file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | waves/SIMD | LDS B/workgroup |\n|---|---|---|---|---|---|---|---|\n")
    for r in out:
        w("| " + " | ".join(str(x) for x in r) + " |\n")


if __name__ == "__main__":
    main()
