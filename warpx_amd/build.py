"""Builds libwarpx_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m warpx_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libwarpx_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wall", "-Wno-unused-function", "-Wno-unused-result"]

if os.environ.get("WXA_DEPOSIT_PROFILE") == "1":   # phase clocks in the deposition tile kernel
    COMMON += ["-DWXA_DEPOSIT_PROFILE", "-DWXA_GATHER_PROFILE"]   # ... and in the gather tile kernel
COMMON += os.environ.get("WXA_EXTRA_DEFS", "").split()   # experiment switches (scripts/microbench)
if os.environ.get("WXA_LIB_OUT"):   # an experiment build next to the product: its own objects
    LIB = os.path.abspath(os.environ["WXA_LIB_OUT"])
    OBJ = os.path.join(CSRC, "_obj_" + os.path.basename(LIB).replace(".so", ""))

# (source, extra flags).  The field kernels keep the reference's operation order and
# are compiled without FMA contraction so that they are bit-identical to the CPU path.
SOURCES = [
    ("runtime.hip", []),
    ("fields.hip", ["-ffp-contract=off"]),
    ("particles.hip", []),
    ("deposit_tile.hip", []),
    ("gather_tile.hip", []),
    ("host/warpx_host.hip", []),
    ("host/btd_kernels.hip", []),
    ("host/reduced_kernels.hip", []),
    ("rccl_comm.hip", []),
]


def _deps(src):
    d = [src]
    for root, _, files in os.walk(CSRC):
        if root.endswith("_obj"):
            continue
        for f in files:
            if f.endswith((".hpp", ".h", ".H")):
                d.append(os.path.join(root, f))
    d.append(os.path.join(os.path.dirname(HERE), "include", "warpx_amd.h"))
    return d


def _compile(item, force):
    name, extra = item
    src = os.path.join(CSRC, name)
    obj = os.path.join(OBJ, name.replace("/", "_") + ".o")
    if not force and os.path.exists(obj) and all(
            os.path.getmtime(d) <= os.path.getmtime(obj) for d in _deps(src)):
        return obj, False
    cmd = [HIPCC] + COMMON + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda it: _compile(it, force), SOURCES))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[warpx_amd.build] linked {LIB}")
    elif verbose:
        print(f"[warpx_amd.build] {LIB} is up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
