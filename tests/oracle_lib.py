"""Loads oracle/liboracle.so (building it with the committed Makefile when absent).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this."""
import os
import subprocess

from warpx_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_oracle = None


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("pic_oracle.cpp", "pic_kernels.hpp")]
    srcs.append(os.path.join(ROOT, "include", "warpx_amd.h"))
    stale = (not os.path.exists(ORACLE_LIB)) or any(
        os.path.getmtime(s) > os.path.getmtime(ORACLE_LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return ORACLE_LIB


def available_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota.  The GPU boxes show 256
    logical CPUs under a quota of 16 (cpu.max = "1600000 100000"): an OpenMP team of 256 spinning threads on 16 CPUs'
    worth of time made every parallel region of the oracle 50-100x slower than a team of 16 (profiles/round3/README.md)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:   # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:   # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def load_oracle():
    """The OpenMP team is sized to available_cpus() unless OMP_NUM_THREADS says otherwise (one libgomp per process: the
    setting also covers tests/host_cpu, which links the same kernels)."""
    global _oracle
    if _oracle is None:
        build_oracle()
        _oracle = _capi.CLib(ORACLE_LIB, "orc_", _capi._ORACLE_SIGS)
        if "OMP_NUM_THREADS" not in os.environ:
            _oracle._set_num_threads(available_cpus())
    return _oracle


HOST_CPU_DIR = os.path.join(ROOT, "tests", "host_cpu")
HOST_CPU_LIB = os.path.join(HOST_CPU_DIR, "libhost_cpu.so")
_host_cpu = None


def load_host_cpu():
    """The product's C++ host layer compiled against the oracle kernels (CPU, tests only):
    `hst_sim_*` has the step-level API of include/warpx_amd.h with host pointers."""
    global _host_cpu
    if _host_cpu is None:
        subprocess.check_call(["make", "-C", HOST_CPU_DIR, "libhost_cpu.so"], stdout=subprocess.DEVNULL)
        _host_cpu = _capi.CLib(HOST_CPU_LIB, "hst_", _capi._INPUTS_SIGS, kernels=False)
        load_oracle()   # sizes the OpenMP team, see available_cpus()
    return _host_cpu


HIPCPU_DIR = os.path.join(ROOT, "tests", "hipcpu")
HIPCPU_LIB = os.path.join(HIPCPU_DIR, "libwarpx_amd_hipcpu.so")
_hip_on_cpu = None


def build_hip_on_cpu():
    """Both builds of tests/hipcpu (plain and FMA-contracting), where the host clang of the ROCm install exists."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        return
    for extra in ([], ["FMA=1"]):
        subprocess.check_call(["make", "-C", HIPCPU_DIR, "-j8"] + extra, stdout=subprocess.DEVNULL)


def load_hip_on_cpu():
    """The product's .hip sources, unmodified, compiled against the HIP-on-CPU execution model of
    tests/hipcpu (tests only; a logic check of the kernels and launches where there is no GPU)."""
    global _hip_on_cpu
    if _hip_on_cpu is None:
        # WXA_HIP_ON_CPU_FMA=1: the build that contracts a*b+c in the kernels like the gfx950 compiler does
        fma = os.environ.get("WXA_HIP_ON_CPU_FMA") == "1"
        lib = HIPCPU_LIB.replace(".so", "_fma.so") if fma else HIPCPU_LIB
        if os.environ.get("WXA_HIPCPU_LIB"):   # a hand-made build, e.g. with -DWXA_DEV_VARIANTS (see tests/hipcpu/Makefile)
            lib = os.environ["WXA_HIPCPU_LIB"]
        else:
            subprocess.check_call(["make", "-C", HIPCPU_DIR, "-j8"] + (["FMA=1"] if fma else []), stdout=subprocess.DEVNULL)
        _hip_on_cpu = _capi.CLib(lib, "wxa_", {**_capi._PRODUCT_SIGS, **_capi._INPUTS_SIGS}, memory="cpu:0")
    return _hip_on_cpu
