"""Loader of the product library rl4mm_amd/csrc/liblobsim.so (HIP kernels + C-ABI of include/lobsim.h).

There is no CPU fallback: if the library cannot be built or loaded this raises, and ``lobsim_create`` itself
fails when no GPU is visible.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

from . import _capi

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "liblobsim.so"
SOURCES = [CSRC / "lobsim_kernels.hip", CSRC / "lobsim_host.inc", CSRC / "lobsim_layout.h", CSRC.parent.parent / "include" / "lobsim.h"]
# -amdgpu-atomic-optimizer-strategy=None: the LDS adds in the kernels are issued by one lane on purpose; the optimizer's
# wave-reduction prologue (mbcnt / exec juggling, ~12 instructions per add) only costs scalar-unit time there.
# -amdgpu-sched-strategy=max-ilp: the step kernel is a chain of dependent scalar / LDS / vector work per wave; the scheduler
# tuned for instruction-level parallelism instead of occupancy measured +1.5 % per step (-1 % on the multi-step launch).
# -disable-machine-licm (round 6): the machine-level loop-invariant code motion hoists per-lane address arithmetic out of the kernels' loops - out
# of the multi-step launch's step loop above all - and the hoisted values then ride through the whole message replay in vector registers the
# allocator spills there.  Without it: lob_rollout_kernel<false, 64, 256, true> 91 -> 0 spilled registers (224 -> 0 B of scratch), the multi-step
# launch 21.1 -> 16.6 us per 4096-env step (194 -> 246 M env-steps/s); the reset kernels 57 -> 8, the forward kernel 127 -> 80 registers; the
# per-step kernel 96 -> 86 registers at the same speed (28.0 us, tools/ab_libs.py, five interleaved rounds; profiles/r6/README.md).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-disable-machine-licm",
               "-Wno-pass-failed"]  # (the 128-level kernels are LDS-limited below their register budget's occupancy: expected)
# The kernel file is compiled as N_PARTS translation units in parallel (-DLOBSIM_PART=k: 0 = the C-ABI host side, 1 .. 4 = the engine
# kernels of one (on-chip ladder width, agent table) combination) and linked into one library; as one unit the 56 kernels take minutes.
N_PARTS = 5

_binding = None


def _stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(s.exists() and s.stat().st_mtime > t for s in SOURCES)


def compile_library(out: Path, extra_flags=(), source: Path = None, verbose: bool = False, jobs: int = None) -> None:
    """hipcc --offload-arch=gfx950: the parts of lobsim_kernels.hip -> objects (in parallel) -> one shared library at `out`."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc was not found: cannot build the HIP library")
    src = (Path(source) if source else SOURCES[0]).resolve()
    out = Path(out).resolve()
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    with tempfile.TemporaryDirectory(prefix="lobsim_build_") as td:
        objs = [Path(td) / f"part{k}.o" for k in range(N_PARTS)]

        def one(k):
            cmd = [hipcc, *compile_flags, f"-DLOBSIM_PART={k}", "-c", "-o", str(objs[k]), str(src)]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            subprocess.check_call(cmd, cwd=str(src.parent))

        with ThreadPoolExecutor(max_workers=jobs or min(N_PARTS, os.cpu_count() or 1)) as ex:
            list(ex.map(one, range(N_PARTS)))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", str(out), *map(str, objs)], cwd=str(src.parent))


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> liblobsim.so (in-tree, so it travels with the repo snapshot).  Safe when several
    ranks import at once: the build is serialised by a file lock and the library is replaced atomically."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    import tempfile

    with open(CSRC / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _stale():  # another process may have built it while we waited
                fd, tmp = tempfile.mkstemp(suffix=".so", dir=str(CSRC))
                os.close(fd)
                try:
                    compile_library(Path(tmp), verbose=verbose)
                    os.replace(tmp, LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.unlink(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


BOUNDS_LIB_PATH = CSRC / "liblobsim_bounds.so"
BOUNDS_FLAGS = ["-DLOBSIM_BOUNDS", "-DLOBSIM_REDZONE"]


def build_bounds_library(force: bool = False) -> Path:
    """The diagnostic build of the same sources: every computed index held against its array (LOBSIM_ST_INTERNAL + the line of the check),
    sentinel zones around the per-env arrays (lobsim_debug_check_redzones).  GPU AddressSanitizer is not available on the pool this was
    developed on; this is what stands in for it.  In-tree like the product library (it travels with the repo snapshot); only the tests
    and the fuzzers load it - explicitly, by path."""
    if not force and BOUNDS_LIB_PATH.exists() and all((not s.exists()) or s.stat().st_mtime <= BOUNDS_LIB_PATH.stat().st_mtime for s in SOURCES):
        return BOUNDS_LIB_PATH
    import fcntl
    import tempfile

    with open(CSRC / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            fd, tmp = tempfile.mkstemp(suffix=".so", dir=str(CSRC))
            os.close(fd)
            try:
                compile_library(Path(tmp), BOUNDS_FLAGS)
                os.replace(tmp, BOUNDS_LIB_PATH)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return BOUNDS_LIB_PATH


def bounds_binding() -> _capi.Binding:
    _one_hip_runtime()
    return _capi.Binding(build_bounds_library(), prefix="lobsim_")


def _one_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so (SONAME libamdhip64.so.7, the one liblobsim.so
    names).  With torch imported FIRST the loader resolves liblobsim's dependency to that copy: one HIP runtime in the process.
    The other way round the process ends up with two runtimes, and the one that initialises second finds no GPU ("No HIP GPUs
    are available" from the first torch.cuda call after a handle was created).  torch is the device-memory / stream plumbing of
    this package anyway (env.py's tensor path, bench.py), so it is imported here when it is installed."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def binding() -> _capi.Binding:
    global _binding
    if _binding is None:
        _one_hip_runtime()
        alt = os.environ.get("LOBSIM_LIB")  # diagnostics: a prebuilt variant of the same library (tools/build_variant.sh)
        if alt:
            import sys

            print(f"rl4mm_amd: LOBSIM_LIB is set - loading {alt} instead of {LIB_PATH} (no staleness check)", file=sys.stderr)
            _binding = _capi.Binding(alt, prefix="lobsim_")
            return _binding
        if _stale():
            build()
        _binding = _capi.Binding(LIB_PATH, prefix="lobsim_")
    return _binding


def new_sim(cfg) -> _capi.Sim:
    """lobsim_create on the HIP library; raises LobsimError when no GPU is available."""
    return _capi.Sim(binding(), cfg)
