"""Build recipe for libmmdfn_hip.so (hipcc, gfx950 only, in-tree output).

``python -m mm_dfn_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the resulting shared object is git-ignored but
travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libmmdfn_hip.so")
# the same sources with -DMMDFN_TUNING: ablation / tiling overrides read from the environment (tools/ only; the
# production library above has none of them compiled in)
TUNING_LIBPATH = os.path.join(LIBDIR, "libmmdfn_hip_tuning.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
ARCH = "gfx950"
# kernels whose vector-memory requests are asm statements with hand-counted waits: a register spill between a request and its wait
# would store a register the load has not written yet.  (source file, substring of the mangled kernel name) -> the build fails
# unless hipcc reports ScratchSize 0 for every such kernel.
NO_SPILL = {}        # (round 6's hand-counted form of linear_planes.hip measured equal to hipcc's schedule and was removed)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files.append(os.path.join(INCLUDE, "mmdfn_hip.h"))
    for f in files:
        h.update(f.encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmmdfn_hip.so cannot be built")
    return exe


def build(force=False, verbose=True, tuning=False):
    """Compile every csrc/*.hip for gfx950 into lib/libmmdfn_hip.so (serialised across processes);
    ``tuning=True`` builds lib/libmmdfn_hip_tuning.so (-DMMDFN_TUNING) instead."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)          # one builder at a time (data-parallel ranks share the tree)
        try:
            return _build_locked(force, verbose, tuning)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _paths(tuning):
    lib = TUNING_LIBPATH if tuning else LIBPATH
    return lib, lib[:-3] + ".sha256", ("tuning_" if tuning else "")


def _build_locked(force, verbose, tuning):
    lib, stamp, prefix = _paths(tuning)
    digest = _digest()
    if not force and os.path.exists(lib) and os.path.exists(stamp):
        with open(stamp) as fh:
            if fh.read().strip() == digest:
                return lib
    objs = []
    jobs = []
    for src in sources():
        obj = os.path.join(LIBDIR, prefix + os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-c", src, "-o", obj]
        if tuning:
            cmd.insert(1, "-DMMDFN_TUNING")
        if verbose:
            print(" ".join(cmd), flush=True)
        watch = NO_SPILL.get(os.path.basename(src))
        if watch:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            jobs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True), watch))
        else:
            jobs.append((cmd, subprocess.Popen(cmd), None))  # the translation units are independent: compile in parallel
        objs.append(obj)
    for cmd, proc, watch in jobs:
        err = proc.communicate()[1] if watch else None
        if proc.wait() != 0:
            if err:
                sys.stderr.write(err)
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        if watch:
            _check_no_spill(err, watch, cmd)
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(digest)
    return lib


def _check_no_spill(remarks, watch, cmd):
    name = None
    seen = 0
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize [bytes/lane]:" in line and name and watch in name:
            seen += 1
            if int(line.split("ScratchSize [bytes/lane]:")[1].split()[0]) != 0:
                raise RuntimeError("%s spills registers (%s): its hand-counted vector-memory waits are only valid without "
                                   "scratch traffic" % (name, line.strip()))
    if not seen:
        raise RuntimeError("no resource-usage remark for kernels matching %r (%s)" % (watch, " ".join(cmd)))


def is_stale(tuning=False):
    lib, stamp, _ = _paths(tuning)
    if not (os.path.exists(lib) and os.path.exists(stamp)):
        return True
    with open(stamp) as fh:
        return fh.read().strip() != _digest()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, tuning="--tuning" in sys.argv))
