"""Build libmetropolis_hip.so for gfx950 with hipcc (no cmake, no JIT; the .so is built in-tree).

    python -m metropolisengine_amd.build [--force] [--jobs N]

One object per (n_real, n_complex) pair from ``csrc/me_kernels.hip`` (the chain state is register-resident, so
the dimensions are compile-time), plus the dimension-independent kernels and the C-ABI layer.
"""
import argparse
import concurrent.futures
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "_build")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmetropolis_hip.so")
ARCH = "gfx950"

# (n_real, n_complex, dense energy, per-chain covariance kernels)
#   the BASELINE.json configurations: (1,0) (2,0) (16,0) (4,4) (2,1) (64,0) (2,7); the rest are small sizes the
#   tests and the reference's demos use.
KERNEL_DIMS = [
    (1, 0, 0, 1), (2, 0, 1, 1), (3, 0, 0, 1), (4, 0, 1, 1), (8, 0, 0, 1), (16, 0, 1, 1),
    (64, 0, 1, 0),
    (0, 1, 0, 1), (0, 2, 0, 1), (0, 3, 0, 1), (0, 4, 0, 1),
    (1, 1, 0, 1), (1, 2, 0, 1), (2, 1, 0, 1), (2, 2, 1, 1), (4, 4, 0, 1), (2, 7, 0, 1),
]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmetropolis_hip.so cannot be built")
    return exe


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), proc.stdout))
    return proc.stdout


def build(force=False, jobs=None, verbose=True, extra_flags=()):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(PKG_DIR), "include", "metropolis_engine.h"))
    headers.append(os.path.abspath(__file__))
    base = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
            "-I", os.path.join(os.path.dirname(PKG_DIR), "include")] + list(extra_flags)
    units = []
    for nr, nc, dense, per_chain in KERNEL_DIMS:
        obj = os.path.join(OBJ_DIR, "me_kernels_%d_%d.o" % (nr, nc))
        units.append((obj, os.path.join(CSRC, "me_kernels.hip"),
                      ["-DME_NR=%d" % nr, "-DME_NC=%d" % nc, "-DME_DENSE=%d" % dense, "-DME_PER_CHAIN=%d" % per_chain]))
    for name in ("me_generic", "me_api"):
        units.append((os.path.join(OBJ_DIR, name + ".o"), os.path.join(CSRC, name + ".hip"), []))

    todo = [(obj, src, flags) for obj, src, flags in units if force or not _newer(obj, [src] + headers)]
    if verbose and todo:
        print("[metropolisengine_amd.build] compiling %d of %d objects for %s" % (len(todo), len(units), ARCH), flush=True)
    jobs = jobs or min(8, os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        futs = {pool.submit(_run, base + flags + ["-c", src, "-o", obj]): obj for obj, src, flags in todo}
        for fut in concurrent.futures.as_completed(futs):
            out = fut.result()
            if verbose:
                print("[metropolisengine_amd.build]   %s" % os.path.basename(futs[fut]), flush=True)
                if out.strip():
                    print(out)
    objs = [u[0] for u in units]
    if force or todo or not _newer(LIB_PATH, objs):
        _run([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs)
        if verbose:
            print("[metropolisengine_amd.build] linked %s" % LIB_PATH, flush=True)
    return LIB_PATH


def user_plugin_path(name, n_real, n_complex):
    return os.path.join(LIB_DIR, "libme_user_%s_%d_%d.so" % (name, n_real, n_complex))


def build_user_energy(source, name, n_real, n_complex, force=False, per_chain=True):
    """Compile a user device energy (include/metropolis_user_energy.h) around the engine's kernels into a plugin.

    Returns the plugin path (``lib/libme_user_<name>_<nr>_<nc>.so``); load it with ``me_load_plugin``.
    """
    source = os.path.abspath(source)
    if not name.isidentifier():
        raise ValueError("plugin name must be an identifier")
    build(verbose=False)                      # the plugin links against libmetropolis_hip.so
    out = user_plugin_path(name, n_real, n_complex)
    deps = [source, os.path.join(CSRC, "me_kernels.hip"), LIB_PATH] + \
           [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and _newer(out, deps):
        return out
    inc = os.path.join(os.path.dirname(PKG_DIR), "include")
    _run([hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-I", inc,
          "-I", os.path.dirname(source),
          "-DME_NR=%d" % n_real, "-DME_NC=%d" % n_complex, "-DME_PER_CHAIN=%d" % int(per_chain),
          "-DME_USER_SOURCE=\"%s\"" % source, "-DME_USER_NAME=\"%s\"" % name,
          os.path.join(CSRC, "me_kernels.hip"), "-o", out,
          "-L", LIB_DIR, "-lmetropolis_hip", "-Wl,-rpath,$ORIGIN"])
    return out


def dims_plugin_path(n_real, n_complex):
    return os.path.join(LIB_DIR, "libme_dims_%d_%d.so" % (n_real, n_complex))


MAX_PACKED_IN_REGISTERS = 160     # per-chain covariance / factor kernels keep the packed matrix in registers


def build_dims(n_real, n_complex, force=False):
    """Compile the kernel set for a (n_real, n_complex) pair that is not in KERNEL_DIMS into a plugin library.

    The chain state is register-resident, so the dimensions are compile-time constants; any other size is one
    hipcc run (~10-30 s, cached in lib/) away.  Returns the plugin path; load it with ``me_load_plugin``.
    """
    build(verbose=False)
    out = dims_plugin_path(n_real, n_complex)
    deps = [os.path.join(CSRC, "me_kernels.hip"), LIB_PATH] + \
           [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and _newer(out, deps):
        return out
    d = n_real + 2 * n_complex
    packed = n_real * (n_real + 1) // 2 + n_complex * n_complex
    if d > 96:
        raise RuntimeError("register-resident kernels support at most 96 real degrees of freedom (got %d)" % d)
    inc = os.path.join(os.path.dirname(PKG_DIR), "include")
    _run([hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-I", inc,
          "-DME_NR=%d" % n_real, "-DME_NC=%d" % n_complex, "-DME_DENSE=%d" % int(d <= 24),
          "-DME_PER_CHAIN=%d" % int(packed <= MAX_PACKED_IN_REGISTERS),
          os.path.join(CSRC, "me_kernels.hip"), "-o", out, "-L", LIB_DIR, "-lmetropolis_hip", "-Wl,-rpath,$ORIGIN"])
    return out


def build_examples(force=False):
    """The shipped plugins: the cylinder-style user energy for BASELINE config 5 (2 real + 7 complex) and one
    kernel set outside KERNEL_DIMS, (3, 2), which exercises the compile-on-demand path of build_dims."""
    src = os.path.join(os.path.dirname(PKG_DIR), "examples", "user_energy_cylinder.h")
    build(verbose=False)
    with concurrent.futures.ThreadPoolExecutor(max_workers=2) as pool:
        jobs = [pool.submit(build_user_energy, src, "cylinder", 2, 7, force), pool.submit(build_dims, 3, 2, force)]
        return [job.result() for job in jobs]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    args = ap.parse_args()
    try:
        build(force=args.force, jobs=args.jobs)
        build_examples(force=args.force)
    except RuntimeError as exc:
        print(exc, file=sys.stderr)
        sys.exit(1)
