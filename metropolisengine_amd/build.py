"""Build libmetropolis_hip.so (and its plugins) for gfx950 with hipcc -- no cmake; everything is built in-tree.

    python -m metropolisengine_amd.build [--force] [--jobs N]

One object per (n_real, n_complex) pair from ``csrc/me_kernels.hip`` (the chain state is register-resident, so the
dimensions are compile-time), plus the dimension-independent kernels and the C-ABI layer.  Up-to-date checks use
content hashes (``<output>.stamp`` = sha256 of the command line and of every input file), not modification times, so
a tree that was copied to another machine is not rebuilt.
"""
import argparse
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "_build")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmetropolis_hip.so")
INCLUDE_DIR = os.path.join(REPO_DIR, "include")
ARCH = "gfx950"

# (n_real, n_complex, dense energy, per-chain covariance kernels)
#   the BASELINE.json configurations: (1,0) (2,0) (16,0) (4,4) (2,1) (64,0) (2,7); the rest are small sizes the
#   tests and the reference's demos use.  Any other pair is compiled on demand by build_dims().
KERNEL_DIMS = [
    (1, 0, 0, 1), (2, 0, 1, 1), (3, 0, 0, 1), (4, 0, 1, 1), (8, 0, 0, 1), (16, 0, 1, 1),
    (64, 0, 1, 2),
    (0, 1, 0, 1), (0, 2, 0, 1), (0, 3, 0, 1), (0, 4, 0, 1),
    (1, 1, 0, 1), (1, 2, 0, 1), (2, 1, 0, 1), (2, 2, 1, 1), (4, 4, 0, 1), (2, 7, 0, 1),
]
MAX_PACKED_IN_REGISTERS = 160     # per-chain covariance / factor kernels keep the packed matrix in registers
MAX_REGISTER_DOF = 96             # largest n_real + 2 n_complex the register-resident kernels are built for BY DEFAULT: beyond
#                                   it built-in energies run on the runtime-dimension set (no build, identity / shared shape)
MAX_COMPILED_DOF = 128            # ... unless per-chain shapes are asked for (cov_mode="reference"): up to here the kernel set of
#                                   the space is compiled on demand after all (minutes of hipcc; float64 keeps part of a chain's
#                                   state in scratch there), so that the reference's semantics exist beyond 96 too


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmetropolis_hip.so cannot be built")
    return exe


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
        sorted(os.path.join(INCLUDE_DIR, f) for f in os.listdir(INCLUDE_DIR) if f.endswith(".h"))


def _key(args, inputs):
    """sha256 over the (path-independent) command arguments and the contents of the input files."""
    h = hashlib.sha256()
    for a in args:
        h.update(os.path.basename(a).encode() if os.path.isabs(a) else a.encode())
        h.update(b"\0")
    for path in inputs:
        with open(path, "rb") as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    return h.hexdigest()


def _fresh(target, key):
    try:
        with open(target + ".stamp") as fh:
            return os.path.exists(target) and fh.read().strip() == key
    except OSError:
        return False


def _run(cmd, target=None, key=None):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), proc.stdout))
    if target is not None:
        with open(target + ".stamp", "w") as fh:
            fh.write(key + "\n")
    return proc.stdout


def _compile_flags():
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", INCLUDE_DIR]


def build(force=False, jobs=None, verbose=True):
    """Compile and link libmetropolis_hip.so; returns its path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = _headers()
    units = []
    # the largest sets first: they take minutes, the small ones seconds
    for nr, nc, dense, per_chain in sorted(KERNEL_DIMS, key=lambda dims: -(dims[0] + 2 * dims[1])):
        for bits in (32, 64):       # one object per precision: the halves of a large set (64 parameters) build in parallel
            units.append((os.path.join(OBJ_DIR, "me_kernels_%d_%d_f%d.o" % (nr, nc, bits)), os.path.join(CSRC, "me_kernels.hip"),
                          ["-DME_NR=%d" % nr, "-DME_NC=%d" % nc, "-DME_DENSE=%d" % dense, "-DME_PER_CHAIN=%d" % per_chain,
                           "-DME_ONLY_DTYPE=%d" % bits]))
    for name in ("me_generic", "me_statistics", "me_runtime_dims", "me_api"):
        units.append((os.path.join(OBJ_DIR, name + ".o"), os.path.join(CSRC, name + ".hip"), []))

    todo = []
    for obj, src, flags in units:
        args = _compile_flags() + flags
        key = _key(args, [src] + headers)
        if force or not _fresh(obj, key):
            todo.append((obj, [hipcc()] + args + ["-c", src, "-o", obj], key))
    if verbose and todo:
        print("[metropolisengine_amd.build] compiling %d of %d objects for %s" % (len(todo), len(units), ARCH), flush=True)
    jobs = jobs or min(8, os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        futs = {pool.submit(_run, cmd, obj, key): obj for obj, cmd, key in todo}
        for fut in concurrent.futures.as_completed(futs):
            out = fut.result()
            if verbose:
                print("[metropolisengine_amd.build]   %s" % os.path.basename(futs[fut]), flush=True)
                if out.strip():
                    print(out)
    objs = [u[0] for u in units]
    link_key = _key(["link"] + [os.path.basename(o) for o in objs], [o + ".stamp" for o in objs])
    if force or todo or not _fresh(LIB_PATH, link_key):
        _run([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs, LIB_PATH, link_key)
        if verbose:
            print("[metropolisengine_amd.build] linked %s" % LIB_PATH, flush=True)
    return LIB_PATH


def _build_plugin(out, defines, extra_inputs, extra_includes=(), force=False):
    """One hipcc run: csrc/me_kernels.hip + defines -> a plugin library linked against libmetropolis_hip.so."""
    build(verbose=False)
    src = os.path.join(CSRC, "me_kernels.hip")
    args = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-I", INCLUDE_DIR] + list(defines)
    key = _key(args, [src, LIB_PATH + ".stamp"] + _headers() + list(extra_inputs))
    if not force and _fresh(out, key):
        return out
    includes = []
    for inc in extra_includes:
        includes += ["-I", inc]
    # one object per precision, compiled side by side (a large set is minutes of hipcc per precision), then the link
    compile_args = [a for a in args if a != "-shared"] + includes
    objs = [os.path.join(OBJ_DIR, os.path.basename(out) + ".f%d.o" % bits) for bits in (32, 64)]
    os.makedirs(OBJ_DIR, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=2) as pool:
        jobs = [pool.submit(_run, [hipcc()] + compile_args + ["-DME_ONLY_DTYPE=%d" % bits, "-c", src, "-o", obj])
                for bits, obj in zip((32, 64), objs)]
        for job in jobs:
            job.result()
    _run([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs +
         ["-L", LIB_DIR, "-lmetropolis_hip", "-Wl,-rpath,$ORIGIN"], out, key)
    return out


def user_plugin_path(name, n_real, n_complex):
    return os.path.join(LIB_DIR, "libme_user_%s_%d_%d.so" % (name, n_real, n_complex))


def build_user_energy(source, name, n_real, n_complex, force=False, per_chain=True):
    """Compile a user device energy (include/metropolis_user_energy.h) around the engine's kernels into a plugin.

    Returns the plugin path (``lib/libme_user_<name>_<nr>_<nc>.so``); load it with ``me_load_plugin``.
    """
    source = os.path.abspath(source)
    if not name.isidentifier():
        raise ValueError("plugin name must be an identifier")
    packed = n_real * (n_real + 1) // 2 + n_complex * n_complex
    defines = ["-DME_NR=%d" % n_real, "-DME_NC=%d" % n_complex,
               "-DME_PER_CHAIN=%d" % ((1 if packed <= MAX_PACKED_IN_REGISTERS else 2) if per_chain else 0),
               "-DME_USER_SOURCE=\"%s\"" % source, "-DME_USER_NAME=\"%s\"" % name]
    # the absolute source path is part of the command line but not of the cache key: hash its basename + contents
    key_defines = defines[:3] + ["-DME_USER_SOURCE=" + os.path.basename(source), defines[4]]
    out = user_plugin_path(name, n_real, n_complex)
    build(verbose=False)
    src = os.path.join(CSRC, "me_kernels.hip")
    base = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-I", INCLUDE_DIR]
    key = _key(base + key_defines, [src, LIB_PATH + ".stamp", source] + _headers())
    if not force and _fresh(out, key):
        return out
    _run([hipcc()] + base + defines + ["-I", os.path.dirname(source), src, "-o", out,
                                       "-L", LIB_DIR, "-lmetropolis_hip", "-Wl,-rpath,$ORIGIN"], out, key)
    return out


def dims_plugin_path(n_real, n_complex):
    return os.path.join(LIB_DIR, "libme_dims_%d_%d.so" % (n_real, n_complex))


def build_dims(n_real, n_complex, force=False):
    """Compile the kernel set for a (n_real, n_complex) pair that is not in KERNEL_DIMS into a plugin library.

    The chain state is register-resident, so the dimensions are compile-time constants; any other size is one
    hipcc run (~1 min, cached in lib/) away.  Returns the plugin path; load it with ``me_load_plugin``.
    """
    d = n_real + 2 * n_complex
    packed = n_real * (n_real + 1) // 2 + n_complex * n_complex
    if d > MAX_COMPILED_DOF:
        raise RuntimeError("register-resident kernels are compiled for at most %d real degrees of freedom (got %d); larger "
                           "spaces run on the runtime-dimension kernels of the main library (csrc/me_runtime_dims.hip), no "
                           "build needed, with cov_mode='fixed' or 'pooled'" % (MAX_COMPILED_DOF, d))
    defines = ["-DME_NR=%d" % n_real, "-DME_NC=%d" % n_complex, "-DME_DENSE=%d" % int(d <= 24),
               "-DME_PER_CHAIN=%d" % (1 if packed <= MAX_PACKED_IN_REGISTERS else 2)]   # 2: packed matrices streamed
    return _build_plugin(dims_plugin_path(n_real, n_complex), defines, [], force=force)


def build_examples(force=False):
    """The shipped plugins: the cylinder-style user energy for BASELINE config 5 (2 real + 7 complex), the term-wise
    Landau plugin (energy dictionary, 2 real + 1 complex) and one kernel set outside KERNEL_DIMS, (3, 2), which
    exercises the compile-on-demand path of build_dims."""
    src = os.path.join(REPO_DIR, "examples", "user_energy_cylinder.h")
    terms = os.path.join(REPO_DIR, "examples", "user_energy_landau_terms.h")
    build(verbose=False)
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        jobs = [pool.submit(build_user_energy, src, "cylinder", 2, 7, force), pool.submit(build_dims, 3, 2, force),
                pool.submit(build_user_energy, terms, "landau_terms", 2, 1, force),
                pool.submit(build_dims, 1, 13, force),     # 170 packed entries WITH a complex block: streamed mixed shapes
                pool.submit(build_dims, 0, 13, force),     # ... and a pure complex space beyond the register-resident size
                pool.submit(build_dims, 24, 0, force),     # streamed per-chain shapes with two chains per wavefront
                pool.submit(build_dims, 96, 0, force),     # the largest register-resident parameter space built by default
                pool.submit(build_dims, 100, 0, force)]    # beyond it: compiled for cov_mode="reference" (per-chain shapes)
        return [job.result() for job in jobs]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    cli = ap.parse_args()
    try:
        build(force=cli.force, jobs=cli.jobs)
        build_examples(force=cli.force)
    except RuntimeError as exc:
        print(exc, file=sys.stderr)
        sys.exit(1)
