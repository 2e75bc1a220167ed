"""Builds pies_amd/lib/libpies_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

The library is built in-tree so that it travels with the repository snapshot to the GPU box; object
files are cached under pies_amd/lib/obj and rebuilt when a source or header is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libpies_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

ARCH = "gfx950"
# -ffp-contract=off: device arithmetic is the plain IEEE sequence in the source (no implicit FMA), which
# makes results reproducible on a host bit for bit; see DESIGN.md "Numerics".
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
          "-I", INCLUDE]
DEVICE = [f"--offload-arch={ARCH}"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") or f.endswith(".cpp"):
            out.append(os.path.join(CSRC, f))
    return out


def _newest_header():
    t = os.path.getmtime(os.path.join(INCLUDE, "pies_hip.h"))
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            t = max(t, os.path.getmtime(os.path.join(CSRC, f)))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def _compile(src, force, exp=False):
    # exp: False (product), True (-DPIES_EXPERIMENTS), "bounds" (-DPIES_BOUNDS: device-side bounds checks, dev_math.h)
    obj = os.path.join(OBJDIR + ("_bounds" if exp == "bounds" else "_exp" if exp else ""), os.path.basename(src) + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _newest_header()):
        return obj, False
    extra = ["-DPIES_BOUNDS"] if exp == "bounds" else ["-DPIES_EXPERIMENTS"] if exp else []
    extra += os.environ.get("PIES_EXTRA_FLAGS", "").split()  # compiler-flag experiments (development aid)
    if src.endswith(".hip"):
        cmd = [_hipcc(), "-c", src, "-o", obj] + COMMON + DEVICE + extra
    else:  # host-only translation units: plain C++ against the HIP runtime API
        cmd = [_hipcc(), "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-c", src, "-o", obj] + COMMON + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=False, exp=False):
    """exp=True: the diagnostic library libpies_hip_exp.so (-DPIES_EXPERIMENTS: timing experiments that change the work done,
    in-kernel time stamps).  Never loaded by the product or the tests; tools select it with PIES_LIB."""
    lib = LIB.replace(".so", "_bounds.so") if exp == "bounds" else LIB.replace(".so", "_exp.so") if exp else LIB
    os.makedirs(OBJDIR + ("_bounds" if exp == "bounds" else "_exp" if exp else ""), exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, exp), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(c for _, c in res)
    if rebuilt or not os.path.exists(lib):
        cmd = [_hipcc(), "-shared", "-o", lib] + objs + DEVICE + ["-Wl,--no-undefined"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", lib)
    return lib


def build_asan(verbose=False):
    """libpies_hip_asan.so: the HOST side of the library (scene construction, schedules, planners, PD set-up, the C ABI) compiled
    by g++ with -fsanitize=address,undefined and linked with the kernels' ordinary objects.  Loaded only by
    tests/test_sanitizers.py, in a child process with the sanitizer runtime preloaded, through host-only handles
    (PIES_DEVICE_NONE): no GPU is needed, nothing on a device is instrumented."""
    build()  # the kernels' objects
    lib = LIB.replace(".so", "_asan.so")
    objdir = OBJDIR + "_asan"
    os.makedirs(objdir, exist_ok=True)
    gxx = shutil.which("g++")
    if not gxx:
        raise RuntimeError("g++ not found: the sanitizer build needs it")
    objs, rebuilt = [], False
    for src in sources():
        if src.endswith(".hip"):
            objs.append(os.path.join(OBJDIR, os.path.basename(src) + ".o"))
            continue
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _newest_header()):
            continue
        cmd = [gxx, "-std=c++17", "-O1", "-g", "-fPIC", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-ffp-contract=off",
               "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", INCLUDE, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        rebuilt = True
    if rebuilt or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(LIB):
        cmd = [gxx, "-shared", "-fsanitize=address,undefined", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,--no-undefined",
                                                                                     "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", lib)
    return lib


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan(verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True, exp="bounds" if "--bounds" in sys.argv else "--exp" in sys.argv))
