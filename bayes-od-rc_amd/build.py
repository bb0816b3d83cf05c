"""Builds libbayesod_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbayesod_hip.so")

SOURCES = [
    ("conv_igemm.hip", []),
    ("conv_igemm_f32.hip", []),
    ("conv_pointwise.hip", []),
    ("aux_kernels.hip", []),
    # the Bayesian stages are compared against a NumPy oracle: no FMA contraction.  -fno-slp-vectorize: no packed fp32 instructions
    # (v_pk_*_f32) in these kernels -- with them their 4x4 inverses / matrix products come out wrong in lanes 48-63 of a wave while a
    # convolution kernel of the library shares the compute unit (same IEEE operations, bit-identical results)
    ("post_kernels.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    ("loss_kernels.hip", ["-ffp-contract=off", "-fno-slp-vectorize"]),
    ("train_kernels.hip", []),
    ("engine.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libbayesod_hip.so)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


HEADER = os.path.join(os.path.dirname(HERE), "include", "bayesod.h")
CDEF = os.path.join(os.path.dirname(HERE), "include", "bayesod_cdef.h")


def cdef_text():
    """include/bayesod.h reduced to what ``cffi.FFI().cdef()`` (or any pycparser-based binder) accepts: comments,
    preprocessor lines and the extern "C" bracket removed, declarations untouched."""
    import re
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for line in text.splitlines():
        t = line.strip()
        if not t or t.startswith("#") or t == 'extern "C" {' or t == "}":
            continue
        out.append(line.rstrip())
    return ("/* GENERATED from include/bayesod.h by bayes_od_rc_amd.build.write_cdef(): the same declarations without\n"
            " * comments / preprocessor lines, for ffi.cdef(open('include/bayesod_cdef.h').read()).  Do not edit. */\n"
            + "\n".join(out) + "\n")


def write_cdef():
    text = cdef_text()
    if not os.path.exists(CDEF) or open(CDEF).read() != text:
        with open(CDEF, "w") as fp:
            fp.write(text)
    return CDEF


def _load_guard():
    """kernel_guard.py next to this file (build.py also runs as a plain script, outside the package)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bayesod_kernel_guard", os.path.join(HERE, "kernel_guard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(force=False, verbose=True):
    hipcc = _hipcc()
    try:
        write_cdef()                     # developer convenience: an installed / read-only tree must still build the library
    except OSError as e:
        if verbose:
            print("include/bayesod_cdef.h not refreshed (%s)" % e, flush=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "bayesod.h"))
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB_PATH, objs):
        # guards on the object about to be linked: no spills in the production kernels, no hazard around the inline-asm MFMAs of
        # the tower loop (kernel_guard.py) -- a compiler bump that breaks either fails the build instead of shipping
        if os.environ.get("BOD_SKIP_KERNEL_GUARD") == "1":
            # explicit opt-out (a toolchain without the llvm binary tools, an experiment): never silent
            print("WARNING: BOD_SKIP_KERNEL_GUARD=1 -- linking libbayesod_hip.so WITHOUT the disassembly guards (spills, inline-asm MFMA "
                  "hazard windows and hand-counted LDS waits are UNCHECKED)", file=sys.stderr, flush=True)
        else:
            kernel_guard = _load_guard()
            regs = kernel_guard.verify(os.path.join(obj_dir, "conv_igemm.o"))
            regs.update(kernel_guard.verify_aux(os.path.join(obj_dir, "aux_kernels.o")))
            regs.update(kernel_guard.verify_pointwise(os.path.join(obj_dir, "conv_pointwise.o")))
            if verbose:
                print("kernel guards ok: %d production kernels, no spills, inline-asm MFMA windows clean" % len(regs), flush=True)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
