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
    # the Bayesian stages are compared against a NumPy oracle: no FMA contraction
    ("post_kernels.hip", ["-ffp-contract=off"]),
    ("loss_kernels.hip", ["-ffp-contract=off"]),
    ("train_kernels.hip", []),
    ("engine.hip", []),
]
# -fno-slp-vectorize (round 6): NO packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in any kernel of
# the library.  With them, post_fuse_kernel / cluster_fuse_kernel -- whose 4x4 inverses and matrix products the SLP vectoriser packs two
# floats at a time -- returned wrong results in lanes 48-63 of a wave whenever a convolution kernel of the library shared the compute
# unit (tests/tools/selfcheck_probe.py: 0.2-10 % of the self-checked waves; without them 0 of 56 million, and tests/test_gpu_zz_canary.py
# is green).  Same IEEE operations either way (bit-identical results when nothing runs beside the kernel); the convolution kernels'
# epilogues measure the same speed without them (profiles/round6_mx_ablations.txt: headline 1 877-1 881 against 1 873-1 876 frames/s,
# f16mx 893-895 against 894-896 on one box).
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result", "-Wno-unused-value", "-fno-slp-vectorize"]


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
            n_checked = sum(kernel_guard.check_no_packed_fp32(o) for o in objs)          # (-fno-slp-vectorize took effect everywhere)
            if verbose:
                print("no packed fp32 instruction in %d kernels" % n_checked, flush=True)
            if verbose:
                print("kernel guards ok: %d production kernels, no spills, inline-asm MFMA windows clean" % len(regs), flush=True)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
