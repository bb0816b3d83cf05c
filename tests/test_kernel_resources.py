"""CPU (hipcc cross-compile): guards on the compiled gfx950 code object of the implicit-GEMM kernel.  The checks themselves
live in bayes-od-rc_amd/kernel_guard.py and are a HARD part of build.build() (a failing guard refuses to link); here they
run against the object the build just produced, plus negative tests of the guard's own logic."""
import os
import shutil

import pytest

from bayes_od_rc_amd import build as build_mod
from bayes_od_rc_amd import kernel_guard as guard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def conv_object():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    build_mod.build(verbose=False)                   # no-op when the tree is built; runs the guards before linking otherwise
    return os.path.join(build_mod.LIB_DIR, "obj", "conv_igemm.o")


def test_production_conv_kernels_do_not_spill(conv_object, tmp_path):
    co = guard.extract_device_object(conv_object, str(tmp_path))
    meta = guard.kernel_metadata(co)
    guard.check_no_spills(meta)
    for want in guard.PRODUCTION:
        assert any(want in n for n in meta), want
    # the bottleneck-chain builds must keep the residency of the kernels they replace: 3 / 2 workgroups (waves per SIMD) per CU
    for n, f in meta.items():
        if "ELi64ELi128ELi1ELi4ELi10E" in n:
            assert f["vgpr_count"] <= 168, (n, f)
        if "ELi128ELi128ELi2ELi2ELi10E" in n:
            assert f["vgpr_count"] <= 256, (n, f)
    # the 8-wave 256x256 tiles run two waves per SIMD: at most 256 registers per lane
    for n, f in meta.items():
        if any(w in n for w in guard.PRODUCTION) and "ILi256ELi256ELi2ELi4E" in n:
            assert f["vgpr_count"] + f.get("agpr_count", 0) <= 256, (n, f)


def test_row_task_stem_kernel_does_not_spill():
    """The bf16 stem kernels keep their weight fragments in registers (one block per CU): the four-wave forms hold all 44 and sit
    close to the register file's limit; the eight-wave stem + pool kernel (round 4) holds half of them per wave and must stay
    within the 256 registers two waves per SIMD leave each -- and so must its (hi, lo) twin of round 6, which also holds the lo
    weights of five more k-steps."""
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    build_mod.build(verbose=False)
    regs = guard.verify_aux(os.path.join(build_mod.LIB_DIR, "obj", "aux_kernels.o"))
    assert len(regs) == 4, regs
    for name, v in regs.items():
        if "stem_pool_fused_kernelILi2E" in name or "stem_pool_fused_split_kernel" in name:
            assert v <= 256, regs
        else:
            assert 256 < v <= 512, regs


def test_pointwise_kernels_keep_their_residency():
    """The streaming 1x1 kernels (conv_pointwise.hip): no spills, and few enough registers for 2 / 3 workgroups per CU."""
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    build_mod.build(verbose=False)
    regs = guard.verify_pointwise(os.path.join(build_mod.LIB_DIR, "obj", "conv_pointwise.o"))
    assert len(regs) == 11, regs            # eight pw_conv_kernel instantiations (incl. the NEXT and dual forms), two slide3x3 of 64 channels, one of 128


def test_inline_asm_mfma_accumulators_are_untouched_inside_the_tower_loop(conv_object, tmp_path):
    """The row-reuse tower loop issues its MFMAs as inline asm with the accumulators tied in place.  The compiler neither knows
    their result latency nor inserts the wait states a real MFMA would get, so any compiler-generated instruction that reads or
    writes an accumulator register right behind an MFMA (a v_mov that merges two live ranges, say -- seen when a branch was put
    around the last fragment's MFMAs), or a VALU write of a source two slots in front of one, silently corrupts results."""
    co = guard.extract_device_object(conv_object, str(tmp_path))
    funcs = guard.disassemble(co)
    checked = 0
    for want in guard.INLINE_ASM_MFMA:
        names = [n for n in funcs if want in n]
        assert names, want
        for n in names:
            checked += guard.check_inline_asm_mfma(funcs[n], want)
    assert checked >= 3 * 192


def test_hand_counted_lds_waits_cover_every_fragment_read(conv_object, tmp_path):
    """The mid-tile-barrier tower loop (production tower and fan-out kernels) reads its MFMA fragments with inline-asm ds_read_b128 and waits with hand-written
    `s_waitcnt lgkmcnt(n)`: walk the loop's disassembly with the in-order LDS return queue (two trips, across the back edge)."""
    co = guard.extract_device_object(conv_object, str(tmp_path))
    funcs = guard.disassemble(co, with_addr=True)
    walked = 0
    for want in guard.INLINE_ASM_LDS:
        names = [n for n in funcs if want in n]
        assert names, want
        for n in names:
            walked += guard.check_asm_lds_reads(funcs[n], want)
    assert walked >= 2 * (6 + 2 * 72)


def test_the_guard_catches_both_hazards():
    mf = "v_mfma_f32_16x16x32_bf16 v[0:3], v[10:13], v[20:23], v[0:3]"
    pad = ["s_nop 15"]
    ok = [mf] * 192 + pad
    assert guard.check_inline_asm_mfma(ok, "k") == 192
    with pytest.raises(guard.GuardError, match="touches the destination"):
        guard.check_inline_asm_mfma([mf] * 191 + [mf, "v_mov_b32_e32 v40, v2"], "k")
    with pytest.raises(guard.GuardError, match="writes a source"):
        guard.check_inline_asm_mfma([mf] * 100 + ["v_mov_b32_e32 v11, v50", "s_nop 0", mf] + [mf] * 91 + pad, "k")
    # two wait states in between are enough
    assert guard.check_inline_asm_mfma([mf] * 100 + ["v_mov_b32_e32 v11, v50", "s_nop 1", mf] + [mf] * 91 + pad, "k") == 192
    with pytest.raises(guard.GuardError, match="changed shape"):
        guard.check_inline_asm_mfma([mf] * 10, "k")
    # the LDS queue walk: an MFMA in front of the wait that covers its operand, and a correct sequence
    rd = lambda r: "ds_read_b128 v[%d:%d], v200" % (r, r + 3)
    mm = lambda a_: "v_mfma_f32_16x16x32_bf16 v[0:3], v[%d:%d], v[20:23], v[0:3]" % (a_, a_ + 3)
    good = [(4 * i, l) for i, l in enumerate(["s_barrier", rd(100), rd(104), "s_waitcnt lgkmcnt(1)", mm(100), "s_waitcnt lgkmcnt(0)", mm(104)])]
    assert guard.check_asm_lds_reads(good, "k", min_reads=2) == 2
    bad = [(4 * i, l) for i, l in enumerate(["s_barrier", rd(100), rd(104), "s_waitcnt lgkmcnt(1)", mm(104), "s_waitcnt lgkmcnt(0)", mm(100)])]
    with pytest.raises(guard.GuardError, match="may still be in flight"):
        guard.check_asm_lds_reads(bad, "k", min_reads=2)
    with pytest.raises(guard.GuardError, match="spills"):
        guard.check_no_spills({"_Z17" + guard.PRODUCTION[0] + "v8ConvArgs": {"vgpr_spill_count": 3, "private_segment_fixed_size": 16}},
                              wanted=guard.PRODUCTION[:1])


def test_no_kernel_of_the_library_contains_packed_fp32_instructions():
    """DESIGN.md 8.4 (round 6): with packed fp32 instructions (what the SLP vectoriser makes of 4x4 inverses and matrix products) the
    posterior's fusion kernels came out wrong in lanes 48-63 of a wave while a convolution kernel shared the compute unit; the library
    is built with -fno-slp-vectorize and the build refuses an object that contains one (kernel_guard.check_no_packed_fp32).  Here: the
    objects of the in-tree build are clean, and the check does fire on a source compiled with the vectoriser on."""
    import subprocess
    import tempfile
    obj_dir = os.path.join(ROOT, "bayes-od-rc_amd", "lib", "obj")
    objs = [os.path.join(obj_dir, f) for f in sorted(os.listdir(obj_dir)) if f.endswith(".o")]
    if not objs:
        pytest.skip("library not built in-tree")
    assert sum(guard.check_no_packed_fp32(o) for o in objs) > 50
    packed = ("#include <hip/hip_runtime.h>\n"
              "typedef float f2 __attribute__((ext_vector_type(2)));\n"
              "__global__ void k(const f2* a, const f2* b, f2* c) { c[threadIdx.x] = a[threadIdx.x] * b[threadIdx.x] + a[threadIdx.x]; }\n")
    plain = ("#include <hip/hip_runtime.h>\n"
             "__global__ void k(const float* a, const float* b, float* c) { c[threadIdx.x] = a[threadIdx.x] * b[threadIdx.x] + a[threadIdx.x]; }\n")
    with tempfile.TemporaryDirectory() as d:
        for name, src in (("packed", packed), ("plain", plain)):
            with open(os.path.join(d, name + ".hip"), "w") as fp:
                fp.write(src)
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-c", os.path.join(d, name + ".hip"), "-o", os.path.join(d, name + ".o")])
        with pytest.raises(guard.GuardError):
            guard.check_no_packed_fp32(os.path.join(d, "packed.o"))          # (an explicit two-float vector operation: v_pk_fma_f32 / v_pk_mul_f32)
        assert guard.check_no_packed_fp32(os.path.join(d, "plain.o")) == 1
