"""Build-time guards on the compiled gfx950 code object of csrc/conv_igemm.hip.

Two properties no functional test sees are checked on the DISASSEMBLY / metadata of the object that is about to be linked
into libbayesod_hip.so (``build.build()`` calls ``verify()`` and refuses to link when it fails; tests/test_kernel_resources.py
calls the same functions):

* the production instantiations of ``conv_igemm_kernel`` do not spill (a spill in the 256x256 tile costs ~15 % of the
  pipeline);
* the row-reuse tower loop issues ``v_mfma_f32_16x16x32_bf16`` as inline asm with the accumulators tied in place.  The
  compiler neither knows the result latency of an inline-asm MFMA nor inserts the wait states a real one gets, so
  (a) no other instruction may touch an MFMA's destination in the 16 issue slots behind it and (b) no VALU instruction may
  write one of its sources in the two slots in front of it (tests/tools/mfma_war_probe.hip: 18 % wrong results otherwise).
  A compiler bump that re-orders the loop must fail the BUILD, not ship an unchecked kernel.
"""
import os
import re
import subprocess
import tempfile

import shutil


def _llvm_bin_candidates():
    """Directories that may hold llvm-objcopy / clang-offload-bundler / llvm-readelf / llvm-objdump: next to the hipcc the build
    uses ($HIPCC, PATH), under $ROCM_PATH, the unversioned /opt/rocm, any /opt/rocm-x.y.z."""
    import glob
    roots = []
    for hipcc in (os.environ.get("HIPCC"), shutil.which("hipcc")):
        if hipcc:
            roots.append(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))))      # <root>/bin/hipcc
    roots += [os.environ.get("ROCM_PATH"), os.environ.get("ROCM_HOME"), "/opt/rocm"] + sorted(glob.glob("/opt/rocm-*"), reverse=True)
    out = []
    for r in roots:
        if r:
            for sub in ("lib/llvm/bin", "llvm/bin", "bin"):
                d = os.path.join(r, sub)
                if d not in out:
                    out.append(d)
    return out

PRODUCTION = [     # <BC, BP, WC, WP, ABL, XR, SPLIT>
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi0ELb1ELb0EE",     # head towers (row reuse)
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi5ELb1ELb0EE",     # first tower layer, N-way fan-out, on the row-reuse loop
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi6ELb1ELb0EE",     # backbone / FPN 3x3 256 -> 256 layers on the row-reuse loop (no dropout build)
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi0ELb0ELb0EE",     # backbone / FPN, big tile
    "conv_igemm_kernelILi128ELi128ELi2ELi2ELi0ELb0ELb0EE",     # fan-out layer, small layers, split-K
    "conv_igemm_kernelILi64ELi128ELi1ELi4ELi0ELb0ELb0EE",
    "conv_igemm_kernelILi64ELi128ELi1ELi4ELi10ELb0ELb0EE",     # bottleneck chain (2b -> 2c + shortcut -> next 2a), stage 2 / stage 3
    "conv_igemm_kernelILi128ELi128ELi2ELi2ELi10ELb0ELb0EE",
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi0ELb0ELb1EE",     # bf16x3 precision: the three tile configurations
    "conv_igemm_kernelILi128ELi128ELi2ELi2ELi0ELb0ELb1EE",
    "conv_igemm_kernelILi64ELi128ELi1ELi4ELi0ELb0ELb1EE",
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi0ELb1ELb1EE",     # bf16x3 on the row-reuse loop (fused 1x1 + MC aggregation)
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi9ELb1ELb0EE",     # head towers on the round-2 loop (top-of-K-tile barrier): A/B twin, BOD_TOWER_MIDBAR=0
    "conv_igemm_mx_kernelILi1EE",                              # f16mx precision: head towers, hx rows in (f16 + block-scaled e2m3 products, tied inline asm)
    "conv_igemm_mx_kernelILi2EE",                              # f16mx precision: first tower layer ((hi, lo) pairs in, hx rows out)
    "conv_igemm_mx_kernelILi3EE",                              # f16mx4 precision: head towers, h4 rows in (f16 + block-scaled e2m1 products)
]
MX_ASM_MFMA = ["conv_igemm_mx_kernelILi1EE", "conv_igemm_mx_kernelILi3EE"]                   # ... whose two MFMA flavours are tied inline asm on the 32x32 accumulators
MX_MFMA_RE = r"(?:v_mfma_scale_f32_32x32x64_f8f6f4|v_mfma_f32_32x32x16_f16)"
def _select(*markers):
    """Production kernels by a piece of their MANGLED NAME (never by list position: an insert would silently guard the wrong kernel)."""
    out = [k for k in PRODUCTION if any(m in k for m in markers)]
    assert len(out) == len(markers), (markers, out)
    return out


# the kernels on the 16x16x32 inline-asm loop: row reuse (XR = true, SPLIT = false) with ABL 0 / 5 / 6, and the round-2 twin ABL 9
INLINE_ASM_MFMA = _select("ELi0ELb1ELb0E", "ELi5ELb1ELb0E", "ELi6ELb1ELb0E", "ELi9ELb1ELb0E")
# ... whose fragment reads and lgkmcnt waits are hand-written too (mid-tile barrier)
INLINE_ASM_LDS = _select("ELi0ELb1ELb0E", "ELi5ELb1ELb0E", "ELi6ELb1ELb0E")
# scratch bytes per lane a kernel may park ACROSS its main loop (checked on the disassembly: none inside it); growth fails the guard.
# Empty since the end of round 6: the f16mx kernels <1> and <3> parked 16 bytes until their epilogue's ReLU + clamp became one v_med3_f32.
TOLERATED_SCRATCH = {}


class GuardError(RuntimeError):
    pass


def _tool(name):
    for d in _llvm_bin_candidates():
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    p = shutil.which(name)
    if p:
        return p
    raise GuardError("%s not found (looked in %s and PATH): the ROCm llvm tools are needed for the kernel guards; "
                     "BOD_SKIP_KERNEL_GUARD=1 builds without them, unchecked" % (name, ", ".join(_llvm_bin_candidates())))


def extract_device_object(host_obj, workdir):
    """gfx950 code object out of a hipcc host object (.hip_fatbin section -> offload bundle -> hipv4 entry)."""
    fat = os.path.join(workdir, "fat.bin")
    co = os.path.join(workdir, "dev.co")
    subprocess.check_call([_tool("llvm-objcopy"), "--dump-section=.hip_fatbin=" + fat, host_obj, os.path.join(workdir, "stripped.o")])
    subprocess.check_call([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return co


def kernel_metadata(code_object):
    """{kernel name: {field: int}} from the code object's AMDGPU metadata note."""
    text = subprocess.run([_tool("llvm-readelf"), "--notes", code_object], capture_output=True, text=True, check=True).stdout
    out = {}
    for block in re.split(r"\n  - ", text)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name or ".symbol:" not in block:
            continue
        fields = {}
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                    "group_segment_fixed_size"):
            m = re.search(r"\.%s:\s+(\d+)" % key, block)
            if m:
                fields[key] = int(m.group(1))
        out[name.group(1)] = fields
    return out


def check_no_spills(meta, wanted=PRODUCTION, tolerated=()):
    """tolerated: kernels whose (small) scratch use is checked on the disassembly instead -- check_no_scratch_in_loop."""
    for want in wanted:
        hits = [(n, f) for n, f in meta.items() if want in n]
        if not hits:
            raise GuardError("kernel %s not found in the code object" % want)
        for n, f in hits:
            if want in tolerated and f.get("private_segment_fixed_size", 0) <= TOLERATED_SCRATCH.get(want, 0):
                continue
            if f.get("vgpr_spill_count", 0) or f.get("private_segment_fixed_size", 0):
                raise GuardError("%s spills %d VGPRs (%d B scratch/lane)" % (n, f.get("vgpr_spill_count", 0), f.get("private_segment_fixed_size", 0)))


def check_no_scratch_in_loop(body, want, mfma_re):
    """A kernel at the register file's edge may park a value that lives ACROSS its main loop (written in front of it, read behind it) in
    scratch: harmless.  A scratch access INSIDE the loop -- between the first and the last of its tied MFMAs -- is a spill in the hot
    path (and a vmcnt the hand-counted waits do not know): refused."""
    mf = [i for i, l in enumerate(body) if re.match(mfma_re, l)]
    if not mf:
        raise GuardError("%s: no MFMA of the main loop found" % want)
    bad = [l for l in body[mf[0]:mf[-1] + 1] if l.startswith("scratch_")]
    if bad:
        raise GuardError("%s: %d scratch access(es) inside the main loop, e.g. `%s`" % (want, len(bad), bad[0]))
    return len([l for l in body if l.startswith("scratch_")])


def disassemble(code_object, with_addr=False):
    """{symbol: [instruction lines]} (comments and encodings stripped); with_addr: [(byte address, instruction)]."""
    text = subprocess.run([_tool("llvm-objdump"), "-d", "--no-show-raw-insn", code_object], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        parts = line.split("//")
        ins = parts[0].strip()
        if ins:
            if with_addr:
                am = re.match(r"\s*([0-9A-Fa-f]+):", parts[1]) if len(parts) > 1 else None
                cur.append((int(am.group(1), 16) if am else -1, ins))
            else:
                cur.append(ins)
    return funcs


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


MFMA = "v_mfma_f32_16x16x32_bf16"


def check_inline_asm_mfma(body, want, window=16, min_mfma=192, mfma_re=None):
    """`body`: instruction list of one kernel.  Returns the number of MFMAs checked.  mfma_re: regex of the tied inline-asm MFMA
    mnemonics (default: the 16x16x32 bf16 form of the tower loop)."""
    MFMA = mfma_re or globals()["MFMA"]
    is_mfma = (lambda l: re.match(MFMA, l) is not None) if mfma_re else (lambda l: l.startswith(MFMA))
    mf = [i for i, l in enumerate(body) if is_mfma(l)]
    if len(mf) < min_mfma:                          # three unrolled K-tiles of 64
        raise GuardError("%s: only %d %s found (the loop changed shape: re-derive the guard)" % (want, len(mf), MFMA))
    for i in mf:
        srcs = set()
        for tok in re.findall(r"v\[\d+:\d+\]", body[i]):
            srcs |= _regs(tok)
        # (b) a VALU write of a source register needs two wait states before the MFMA
        slots, k = 0, i - 1
        while k >= 0 and slots < 2:
            l = body[k]
            mm = re.match(r"s_nop (\d+)", l)
            if mm:
                slots += int(mm.group(1)) + 1
            else:
                if l.startswith("v_") and not l.startswith("v_mfma") and not l.startswith("v_cmp"):
                    toks = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l)
                    if toks and (_regs(toks[0]) & srcs):
                        raise GuardError("%s: `%s` writes a source of `%s` %d slot(s) before it" % (want, l, body[i], slots))
                slots += 1
            k -= 1
        # (a) nothing but MFMAs on the same accumulator may touch the destination in the next `window` slots
        dst = _regs(re.match((MFMA if mfma_re else re.escape(MFMA)) + r" (v\[\d+:\d+\])", body[i]).group(1))
        slots, k = 0, i + 1
        while k < len(body) and slots < window:
            l = body[k]
            mm = re.match(r"s_nop (\d+)", l)
            slots += int(mm.group(1)) + 1 if mm else 1
            if not is_mfma(l) and re.match(r"(v_|ds_|buffer_|global_|flat_)", l):
                toks = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l)
                touched = set().union(*[_regs(t) for t in toks]) if toks else set()
                if touched & dst:
                    raise GuardError("%s: `%s` touches the destination of `%s` %d slots behind it" % (want, l, body[i], slots))
            if l.startswith("s_cbranch") or l.startswith("s_branch") or l.startswith("s_endpgm"):
                break                                # (fall-through only: the check is per straight-line run)
            k += 1
    return len(mf)


def check_asm_lds_reads(body, want, min_reads=60):
    """The mid-tile-barrier tower loop (conv_igemm.hip, ABL = 6) issues every fragment read as inline asm and places every
    `s_waitcnt lgkmcnt(n)` by hand; the compiler believes a fragment register is written the moment its ds_read issues.  Walk the
    loop -- every forward conditional branch around a block WITHOUT fragment reads or MFMAs taken (LDS-DMA issues, Philox rounds:
    the path with the fewest waits) -- with the in-order LDS return queue and fail if any instruction reads or writes
    a fragment register while its ds_read may still be in flight (an MFMA placed in front of the wait that covers its operand, a
    compiler copy of a register across the loop's back edge, a register reused for something else ...).
    `body`: [(address, instruction)] of one kernel.  Returns the number of fragment reads checked."""
    addr_index = {a: i for i, (a, _) in enumerate(body)}
    mf = [i for i, (_, l) in enumerate(body) if l.startswith(MFMA)]
    if not mf:
        raise GuardError("%s: no %s" % (want, MFMA))
    # start at the first fragment read in front of the first MFMA's prologue (the reads issued behind the prologue barrier)
    start = mf[0]
    while start > 0 and not body[start - 1][1].startswith("s_barrier"):
        start -= 1
    queue, checked, i, steps = [], 0, start, 0
    end = mf[-1] + 1
    seen_back_edge = False
    while i < len(body) and steps < 200000:
        steps += 1
        a, l = body[i]
        op = l.split()[0]
        if op == "ds_read_b128":
            dst = _regs(re.match(r"ds_read_b128 (v\[\d+:\d+\])", l).group(1))
            addr_regs = set().union(*[_regs(t) for t in re.findall(r"\bv\d+\b", l)]) if re.findall(r"\bv\d+\b", l) else set()
            busy = set().union(*[q for q in queue if q]) if queue else set()
            if (dst | addr_regs) & busy:
                raise GuardError("%s: `%s` touches a fragment register still in flight" % (want, l))
            queue.append(dst)
            checked += 1
        elif op.startswith("ds_"):
            queue.append(set())                      # another LDS operation: returns in order with the fragment reads, no fragment register
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            # SMEM shares the counter but returns OUT OF ORDER: an s_load still in flight may be what keeps lgkmcnt at n while an older
            # ds_read has not landed, and one that returned early lowers the count without any ds_read having landed.  It is NOT put
            # in the queue: `lgkmcnt(n)` passing means (LDS ops + SMEM ops in flight) <= n, hence at most n LDS operations in
            # flight whatever the SMEM ones do -- retiring all but the n youngest LDS entries is the sound bound.
            pass
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
        elif op.startswith("s_cbranch") or op == "s_branch":
            off = int(l.split()[1])
            if off >= 32768:
                off -= 65536
            target = a + 4 + 4 * off
            if off < 0:                              # the loop's back edge: one more trip with the queue as it is, then stop
                if seen_back_edge:
                    i += 1                           # second visit: leave the loop, drain behind it
                    continue
                seen_back_edge = True
                if target not in addr_index:
                    raise GuardError("%s: branch target %x not found" % (want, target))
                i = addr_index[target]
                continue
            if target not in addr_index:
                raise GuardError("%s: branch target %x not found" % (want, target))
            skipped = [sk for _, sk in body[i + 1:addr_index[target]]]

            def backward(sk):
                return (sk.startswith("s_cbranch") or sk.startswith("s_branch ")) and int(sk.split()[1]) >= 32768
            if not seen_back_edge and any(backward(sk) for sk in skipped):
                i += 1                               # the loop's exit test in front of its back edge: stay in the loop once more
                continue
            if any(sk.startswith(MFMA) or sk.startswith("ds_read_b128") for sk in skipped):
                i += 1                               # a branch around the loop itself (trip-count check): fall through
                continue
            i = addr_index[target]                   # a conditional block of DMA issues / Philox rounds: the path without it
            continue
        else:
            busy = set().union(*[q for q in queue if q]) if queue else set()
            if busy:
                toks = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l)
                touched = set().union(*[_regs(t) for t in toks]) if toks else set()
                if touched & busy:
                    raise GuardError("%s: `%s` touches a fragment register whose ds_read may still be in flight (%d reads outstanding)"
                                     % (want, l, len([q for q in queue if q])))
        if i >= end and seen_back_edge and not queue:
            break
        i += 1
    if checked < min_reads:
        raise GuardError("%s: only %d fragment reads walked" % (want, checked))
    return checked


# kernels of aux_kernels.o that run close to the register file's limit (one block per CU, ~490 VGPRs + AGPRs): no spills either
AUX_PRODUCTION = ["stem_conv_bf16_row_kernel", "stem_pool_fused_kernel", "stem_pool_fused_split_kernel"]


def verify_aux(host_obj, wanted=AUX_PRODUCTION):
    """No-spill guard on a built aux_kernels.o.  Raises GuardError; returns {kernel: registers}."""
    with tempfile.TemporaryDirectory() as wd:
        co = extract_device_object(host_obj, wd)
        meta = kernel_metadata(co)
        check_no_spills(meta, wanted)
        # (.vgpr_count is the unified total on gfx950: architectural + accumulator registers)
        return {n: f.get("vgpr_count", -1) for n, f in meta.items() if any(w in n for w in wanted)}


# conv_pointwise.o: the streaming 1x1 kernels share a CU two or three at a time, which only works while they stay inside their
# register budget without spilling
POINTWISE_PRODUCTION = ["pw_conv_kernelILi64ELi64E", "pw_conv_kernelILi128ELi64E", "pw_conv_kernelILi256ELi32E", "pw_conv_kernelILi512ELi32E",
                        "slide3x3_c64_kernelILi1E", "slide3x3_c128_kernel"]


def verify_pointwise(host_obj):
    regs = verify_aux(host_obj, POINTWISE_PRODUCTION)
    for n, v in regs.items():
        # pw_conv_kernel<K, BP, RES, WGS, BC, NEXT>: WGS workgroups of BC/32 waves per CU must stay co-resident
        m = re.search(r"pw_conv_kernelILi\d+ELi\d+ELb[01]ELi(\d+)ELi(\d+)ELi[012]E", n)
        if m:
            waves_per_simd = int(m.group(1)) * (int(m.group(2)) // 32) / 4.0
            cap = min(512, int(512 / waves_per_simd) // 8 * 8)
        else:
            cap = 256                                          # slide3x3: two workgroups of four waves
        if v > cap:
            raise GuardError("%s uses %d registers: more than the %d that keep its workgroups co-resident" % (n, v, cap))
    return regs


PACKED_FP32_RE = r"v_pk_(?:mul|add|fma)_f32\b"


def check_no_packed_fp32(host_obj):
    """Round 6: no kernel of the library may contain packed fp32 instructions -- with them the posterior's 4x4 inverses
    came out wrong in lanes 48-63 of a wave while a convolution kernel shared the compute unit.  The sources are built with
    -fno-slp-vectorize; a toolchain that packs anyway must fail the BUILD.  Returns the number of kernels checked."""
    sections = subprocess.run([_tool("llvm-readelf"), "-S", host_obj], capture_output=True, text=True, check=True).stdout
    if ".hip_fatbin" not in sections:
        return 0                                  # host code only (engine.hip)
    with tempfile.TemporaryDirectory() as d:
        funcs = disassemble(extract_device_object(host_obj, d))
    for name, body in funcs.items():
        hits = [l for l in body if re.match(PACKED_FP32_RE, l)]
        if hits:
            raise GuardError("%s: %d packed fp32 instruction(s) in %s, e.g. `%s` (build with -fno-slp-vectorize: DESIGN.md 8.4)"
                             % (os.path.basename(host_obj), len(hits), name, hits[0]))
    return len(funcs)


def verify(host_obj, wanted=PRODUCTION, asm_kernels=INLINE_ASM_MFMA):
    """All guards on a built conv_igemm.o.  Raises GuardError; returns {kernel: vgpr_count} of the production kernels."""
    with tempfile.TemporaryDirectory() as wd:
        co = extract_device_object(host_obj, wd)
        meta = kernel_metadata(co)
        check_no_spills(meta, wanted, tolerated=[w for w in MX_ASM_MFMA if w in wanted])
        funcs = disassemble(co)
        for want in asm_kernels:
            names = [n for n in funcs if want in n]
            if not names:
                raise GuardError("kernel %s not found in the disassembly" % want)
            for n in names:
                check_inline_asm_mfma(funcs[n], want)
        for want in MX_ASM_MFMA:
            if want in wanted:
                for n in [n for n in funcs if want in n]:
                    # (a K-tile: 32 f16 or 16 block-scaled MFMAs per wave; three unrolled K-tiles of each flavour)
                    check_inline_asm_mfma(funcs[n], want, min_mfma=144, mfma_re=MX_MFMA_RE)
                    check_no_scratch_in_loop(funcs[n], want, MX_MFMA_RE)
        if any(w in INLINE_ASM_LDS for w in asm_kernels):
            funcs_a = disassemble(co, with_addr=True)
            for want in INLINE_ASM_LDS:
                for n in [n for n in funcs_a if want in n]:
                    check_asm_lds_reads(funcs_a[n], want)
        return {n: f.get("vgpr_count", -1) + f.get("agpr_count", 0) for n, f in meta.items() if any(w in n for w in wanted)}
