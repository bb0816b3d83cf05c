"""CPU (hipcc cross-compile): the production configurations of the implicit-GEMM kernel must not spill registers.
A spill in the 256x256 tile costs ~15 % of the whole pipeline and does not show up in any functional test."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCTION = [     # <BC, BP, WC, WP, ABL=0, XR, SPLIT>
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi0ELb1ELb0EE",     # head towers (row reuse)
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi5ELb1ELb0EE",     # first tower layer, N-way fan-out, on the row-reuse loop
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi0ELb0ELb0EE",     # backbone / FPN, big tile
    "conv_igemm_kernelILi128ELi128ELi2ELi2ELi0ELb0ELb0EE",     # fan-out layer, small layers, split-K
    "conv_igemm_kernelILi64ELi128ELi1ELi4ELi0ELb0ELb0EE",
    "conv_igemm_kernelILi256ELi256ELi2ELi4ELi0ELb0ELb1EE",     # bf16x3 precision: the three tile configurations
    "conv_igemm_kernelILi128ELi128ELi2ELi2ELi0ELb0ELb1EE",
    "conv_igemm_kernelILi64ELi128ELi1ELi4ELi0ELb0ELb1EE",
]


def test_production_conv_kernels_do_not_spill(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "bayes-od-rc_amd", "csrc", "conv_igemm.hip")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.dirname(src), "-c", src,
                          "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", out.stderr)[1:]
    seen = {}
    for b in blocks:
        name = b.split()[0]
        m = re.search(r"VGPRs Spill: (\d+)", b)
        s = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
        seen[name] = (int(m.group(1)), int(s.group(1)))
    for want in PRODUCTION:
        hits = [(n, v) for n, v in seen.items() if want in n]
        assert hits, "kernel %s not found in the compile remarks" % want
        for n, (spill, scratch) in hits:
            assert spill == 0 and scratch == 0, "%s spills %d VGPRs (%d B scratch/lane)" % (n, spill, scratch)


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def test_inline_asm_mfma_accumulators_are_untouched_inside_the_tower_loop(tmp_path):
    """The row-reuse tower loop issues its MFMAs as inline asm with the accumulators tied in place.  The compiler neither knows
    their result latency nor inserts the wait states a real MFMA would get, so any compiler-generated instruction that reads or
    writes an accumulator register between the loop's first and last MFMA (a v_mov that merges two live ranges, say -- seen when
    a branch was put around the last fragment's MFMAs) silently corrupts results.  The disassembly of the production kernels must
    not contain one; the epilogue reads the accumulators behind explicit s_nops."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "bayes-od-rc_amd", "csrc", "conv_igemm.hip")
    asm = str(tmp_path / "conv.s")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.dirname(src), "-S", "--cuda-device-only",
                          src, "-o", asm], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    text = open(asm).read()
    checked = 0
    for want in PRODUCTION[:2]:                      # the two kernels on the 16x16x32 row-reuse loop
        m = re.search(r"^(_Z17%sv8ConvArgs):" % want, text, re.M)
        assert m, want
        body = [l.strip() for l in text[m.end():text.find(".Lfunc_end", m.end())].split("\n")]
        body = [l for l in body if l and not l.startswith(";")]
        mf = [i for i, l in enumerate(body) if l.startswith("v_mfma_f32_16x16x32_bf16")]
        assert len(mf) >= 192, (want, len(mf))       # three unrolled K-tiles of 64
        # For every MFMA: no other instruction may read or write its destination in the next WINDOW issue slots (an s_nop k counts
        # k + 1 slots; the epilogue reads the accumulators behind two s_nop 15).  MFMAs accumulating into the same registers are the
        # only legal users.  (The register ROLES may differ between a peeled first K-tile and the steady-state loop, so the check
        # follows each MFMA's own destination instead of one global accumulator set.)
        WINDOW = 16
        # ... and the other direction: a VALU instruction that writes a register the MFMA reads (SrcA, SrcB or the tied SrcC) needs two
        # wait states before the MFMA (tests/tools/mfma_war_probe.hip: 18 % wrong results with 0 or 1 slot in between, none from 2 on);
        # the compiler inserts them for a real MFMA and cannot for inline asm.  (An LDS load that RETURNS into a source register right
        # behind the MFMA is safe: same probe.)
        for i in mf:
            srcs = set()
            for tok in re.findall(r"v\[\d+:\d+\]", body[i]):
                srcs |= _regs(tok)
            slots, k = 0, i - 1
            while k >= 0 and slots < 2:
                l = body[k]
                if l.endswith(":") or l.startswith("."):  # a label: the fall-through predecessor is checked (other paths end in a branch)
                    k -= 1
                    continue
                mm = re.match(r"s_nop (\d+)", l)
                if mm:
                    slots += int(mm.group(1)) + 1
                else:
                    if l.startswith("v_") and not l.startswith("v_mfma") and not l.startswith("v_cmp"):
                        toks = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l)
                        if toks:
                            assert not (_regs(toks[0]) & srcs), "%s: `%s` writes a source of `%s` %d slot(s) before it" % (want, l, body[i], slots)
                    slots += 1
                k -= 1
        for i in mf:
            dst = _regs(re.match(r"v_mfma_f32_16x16x32_bf16 (v\[\d+:\d+\])", body[i]).group(1))
            slots, k = 0, i + 1
            while k < len(body) and slots < WINDOW:
                l = body[k]
                mm = re.match(r"s_nop (\d+)", l)
                slots += int(mm.group(1)) + 1 if mm else 1
                if not l.startswith("v_mfma_f32_16x16x32_bf16") and re.match(r"(v_|ds_|buffer_|global_|flat_)", l):
                    toks = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l)
                    touched = set().union(*[_regs(t) for t in toks]) if toks else set()
                    assert not (touched & dst), "%s: `%s` touches the destination of `%s` %d slots behind it" % (want, l, body[i], slots)
                if l.startswith("s_cbranch") or l.startswith("s_branch") or l.startswith("s_endpgm"):
                    break                                # (fall-through only: the check is per straight-line run)
                k += 1
            checked += 1
    assert checked >= 2 * 192
