"""No hot kernel may spill registers to scratch memory — checked at build time, on the CPU (hipcc cross-compiles gfx950 here).

Round 5 lost 0.8 ms of the AHDS step for an afternoon to one refactor (a wave-uniform branch hoisted out of an eight-element loop
inside a device function that is inlined into the convolution kernels): the register allocator answered with 144-192 bytes of scratch
per lane in `conv3x3_kernel`, the kernels still passed every numerics test, and only a same-box A/B showed it.  This test recompiles
the kernel sources with `-Rpass-analysis=kernel-resource-usage` (the Makefile's flags) and fails on any kernel whose `ScratchSize` is
not zero, except the listed ones."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gaussianip_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernel-name substring -> why scratch is accepted there
ALLOWED = {
    "gip_tile_sort_kernel": "16 bytes in the >= 2048-entry phase of a 1024-thread workgroup (128-register budget); off the critical path",
    "attn_fwd_kernelILi80ELi2E": "two-accumulator form for a second key set longer than 64 keys: never launched by the reference's shapes",
    "attn_fwd_kernelILi160ELi2E": "same",
}
# source -> extra flags (the Makefile's EXACT / FAST / NOSLP_* / NNFLAGS_* lines)
SOURCES = {
    "preprocess.hip": ["-ffp-contract=off"], "binning.hip": ["-ffp-contract=off"], "sh_mfma.hip": ["-ffp-contract=off"],
    "render_forward.hip": ["-ffp-contract=fast", "-fno-slp-vectorize"], "render_backward.hip": ["-ffp-contract=fast", "-fno-slp-vectorize"],
    "gather_backward.hip": ["-ffp-contract=fast"], "conv3x3.hip": ["-ffp-contract=fast"], "groupnorm.hip": ["-ffp-contract=fast"],
    "attention.hip": ["-ffp-contract=fast"], "conv_small.hip": ["-ffp-contract=fast"], "winograd.hip": ["-ffp-contract=fast"],
    "softmax.hip": ["-ffp-contract=fast"], "guidance_glue.hip": ["-ffp-contract=off"],
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src", sorted(SOURCES))
def test_kernels_do_not_spill(src, tmp_path):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + SOURCES[src] + [
        "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", str(tmp_path / "o.o")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    name, found, bad = None, 0, []
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
        if m and name:
            found += 1
            if int(m.group(1)) and not any(k in name for k in ALLOWED):
                bad.append((name, int(m.group(1))))
    assert found > 0, "no kernel-resource-usage remarks: " + r.stderr[-500:]
    assert not bad, "kernels spilling to scratch: %s" % bad
