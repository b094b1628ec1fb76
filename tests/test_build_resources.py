"""band_diag_kernel must not spill: it is built at 127 - 128 VGPRs (four wavefronts per SIMD), and a single spilled dword makes the runtime
set up private memory at every launch of every context.  hipcc cross-compiles without a GPU: this reads the compiler's own resource remarks
for every instantiation of the kernel (entry width x mask words) in the production build of vtx_band.hip."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_band_diag_kernel_has_no_scratch_and_keeps_four_wavefronts():
    src = os.path.join(ROOT, "vartrix_amd", "csrc", "vtx_band.hip")
    with tempfile.TemporaryDirectory() as td:
        p = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result",
                            "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.path.join(td, "b.o"), src],
                           capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", p.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "band_diag_kernel" not in name:
            continue
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert scratch == 0 and vgprs <= 128 and occ >= 4, (name, vgprs, scratch, occ)
        seen += 1
    assert seen == 4, seen          # two-byte / four-byte match entries x three / four mask words
