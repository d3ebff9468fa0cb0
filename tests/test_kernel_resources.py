"""Every kernel in the built library keeps its vector registers: no VGPR spill and no scratch (a few SGPRs parked in
VGPR lanes -- the feature-split backward variants -- are tolerated and counted).  (A structural edit once
made the pair kernels' address arithmetic loop-invariant, LLVM hoisted it across both wave roles, and the backward
kernel spilled 60 registers -- one reloaded per tile on its critical path -- without any test noticing.)  The code objects
are cut out of the library's clang offload bundles and read with llvm-readelf; no GPU needed."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "flash-molecular-dynamics_amd", "flashmd_amd", "libflashmd_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    """(triple, bytes) of every device code object in the uncompressed offload bundles of `blob`."""
    out, at = [], 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            return out
        (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, blob[at + off:at + off + size]))
        at += len(MAGIC)


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="library or llvm-readelf missing")
def test_no_kernel_spills_or_uses_scratch(tmp_path):
    objs = _code_objects(open(LIB, "rb").read())
    assert objs, "no device code objects found in the library"
    kernels, bad = 0, []
    for k, (triple, data) in enumerate(objs):
        assert "gfx950" in triple, triple
        path = tmp_path / f"co{k}.o"
        path.write_bytes(data)
        notes = subprocess.run([READELF, "--notes", str(path)], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            vals = {key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
                    for key in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
            kernels += 1
            if vals["vgpr_spill_count"] or vals["private_segment_fixed_size"] or vals["sgpr_spill_count"] > 8:
                bad.append((name, vals))
    assert kernels > 100, kernels      # 118 kernels at the time of writing: operators and engine variants
    assert not bad, bad
