"""Properties of the COMPILED product kernel that its speed rests on and that no numerical test sees: the dominant
kernel must fit four workgroups per CU (<= 128 VGPRs, <= 40 KB of LDS), spill nothing, and address the image planes
through buffer resources held in SCALAR registers -- under register pressure the allocator is free to move a
descriptor into vector registers, and every load through it then becomes a waterfall loop (v_readfirstlane /
s_cbranch_execnz around each buffer_load: +30 % instructions, seen while experimenting in round 3).  hipcc
cross-compiles gfx950 without a GPU; the listing is parsed here.  Skipped where there is no hipcc."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sc-sfmlearner-release_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
PRODUCT = "_ZN5scsfm20pair_fwd_spec_kernelIfLb1ELj7ELb0ELb0"  # <float, ssim, training flags, full resolution, no forward staging>
PLAIN = "_ZN5scsfm15pair_fwd_kernelIfLb1ELb0"


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc on this machine")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "sc-sfmlearner-release_amd"))
    from scsfm_hip import build
    out = tmp_path_factory.mktemp("isa") / "pair.s"
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", str(out),
                    os.path.join(CSRC, "scsfm_pair.hip")], check=True, capture_output=True)
    return open(out).read().split("\n")


def _kernel(lines, prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().split(":")[0].startswith(prefix))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    meta = {}
    for l in lines[end:end + 400]:
        m = re.match(r";\s*(NumVgprs|ScratchSize|LDSByteSize|Occupancy|NumSGPRsForWavesPerEU):\s*(\d+)", l)
        if m and m.group(1) not in meta:
            meta[m.group(1)] = int(m.group(2))
        if l.startswith("; Occupancy"):
            break
    return lines[start:end], meta


def _waterfalls(body):
    """buffer loads that sit in a loop which re-reads its descriptor lane by lane"""
    n = 0
    for i, l in enumerate(body):
        if "buffer_load" in l and any("v_readfirstlane_b32" in x for x in body[max(0, i - 12):i]) and \
                any("s_cbranch_execnz" in x for x in body[i:i + 4]):
            n += 1
    return n


def test_the_dominant_kernel_fits_four_workgroups_per_cu_and_spills_nothing(listing):
    body, meta = _kernel(listing, PRODUCT)
    assert meta["NumVgprs"] <= 128 and meta["ScratchSize"] == 0, meta
    assert meta["LDSByteSize"] <= 160 * 1024 // 4, meta
    assert meta["Occupancy"] >= 4, meta
    loads = sum("buffer_load" in l for l in body)
    assert loads >= 90, "the image planes are read through buffer resources"
    assert _waterfalls(body) == 0
    # (the global atomics of the scatter form their 64-bit addresses in vector registers -- about 18; the image loads
    # used to add 75 to that)
    # (round 4: the window flush addresses the scatter plane with 32-bit byte offsets from its scalar base -- 16 left,
    # all in the direct-atomic fall-back of taps that miss the window)
    assert sum(l.strip().startswith("v_lshl_add_u64") for l in body) <= 20, "64-bit vector address arithmetic is back"
    flush_atomics = [l for l in body if "global_atomic_add_f32" in l and re.search(r",\s*s\[\d+:\d+\]", l)]
    assert len(flush_atomics) >= 2, "the flush's atomics no longer use the scalar-base addressing mode"
    # the kernel sits at the limit of the scalar file: anything that adds live scalars (a run-time XCD chunk size did)
    # pushes the image descriptors into vector registers -- caught by _waterfalls above; the count itself is recorded
    # (the figure includes VCC, FLAT_SCRATCH and XNACK_MASK: 102 general-purpose scalars + 6; round 6: 100 + 6)
    assert 0 < meta["NumSGPRsForWavesPerEU"] <= 108, meta
    valu = sum(1 for l in body if l.startswith("\tv_"))
    # (static count: 2599 on the path of a tile with a coherent footprint + the uniformly skipped code of the wide
    # scatter window; round 6: + ~60 for the bound of the scatter cells' unit, + ~260 behind the flush that only the
    # pair-directions carrying their target frame's smooth loss execute -- uniformly skipped by the others)
    assert valu <= 3200, f"{valu} vector instructions per thread (round 3: 2773, round 5: 2802)"


def test_the_plain_forward_spills_nothing(listing):
    body, meta = _kernel(listing, PLAIN)
    assert meta["ScratchSize"] == 0 and meta["NumVgprs"] <= 128, meta
    assert _waterfalls(body) == 0
