"""The built gfx950 code objects inside tvidz_amd/libtvz.so, read without a GPU: the hot kernels must
not touch scratch (`ts_match_index_kernel`, `ts_find_fused_kernel` and `ts_match_join_kernel`
carried 4..12-byte spill pairs at the 64-VGPR cap) and must keep the register budget their launch shapes
assume (8 waves per SIMD = at most 64 VGPRs).  Parsed here: the clang offload bundles in .hip_fatbin ->
the gfx950 ELF -> its NT_AMDGPU_METADATA note (msgpack)."""
import os
import struct

import msgpack
import pytest

from tvidz_amd import build as tbuild

HOT = ("ts_match_index_kernel", "ts_match_index_topk_kernel", "ts_find_fused_kernel", "ts_match_join_kernel",
       "ts_match_q1_kernel", "ts_match_tile_kernel", "luma_sad_flat_kernel", "ts_topk_wave_kernel",
       "ts_topk_merge_sorted_kernel", "ts_match_wq_topk_kernel", "bk_slice_build_kernel")
AT_MOST_64_VGPRS = ("ts_match_index_kernel", "ts_match_index_topk_kernel", "ts_find_fused_kernel", "ts_match_join_kernel")


def _code_objects(blob: bytes):
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = blob.find(magic)
    while at >= 0:
        n = struct.unpack_from("<Q", blob, at + 24)[0]
        o = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tlen].decode()
            o += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at = blob.find(magic, at + 1)


def _kernel_metadata(elf: bytes):
    assert elf[:4] == b"\x7fELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = elf[shoff + i * shentsize: shoff + (i + 1) * shentsize]
        sh_type, = struct.unpack_from("<I", sh, 4)
        off, size = struct.unpack_from("<QQ", sh, 0x18)
        if sh_type != 7:                                    # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + ((namesz + 3) & ~3)
            if name == b"AMDGPU" and ntype == 32:           # NT_AMDGPU_METADATA
                md = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                yield from md["amdhsa.kernels"]
            p = d0 + ((descsz + 3) & ~3)


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(tbuild.SO):
        pytest.skip("libtvz.so is not built")
    out = {}
    for co in _code_objects(open(tbuild.SO, "rb").read()):
        for k in _kernel_metadata(co):
            out[k[".name"]] = k
    assert out, "no gfx950 code object found in libtvz.so"
    return out


def test_hot_kernels_use_no_scratch(kernels):
    seen = set()
    for name, k in kernels.items():
        for h in HOT:
            if h in name:
                seen.add(h)
                assert k[".private_segment_fixed_size"] == 0, (name, k[".private_segment_fixed_size"])
                assert k.get(".vgpr_spill_count", 0) == 0, (name, k)      # (SGPRs parked in VGPR lanes are not scratch)
    assert seen == set(HOT), set(HOT) - seen


def test_eight_waves_per_simd_kernels_fit_64_vgprs(kernels):
    n = 0
    for name, k in kernels.items():
        if any(h in name for h in AT_MOST_64_VGPRS):
            n += 1
            assert k[".vgpr_count"] <= 64, (name, k[".vgpr_count"])
            assert k.get(".vgpr_spill_count", 0) == 0, (name, k)
    assert n >= 10          # index x {host, device} x {M2, Top5, Count}, top-k x 2, fused x 2, join


def test_one_wave_lookup_keeps_three_waves_per_simd(kernels):
    """ts_match_wq_topk_kernel holds a query's postings in registers between its passes; its LDS (~16 KiB per wave) admits
    ten waves per CU, which needs three per SIMD: at most 168 VGPRs, and no scratch (checked above)."""
    n = 0
    for name, k in kernels.items():
        if "ts_match_wq_topk_kernel" in name:
            n += 1
            assert k[".vgpr_count"] <= 168, (name, k[".vgpr_count"])
    assert n == 2


def test_static_lds_of_the_single_query_sweep_is_what_its_launch_guard_assumes(kernels):
    """launch_q1 (tvz_match.hip) refuses a launch whose dynamic + static LDS exceeds the 160 KiB of a
    gfx950 workgroup, with kQ1StaticLds = kQ1Stage * 12 + 64 = 3,136 B for the static part."""
    seen = 0
    for name, k in kernels.items():
        if "ts_match_q1_kernel" in name:
            seen += 1
            assert k[".group_segment_fixed_size"] <= 256 * 12 + 64, (name, k[".group_segment_fixed_size"])
    assert seen == 6
