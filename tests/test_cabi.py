"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/fastvim_hip.h declares, and the product has no CPU fallback."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fastvim_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import fastvim_amd.build as fb
    fb.build()
    from fastvim_amd import _lib
    lib = _lib.lib()
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in include/fastvim_hip.h but not exported"
    assert sorted(_lib.C_ABI_SYMBOLS) == declared
    # the library and the header it was built from agree on the ABI version (bumped with every signature change)
    hdr = open(os.path.join(ROOT, "include", "fastvim_hip.h")).read()
    assert lib.fv_version() == int(re.search(r"#define\s+FV_ABI_VERSION\s+(\d+)", hdr).group(1)) == 3


def test_no_cpu_fallback():
    from fastvim_amd.selective_scan_interface import selective_scan_fn
    u = torch.randn(1, 2, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        selective_scan_fn(u, u, -torch.rand(2, 4), torch.randn(1, 4, 4), torch.randn(1, 4, 4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "fastvim_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_host_side_launch_planning_functions():
    """The size / plan queries of the C ABI are plain host code (no HIP call): checked here without a GPU."""
    import ctypes
    from fastvim_amd import _lib
    lib = _lib.lib()
    i = ctypes.c_int
    # segment-parallel forward scan: only long sequences on few batch elements are cut, into >= 8-chunk segments
    seg = lambda b, lc, d, r: lib.fv_mixer_scan_fwd_segments(i(b), i(lc), i(d), i(r))
    s_vim = seg(8, 16392, 384, 12)                      # un-pooled Vim-T at 2048 px, batch 8: 96 workgroups, 1025 chunks
    assert 8 <= s_vim <= 1025 // 8 and s_vim * 96 >= 1024
    assert seg(128, 200, 384, 12) == 1                  # Vim-T 224 px batch 128: 1536 workgroups already
    assert seg(8, 128, 1536, 48) == 1                   # FastVim-B 2048 px: 8 chunks, nothing to cut
    assert seg(128, 14, 384, 12) == 1 and seg(1, 4104, 192, 80) == 1      # short; dt_rank > 48 (generic kernel)
    assert seg(1, 4104, 192, 6) == 29                   # 257 chunks: at most 32 segments -> 9 chunks each -> 29 non-empty ones
    # the backward scan follows the forward rule on the REAL d_inner; one partial row per (batch, segment)
    assert lib.fv_mixer_scan_bwd_segments(i(8), i(16392), i(384), i(12)) == s_vim
    assert lib.fv_mixer_scan_bwd_seg_partials(i(8), i(16392), i(384), i(12)) == 8 * s_vim
    # d_inner != 32 dt_rank (explicit dt_rank / expand != 2): every size query describes the launch that is made --
    # segment-parallel launches walk 64-channel workgroups, and dx_dbl has exactly one slice per workgroup column
    for b, lc, d, r in ((32, 1000, 768, 4), (2, 1000, 768, 4), (8, 16392, 384, 12), (64, 37, 384, 12), (2, 4104, 1536, 8)):
        S = lib.fv_mixer_scan_bwd_segments(i(b), i(lc), i(d), i(r))
        assert S == seg(b, lc, d, r)
        assert lib.fv_mixer_scan_bwd_seg_floats(i(b), i(lc), i(d), i(16), i(r)) == (2 * b * S * d * 33 if S > 1 else 0)
        ch = lib.fv_mixer_scan_bwd_seg_chunks(i(b), i(d), i(lc), i(r), i(int(S > 1)))
        assert ch == (-(-d // 64) if S > 1 else lib.fv_mixer_scan_bwd_chunks_b(i(b), i(d), i(lc), i(r)))
    assert seg(32, 1000, 768, 4) == 1 and seg(2, 1000, 768, 4) > 1
    assert lib.fv_mixer_scan_bwd_seg_floats(i(128), i(200), i(384), i(16), i(12)) == 0
    n = lib.fv_mixer_scan_fwd_seg_floats(i(8), i(16392), i(384), i(16), i(12))
    assert n == 2 * 8 * s_vim * 384 * (2 * 16 + 1)
    assert lib.fv_mixer_scan_fwd_seg_floats(i(128), i(200), i(384), i(16), i(12)) == 0
    # checkpoints of the chunked scan: one state per 16-step chunk
    assert lib.fv_mixer_scan_ckpt_floats(i(8), i(16392), i(384), i(16), i(12)) == 2 * 8 * 1025 * 384 * 16
    # the fused in_proj-data-gradient kernel: one partial row of the norm weight's gradient per 64-row workgroup
    assert lib.fv_gemm_bf16_dgrad_addnorm_blocks(i(25088)) == 392 and lib.fv_gemm_bf16_dgrad_addnorm_blocks(i(65)) == 2
