"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads here (no GPU) and exports
every symbol include/lgd_hip.h declares; kernel calls without a GPU fail loudly (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from lgd_amd import hip
    return hip.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "lgd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lgd_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 10
    raw = ctypes.CDLL(os.path.join(ROOT, "lgd_amd", "_lib", "liblgd_hip.so"))
    for n in names:
        assert hasattr(raw, n), "missing export: " + n


def test_binding_table_matches_header(lib):
    from lgd_amd import hip
    assert sorted(hip.SIGNATURES) == _declared()


def test_version_and_arch(lib):
    from lgd_amd import hip
    assert lib.lgd_abi_version() == hip.ABI_VERSION
    assert lib.lgd_arch() == b"gfx950"


def test_size_helpers_no_gpu_needed(lib):
    L, B, T, mx = 5, 2, 22, 11
    assert lib.lgd_geom_ints(L, B, T, mx) == L * B * mx * 4 + L * B + L * B * (2 * mx + 2)
    from lgd_amd import hip
    hw = hip.int_array([64, 64, 32, 32])
    # chunks of 4096 elements: 1 chunk per plane at both levels; 5 doubles per chunk + block terms
    assert lib.lgd_distill_ws_doubles(hw, 2, 2, 256) == 2 * 2 * 256 * 5 + (2 * 2 * 256 + 255) // 256


def test_invalid_arguments_are_rejected(lib):
    from lgd_amd import hip
    hw = hip.int_array([8, 8])
    assert lib.lgd_box_prep(None, None, 1, 1, 1, 64, 64, hw, 1, None, None) == -1
    assert lib.lgd_box_sum(None, hw, 1, 1, 256, 1, 1, None, None, None, None, 1, 0, None) == -1
    assert lib.lgd_box_sum(None, hw, 17, 1, 256, 1, 1, None, None, None, None, 1, 0, None) == -1  # L > LGD_MAX_LEVELS


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu(lib):
    from lgd_amd import hip, ops
    with pytest.raises(hip.LgdHipError):
        ops.BoxGeometry(torch.zeros(1, 4), [1], (64, 64), [(8, 8)])
    with pytest.raises(hip.LgdHipError):
        ops.distill_in_mse([torch.zeros(1, 4, 8, 8)], [torch.zeros(1, 4, 8, 8)], 1.0)


def test_no_packed_fp32_in_the_library(lib):
    """Round 6: a wave's packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) return wrong low bits while a wave of ANOTHER kernel
    on the same CU issues f16 MFMAs -- found as wino6_out changing a few hundred elements per step beside h2_fwd on a side stream
    (profiles/r06_packed_fp32_beside_mfma.txt; tools/conv_stage_probe.py: 59 of 60 runs differ from the quiet run with the packed forms, 0 of 100
    without).  The library is built with `-target-feature -packed-fp32-ops` (__graft_entry__.FLAGS): the gfx950 code of the BUILT .so must not
    hold a single one of them, whatever a later edit or compiler picks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("scan_packed_fp32", os.path.join(ROOT, "tools", "scan_packed_fp32.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    found, n = mod.scan(os.path.join(ROOT, "lgd_amd", "_lib", "liblgd_hip.so"))
    assert n > 100000, n          # (every translation unit's code object was scanned)
    assert not found, sorted(found.items(), key=lambda kv: -kv[1])[:5]


# ---- the counted waits of the LDS-DMA pipelines ("nothing enforces this at build time")
def _asm_of(src):
    """hipcc -S of one csrc file with the library's flags (cached under build/asm while the source is older)"""
    import subprocess
    import __graft_entry__ as g
    out_dir = os.path.join(ROOT, "build", "asm")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(ROOT, "lgd_amd", "csrc", src)
    import hashlib
    out = os.path.join(out_dir, src.replace(".hip", "") + "." + hashlib.sha1(" ".join(g.FLAGS).encode()).hexdigest()[:8] + ".s")   # (the flags are part of the cache key)
    deps = [path, os.path.join(ROOT, "lgd_amd", "csrc", "common.h"), os.path.join(ROOT, "lgd_amd", "csrc", "winograd.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        flags = [f for f in g.FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call([g.HIPCC] + flags + ["-S", "--cuda-device-only", path, "-o", out])
    return open(out).read()


def _kernels(asm, pattern):
    """{mangled name: (instruction lines, ScratchSize)} of the kernels whose name matches"""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:(.*?)^; ScratchSize: (\d+)", asm, flags=re.S | re.M):
        if re.search(pattern, m.group(1)):
            lines = [l.strip() for l in m.group(2).splitlines() if l.strip() and not l.strip().startswith(";")]
            out[m.group(1)] = (lines, int(m.group(4)))
    return out


def _mfma_loops(lines):
    """bodies (lists of lines) of the loops that contain MFMAs: from a label to the LAST branch (conditional or not) back to it"""
    pos = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    last = {}
    for i, l in enumerate(lines):
        m = re.match(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in pos and pos[m.group(1)] < i:
            last[m.group(1)] = i
    loops = []
    for lab, i in sorted(last.items(), key=lambda kv: pos[kv[0]]):
        body = lines[pos[lab]:i]
        if any("v_mfma" in b for b in body) and not any(pos[o] <= pos[lab] and last[o] >= i and o != lab for o in last):
            loops.append(body)
    return loops


_VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")


def test_h2_kloop_has_only_counted_waits():
    """csrc/h2.hip issues its LDS-DMA from inline asm and counts its own waits: inside each k-loop there must be exactly DPW LDS-DMA
    instructions per wave, NO other vector-memory instruction, exactly one vmcnt wait -- for DPW (forward: the youngest group may still be in
    flight) / 2 DPW (weight gradient, four buffers) -- no scratch, and the transposing LDS reads must have survived."""
    asm = _asm_of("h2.hip")
    want = {"h2_fwd_kernelILi256ELb0": (6, 6), "h2_fwd_kernelILi256ELb1": (6, 6), "h2_fwd_kernelILi128ELb0": (4, 4), "h2_fwd_kernelILi128ELb1": (4, 4),
            "h2_dw_kernel": (4, 8)}
    ks = _kernels(asm, r"h2_fwd_kernel|h2_dw_kernel")
    assert len(ks) == 5, sorted(ks)
    for name, (lines, scratch) in ks.items():
        key = next(k for k in want if k in name)
        dpw, wait = want[key]
        assert scratch == 0, (name, scratch)
        loops = _mfma_loops(lines)
        assert len(loops) == 1, (name, len(loops))
        body = loops[0]
        dma = [l for l in body if re.match(r"buffer_load_dwordx4 .* lds$", l)]
        vmem = [l for l in body if _VMEM.match(l)]
        waits = [int(m.group(1)) for l in body for m in [re.search(r"vmcnt\((\d+)\)", l)] if m and l.startswith("s_waitcnt")]
        assert len(dma) == dpw and len(vmem) == dpw, (name, len(dma), len(vmem))
        assert waits == [wait], (name, waits)
        assert sum("v_mfma_f32_32x32x16_f16" in l for l in body) == (12 if "ILi128E" in name else 24), name
        if "fwd" in name:
            assert sum("ds_read_b64_tr_b16" in l for l in body) == 8, name


def test_h2_dma_statements_own_m0():
    """the inline-asm LDS-DMA of csrc/h2.hip writes m0.  (1) Source: every asm statement under csrc/ whose text names m0 lists
    it among its clobbers.  (2) Object: in the product kernels every LDS-DMA instruction has its own `s_mov_b32 m0` directly in front of it (at most
    the hazard nop between), and nothing else in those kernels writes or implicitly reads m0 (no other `... lds` form, v_movrel, s_sendmsg,
    ds_gws): the compiler keeps no value of its own in m0 across the statements."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "lgd_amd", "csrc", "*.h*")):
        src = open(path).read()
        for m in re.finditer(r"asm\s*(?:volatile)?\s*\((.*?)\);", src, flags=re.S):
            body = m.group(1)
            if re.search(r"\bm0\b", body.split(":")[0]):
                clobbers = body.rsplit(":", 1)[1]
                assert '"m0"' in clobbers, (os.path.basename(path), body[:80])
    asm = _asm_of("h2.hip")
    ks = _kernels(asm, r"h2_fwd_kernel|h2_dw_kernel")
    assert len(ks) == 5
    for name, (lines, _) in ks.items():
        dma = [i for i, l in enumerate(lines) if re.match(r"buffer_load_dwordx4 .* lds$", l)]
        assert dma, name
        for i in dma:
            prev = [l for l in lines[max(0, i - 2):i]]
            assert any(l.startswith("s_mov_b32 m0,") for l in prev), (name, lines[max(0, i - 3):i + 1])
        writes = [l for l in lines if re.match(r"s_\w+ m0,", l)]
        assert len(writes) == len(dma) and all(l.startswith("s_mov_b32 m0,") for l in writes), (name, len(writes), len(dma))
        others = [l for l in lines if re.match(r"(v_movrel|s_sendmsg|ds_gws|ds_\w+_gs_reg|global_load_lds|buffer_load_(dword|ubyte|ushort) .* lds$)", l)]
        assert not others, (name, others[:3])


def test_gemm3_kernels_keep_their_counted_waits_and_no_scratch():
    """csrc/gemm3.hip: the k-loop waits `vmcnt(8)` for the image DMA of the next k-step, relying on exactly 8 B loads having been issued behind
    it.  Build-time guard on every instance: no scratch traffic inside the counted window (a spill reloaded there would be an extra load), the counted
    wait still stands directly in front of the k-loop's barriers, the LDS-DMA and the 8-load groups are there in the expected numbers.  (The
    run-time side -- products under memory pressure from a second stream, bit-equal to the quiet run -- is tests/test_kernels_gpu.py::
    test_gemm3_and_h2_products_under_load.)"""
    asm = _asm_of("gemm3.hip")
    ks = _kernels(asm, r"gemm3_kernel")
    assert len(ks) == 15, sorted(ks)   # bf16x3: {256, 128 rows} x 5 epilogues; f16x2: 128 rows x 5 epilogues (it never takes the 256-row tile)
    for name, (lines, scratch) in ks.items():
        pcs = 2 if name.endswith("ELi2EEEvNS0_6ParamsE") else 3
        ch = pcs * (8 if "ILi256E" in name else 4) // 4   # LDS-DMA instructions per wave and k-step (pieces x BM / 32 row blocks / 4 waves)
        dma = [i for i, l in enumerate(lines) if re.match(r"buffer_load_dwordx4 .* lds$", l)]
        assert dma and len(dma) % ch == 0, (name, len(dma))
        def to_barrier(i):   # the wait reaches a barrier within a few instructions, none of them a vector-memory one
            for x in lines[i + 1:i + 9]:
                if x.startswith("s_barrier"):
                    return True
                if _VMEM.match(x):
                    return False
            return False
        counted = [i for i, l in enumerate(lines) if l.startswith("s_waitcnt vmcnt(8)") and to_barrier(i)]
        assert len(counted) >= 2, (name, len(counted))   # prologue + k-loop
        # no scratch traffic inside the counted window (first LDS-DMA .. the k-loop's last counted wait): a spill reloaded there would be one more load
        # in the memory pipe than the waits count.  (Round 6: without packed fp32 instructions one epilogue instance -- <128, residual, f16x2>, held to
        # 168 registers -- spills 28 bytes in its residual prologue and its epilogue; that is outside the window and stays allowed, but small.)
        spills = [i for i, l in enumerate(lines) if l.startswith("scratch_")]
        assert all(i < dma[0] or i > counted[-1] for i in spills), (name, [i for i in spills if dma[0] <= i <= counted[-1]][:4], dma[0], counted[-1])
        assert scratch <= 64, (name, scratch)
        # every DMA group is followed by exactly 8 dword loads of B before anything else enters the memory pipe or a wait is taken
        groups = [dma[j] for j in range(ch - 1, len(dma), ch)]
        ok = 0
        for g in groups:
            n = 0
            for l in lines[g + 1:g + 200]:
                if re.match(r"buffer_load_dword v", l):
                    n += 1
                elif _VMEM.match(l) or l.startswith("s_waitcnt vmcnt") or l.startswith("s_barrier"):
                    break
            ok += n == 8
        assert ok >= 1, (name, ok, len(groups))
