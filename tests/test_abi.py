"""CPU-side checks of the drop-in boundary: libgfmatch.so loads, exports every
symbol include/gfmatch.h declares, and fails loudly (no CPU fallback) when no
HIP device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "gfmatch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(gf_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_boundary():
    names = _declared_functions()
    for must in ("gf_index_build", "gf_index_free", "gf_map_reads", "gf_map_read", "gf_map_reads_device",
                 "gf_compact_hits_device", "gf_in_required_direction", "gf_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from genefuserust_amd import _lib
    L = _lib.lib()
    for name in _declared_functions():
        assert hasattr(L, name), "libgfmatch.so does not export %s" % name
    assert b"gfx950" in L.gf_version()


def test_integration_doc_binds_every_entry_point():
    """INTEGRATION.md's Rust `extern "C"` block names every function include/gfmatch.h declares."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index('extern "C" {'):doc.index("pub fn last_error()")]
    bound = set(re.findall(r"pub fn (gf_[a-z0-9_]+)\(", block))
    missing = [n for n in _declared_functions() if n not in bound]
    assert not missing, missing


def test_struct_layouts_match_header():
    from genefuserust_amd import _lib
    assert C.sizeof(_lib.GfSeqMatch) == 16 and C.sizeof(_lib.GfHit) == 48 and C.sizeof(_lib.GfPairHit) == 64
    assert _lib.PAIR_HIT_DTYPE.itemsize == 64
    assert _lib.SEQMATCH_DTYPE.itemsize == 16 and _lib.HIT_DTYPE.itemsize == 48
    assert C.sizeof(_lib.GfOptions) == 32
    assert C.sizeof(_lib.GfIndexInfo) == 88


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from genefuserust_amd import Indexer, _lib
    ix = Indexer.from_gene_slices([b"ACGT" * 100])
    with pytest.raises(_lib.GfError) as e:
        ix.make_index()
    assert e.value.code in (_lib.GF_ERR_NO_DEVICE, _lib.GF_ERR_HIP)
    with pytest.raises(RuntimeError):
        ix.map_read(b"ACGT" * 40)


def test_argument_errors_without_gpu():
    from genefuserust_amd import _lib
    L = _lib.lib()
    assert L.gf_index_build(None, None, -1, None, None) == _lib.GF_ERR_ARG
    assert L.gf_compact_workspace_bytes(0) >= 0
    assert L.gf_compact_workspace_bytes(4096 * 10) >= 10 * 12
    m = (_lib.GfSeqMatch * 2)()
    assert L.gf_in_required_direction(m, 1, None, 0) == 0


def test_in_required_direction_host_logic_matches_oracle(oracle):
    """indexer.rs:541-608 is pure host logic: exhaustive small grid vs both restatements."""
    from genefuserust_amd import Fusion, Gene, GenePos, Indexer, SeqMatch
    from oracle import indexer_model as M
    flags = [False, True, False, True]
    ix = Indexer(None, [Fusion(Gene("g%d" % i, "", 0, 0, f)) for i, f in enumerate(flags)])
    rng = np.random.default_rng(0)
    for _ in range(400):
        m = []
        for k in range(int(rng.integers(0, 3))):
            s = int(rng.integers(0, 100))
            m.append((s, s + 30, int(rng.integers(0, 4)), int(rng.choice([-500, -1, 0, 1, 700]))))
        want = oracle.in_required_direction(m, flags)
        assert want == M.in_required_direction(m, flags)
        got = ix.in_required_direction([SeqMatch(a, b, GenePos(c, p)) for a, b, c, p in m])
        assert got == want, m
