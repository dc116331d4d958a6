"""CPU-side checks of the C ABI: the library loads, exports every declared symbol, and the
host-only planners tile every row of every tensor exactly once."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from spice_amd import _native as nat
from spice_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "spice_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(spice_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 14
    handle = ctypes.CDLL(nat.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/spice_amd.h but not exported"
        assert n in nat.SIGNATURES, f"{n} has no ctypes signature"
    assert nat.lib().spice_abi_version() == nat.ABI_VERSION
    assert b"argument" in nat.lib().spice_error_string(nat.ERR_ARG)


def test_struct_sizes_match_header_layout():
    assert ctypes.sizeof(nat.RowsItem) == 40
    assert ctypes.sizeof(nat.ColsItem) == 32
    assert ctypes.sizeof(nat.ColsFold) == 32
    assert ctypes.sizeof(nat.GatherItem) == 64
    assert ctypes.sizeof(nat.KdliteParams) == 32


def _items(plan, struct):
    raw = plan.items.cpu().numpy().tobytes()
    n = plan.n_items
    arr = (struct * n).from_buffer_copy(raw[: ctypes.sizeof(struct) * n])
    return list(arr)


def test_rows_plan_covers_every_row_once():
    layers = [ops.RowsLayout(0, 0, 0, 64, 147), ops.RowsLayout(9408, 9408, 64, 512, 4608),
              ops.RowsLayout(9408 + 512 * 4608, 9408 + 512 * 4608, 576, 2048, 9),
              ops.RowsLayout(3_000_000, 3_000_000, 2624, 5, 256), ops.RowsLayout(3_100_000, 3_100_000, 2629, 0, 7)]
    plan = ops.RowsPlan(layers, torch.device("cpu"))
    seen = np.zeros(2629, dtype=np.int32)
    for it in _items(plan, nat.RowsItem):
        lay = [l for l in layers if l.rows and l.out_off <= it.out_off < l.out_off + l.rows][0]
        r0 = it.out_off - lay.out_off
        assert it.row_len == lay.row_len
        assert it.w_off == lay.w_off + r0 * lay.row_len
        assert it.lanes in (1, 2, 4, 8, 16, 32, 64) and it.n_rows > 0
        if it.flags & 1:
            assert it.row_len % 4 == 0 and it.w_off % 4 == 0
        seen[it.out_off: it.out_off + it.n_rows] += 1
    assert (seen == 1).all()
    assert plan.weight_elems == 64 * 147 + 512 * 4608 + 2048 * 9 + 5 * 256


def test_rows_plan_rejects_bad_input():
    with pytest.raises(ValueError):
        ops.RowsPlan([ops.RowsLayout(0, 0, 0, -1, 4)], torch.device("cpu"))


def test_gather_plan_covers_every_row_once():
    specs = [ops.GatherSpec(0, 0, 0, -1, 100, 64, 64, 9), ops.GatherSpec(40000, 60000, -1, 100, 256, 30, 64, 1),
             ops.GatherSpec(90000, 90000, 130, -1, 17, 1, 1, 1)]
    plan = ops.GatherPlan(specs, torch.device("cpu"))
    counts = {0: np.zeros(100, int), 1: np.zeros(256, int), 2: np.zeros(17, int)}
    for it in _items(plan, nat.GatherItem):
        k = [i for i, s in enumerate(specs) if s.src_off == it.src_off][0]
        counts[k][it.row0: it.row0 + it.n_rows] += 1
        s = specs[k]
        assert it.dst_off == s.dst_off + it.row0 * s.dst_cols * s.inner
        assert (it.idx_out_off == -1) == (s.idx_out_off == -1)
    for v in counts.values():
        assert (v == 1).all()
    assert plan.algorithmic_bytes() == 8 * (100 * 64 * 9 + 256 * 30 + 17) + 8 * (100 + 30 + 17)


def test_cols_plan_partial_layout():
    layers = [ops.ColsLayout(0, 0, 512, 512, 9), ops.ColsLayout(512 * 4608, 512, 37, 10, 1)]
    plan = ops.ColsPlan(layers, torch.device("cpu"))
    folds = (nat.ColsFold * 2).from_buffer_copy(plan.folds.cpu().numpy().tobytes()[:64])
    assert folds[0].n_flat == 4608 and folds[0].inner == 9 and folds[0].cols == 512
    assert 1 <= folds[0].n_slabs <= 16
    cover = np.zeros((folds[0].n_slabs, 4608), int)
    rows = 0
    for it in _items(plan, nat.ColsItem):
        if it.out_off < folds[1].part_off:
            s, c0 = divmod(it.out_off - folds[0].part_off, 4608)
            cover[s, c0: c0 + it.n_flat] += 1
            if c0 == 0:
                rows += it.n_rows
    assert (cover == 1).all() and rows == 512


def test_ops_refuse_cpu_tensors():
    conf = torch.zeros(3, 3, dtype=torch.int64)
    x = torch.zeros(10, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ops.confusion_update(conf, x, x, 3)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ops.channel_importance([torch.zeros(4, 4)], nat.IMP_L2)
