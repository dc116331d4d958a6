"""The entry point that allocates nothing, spice_confusion_update_static, called through the C ABI:
same matrix as spice_confusion_update (bit for bit) and as the CPU oracle on two calls in a row, the
second on a smaller input, plus its limits as return codes.  Collected last on purpose."""
import numpy as np
import pytest
import torch

from oracle import confusion as oc
from spice_amd import _native as nat

pytestmark = pytest.mark.gpu


def _labels(seed, n, classes):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(0, classes, (n,), generator=g)
    p = torch.randint(0, classes, (n,), generator=g)
    t[torch.rand(n, generator=g) < 0.07] = 255
    t[torch.rand(n, generator=g) < 0.01] = -1
    p[torch.rand(n, generator=g) < 0.002] = classes + 3   # a valid target with such a prediction is counted, not binned
    return p, t


def test_static_entry_point_equals_workspace_entry_point():
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    L = nat.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    for classes, sizes in ((21, (8 * 512 * 512, 3 * 211 * 97)), (32, (70001, 513)), (1, (5000, 17))):
        conf_static = torch.zeros(classes, classes, dtype=torch.int64, device=dev)
        conf_ws = torch.zeros(classes, classes, dtype=torch.int64, device=dev)
        want = np.zeros((classes, classes), dtype=np.int64)
        bad_static = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = torch.zeros(L.spice_confusion_workspace_bytes(max(sizes), classes), dtype=torch.uint8, device=dev)
        want_bad = 0
        for k, n in enumerate(sizes):          # the second call is the smaller one
            p, t = _labels(100 * classes + k, n, classes)
            pd, td = p.to(dev), t.to(dev)
            torch.cuda.synchronize()
            before = torch.cuda.memory_allocated(dev)
            rc = L.spice_confusion_update_static(pd.data_ptr(), td.data_ptr(), n, classes, conf_static.data_ptr(),
                                                 bad_static.data_ptr(), stream)
            assert rc == 0, L.spice_error_string(rc)
            assert torch.cuda.memory_allocated(dev) == before
            rc = L.spice_confusion_update(pd.data_ptr(), td.data_ptr(), n, classes, conf_ws.data_ptr(), ws.data_ptr(),
                                          ws.numel(), stream)
            assert rc == 0, L.spice_error_string(rc)
            oc.update_confusion(want, np.where((p.numpy() >= 0) & (p.numpy() < classes), p.numpy(), 0), np.where(
                (p.numpy() >= 0) & (p.numpy() < classes), t.numpy(), 255), classes)
            valid = (t.numpy() >= 0) & (t.numpy() < classes)
            want_bad += int((valid & ((p.numpy() < 0) | (p.numpy() >= classes))).sum())
            torch.cuda.synchronize()
            assert torch.equal(conf_static, conf_ws), (classes, n)
            np.testing.assert_array_equal(conf_static.cpu().numpy(), want)
            assert int(bad_static.item()) == want_bad == int(ws[:8].view(torch.int64).item())
    # without a counter for out-of-range predictions
    p, t = _labels(5, 4099, 19)
    conf = torch.zeros(19, 19, dtype=torch.int64, device=dev)
    assert L.spice_confusion_update_static(p.to(dev).data_ptr(), t.to(dev).data_ptr(), 4099, 19, conf.data_ptr(), None,
                                           stream) == 0
    want = np.zeros((19, 19), dtype=np.int64)
    ok = (p.numpy() >= 0) & (p.numpy() < 19)
    oc.update_confusion(want, np.where(ok, p.numpy(), 0), np.where(ok, t.numpy(), 255), 19)
    np.testing.assert_array_equal(conf.cpu().numpy(), want)


def test_static_entry_point_limits_are_return_codes():
    L = nat.lib()
    dev = torch.device("cuda:0")
    x = torch.zeros(64, dtype=torch.int64, device=dev)
    conf = torch.zeros(33 * 33, dtype=torch.int64, device=dev)
    rc = L.spice_confusion_update_static(x.data_ptr(), x.data_ptr(), 64, 33, conf.data_ptr(), None, None)
    assert rc == nat.ERR_STATIC_LIMIT
    assert b"at most 32 classes" in L.spice_error_string(rc)
    assert L.spice_confusion_update_static(x.data_ptr(), x.data_ptr(), (1 << 40) + 1, 4, conf.data_ptr(), None,
                                           None) == nat.ERR_STATIC_LIMIT
    assert L.spice_confusion_update_static(None, None, 64, 4, conf.data_ptr(), None, None) == nat.ERR_ARG
    assert L.spice_confusion_update_static(x.data_ptr(), x.data_ptr(), 0, 4, conf.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    assert int(conf.sum().item()) == 0


def test_static_torch_op_from_two_streams_in_turn():
    """torch.ops.spice.confusion_update_static called from two streams alternately: the calls share the GPU's one
    table, and the operator drains the previous stream when the stream changes, so the matrices are right."""
    from spice_amd import ops
    dev = torch.device("cuda:0")
    classes, n = 21, 1 << 20
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    confs = [torch.zeros(classes, classes, dtype=torch.int64, device=dev) for _ in streams]
    wants = [np.zeros((classes, classes), dtype=np.int64) for _ in streams]
    data = []
    for k in range(6):
        p, t = _labels(900 + k, n, classes)
        p.clamp_(0, classes - 1)
        data.append((p.to(dev), t.to(dev)))
        oc.update_confusion(wants[k % 2], p.numpy(), t.numpy(), classes)
    torch.cuda.synchronize()
    for k, (pd, td) in enumerate(data):
        with torch.cuda.stream(streams[k % 2]):
            ops.spice_ops().confusion_update_static(confs[k % 2], pd, td, classes, None)
    torch.cuda.synchronize()
    for conf, want in zip(confs, wants):
        np.testing.assert_array_equal(conf.cpu().numpy(), want)
