"""saf_stage_frame (csrc/saf_misc.hip): one launch copies a frame's inputs into a slot of the staging ring -- 16-byte pieces where
source and destination allow, 4-byte copies elsewhere (images whose size is not a multiple of four pixels put every second frame
off a 16-byte boundary), a strided gather for a permuted feature map, nothing for images the caller lends (NULL on both sides)."""
import ctypes as C

import pytest
import torch

from spatially_aware_ai_amd import _abi
from spatially_aware_ai_amd._lib import check, current_stream_ptr, lib

pytestmark = pytest.mark.gpu


def _frame(h, w, depth, rgb, pose, K, feat, npy, npx, labels):
    p = lambda t: None if t is None else t.data_ptr()
    return _abi.SafFrame(h, w, p(depth), p(rgb), p(pose), p(K), p(feat), npy, npx, p(labels), 0)


@pytest.mark.parametrize("h,w", [(48, 64), (33, 31), (7, 5), (480, 640)])
@pytest.mark.parametrize("lend", [False, True])
def test_stage_frame_copies_every_segment(h, w, lend):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(h * 1000 + w)
    npy, npx, ch = 5, 7, 40
    n = 3  # frame 1 of a batch starts h * w * 4 bytes into the tensor: off a 16-byte boundary for odd sizes
    src = {"depth": torch.rand((n, h, w), generator=g, device=dev), "rgb": torch.rand((n, h, w, 3), generator=g, device=dev),
           "pose": torch.rand((n, 4, 4), generator=g, device=dev), "K": torch.rand((n, 3, 3), generator=g, device=dev),
           "feat": torch.rand((n, npy, npx, ch), generator=g, device=dev).permute(0, 3, 1, 2),  # a permuted view [n, C, npy, npx]
           "labels": torch.rand((n, h, w), generator=g, device=dev)}
    dst = {"depth": torch.full((n, h, w), -1.0, device=dev), "rgb": torch.full((n, h, w, 3), -1.0, device=dev),
           "pose": torch.full((n, 4, 4), -1.0, device=dev), "K": torch.full((n, 3, 3), -1.0, device=dev),
           "feat": torch.full((n, ch, npy, npx), -1.0, device=dev), "labels": torch.full((n, h, w), -1.0, device=dev)}
    for i in range(n):
        s = _frame(h, w, None if lend else src["depth"][i], None if lend else src["rgb"][i], src["pose"][i], src["K"][i], src["feat"][i],
                   npy, npx, None if lend else src["labels"][i])
        d = _frame(h, w, None if lend else dst["depth"][i], None if lend else dst["rgb"][i], dst["pose"][i], dst["K"][i], dst["feat"][i],
                   npy, npx, None if lend else dst["labels"][i])
        fs = src["feat"][i].stride()
        check(lib().saf_stage_frame(C.byref(s), ch, fs[0], fs[1], fs[2], C.byref(d), current_stream_ptr()), "saf_stage_frame")
    torch.cuda.synchronize()
    for k in ("pose", "K"):
        assert torch.equal(dst[k], src[k]), k
    assert torch.equal(dst["feat"], src["feat"].contiguous())
    for k in ("depth", "rgb", "labels"):
        if lend:
            assert bool((dst[k] == -1.0).all()), f"{k} was lent, not copied"
        else:
            assert torch.equal(dst[k], src[k]), k


def test_stage_frame_rejects_half_lent_images():
    dev = torch.device("cuda", 0)
    t = torch.zeros((4, 4), device=dev)
    p, k = torch.zeros((4, 4), device=dev), torch.zeros((3, 3), device=dev)
    s = _frame(4, 4, t, None, p, k, None, 0, 0, None)   # depth without rgb on the source side only ...
    d = _frame(4, 4, t, t.new_zeros((4, 4, 3)), p, k, None, 0, 0, None)
    assert lib().saf_stage_frame(C.byref(s), 0, 0, 0, 0, C.byref(d), current_stream_ptr()) == _abi.SAF_E_INVALID
