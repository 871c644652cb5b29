"""RCCL has to execute the collectives of the merge at least once on the hardware the tests see: one GPU.  A child process
brings up backend ``nccl`` with world_size 1 and drives the collectives of distributed.py (the striped IN-PLACE
reduce-scatter / all-gather into the volume, piece by piece; the staged frame exchange) with the ``world == 1`` early return
of ``merge_sums`` bypassed; with one rank every sum is the identity, so the tensors must come back unchanged -- what is
tested is that torch's and RCCL's argument checks accept the calls (output aliasing its input's slice, several pieces, a
ragged last piece).  The layout arithmetic itself is tested at world 2, 4 and 8 under gloo (tests/test_distributed_cpu.py:
the same calls -- gloo implements reduce_scatter_tensor / all_gather_into_tensor)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

_CHILD = textwrap.dedent('''
    import os, sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["SAF_REPO"])
    from spatially_aware_ai_amd import distributed as sd

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["SAF_PORT"], world_size=1, rank=0)
    dev = torch.device("cuda", 0)
    sd._CHUNK_ELEMS = 1 << 12          # many pieces, a ragged last one
    g = torch.Generator(device=dev).manual_seed(3)
    for shape in ((1000, 24), (257, 3), (4099,), (130, 143)):
        t = torch.randn(shape, generator=g, device=dev) if len(shape) > 1 or True else None
        if shape == (130, 143):
            t = torch.randint(0, 9, shape, generator=g, device=dev, dtype=torch.int32)
        ref = t.clone()
        plan = sd.stripe_plan(t.shape[0], 1, 61)   # many pieces, a ragged last one
        sd._reduce_scatter_striped(t, plan, None, 0, 1)
        assert torch.equal(t, ref), ("reduce-scatter changed a one-rank tensor", shape)
        sd._all_gather_striped(t, plan, None, 0, 1)
        assert torch.equal(t, ref), ("all-gather changed a one-rank tensor", shape)
    fr = [torch.randn((9, 16, 12), generator=g, device=dev), None, torch.randn((9, 4, 4), generator=g, device=dev)]
    out = sd.gather_frames(fr)
    assert out[1] is None and torch.equal(out[0], fr[0]) and torch.equal(out[2], fr[2])
    assert sd.probe_collectives(dev) is None
    # the packed route (round 6: scan / pack / add are HIP passes): weight all-reduced, touched rows through all_to_all_single
    assert sd.probe_all_to_all(dev) is None
    n = 5000
    w = (torch.rand(n, generator=g, device=dev) < 0.2).int() * torch.randint(1, 9, (n,), generator=g, device=dev, dtype=torch.int32)
    tens = {"weight": w.clone(), "clip_feat": torch.randn((n, 512), generator=g, device=dev) * (w > 0)[:, None],
            "rgb": torch.rand((n, 3), generator=g, device=dev) * (w > 0)[:, None],
            "labels_one_hot": torch.randint(0, 5, (n, 143), generator=g, device=dev, dtype=torch.int32) * (w > 0)[:, None].int(),
            "tsdf": torch.randn(n, generator=g, device=dev)}
    ref = {k: v.clone() for k, v in tens.items()}
    plan = sd.stripe_plan(n, 1, 700)
    packed = sd._merge_rows(tens, plan, None, 0, 1, 1.0)
    torch.cuda.synchronize()
    assert packed == len(plan) and sd.last_merge["touched_rows"] == int((w > 0).sum()), sd.last_merge
    for k in tens:
        assert torch.equal(tens[k], ref[k]), ("the packed route changed a one-rank tensor", k)
    x = torch.ones(5, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("NCCL_WORLD1_OK")
''')


def test_nccl_collectives_of_the_merge_run_at_world_size_one(tmp_path):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, SAF_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), SAF_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
