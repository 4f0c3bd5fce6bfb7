"""GPU: the one-process-per-GPU path of the stream driver (ovhip_stream_xfer callbacks on the C communication thread) executed by two
ranks -- on ONE GPU, pictures exchanged through host memory over gloo (OVVC_BENCH_DEBUG_GLOO=1: gpurun boxes have one GPU; the RCCL
leg of the same callbacks first runs on the driver's multi-GPU node).  Both dealings; every rank's pictures are checked against the
oracle / the one-at-a-time decode by bench.py --check inside the run."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _run(dealing):
    port = _free_port()
    env = dict(os.environ, OVVC_BENCH_DEBUG_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--width", "832", "--height", "480", "--no-cpu-baseline",
           "--no-isolated-survey", "--check", "3", "--dealing", dealing, "--both-dealings"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("dealing", ["gop", "picture"])
def test_two_ranks_on_one_gpu(built_lib, dealing):
    d = _run(dealing)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["dealing"] == dealing and c["ordered_pass_second_passes"] == 0
    assert c["check"]["differ"] == 0 and c["check"]["differ_in_flight"] == 0
    assert c["transfers_this_rank"]["sent"] > 0 and c["transfers_this_rank"]["bytes_sent"] > 0
    # the line checks itself: every rank's account of the timed run is in it (what it decoded, sent, received; what the dealing says it
    # should have sent), and the sums agree
    assert len(c["ranks"]) == 2 and [a["rank"] for a in c["ranks"]] == [0, 1] and c["exchange_consistent"] is True
    assert all(a["pictures_sent"] == a["pictures_sent_expected_from_the_dealing"] and a["pictures_decoded"] > 0 for a in c["ranks"])
    assert c["other_dealing"]["dealing"] != dealing and c["other_dealing"]["fps"] > 0
    # the picture-interleaved dealing moves (nearly) a picture per picture, a GOP per GPU one picture per GOP
    sent_pic = c["transfers_this_rank"]["sent"] if dealing == "picture" else c["other_dealing"]["pictures_sent_by_this_rank"]
    sent_gop = c["transfers_this_rank"]["sent"] if dealing == "gop" else c["other_dealing"]["pictures_sent_by_this_rank"]
    assert sent_pic > 4 * sent_gop


@pytest.mark.gpu
def test_gpus_flag_without_torchrun_drives_that_many_devices(built_lib):
    """`python bench.py --gpus 2` launched WITHOUT torch.distributed.run (WORLD_SIZE unset) must not decode on one GPU and print
    n_gpus: 1: it takes the one-process path (C stream driver over two logical devices, a set of pre-recorded jobs per device, peer
    copies through the device DPB) -- here both on GPU 0 (--same-gpu) -- and says n_gpus: 2; with more devices than visible it refuses."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--same-gpu", "--steps", "1", "--warmup", "1", "--width", "832", "--height", "480",
           "--no-cpu-baseline", "--no-isolated-survey", "--check", "3"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["local_devices"] == 2 and d["scaling"] == "weak"
    assert c["pictures_timed"] == 2 * 64 and c["pictures_per_step"] == 2 * 64 and c["check"]["differ"] == 0
    assert c["dpb"]["peer_copies"] > 0 and c["other_dealing"]["dealing"] == "picture" and c["other_dealing"]["peer_copies"] > c["dpb"]["peer_copies"]
    assert c["step_fps"]["min"] > 0
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64", "--steps", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "devices asked for" in p.stderr


@pytest.mark.gpu
def test_rccl_transport_on_one_rank(built_lib):
    """The product's multi-process exchange (ovvc_rccl.hip: librccl opened at run time, ncclCommInitRank, the three planes of a picture as
    ncclSend / ncclRecv in one ncclGroup on the transport's own stream) on a one-GPU box: a communicator of ONE rank sends a picture to
    itself.  (Two ranks on one GPU are refused by RCCL -- "Duplicate GPU detected" -- so the two-rank tests above exchange through gloo.)"""
    import numpy as np
    from openvvc_amd import engine
    ctx = engine.Context(0)
    t = engine.RcclTransport(engine.RcclTransport.unique_id(), 0, 1, 0)
    rs = np.random.RandomState(7)
    w, h = 416, 240
    y, cb, cr = (rs.randint(0, 1024, size=s).astype(np.uint16) for s in ((h, w), (h // 2, w // 2), (h // 2, w // 2)))
    src, dst = ctx.upload_pic(y, cb, cr), ctx.new_pic(w, h)
    ctx.sync()
    t.self_exchange(src, dst)
    gy, gcb, gcr = dst.download()
    assert np.array_equal(gy, y) and np.array_equal(gcb, cb) and np.array_equal(gcr, cr)
    t.close(); ctx.close()
