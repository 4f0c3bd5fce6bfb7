"""CPU: the shim, the device DPB's state machine and the frame layer under SEVERAL frame threads, without a GPU.

`gen_pipe ... device threads N` (oracle/ref_harness/gen_pipe.c; built only where /root/reference exists) runs the reference's own slice
decoder + parser over the installed shim on N frame threads -- one OVSliceDec + OVCTUDec each, pictures taken in decoding order by the
next free thread, OVFrames handed out and taken back by a frame pool (so a recycled OVFrame meets its previous owner's DPB slot), the
collocated motion field read under the reference's row synchronisation -- on DRY frames (a DPB with a test memory back-end: every
ovhip_frame_* call and every wait happens, nothing is launched).  The refined vectors no device computes are fed from the reference
pass per call, and the harness requires the DMVR call sequence of every picture to be the reference pass's: a parse that diverged
(a vector read too early, a plane patched too late) ends the run.  The live run of the same harness on a GPU is tests/test_gpu_live.py."""
import json
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GEN_PIPE = ROOT / "oracle" / "_ref" / "gen_pipe"


def dry(threads, *args):
    if not GEN_PIPE.exists():
        pytest.skip("oracle/_ref/gen_pipe is built only where /root/reference exists")
    p = subprocess.run([str(GEN_PIPE), "/tmp", "device", "threads", str(threads)] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert rows and p.returncode == 0, f"rc {p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-2500:]}"
    return rows


@pytest.mark.parametrize("args", [("pics", 9), ("pics", 33, "size", 832, 480, "reps", 2), ("seed", 5, "size", 264, 392, "tiles", 2, 2, "pics", 5)])
def test_frame_threads_on_dry_frames(args):
    n = int(args[args.index("pics") + 1])
    for r in dry("1,4,8", *args):
        assert r["mode"] == "device_dry_threads" and r["pictures_decoded"] == n and r["shim_error"] == 0
        assert r["collocated_motion_entries_differing"] == 0 and r["collocated_motion_entries_compared"] > 1000
        if r["frame_threads"] == 1:
            assert r["host_frames_recycled"] > 0          # the pool handed frames out again: keys came back to the DPB


@pytest.mark.parametrize("bands", (1, 2))
def test_frame_threads_on_dry_frames_band_by_band(bands):
    """the same with band-wise submission switched on (ovhip_frame_band on dry frames: the DPB's row progress, bands left to later hooks,
    the eager DMVR rows' row-granular readiness): the parse of every picture still sees the reference pass's DMVR vectors in time"""
    for r in dry("1,4,8", "pics", 33, "size", 832, 480, "bands", bands):
        assert r["pictures_decoded"] == 33 and r["shim_error"] == 0 and r["collocated_motion_entries_differing"] == 0
        assert r["bands_sent"] > 33
