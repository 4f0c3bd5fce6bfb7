"""GPU: a LIVE decode -- the reference's own slice decoder and parser (libovvc/slicedec.c + vcl_*.c, compiled where they lay into the
prebuilt harness oracle/_ref/gen_pipe, which travels with the snapshot; /root/reference does not) driving the INSTALLED shim
(shim/rcn_hip.c, rcn_init_functions_hip) on the real device, in one process, on N frame threads.

What the replay tests (test_gpu_pipe.py, test_gpu_shim_replay.py) cannot show and this one does: the timing-dependent half of the
boundary.  In `gen_pipe ... live`
  * the shim is neither record-only nor dry: its own device DPB (ovhip_dpb_create), ovhip_frame_begin_tag / _ref_tag /
    _dmvr_rows_collect / _dmvr_rows_begin / _submit launching on the GPU, OVHIP_OUT_PLANES copying every picture into its OVFrame;
  * the DMVR slot hands the caller what the shim returns (unrefined vectors); the refined vectors reach the picture's collocated
    motion planes through ovhip_frame_dmvr_rows_collect + ovhip_shim_apply_tmvp_cells BEFORE ovdpb_report_decoded_ctu_line reports
    the row (slicedec.c:934-956) -- nothing is fed back from the reference pass;
  * later pictures' parse threads read those planes under the reference's row synchronisation (drv_mvp.c:281-294; dpb.c:1242-1323)
    while the producing thread is still parsing: a late or wrong vector changes the parse of every picture that follows;
  * pictures are taken in decoding order by the next free frame thread (ovdec.c:188-248), each with its own OVSliceDec + OVCTUDec, and
    a device picture is released (ovhip_shim_frame_released) when its last reader is done.
The harness compares, in process, every picture's OVFrame and every meaningful entry of both collocated motion planes with the
reference pass (scalar slots, one thread) and prints one JSON line; it exits non-zero on any difference."""
import json
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
GEN_PIPE = ROOT / "oracle" / "_ref" / "gen_pipe"


def live(threads, *args, timeout=900):
    if not GEN_PIPE.exists():
        pytest.fail("oracle/_ref/gen_pipe is missing: the prebuilt harness (make -C oracle, in the build container) must travel with the snapshot -- a GPU run without it would silently drop the live / full-size parity tests")
    cmd = [str(GEN_PIPE), "/tmp", "live", "threads", str(threads)] + [str(a) for a in args]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines, f"gen_pipe live printed no result (rc {p.returncode}):\n{p.stderr[-3000:]}"
    r = json.loads(lines[-1])
    assert p.returncode == 0, f"gen_pipe live rc {p.returncode}: {r}\n{p.stderr[-3000:]}"
    return r


def check(r, n_pic, threads):
    assert r["mode"] == "live" and r["frame_threads"] == threads
    assert r["pictures_decoded"] == n_pic and r["shim_error"] == 0
    assert r["frames_differing"] == 0 and r["samples_differing"] == 0
    assert r["collocated_motion_entries_differing"] == 0 and r["collocated_motion_entries_compared"] > 100


# the committed fixtures' streams (tests/golden/pipe.ovg, pipe_b.ovg, tiles.ovg, tiles_b.ovg: same seeds, sizes, variants), now live
STREAMS = {
    "pipe":    ("pics", 5),
    "pipe_b":  ("seed", 3, "variant", 1, "size", 264, 392, "pics", 5),
    "tiles":   ("seed", 5, "size", 264, 392, "tiles", 2, 2, "pics", 5),
    "tiles_b": ("seed", 7, "size", 416, 240, "tiles", 2, 2, "pics", 3),
}


@pytest.mark.parametrize("threads", (1, 4))
@pytest.mark.parametrize("name", sorted(STREAMS))
def test_live_decode_of_the_fixture_streams(name, threads):
    args = STREAMS[name]
    r = live(threads, *args)
    check(r, int(args[args.index("pics") + 1]), threads)
    assert r["dmvr_calls"] > 50 or name == "tiles_b"


@pytest.mark.parametrize("threads", (1, 4))
def test_live_decode_of_a_whole_gop_416x240(threads):
    r = live(threads, "pics", 9)
    check(r, 9, threads)


@pytest.mark.parametrize("threads", (1, 4))
def test_live_decode_1080p(threads):
    """BASELINE configs[2]: 1920x1080, the full rcn path on the device under the real parser"""
    r = live(threads, "size", 1920, 1080, "pics", 9, "seed", 31)
    check(r, 9, threads)
    assert r["dmvr_calls"] > 5000


@pytest.mark.parametrize("threads", (1, 4))
def test_live_decode_of_the_nine_4k_pictures(threads):
    """BASELINE configs[3]: the nine 3840x2160 pictures bench.py's config.reference_stream replays, decoded live"""
    r = live(threads, "size", 3840, 2160, "pics", 9, timeout=1500)
    check(r, 9, threads)
    assert r["dmvr_calls"] > 20000


def test_live_decode_of_four_gops_on_eight_threads():
    """33 pictures (I + 4 GOPs of 8), 8 frame threads: more pictures in flight than a GOP holds, key pictures of the next GOP parsed
    while the leaves of the previous one still run; device pictures released as their last reader finishes"""
    r = live(8, "size", 832, 480, "pics", 33)
    check(r, 33, 8)


# ---- the same with shim/caller.patch applied to the reference (a recorder-friendly caller): oracle/_ref/patched/gen_pipe is the
# harness built against the patched rcn_structures.h / vcl_coding_unit.c / drv_affine_mvp.c, the shim with -DOVVC_HIP_CALLER_PATCH.  The caller
# then hands over whole BDOF / DMVR / affine coding units (rcn_cu_inter_b, rcn_affine_cu -> ovhip_rec_cu_inter): a third of the hook calls,
# no stitching in the shim -- and the same pictures and collocated motion planes as the UNPATCHED reference pass, bit for bit.
GEN_PIPE_PATCHED = ROOT / "oracle" / "_ref" / "patched" / "gen_pipe"


def live_patched(threads, *args, timeout=900):
    if not GEN_PIPE_PATCHED.exists():
        pytest.fail("oracle/_ref/patched/gen_pipe is missing: the prebuilt patched-caller harness (make -C oracle patched) must travel with the snapshot")
    cmd = [str(GEN_PIPE_PATCHED), "/tmp", "live", "threads", str(threads), "profile"] + [str(a) for a in args]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines and p.returncode == 0, f"patched gen_pipe live rc {p.returncode}:\n{p.stdout[-1500:]}\n{p.stderr[-3000:]}"
    return json.loads(lines[-1])


@pytest.mark.parametrize("threads", (1, 4))
@pytest.mark.parametrize("name", sorted(STREAMS))
def test_live_decode_with_the_patched_caller_fixture_streams(name, threads):
    args = STREAMS[name]
    check(live_patched(threads, *args), int(args[args.index("pics") + 1]), threads)


@pytest.mark.parametrize("w,h,threads,pics", [(1920, 1080, 4, 17), (3840, 2160, 1, 9), (3840, 2160, 8, 17)])
def test_live_decode_with_the_patched_caller(w, h, threads, pics):
    r = live_patched(threads, "size", w, h, "pics", pics, "seed", 31, timeout=1500)
    check(r, pics, threads)
    u = live(threads, "size", w, h, "pics", pics, "seed", 31, "profile", timeout=1500)
    # whole coding units instead of <= 16x16 / 4x4 sub-block calls: far fewer hook calls for the same pictures
    assert r["shim_hook_calls"] * 2 < u["shim_hook_calls"]


@pytest.mark.parametrize("patched", (False, True))
@pytest.mark.parametrize("gop,threads,pics", [(16, 8, 33), (32, 16, 33)])
def test_live_decode_of_deep_hierarchies_with_recycled_frames(gop, threads, pics, patched):
    """GOPs of 16 / 32 (five / six levels of references, decoded depth first) on more frame threads than the hierarchy is deep, the stream
    decoded three times by the same warm threads: every OVFrame comes back from the pool under a new picture while device pictures of
    the repetition before are still being released (the device DPB's tags keep a recycled key from matching a stale entry)"""
    run = live_patched if patched else live
    r = run(threads, "size", 832, 480, "pics", pics, "gop", gop, "reps", 3, "seed", 11)
    check(r, pics, threads)
    assert r["host_frames_recycled"] > pics          # (three repetitions over a pool much smaller than 3 x pics)


@pytest.mark.parametrize("patched", (False, True))
def test_live_decode_of_sequences_back_to_back(patched):
    """`cont 4`: four coded video sequences (I + GOPs of 8 each) as ONE stream of 68 pictures -- the frame threads take the next sequence's
    pictures while the tail of the one before still decodes (the steady state bench.py's config.live_decoder.steady_state measures); every
    copy's frames and collocated motion planes are compared with the one reference pass"""
    run = live_patched if patched else live
    r = run(8, "size", 832, 480, "pics", 17, "cont", 4, "reps", 2)
    check(r, 68, 8)
    assert r["copies_back_to_back"] == 4


# ---- band-wise submission (ovhip_frame_band; OFF by default -- whole pictures are faster on this decoder): the picture enters the
# device CTU row by CTU row while it is parsed, its rows are posted to the device DPB as their filters finish, dependent pictures' bands and
# eager DMVR rows go when the rows they read are final, and a picture whose parse ran ahead of its references works through its rows in its
# last hook as they arrive.  Same bar: every frame and every collocated motion entry equals the reference pass.
@pytest.mark.parametrize("bands", (1, 2))
@pytest.mark.parametrize("threads", (1, 4))
@pytest.mark.parametrize("name", sorted(STREAMS))
def test_live_decode_band_by_band_fixture_streams(name, threads, bands):
    args = STREAMS[name]
    n = int(args[args.index("pics") + 1])
    r = live(threads, *args, "bands", bands)
    check(r, n, threads)
    if "tiles" not in name:
        assert r["bands_sent"] >= n - 1              # (pictures cut into rect entries go whole)
    check(live_patched(threads, *args, "bands", bands), n, threads)


@pytest.mark.parametrize("w,h,threads,pics,bands", [(1920, 1080, 4, 17, 1), (3840, 2160, 8, 17, 1), (3840, 2160, 16, 33, 3), (832, 480, 8, 33, 1)])
def test_live_decode_band_by_band(w, h, threads, pics, bands):
    """deep hierarchies on more threads than the hierarchy is wide: bands left to later hooks, the last hook working through the rows as the
    references deliver them (final_progressive), I pictures band by band under the device-is-behind rule"""
    for fn in (live_patched, live):
        r = fn(threads, "size", w, h, "pics", pics, "seed", 31, "bands", bands, timeout=1500)
        check(r, pics, threads)
        assert r["bands_sent"] > pics
