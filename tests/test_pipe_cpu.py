"""CPU: the stage COMPOSITION is pinned on the reference's own slice decoder.

tests/golden/pipe.ovg / pipe_b.ovg hold five chained pictures each (I, B, B, B, P in decoding order) exactly as
/root/reference/libovvc/slicedec.c left them: its parser turned seeded slice data into coding units and called the rcn slots in
decode_ctu / decode_ctu_line / decode_ctu_last_line order (slicedec.c:712-1082) with the reference's own CTU scratch, intra line,
dbf_load_info / dbf_store_info, store_inter_maps and SAO / ALF line buffers live (oracle/ref_harness/gen_pipe.c).
shim_pipe*.ovg hold what the installed slots of shim/rcn_hip.c recorded while the same parser drove THEM.

Here the oracle decodes picture k of the recorded stream from the pictures IT decoded before and must end, byte for byte, with
the reference's frames -- deblocking after the inverse luma mapping, SAO on deblocked samples across CTU corners, ALF / CC-ALF with
its virtual boundaries over SAO output, CIIP / intra blocks next to inter CUs, DMVR's refined vectors feeding the next pictures'
temporal candidates, ragged last CTU column and row (416x240: 4 x 2 CTUs, 32 wide / 112 high; 264x392: 3 x 4 CTUs, 8 / 8).

tiles.ovg / tiles_b.ovg: the same with every picture cut into 2 x 2 rect entries (tiles; slicedec.c:636-657, one entry after the other
on one OVCTUDec as slicedec.c:649-653 runs them): prediction and CABAC contexts stop at the tile borders in the parser; the
reference deblocks, SAO-filters and ALF-filters every rect entry on its own (is_border from the entry-local CTU index), which the
picture-wide stages reproduce from the per-CTU border flags the shim records (ovhip_sao_ctu.border / ovhip_alf_ctu.border).
tiles: (2 + 1) x (2 + 2) CTUs; tiles_b: tiles of ONE CTU row (rcn_sao_first_pix_rows' single-row quirk per entry)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_pipeline
import pipe_cases
from openvvc_amd import capi

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/libovvc/slicedec.c")
STREAMS = ("pipe", "pipe_b", "tiles", "tiles_b")
# gen_pipe's arguments per stream (oracle/Makefile, `golden`)
ARGS = {"pipe": [], "pipe_b": "name pipe_b seed 3 variant 1 size 264 392".split(),
        "tiles": "name tiles seed 5 size 264 392 tiles 2 2".split(), "tiles_b": "name tiles_b seed 7 size 416 240 tiles 2 2 pics 3".split()}


@pytest.mark.parametrize("name", STREAMS)
def test_oracle_chain_equals_the_reference_slice_decoder(built_lib, name):
    P = pipe_cases.Pipe(name)
    assert [int(v) for v in P.info[:, 1]] == [2, 0, 0, 0, 1][:P.n] and P.n == (3 if name == "tiles_b" else 5)          # I B B B P
    decoded, seen = {}, dict(mc=0, mcx=0, dmvr=0, bdof=0, aff=0, prof=0, gpm=0, ciip=0, itask=0, region=0, tb=0, res_store=0)
    for k in range(P.n):
        wl = P.workload(k, decoded)
        dst, mvs = oracle_pipeline.decode(wl, want_mvs=True)
        for plane, got, want in zip("Y Cb Cr".split(), (dst.y, dst.cb, dst.cr), P.frames[k]):
            assert np.array_equal(got, want), f"{name} picture {k} plane {plane}: {int((got != want).sum())} samples differ from the reference"
        calls = P.dmvr_calls(k)
        is_dmvr = (wl.mcx_units["flags"] & 64) != 0
        assert is_dmvr.sum() == len(calls)
        if len(calls):
            # every rcn_dmvr_mv_refine call of the reference is one recorded unit, in order: position, size, vectors in, vectors out
            u = wl.mcx_units[is_dmvr]
            assert np.array_equal(u["x"], calls[:, 0]) and np.array_equal(u["y"], calls[:, 1])
            assert np.array_equal(mvs[is_dmvr], calls[:, 8:12]), f"{name} picture {k}: refined vectors differ from rcn_dmvr_mv_refine's"
        decoded[k] = (dst.y, dst.cb, dst.cr)                                        # the chain: OUR picture is the next one's reference
        f = wl.mc_units["flags"]
        seen["mc"] += len(f); seen["gpm"] += int(((f & 128) != 0).sum()); seen["ciip"] += 0 if wl.itasks is None else int((wl.itasks["ciip_wt"] != 0).sum())
        fx = wl.mcx_units["flags"]
        seen["mcx"] += len(fx); seen["dmvr"] += int(is_dmvr.sum()); seen["bdof"] += int(((fx & 32) != 0).sum())
        seen["aff"] += len(wl.aff_units); seen["prof"] += int(((wl.aff_units["flags"] & capi.AFF_PROF) != 0).sum())
        seen["itask"] += 0 if wl.itasks is None else len(wl.itasks); seen["region"] += len(wl.lmcs_regions); seen["tb"] += len(wl.tb_cmds)
        seen["res_store"] += int((wl.tb_cmds["res_mode"] != 0).sum())
    # the parse is a random walk through the reference's caller code: the prediction families must have come up
    least = {"pipe": dict(mc=1000, gpm=50, mcx=400, dmvr=400, bdof=400, aff=200, prof=100, itask=700, region=80, tb=500, res_store=300, ciip=4),
             "pipe_b": dict(mc=800, mcx=400, dmvr=300, aff=150, itask=300, region=80, tb=300),
             "tiles": dict(mc=800, mcx=400, dmvr=300, aff=150, itask=300, region=80, tb=200),
             "tiles_b": dict(mc=500, mcx=200, dmvr=200, aff=80, itask=300, region=50, tb=200)}[name]
    for key, n in least.items():
        assert seen[key] >= n, (name, key, seen)
    if name == "pipe":
        t = np.concatenate([P.s.case(k)["itask"] for k in range(P.n)])
        tb = np.concatenate([P.s.case(k)["tb"] for k in range(P.n)])
        for what, n in (("ISP", ((t["flags"] & capi.IF_ISP) != 0).sum()), ("MIP", ((t["flags"] & capi.IF_MIP) != 0).sum()), ("MRL", (t["mrl_idx"] != 0).sum()),
                        ("BDPCM", ((t["flags"] & capi.IF_BDPCM) != 0).sum()), ("LM", ((t["kind"] == capi.IT_CHROMA) & (t["mode"] >= 67)).sum()),
                        ("LFNST", (tb["lfnst"] != 0).sum()), ("DST-VII / DCT-VIII", (tb["tr_h"] != 0).sum())):
            assert n >= 20, (what, int(n))


def test_chain_breaks_when_a_stage_is_left_out(built_lib):
    """the comparison has teeth: without the deblocking stage, or from a wrong reference picture, the frames differ"""
    P = pipe_cases.Pipe("pipe")
    ref = {k: P.frames[k] for k in range(P.n)}
    wl = P.workload(2, ref)
    assert not np.array_equal(oracle_pipeline.decode(wl, stages=("mc", "itx", "sao", "alf")).y, P.frames[2][0])
    wrong = dict(ref); wrong[1] = ref[0]
    assert not np.array_equal(oracle_pipeline.decode(P.workload(2, wrong)).y, P.frames[2][0])
    assert np.array_equal(oracle_pipeline.decode(wl).y, P.frames[2][0])


@pytest.mark.skipif(not REF.exists(), reason="the reference tree is only present in the build container")
def test_pipe_fixtures_regenerate_identically(built_lib, tmp_path):
    """pipe*.ovg / shim_pipe*.ovg are what the harness makes from the reference's sources and the current shim + recorder."""
    subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
    gen = str(ROOT / "oracle" / "_ref" / "gen_pipe")
    for name in STREAMS:
        subprocess.check_call([gen, str(tmp_path)] + ARGS[name], stderr=subprocess.DEVNULL)
        subprocess.check_call([gen, str(tmp_path), "shim"] + ARGS[name], stderr=subprocess.DEVNULL)
        subprocess.check_call([gen, str(tmp_path), "device"] + ARGS[name], stderr=subprocess.DEVNULL)
        for f in (f"{name}.ovg", f"shim_{name}.ovg", f"shim_{name}_dev.ovg"):
            assert (tmp_path / f).read_bytes() == (ROOT / "tests" / "golden" / f).read_bytes(), f
    # A second witness for the legal-syntax stream: the reference's own SSE4.1 / AVX2 back-end (installed over the scalar table in
    # rcn.c:216-254's order) decodes `pipe` to the same bytes as its scalar slots.  (Not so on 264-wide pictures: chroma rows of
    # 132 samples -- its vector stores run 4 samples past the row's end into the next row's left edge; the scalar slots stay the
    # oracle, as for the fixtures of test_oracle_golden.py.)
    simd = tmp_path / "simd"
    simd.mkdir()
    subprocess.check_call([gen, str(simd), "simd"], stderr=subprocess.DEVNULL)
    assert (simd / "pipe.ovg").read_bytes() == (ROOT / "tests" / "golden" / "pipe.ovg").read_bytes()


@pytest.mark.parametrize("name", ("tiles", "tiles_b"))
def test_tile_borders_are_what_the_filters_stop_at(built_lib, name):
    """the per-CTU border flags are the rect entries' (entry-local first / last column and row) and the comparison needs them: with
    the flags dropped, SAO / ALF filter across the tile borders and the pictures differ from the reference's"""
    P = pipe_cases.Pipe(name)
    nw, nh = (P.w + 127) // 128, (P.h + 127) // 128
    cols, rows = ((0, 2), (2, 3)) if name == "tiles" else ((0, 2), (2, 4)), ((0, 2), (2, 4)) if name == "tiles" else ((0, 1), (1, 2))
    want = np.zeros((nh, nw), np.uint8)
    for y0, y1 in rows:
        for x0, x1 in cols:
            want[y0:y1, x0] |= capi.BORDER_LEFT; want[y0:y1, x1 - 1] |= capi.BORDER_RIGHT
            want[y0, x0:x1] |= capi.BORDER_UPPER; want[y1 - 1, x0:x1] |= capi.BORDER_BOTTOM
            if y1 - y0 == 1:
                want[y0:y1, x0:x1] |= capi.BORDER_ONE_ROW
    ref = {k: P.frames[k] for k in range(P.n)}
    for k in range(P.n):
        wl = P.workload(k, ref)
        assert np.array_equal(wl.alf["ctus"]["border"].reshape(nh, nw), want)
        on = (wl.sao_params["type"] != 0).any(axis=1)
        assert np.array_equal(wl.sao_params["border"][on], want.reshape(-1)[on])          # (CTUs no SAO row call covered stay zero)
    wl = P.workload(0, ref)
    wl.alf["ctus"] = wl.alf["ctus"].copy(); wl.alf["ctus"]["border"] = 0
    assert not np.array_equal(oracle_pipeline.decode(wl).y, P.frames[0][0])
    wl = P.workload(0, ref)
    wl.sao_params = wl.sao_params.copy(); wl.sao_params["border"] = 0
    assert not np.array_equal(oracle_pipeline.decode(wl).y, P.frames[0][0])
