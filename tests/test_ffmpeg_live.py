"""Live cross-check against a real ffmpeg — runs only where an `ffmpeg` binary exists (none in
the build image or on the GPU box today, so this auto-skips; the scene half stays PARITY
UNPINNED until it has run somewhere).  Uses the exact argv of inspector/app.py:202-208 on a
seeded Y4M clip and the reference's own line parser."""
import shutil
import subprocess

import pytest

from tvidz_amd import feeder, scene, synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(shutil.which("ffmpeg") is None, reason="no ffmpeg binary on this host")]


def test_cuts_match_live_ffmpeg(tmp_path):
    T, H, W = 300, 480, 854
    frames, _ = synth.synth_luma(T, H, W, device="cuda:0", seed=21, min_scene=25, max_scene=70)
    path = str(tmp_path / "clip.y4m")
    feeder.write_y4m(path, frames.cpu().numpy(), fps=(30, 1), chroma="420jpeg")
    cmd = ["ffmpeg", "-hide_banner", "-loglevel", "info", "-i", path,
           "-vf", "select=gt(scene\\,0.3),showinfo", "-f", "null", "-"]          # app.py:202-208
    if shutil.which("stdbuf"):
        cmd = ["stdbuf", "-oL", "-eL"] + cmd
    err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
    ref = []
    for line in err.splitlines():
        ts = scene.parse_showinfo_line(line)
        if ts is not None and (not ref or ts != ref[-1]):                           # app.py:231
            ref.append(ts)
    ver = subprocess.run(["ffmpeg", "-version"], stdout=subprocess.PIPE, text=True).stdout.splitlines()[0]
    got = {}
    for policy in (scene.PTS_POLICY_G6, scene.PTS_POLICY_F6TRIM):
        got[policy] = list(scene.detect_scene_cuts(frames, time_base=(1, 30), pts_policy=policy, batch=64))
    idx = [n for n, _ in got[scene.PTS_POLICY_G6]]
    ref_idx = [round(t * 30) for t in ref]
    print(ver, "ffmpeg cuts:", ref, "gpu frames:", idx)
    assert len(idx) == len(ref_idx)
    assert all(abs(a - b) <= 1 for a, b in zip(idx, ref_idx))                      # north_star: +-1 frame
    exact = [p for p in got if [ts for _, ts in got[p]] == ref]
    assert exact, f"neither pts_time policy reproduces {ver}'s text: {ref}"
