"""The real decode leg (every mp4 upload takes it: inspector/app.py:202-209) exercised WITHOUT an
ffmpeg binary: the test writes stub `ffprobe` / `ffmpeg` executables that speak the same command
lines (ffprobe prints the key=value stream block, the packet pts list and nb_read_frames; ffmpeg
writes planar rawvideo of a known clip to stdout) and checks what FFmpegReader makes of them:
Y-plane slicing and chroma skip for 420/422/444/gray and 10-bit, the stream time base and the REAL
per-frame pts (frame index x time_base is only right for Y4M), the pass-through frame
sync flags per FFmpeg version, the `-count_frames` fallback of app.py:183-188 and the clean-up of
the child on an early stop (app.py:249-252).  CPU only: no GPU call is made."""
import json
import os
import stat
import sys
import time

import numpy as np
import pytest

from tvidz_amd import feeder, scene

STUB = r'''#!{python}
import json, os, sys
cfg = json.load(open(os.environ["TVZ_STUB_CFG"]))
argv = sys.argv[1:]
with open(cfg["log"], "a") as f:
    f.write(json.dumps([os.path.basename(sys.argv[0])] + argv) + "\n")
name = os.path.basename(sys.argv[0])
if name == "ffprobe":
    joined = " ".join(argv)
    if "-count_frames" in argv:
        if cfg.get("count_frames") is None:
            sys.exit(1)
        print(cfg["count_frames"])
    elif "packet=pts" in joined:
        if cfg.get("fail_packets"):
            sys.exit(1)
        for p in cfg["packet_pts"]:
            print(p)
    else:
        print("width=%d" % cfg["W"]); print("height=%d" % cfg["H"]); print("pix_fmt=%s" % cfg["pix_fmt"])
        print("time_base=%s" % cfg["time_base"])
        print("nb_frames=%s" % cfg.get("nb_frames", "N/A"))
elif name == "ffmpeg":
    if "-version" in argv:
        print("ffmpeg version %s Copyright (c) the FFmpeg developers" % cfg["version"])
        sys.exit(0)
    data = open(cfg["raw"], "rb").read()
    out = sys.stdout.buffer
    step = cfg.get("chunk", 4096)
    try:
        for i in range(0, len(data), step):      # short writes: the reader must reassemble frames
            out.write(data[i:i + step]); out.flush()
        if cfg.get("hang"):
            import time
            time.sleep(600)
    except BrokenPipeError:
        pass
'''


def _chroma(fmt, W, H, bps):
    if fmt.startswith("gray"):
        return 0
    cw, ch = (W + 1) // 2, (H + 1) // 2
    if "420" in fmt:
        return 2 * cw * ch * bps
    if "422" in fmt:
        return 2 * cw * H * bps
    return 2 * W * H * bps


@pytest.fixture()
def stub(tmp_path, monkeypatch):
    bindir = tmp_path / "bin"
    bindir.mkdir()
    for name in ("ffmpeg", "ffprobe"):
        p = bindir / name
        p.write_text(STUB.format(python=sys.executable))
        p.chmod(p.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{bindir}{os.pathsep}{os.environ['PATH']}")
    cfg_path = tmp_path / "cfg.json"
    monkeypatch.setenv("TVZ_STUB_CFG", str(cfg_path))

    def make(luma, pix_fmt="yuv420p", time_base="1/15360", pts=None, version="6.1.1", **extra):
        T, H, W = luma.shape
        bps = luma.dtype.itemsize
        raw = tmp_path / "clip.raw"
        with open(raw, "wb") as f:
            for t in range(T):
                f.write(np.ascontiguousarray(luma[t]).astype("<u2" if bps == 2 else np.uint8).tobytes())
                f.write(bytes([0x80 + (t % 7)]) * _chroma(pix_fmt, W, H, bps))   # chroma must be skipped
        cfg = dict(W=W, H=H, pix_fmt=pix_fmt, time_base=time_base, raw=str(raw), version=version,
                   packet_pts=list(pts if pts is not None else range(T)), log=str(tmp_path / "calls.log"))
        cfg.update(extra)
        cfg_path.write_text(json.dumps(cfg))
        media = tmp_path / "upload.mp4"
        media.write_bytes(b"\x00\x00\x00\x18ftypmp42 not really an mp4")
        return str(media)

    def calls():
        log = tmp_path / "calls.log"
        return [json.loads(l) for l in log.read_text().splitlines()] if log.exists() else []
    return make, calls


@pytest.mark.parametrize("pix_fmt,bits", [("yuv420p", 8), ("yuvj420p", 8), ("yuv422p", 8), ("yuv444p", 8),
                                           ("gray", 8), ("yuv420p10le", 10), ("yuv422p10le", 10)])
def test_y_plane_slicing_and_chroma_skip(stub, pix_fmt, bits):
    make, calls = stub
    rng = np.random.default_rng(len(pix_fmt))
    T, H, W = 7, 18, 26          # odd-ish sizes: chroma planes round up
    luma = rng.integers(0, 1 << bits, size=(T, H, W)).astype(np.uint16 if bits > 8 else np.uint8)
    path = make(luma, pix_fmt=pix_fmt, nb_frames=T, chunk=997)
    r = feeder.open_reader(path)                     # not Y4M magic -> FFmpegReader
    assert isinstance(r, feeder.FFmpegReader)
    assert (r.W, r.H, r.bitdepth, r.bps, r.total_frames) == (W, H, bits, 2 if bits > 8 else 1, T)
    buf = np.zeros((5, H, W), dtype=np.int16 if bits > 8 else np.uint8)
    assert r.read_into(buf) == 5
    assert (buf.view(luma.dtype) == luma[:5]).all()
    assert r.read_into(buf) == 2 and (buf[:2].view(luma.dtype) == luma[5:]).all()
    assert r.read_into(buf) == 0
    r.close()
    dec = [c for c in calls() if c[0] == "ffmpeg" and "-i" in c]
    assert len(dec) == 1 and dec[0][dec[0].index("-pix_fmt") + 1] == pix_fmt    # native format: no conversion
    assert "rawvideo" in dec[0]


def test_real_pts_and_stream_time_base(stub):
    """A 30 fps mp4 has time_base 1/15360 and pts = 512*n: frame 300 is 10.0 s, not 300/15360 s."""
    make, calls = stub
    T = 12
    luma = np.zeros((T, 8, 16), dtype=np.uint8)
    # packets arrive in DECODE order (B-frames): presentation order is the sorted list
    decode_order = [0, 3, 1, 2, 6, 4, 5, 9, 7, 8, 11, 10]
    path = make(luma, pts=[512 * n for n in decode_order], time_base="1/15360")
    r = feeder.FFmpegReader(path)
    assert r.time_base == (1, 15360)
    assert [r.pts_of(n) for n in range(T)] == [512 * n for n in range(T)]
    for n in (1, 7, 11):
        assert scene.pts_time_value(r.pts_of(n), r.time_base) == float("%.6g" % (n / 30))
    assert r.total_frames == T                       # nb_frames absent -> the demuxed packet count
    r.close()
    # variable frame rate: the timestamps are whatever the container says
    vfr = [0, 512, 1536, 1600, 4096, 4097, 9000, 9001, 9002, 20000, 20001, 30000]
    r = feeder.FFmpegReader(make(luma, pts=vfr, time_base="1/15360"))
    assert [r.pts_of(n) for n in range(T)] == vfr
    assert r.pts_of(T + 3) == T + 3                  # beyond the list: frame index (raw streams)
    r.close()


@pytest.mark.parametrize("version,flags", [("4.4.2-0ubuntu0.22.04.1", ["-vsync", "0"]), ("5.0.1", ["-vsync", "0"]),
                                            ("5.1.6-0+deb12u1", ["-fps_mode", "passthrough"]),
                                            ("7.1.1", ["-fps_mode", "passthrough"]),
                                            ("N-109468-gd39b34123d", ["-fps_mode", "passthrough"])])
def test_frames_pass_through_untouched(stub, version, flags):
    """The reference's `-f null` passes frames through; `-f rawvideo` would default to constant
    frame rate and duplicate/drop frames of a VFR input (a duplicate has mafd 0)."""
    make, calls = stub
    r = feeder.FFmpegReader(make(np.zeros((2, 4, 16), dtype=np.uint8), version=version))
    assert r.read_into(np.zeros((2, 4, 16), dtype=np.uint8)) == 2       # the child is up and has logged
    r.close()
    dec = [c for c in calls() if c[0] == "ffmpeg" and "-i" in c][0]
    i = dec.index(flags[0])
    assert dec[i:i + 2] == flags and dec.index("-i") < i < dec.index("-f")


def test_count_frames_fallback(stub):
    """app.py:176-188: nb_frames, else `ffprobe -count_frames ... nb_read_frames`, else 0."""
    make, calls = stub
    luma = np.zeros((3, 4, 16), dtype=np.uint8)
    r = feeder.FFmpegReader(make(luma, nb_frames=77))
    assert r.total_frames == 77
    r.close()
    r = feeder.FFmpegReader(make(luma, fail_packets=True, count_frames=41))
    assert r.total_frames == 41 and [r.pts_of(n) for n in range(3)] == [0, 1, 2]
    r.close()
    probe = [c for c in calls() if c[0] == "ffprobe" and "-count_frames" in c][-1]
    assert "stream=nb_read_frames" in probe and probe[probe.index("-select_streams") + 1] == "v:0"
    r = feeder.FFmpegReader(make(luma, fail_packets=True, count_frames=None))
    assert r.total_frames == 0                       # progress then falls back to the cut-count estimate
    r.close()


def test_unsupported_pixel_formats_fail_loudly(stub):
    make, _ = stub
    for fmt in ("rgb24", "nv12", "yuv410p"):
        with pytest.raises(RuntimeError, match="not planar YUV"):
            feeder.FFmpegReader(make(np.zeros((1, 4, 16), dtype=np.uint8), pix_fmt=fmt))


def test_early_stop_terminates_and_reaps_the_decoder(stub):
    """app.py:249-252: the decoder is stopped at the first duplicate.  close() must terminate the
    child, close the pipe and wait() for it (no zombie, no fd leak in a long-running service)."""
    make, _ = stub
    luma = np.zeros((4, 8, 16), dtype=np.uint8)
    r = feeder.FFmpegReader(make(luma, hang=True))   # the stub keeps running after its output
    buf = np.zeros((2, 8, 16), dtype=np.uint8)
    assert r.read_into(buf) == 2
    proc = r.proc
    assert proc.poll() is None
    t0 = time.time()
    r.close()
    assert proc.poll() is not None and time.time() - t0 < 5.0      # reaped, not left as a zombie
    assert proc.stdout.closed and r.proc is None
    r.close()                                        # idempotent


def test_missing_binaries_raise(monkeypatch, tmp_path):
    monkeypatch.setenv("PATH", str(tmp_path))
    with pytest.raises(RuntimeError, match="ffmpeg/ffprobe not found"):
        feeder.FFmpegReader(str(tmp_path / "x.mp4"))
