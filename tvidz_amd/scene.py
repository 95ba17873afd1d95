"""Scene-cut extraction on MI355X — host side of the seam at
/root/reference/inspector/app.py:202-232.

The reference spawns `ffmpeg -vf select=gt(scene\\,0.3),showinfo -f null -` (app.py:202-209)
and parses `pts_time:` from showinfo's stderr lines (app.py:216-232).  Here decoded 8-bit
luma planes are batched into HBM and scored by the HIP kernels in csrc/tvz_scene.hip through
the C ABI (include/tvz.h); torch tensors are used only as device buffers.

  detect_scene_cuts(frames, ...) -> iterator of (frame_idx, pts_time)
      yields, in presentation order, what the reference's parser would have appended to
      `scene_timestamps` BEFORE its consecutive-duplicate check (app.py:231), so the loop body of
      app.py:233-255 can be kept unchanged by a caller.

pts_time is the double recovered from showinfo's text (app.py:230 `float(...)`); which text
ffmpeg prints depends on its major version (see format_pts_time).
"""
from __future__ import annotations

import math
from typing import Iterable, Iterator, Optional, Sequence, Tuple, Union

import torch

from . import _lib

PTS_POLICY_G6 = "g6"          # FFmpeg <= 6.x: "%.6g"
PTS_POLICY_F6TRIM = "f6trim"  # FFmpeg >= 7.0: "%.*f" (6 decimals, more below 1.0), zeros trimmed
DEFAULT_THRESHOLD = 0.3       # inspector/app.py:206


def format_pts_time(pts: int, time_base: Tuple[int, int], policy: str = PTS_POLICY_G6) -> str:
    """Text showinfo prints after `pts_time:` (libavutil av_ts2timestr): av_q2d(tb) * pts."""
    val = (time_base[0] / time_base[1]) * pts
    if policy == PTS_POLICY_G6:
        return "%.6g" % val
    if policy == PTS_POLICY_F6TRIM:
        lg = -math.inf if val == 0 else math.floor(math.log10(abs(val)))
        precision = int(-lg) + 5 if (math.isfinite(lg) and lg < 0) else 6
        s = "%.*f" % (precision, val)
        last = len(s) - 1
        while last and s[last] == "0":
            last -= 1
        while last and s[last] != "f" and not s[last].isdigit():
            last -= 1
        return s[: last + 1]
    raise ValueError(f"unknown pts_time policy {policy!r}")


def pts_time_value(pts: int, time_base: Tuple[int, int], policy: str = PTS_POLICY_G6) -> float:
    """float(text) exactly as inspector/app.py:230 recovers it."""
    return float(format_pts_time(pts, time_base, policy))


def parse_showinfo_line(line: str) -> Optional[float]:
    """The reference's parser for one stderr line (inspector/app.py:218-230); None if the line
    carries no timestamp.  Used by the live-ffmpeg cross-check, not by the GPU path."""
    line = line.strip()
    if "showinfo" in line and "pts_time:" in line:
        try:
            return float(line.split("pts_time:")[1].split()[0])
        except Exception:
            return None
    return None


def _stream_ptr(stream: Optional[torch.cuda.Stream]) -> int:
    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


class SceneScorer:
    """Owns the scratch + output buffers for batches of up to `max_batch` frames of H x W luma,
    so the hot call allocates nothing.  Carries (last frame, last mafd) across batches so a
    stream scored in chunks gives the same scores as one scored whole."""

    def __init__(self, H: int, W: int, max_batch: int, device: Union[str, torch.device] = "cuda:0",
                 threshold: float = DEFAULT_THRESHOLD, keep_scores: bool = True, bitdepth: int = 8):
        """bitdepth 8: uint8 frames.  bitdepth 9..16 (yuv420p10 ...): samples in 16-bit words,
        passed as torch.int16 or torch.uint16 tensors (bit patterns of uint16)."""
        self.lib = _lib.load()
        self.bitdepth = int(bitdepth)
        if not 8 <= self.bitdepth <= 16:
            raise RuntimeError(f"bitdepth {bitdepth} out of range 8..16")
        self.dtype = torch.uint8 if self.bitdepth == 8 else torch.int16
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SceneScorer needs a GPU device (there is no CPU fallback)")
        self.H, self.W, self.max_batch = int(H), int(W), int(max_batch)
        self.threshold = float(threshold)
        self.ws_bytes = int(self.lib.tvz_scene_workspace_bytes(self.max_batch, self.H, self.W))
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
            self.sad = torch.empty(self.max_batch, dtype=torch.int64, device=self.device)
            self.selected = torch.empty(self.max_batch, dtype=torch.uint8, device=self.device)
            self.score = torch.empty(self.max_batch, dtype=torch.float64, device=self.device) if keep_scores else None
            self.mafd = torch.empty(self.max_batch, dtype=torch.float64, device=self.device)
            self.prev_frame = torch.empty((self.H, self.W), dtype=self.dtype, device=self.device)
        self.have_prev = False
        self.prev_mafd = 0.0

    def reset(self) -> None:
        self.have_prev = False
        self.prev_mafd = 0.0

    def _check(self, frames: torch.Tensor) -> None:
        ok_dtype = frames.dtype == torch.uint8 if self.bitdepth == 8 else frames.element_size() == 2
        if not ok_dtype or frames.dim() != 3 or frames.device != self.device:
            raise RuntimeError(f"frames must be a {'uint8' if self.bitdepth == 8 else '16-bit'} "
                               f"[T,H,W] tensor on {self.device}")
        if frames.shape[1] != self.H or frames.shape[2] != self.W:
            raise RuntimeError(f"frames are {tuple(frames.shape[1:])}, scorer is {(self.H, self.W)}")
        if frames.shape[0] > self.max_batch:
            raise RuntimeError(f"batch of {frames.shape[0]} frames exceeds max_batch={self.max_batch}")
        if frames.shape[0] and frames.stride(2) != 1:
            raise RuntimeError("pixels of a row must be contiguous (stride 1)")

    def score_batch(self, frames: torch.Tensor, stream: Optional[torch.cuda.Stream] = None,
                    carry: bool = True):
        """Enqueue scoring of one batch on `stream` (default: torch's current stream).
        Returns views (sad, mafd, score, selected) of length T into the scorer's buffers —
        valid until the next call.  With carry=True the batch continues the stream of the
        previous call (its last frame and mafd are the predecessor of frames[0])."""
        self._check(frames)
        T = int(frames.shape[0])
        if T == 0:
            return self.sad[:0], self.mafd[:0], (self.score[:0] if self.score is not None else None), self.selected[:0]
        use_prev = carry and self.have_prev
        es = frames.element_size()
        fn = self.lib.tvz_scene_scores_u8 if self.bitdepth == 8 else self.lib.tvz_scene_scores_u16
        with torch.cuda.device(self.device):          # launch on the frames' GPU, whatever is current
            rc = self._call(fn, frames, T, es, use_prev, stream)
        _lib.check(rc)
        return (self.sad[:T], self.mafd[:T], self.score[:T] if self.score is not None else None,
                self.selected[:T])

    def _call(self, fn, frames, T, es, use_prev, stream):
        return fn(
            frames.data_ptr(), T, self.H, self.W, frames.stride(0) * es, frames.stride(1) * es,
            self.prev_frame.data_ptr() if use_prev else None,
            self.prev_mafd if use_prev else 0.0, self.bitdepth, self.threshold,
            self.sad.data_ptr(), self.mafd.data_ptr(),
            self.score.data_ptr() if self.score is not None else None,
            self.selected.data_ptr(), self.workspace.data_ptr(), self.ws_bytes,
            (stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)

    def remember_tail(self, frames: torch.Tensor) -> None:
        """Keep the batch's last frame + mafd as the predecessor of the next batch."""
        T = int(frames.shape[0])
        if T == 0:
            return
        self.prev_frame.copy_(frames[T - 1].view(self.dtype) if frames.dtype != self.dtype else frames[T - 1])
        # an unscored first frame reports mafd 0 == ffmpeg's zero-initialised prev_mafd
        self.prev_mafd = float(self.mafd[T - 1].item())
        self.have_prev = True

    def luma_sad(self, frames: torch.Tensor, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
        """uint64 SAD per frame against its predecessor (sad[0] = 0), as int64 tensor view."""
        self._check(frames)
        T = int(frames.shape[0])
        if T:
            if self.bitdepth != 8:
                raise RuntimeError("luma_sad() is the 8-bit entry point; use score_batch for 16-bit")
            with torch.cuda.device(self.device):
                s = stream if stream is not None else torch.cuda.current_stream(self.device)
                _lib.check(self.lib.tvz_luma_sad_u8(frames.data_ptr(), T, self.H, self.W, frames.stride(0),
                                                    frames.stride(1), self.sad.data_ptr(),
                                                    self.workspace.data_ptr(), self.ws_bytes,
                                                    s.cuda_stream))
        return self.sad[:T]


def scene_select(sad: torch.Tensor, H: int, W: int, threshold: float = DEFAULT_THRESHOLD,
                 bitdepth: int = 8, prev_mafd: float = 0.0, have_prev: bool = False,
                 stream: Optional[torch.cuda.Stream] = None):
    """get_scene_score epilogue over a device SAD vector -> (selected u8, score f64, mafd f64)."""
    if sad.dtype != torch.int64 or sad.device.type != "cuda" or not sad.is_contiguous():
        raise RuntimeError("sad must be a contiguous int64 (uint64 bits) CUDA tensor")
    T = sad.numel()
    sel = torch.empty(T, dtype=torch.uint8, device=sad.device)
    score = torch.empty(T, dtype=torch.float64, device=sad.device)
    mafd = torch.empty(T, dtype=torch.float64, device=sad.device)
    if T:
        with torch.cuda.device(sad.device):
            _lib.check(_lib.load().tvz_scene_select(sad.data_ptr(), T, H, W, bitdepth, threshold,
                                                    prev_mafd, int(have_prev), sel.data_ptr(),
                                                    score.data_ptr(), mafd.data_ptr(),
                                                    _stream_ptr(stream)))
    return sel, score, mafd


def detect_scene_cuts(frames: Union[torch.Tensor, Iterable], time_base: Tuple[int, int] = (1, 30),
                      pts: Optional[Sequence[int]] = None, threshold: float = DEFAULT_THRESHOLD,
                      pts_policy: str = PTS_POLICY_G6, batch: int = 256,
                      device: Union[str, torch.device] = "cuda:0",
                      scorer: Optional[SceneScorer] = None) -> Iterator[Tuple[int, float]]:
    """Yield (frame_idx, pts_time) for every frame ffmpeg's `select=gt(scene,threshold)` keeps.

    frames: a uint8 [T,H,W] luma tensor (CUDA or CPU), an iterable of such chunks (a frame
            feeder), or the path of a media file (Y4M natively, anything else via the host's
            ffmpeg; its time base replaces `time_base`); chunks are scored in micro-batches of `batch` frames so a caller can stop
            early (the reference terminates ffmpeg at the first duplicate, app.py:249-255).
    pts:    presentation timestamps in time_base units (default: the frame index, i.e. a
            constant-frame-rate stream with time_base = 1/fps).
    """
    dev = torch.device(device)
    if isinstance(frames, (str, bytes)) or hasattr(frames, "__fspath__"):
        # a media file: Y4M is read directly, anything else through the host's ffmpeg as a raw
        # planar-YUV pipe (the decode half of the reference's single ffmpeg process)
        from .feeder import FFmpegReader, FrameFeeder, Y4MReader
        with open(frames, "rb") as f:
            magic = f.read(9)
        reader = Y4MReader(frames) if magic == b"YUV4MPEG2" else FFmpegReader(frames)
        time_base = reader.time_base
        frames = (d for _, d in FrameFeeder(reader, batch, dev))
    chunks = [frames] if isinstance(frames, torch.Tensor) else frames
    base = 0
    for chunk in chunks:
        if not isinstance(chunk, torch.Tensor):
            chunk = torch.as_tensor(chunk)
        if chunk.dim() != 3:
            raise RuntimeError("each chunk must be [T,H,W]")
        if scorer is None:
            scorer = SceneScorer(chunk.shape[1], chunk.shape[2], batch, dev, threshold,
                                 bitdepth=8 if chunk.dtype == torch.uint8 else 16)
        for s in range(0, chunk.shape[0], scorer.max_batch):
            part = chunk[s:s + scorer.max_batch]
            if part.device != scorer.device:
                part = part.to(scorer.device, non_blocking=True)
            _, _, _, sel = scorer.score_batch(part)
            scorer.remember_tail(part)
            idx = torch.nonzero(sel, as_tuple=False).flatten().cpu().tolist()
            for i in idx:
                n = base + s + i
                p = n if pts is None else int(pts[n])
                yield n, pts_time_value(p, time_base, pts_policy)
        base += chunk.shape[0]
