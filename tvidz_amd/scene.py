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
    """Owns the scratch, output buffers and the DEVICE-RESIDENT stream state for batches of up to
    `max_batch` frames of H x W luma, so the hot call allocates nothing and nothing the scorer
    carries from one batch to the next (previous frame, previous mafd) ever visits the host: a
    stream scored in chunks gives the same scores as one scored whole, and a chain of chunks can
    be captured in one HIP graph."""

    def __init__(self, H: int, W: int, max_batch: int, device: Union[str, torch.device] = "cuda:0",
                 threshold: float = DEFAULT_THRESHOLD, keep_scores: bool = True, bitdepth: int = 8,
                 cuts_cap: Optional[int] = None):
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
        self.cuts_cap = int(cuts_cap if cuts_cap is not None else self.max_batch)
        self.ws_bytes = int(self.lib.tvz_scene_workspace_bytes(self.max_batch, self.H, self.W))
        self.state_bytes = int(self.lib.tvz_scene_state_bytes(self.H, self.W, 1 if self.bitdepth == 8 else 2))
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
            self.sad = torch.empty(self.max_batch, dtype=torch.int64, device=self.device)
            self.selected = torch.empty(self.max_batch, dtype=torch.uint8, device=self.device)
            self.score = torch.empty(self.max_batch, dtype=torch.float64, device=self.device) if keep_scores else None
            self.mafd = torch.empty(self.max_batch, dtype=torch.float64, device=self.device)
            # [0] = number of selected frames of the last batch, then their indices (ascending)
            self.cuts = torch.zeros(1 + self.cuts_cap, dtype=torch.int32, device=self.device)
            self.state = torch.empty(self.state_bytes, dtype=torch.uint8, device=self.device)
            self._cuts_host = torch.empty(1 + self.cuts_cap, dtype=torch.int32, pin_memory=True)
            self._cuts_event = torch.cuda.Event()
            self.reset()

    def reset(self, stream: Optional[torch.cuda.Stream] = None) -> None:
        """Start a new stream: the next frame scored is a first frame (score 0)."""
        with torch.cuda.device(self.device):
            s = stream if stream is not None else torch.cuda.current_stream(self.device)
            _lib.check(self.lib.tvz_scene_state_reset(self.state.data_ptr(), s.cuda_stream))

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
                    carry: bool = True, shape: int = _lib.SHAPE_AUTO):
        """Enqueue scoring of one batch on `stream` (default: torch's current stream).
        Returns views (sad, mafd, score, selected) of length T into the scorer's buffers —
        valid until the next call; `self.cuts` holds the compacted cut list of the batch.
        carry=True: the batch continues the scorer's stream (its device-resident state supplies
        the predecessor of frames[0] and is advanced to this batch's end).  carry=False: a
        self-contained batch (frames[0] is a first frame; the state is untouched).
        `shape`: per-call kernel-shape override (_lib.shape(U, tc, nt)); results never depend on it."""
        self._check(frames)
        T = int(frames.shape[0])
        if T == 0:
            return self.sad[:0], self.mafd[:0], (self.score[:0] if self.score is not None else None), self.selected[:0]
        es = frames.element_size()
        fn = self.lib.tvz_scene_scores_u8 if self.bitdepth == 8 else self.lib.tvz_scene_scores_u16
        with torch.cuda.device(self.device):          # launch on the frames' GPU, whatever is current
            s = stream if stream is not None else torch.cuda.current_stream(self.device)
            rc = fn(frames.data_ptr(), T, self.H, self.W, frames.stride(0) * es, frames.stride(1) * es,
                    self.state.data_ptr() if carry else None, self.bitdepth, self.threshold,
                    self.sad.data_ptr(), self.mafd.data_ptr(),
                    self.score.data_ptr() if self.score is not None else None,
                    self.selected.data_ptr(), self.cuts.data_ptr(), self.cuts_cap,
                    self.workspace.data_ptr(), self.ws_bytes, int(shape), s.cuda_stream)
        _lib.check(rc)
        return (self.sad[:T], self.mafd[:T], self.score[:T] if self.score is not None else None,
                self.selected[:T])

    def fetch_cuts(self, stream: Optional[torch.cuda.Stream] = None) -> list:
        """Indices (within the last batch) of the frames the filter selected: ONE small
        device-to-host copy + one event wait - the only host synchronisation of a micro-batch."""
        with torch.cuda.device(self.device):
            s = stream if stream is not None else torch.cuda.current_stream(self.device)
            with torch.cuda.stream(s):
                self._cuts_host.copy_(self.cuts, non_blocking=True)
                self._cuts_event.record(s)
        self._cuts_event.synchronize()
        n = int(self._cuts_host[0])
        if n > self.cuts_cap:
            raise RuntimeError(f"{n} cuts in one batch exceed cuts_cap={self.cuts_cap}")
        return self._cuts_host[1:1 + n].tolist()

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
                 bitdepth: int = 8, prev_mafd: Optional[torch.Tensor] = None,
                 stream: Optional[torch.cuda.Stream] = None):
    """get_scene_score epilogue over a device SAD vector -> (selected u8, score f64, mafd f64).
    prev_mafd: None (the vector starts a stream) or a 1-element float64 DEVICE tensor holding the
    mafd of the frame before sad[0] (the vector continues a stream; sad[0] is then a real SAD)."""
    if sad.dtype != torch.int64 or sad.device.type != "cuda" or not sad.is_contiguous():
        raise RuntimeError("sad must be a contiguous int64 (uint64 bits) CUDA tensor")
    if prev_mafd is not None and (prev_mafd.dtype != torch.float64 or prev_mafd.device != sad.device
                                  or prev_mafd.numel() != 1):
        raise RuntimeError("prev_mafd must be a 1-element float64 tensor on the same device")
    T = sad.numel()
    sel = torch.empty(T, dtype=torch.uint8, device=sad.device)
    score = torch.empty(T, dtype=torch.float64, device=sad.device)
    mafd = torch.empty(T, dtype=torch.float64, device=sad.device)
    if T:
        with torch.cuda.device(sad.device):
            _lib.check(_lib.load().tvz_scene_select(sad.data_ptr(), T, H, W, bitdepth, threshold,
                                                    prev_mafd.data_ptr() if prev_mafd is not None else None,
                                                    sel.data_ptr(), score.data_ptr(), mafd.data_ptr(),
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
            ffmpeg; the file's time base and per-frame pts then replace `time_base` / `pts`);
            chunks are scored in micro-batches of `batch` frames so a caller can stop early (the
            reference terminates ffmpeg at the first duplicate, app.py:249-255).
    pts:    presentation timestamps in time_base units, indexable by frame number (default: the
            frame index, i.e. a constant-frame-rate stream with time_base = 1/fps).  showinfo
            prints pts * time_base, so a container's stream time base (1/15360, 1/90000 ...) needs
            the real pts, not the frame index.
    """
    dev = torch.device(device)
    if isinstance(frames, (str, bytes)) or hasattr(frames, "__fspath__"):
        # a media file: Y4M is read directly, anything else through the host's ffmpeg as a raw
        # planar-YUV pipe (the decode half of the reference's single ffmpeg process)
        from .feeder import FrameFeeder, open_reader
        reader = open_reader(frames)
        time_base = reader.time_base
        pts = ReaderPts(reader)
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
            scorer.score_batch(part)
            for i in scorer.fetch_cuts():
                n = base + s + i
                p = n if pts is None else int(pts[n])
                yield n, pts_time_value(p, time_base, pts_policy)
        base += chunk.shape[0]


def scene_cuts_chunked(chunks: Sequence[torch.Tensor], threshold: float = DEFAULT_THRESHOLD,
                       gather_device: Optional[Union[str, torch.device]] = None) -> torch.Tensor:
    """ONE long video over several GPUs (scene scoring): `chunks` are consecutive time
    ranges of the video's 8-bit luma, uint8 [T_c,H,W] each on ITS device.  Every device computes the SADs
    of its chunk against a one-frame halo - the last frame of the chunk before it, copied over (2 MB at
    1080p: the only data that crosses devices) - so the SAD kernel still reads each luma byte once (this
    helper, a test vehicle, first copies every chunk behind its halo frame into one buffer: a production feeder
    would leave a frame of headroom in front of the chunk instead); the per-frame SADs
    (8 B per frame) are concatenated on `gather_device` BEFORE the diff step, where the epilogue of
    get_scene_score (mafd, |mafd - prev_mafd|, clip, threshold: inspector/app.py:206's `select` filter) runs
    once over the whole video.  No collective: a host that owns all chunks drives it (one process per GPU
    would ship its int64[T_c] SAD vector).  -> int64 indices of the selected frames, on the host.
    Identical to scoring the whole video on one device (tests/test_scene_gpu.py)."""
    if not chunks:
        return torch.empty(0, dtype=torch.int64)
    H, W = int(chunks[0].shape[1]), int(chunks[0].shape[2])
    out_dev = torch.device(gather_device) if gather_device is not None else chunks[0].device
    sads, prev_last = [], None
    for c, chunk in enumerate(chunks):
        if chunk.dtype != torch.uint8 or chunk.dim() != 3 or tuple(chunk.shape[1:]) != (H, W):
            raise RuntimeError("chunks must be uint8 [T,H,W] of one frame size")
        dev = chunk.device
        if dev.type != "cuda":
            raise RuntimeError("chunks live on their GPUs (no CPU path)")
        T = int(chunk.shape[0])
        if T == 0:
            continue
        with torch.cuda.device(dev):
            if prev_last is None:
                frames = chunk
            else:                                  # halo + chunk, contiguous for the flat kernel
                frames = torch.empty((T + 1, H, W), dtype=torch.uint8, device=dev)
                frames[0].copy_(prev_last, non_blocking=True)
                frames[1:].copy_(chunk, non_blocking=True)
            sc = SceneScorer(H, W, int(frames.shape[0]), dev, threshold)
            sad = sc.luma_sad(frames)              # sad[0] = 0: the halo (or the video's first frame)
            sads.append((sad if prev_last is None else sad[1:]).to(out_dev, non_blocking=True))
            prev_last = chunk[T - 1]
    for chunk in chunks:                           # the copies above are ordered on each source device's stream
        if chunk.device.type == "cuda":
            torch.cuda.current_stream(chunk.device).synchronize()
    if not sads:
        return torch.empty(0, dtype=torch.int64)
    with torch.cuda.device(out_dev):
        whole = torch.cat(sads).contiguous()
        sel, _, _ = scene_select(whole, H, W, threshold)
        return torch.nonzero(sel, as_tuple=False).flatten().cpu()


class ReaderPts:
    """pts[n] of a reader: its own per-frame pts if it has them, else the frame index."""

    def __init__(self, reader):
        self.reader = reader

    def __getitem__(self, n: int) -> int:
        f = getattr(self.reader, "pts_of", None)
        return int(f(n)) if f is not None else int(n)
