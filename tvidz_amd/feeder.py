"""Frame feeder: decoded 8-bit luma planes -> pinned ring buffer -> HBM micro-batches.

The step immediately before the scene kernel.  In the reference the decoder and the scorer are
one child process (`ffmpeg -i f -vf select=...,showinfo -f null -`, inspector/app.py:202-209);
here decode stays on the host (ffmpeg as a raw-video pipe, or a Y4M file) and only the luma
plane — the only plane ffmpeg's scene score reads for planar YUV — crosses PCIe, batched and
double-buffered so the copy of batch i+1 overlaps the scoring of batch i.

Readers yield uint8 [H,W] luma planes plus (time_base, total_frames):
  Y4MReader       YUV4MPEG2 files/streams (no external tool)
  FFmpegReader    `ffmpeg -i <file> -f rawvideo -` pipe in the stream's native 8-bit planar
                  YUV format (Y plane sliced out, no colour conversion); needs an ffmpeg binary
"""
from __future__ import annotations

import io
import os
import queue
import shutil
import subprocess
import threading
from typing import Iterator, Optional, Tuple

import numpy as np
import torch


def _chroma_bytes(tag: str, W: int, H: int) -> int:
    t = tag.lower()
    if t.startswith("mono"):
        return 0
    if t.startswith("420"):
        return 2 * ((W + 1) // 2) * ((H + 1) // 2)
    if t.startswith("422"):
        return 2 * ((W + 1) // 2) * H
    if t.startswith("444"):
        return 2 * W * H
    if t.startswith("411"):
        return 2 * ((W + 3) // 4) * H
    raise RuntimeError(f"unsupported Y4M colourspace C{tag}")


class Y4MReader:
    """Iterates the luma planes of a YUV4MPEG2 stream (8-bit only)."""

    def __init__(self, src):
        self._own = isinstance(src, (str, os.PathLike))
        self.f = open(src, "rb") if self._own else src
        header = self.f.readline()
        if not header.startswith(b"YUV4MPEG2"):
            raise RuntimeError("not a YUV4MPEG2 stream")
        self.W = self.H = 0
        fps = (30, 1)
        cs = "420jpeg"
        for tok in header.split()[1:]:
            k, v = tok[:1], tok[1:].decode()
            if k == b"W":
                self.W = int(v)
            elif k == b"H":
                self.H = int(v)
            elif k == b"F":
                n, d = v.split(":")
                fps = (int(n), int(d))
            elif k == b"C":
                cs = v
        # high bit depth: C420p10, C422p12, C444p16, Cmono16 ... = little-endian 16-bit samples
        self.bitdepth = 8
        base = cs
        for suffix, bd in (("p9", 9), ("p10", 10), ("p12", 12), ("p14", 14), ("p16", 16)):
            if cs.endswith(suffix):
                self.bitdepth, base = bd, cs[: -len(suffix)]
        if cs in ("mono9", "mono10", "mono12", "mono16"):
            self.bitdepth, base = int(cs[4:]), "mono"
        self.bps = 1 if self.bitdepth == 8 else 2
        if self.W <= 0 or self.H <= 0:
            raise RuntimeError("Y4M header without W/H")
        self.time_base = (fps[1], fps[0])       # ffmpeg's yuv4mpegpipe demuxer: 1/fps
        self._luma = self.W * self.H * self.bps
        self._skip = _chroma_bytes(base, self.W, self.H) * self.bps
        self.total_frames = 0
        if self._own:
            size = os.path.getsize(src) - len(header)
            self.total_frames = size // (6 + self._luma + self._skip)

    def __iter__(self) -> Iterator[np.ndarray]:
        while True:
            line = self.f.readline()
            if not line:
                return
            if not line.startswith(b"FRAME"):
                raise RuntimeError("corrupt Y4M stream: expected FRAME")
            buf = self.f.read(self._luma)
            if len(buf) < self._luma:
                return
            if self._skip:
                self.f.seek(self._skip, io.SEEK_CUR) if self.f.seekable() else self.f.read(self._skip)
            yield np.frombuffer(buf, dtype=np.uint8 if self.bps == 1 else "<u2").reshape(self.H, self.W)

    def read_into(self, out: np.ndarray) -> int:
        """Fill out[n,H,W] (uint8, or a 16-bit dtype for high bit depth) with up to n frames;
        returns how many were read."""
        n = 0
        flat = out.reshape(out.shape[0], -1).view(np.uint8)
        while n < out.shape[0]:
            line = self.f.readline()
            if not line:
                break
            if not line.startswith(b"FRAME"):
                raise RuntimeError("corrupt Y4M stream: expected FRAME")
            got = self.f.readinto(memoryview(flat[n]))
            if got < self._luma:
                break
            if self._skip:
                self.f.seek(self._skip, io.SEEK_CUR) if self.f.seekable() else self.f.read(self._skip)
            n += 1
        return n

    def close(self):
        if self._own:
            self.f.close()


def write_y4m(path: str, luma: np.ndarray, fps: Tuple[int, int] = (30, 1), chroma: str = "mono",
              bitdepth: int = 8) -> None:
    """Write [T,H,W] luma as Y4M — test/fixture helper.  uint8 for bitdepth 8 (chroma "mono",
    "420jpeg", "444"...), uint16 for 9..16 (written as C<chroma>p<bitdepth> / Cmono<bitdepth>,
    little-endian); chroma planes are neutral."""
    T, H, W = luma.shape
    bps = 1 if bitdepth == 8 else 2
    tag = chroma if bitdepth == 8 else (f"mono{bitdepth}" if chroma == "mono" else f"{chroma}p{bitdepth}")
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F{fps[0]}:{fps[1]} Ip A1:1 C{tag}\n".encode())
        if bps == 1:
            pad = bytes([128]) * _chroma_bytes(chroma, W, H)
        else:
            pad = np.full(_chroma_bytes(chroma, W, H), 1 << (bitdepth - 1), dtype="<u2").tobytes()
        for t in range(T):
            f.write(b"FRAME\n")
            f.write(np.ascontiguousarray(luma[t], dtype=np.uint8 if bps == 1 else "<u2").tobytes())
            f.write(pad)


class FFmpegReader:
    """Decode with the host's ffmpeg into a raw planar-YUV pipe and slice the Y plane out."""

    PLANAR8 = {"yuv420p": "420", "yuvj420p": "420", "yuv422p": "422", "yuvj422p": "422",
               "yuv444p": "444", "yuvj444p": "444", "gray": "mono"}

    def __init__(self, path: str, ffmpeg: Optional[str] = None, ffprobe: Optional[str] = None):
        self.ffmpeg = ffmpeg or shutil.which("ffmpeg")
        self.ffprobe = ffprobe or shutil.which("ffprobe")
        if not self.ffmpeg or not self.ffprobe:
            raise RuntimeError("ffmpeg/ffprobe not found on this host")
        out = subprocess.check_output(
            [self.ffprobe, "-v", "error", "-select_streams", "v:0", "-show_entries",
             "stream=width,height,pix_fmt,time_base,nb_frames", "-of", "default=noprint_wrappers=1", path],
            text=True)
        info = dict(l.split("=", 1) for l in out.strip().splitlines() if "=" in l)
        self.W, self.H = int(info["width"]), int(info["height"])
        self.pix_fmt = info["pix_fmt"]
        if self.pix_fmt not in self.PLANAR8:
            raise RuntimeError(f"pix_fmt {self.pix_fmt} is not 8-bit planar YUV")
        n, d = info["time_base"].split("/")
        self.time_base = (int(n), int(d))
        self.total_frames = int(info["nb_frames"]) if info.get("nb_frames", "").isdigit() else 0
        self.bitdepth, self.bps = 8, 1
        self._luma = self.W * self.H
        self._skip = _chroma_bytes(self.PLANAR8[self.pix_fmt], self.W, self.H)
        self.proc = subprocess.Popen([self.ffmpeg, "-v", "error", "-i", path, "-f", "rawvideo",
                                      "-pix_fmt", self.pix_fmt, "-"], stdout=subprocess.PIPE)
        self.f = self.proc.stdout

    def read_into(self, out: np.ndarray) -> int:
        n = 0
        flat = out.reshape(out.shape[0], -1)
        while n < out.shape[0]:
            got = self.f.readinto(memoryview(flat[n]))
            if got < self._luma:
                break
            if self._skip:
                self.f.read(self._skip)
            n += 1
        return n

    def close(self):
        try:
            self.proc.terminate()        # app.py:249-252: stop decoding at the first duplicate
        except Exception:
            pass


class FrameFeeder:
    """reader -> pinned ring (filled by a background thread) -> async H2D on a copy stream.
    Iterating yields (first_frame_index, device uint8 [n,H,W]) with the current stream already
    waiting on the copy; a slot is recycled when the consumer asks for the next batch."""

    def __init__(self, reader, batch: int = 256, device="cuda:0", n_slots: int = 3):
        self.reader = reader
        self.H, self.W = reader.H, reader.W
        self.batch = int(batch)
        self.device = torch.device(device)
        self.n_slots = n_slots
        self.bitdepth = getattr(reader, "bitdepth", 8)
        dt = torch.uint8 if self.bitdepth == 8 else torch.int16     # int16 carries the uint16 bits
        # straight from torch's caching host allocator: freed slots of a finished upload are reused
        # by the next one (a .pin_memory() copy would page-lock ~0.5 GB per slot every time)
        self.pinned = [torch.empty((self.batch, self.H, self.W), dtype=dt, pin_memory=True)
                       for _ in range(n_slots)]
        self.dev = [torch.empty((self.batch, self.H, self.W), dtype=dt, device=self.device)
                    for _ in range(n_slots)]
        self.copy_stream = torch.cuda.Stream(self.device)
        self._released = [None] * n_slots   # event on the consumer's stream: slot's kernels enqueued
        self._free: "queue.Queue[int]" = queue.Queue()
        self._full: "queue.Queue[Tuple[int, int]]" = queue.Queue()
        for i in range(n_slots):
            self._free.put(i)
        self._stop = threading.Event()
        self._err: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._fill, daemon=True)
        self._thread.start()

    def _fill(self):
        try:
            while not self._stop.is_set():
                slot = self._free.get()
                if slot < 0:
                    break
                n = self.reader.read_into(self.pinned[slot].numpy())
                self._full.put((slot, n))
                if n < self.batch:
                    break
        except BaseException as e:  # surfaced to the consumer
            self._err = e
            self._full.put((-1, 0))

    def __iter__(self):
        base = 0
        held = None
        done_evt = None
        try:
            while True:
                slot, n = self._full.get()
                if self._err is not None:
                    raise RuntimeError(f"frame reader failed: {self._err}") from self._err
                if n > 0:
                    if self._released[slot] is not None:    # kernels still reading the old contents
                        self.copy_stream.wait_event(self._released[slot])
                    with torch.cuda.stream(self.copy_stream):
                        self.dev[slot][:n].copy_(self.pinned[slot][:n], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(self.copy_stream)
                    torch.cuda.current_stream(self.device).wait_event(ev)
                if held is not None:
                    # the previous batch's kernels were enqueued before we got here; its pinned
                    # slot is free as soon as its H2D copy has finished
                    done_evt.synchronize()
                    rel = torch.cuda.Event()
                    rel.record(torch.cuda.current_stream(self.device))
                    self._released[held] = rel
                    self._free.put(held)
                if n == 0:
                    return
                held, done_evt = slot, ev
                yield base, self.dev[slot][:n]
                base += n
                if n < self.batch:
                    return
        finally:
            self.close()

    def close(self):
        self._stop.set()
        self._free.put(-1)
        try:
            self.reader.close()
        except Exception:
            pass
