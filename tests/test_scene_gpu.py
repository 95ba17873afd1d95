"""GPU parity: HIP scene-cut kernels (through the C ABI) vs the CPU oracle.

Integer SAD is compared bit-exactly; mafd/score are IEEE double/float32 operations in the same
order as the oracle, so they are compared bit-exactly too (tolerance 0).  The oracle's scene
half is a restatement of FFmpeg's algorithm: PARITY UNPINNED against ffmpeg itself (no binary,
no reference fixture) — see oracle/tvz_oracle.c.
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from tvidz_amd import _lib, scene, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_all(frames_np, threshold=0.3):
    T, H, W = frames_np.shape
    sad = oracle.luma_sad(frames_np)
    sel, score, mafd, _ = oracle.scene_select(sad, H, W, threshold)
    return sad, sel, score, mafd


def _gpu_all(frames, threshold=0.3, max_batch=None):
    T, H, W = frames.shape
    sc = scene.SceneScorer(H, W, max_batch or max(T, 1), DEV, threshold)
    sad, mafd, score, sel = sc.score_batch(frames)
    torch.cuda.synchronize()
    return (sad.cpu().numpy().view(np.uint64), sel.cpu().numpy(), score.cpu().numpy(),
            mafd.cpu().numpy())


@pytest.mark.parametrize("shape", [(40, 480, 854), (33, 1080, 1920), (3, 16, 64), (2, 1, 16),
                                   (130, 64, 64), (257, 48, 80), (5, 2160, 3840)])
def test_flat_path_random_frames_bit_exact(shape):
    T, H, W = shape
    g = torch.Generator(device=DEV); g.manual_seed(T * 7 + W)
    frames = torch.randint(0, 256, shape, dtype=torch.uint8, device=DEV, generator=g)
    sad, sel, score, mafd = _gpu_all(frames)
    o_sad, o_sel, o_score, o_mafd = _oracle_all(frames.cpu().numpy())
    assert (sad == o_sad).all()
    assert (mafd == o_mafd).all() and (score == o_score).all() and (sel == o_sel).all()


@pytest.mark.parametrize("shape", [(9, 37, 53), (20, 480, 853), (4, 5, 3), (70, 33, 130), (3, 7, 1)])
def test_generic_path_odd_sizes_bit_exact(shape):
    T, H, W = shape
    g = torch.Generator(device=DEV); g.manual_seed(W)
    frames = torch.randint(0, 256, shape, dtype=torch.uint8, device=DEV, generator=g)
    sad, sel, score, mafd = _gpu_all(frames)
    o_sad, o_sel, o_score, o_mafd = _oracle_all(frames.cpu().numpy())
    assert (sad == o_sad).all() and (score == o_score).all() and (sel == o_sel).all()


def test_padded_rows_and_frames_take_generic_path():
    T, H, W = 12, 30, 100
    g = torch.Generator(device=DEV); g.manual_seed(3)
    big = torch.randint(0, 256, (T, H + 5, W + 28), dtype=torch.uint8, device=DEV, generator=g)
    view = big[:, 2:2 + H, 7:7 + W]          # unaligned base, padded rows and frames
    assert not view.is_contiguous()
    sad, sel, score, _ = _gpu_all(view)
    o_sad, o_sel, o_score, _ = _oracle_all(view.cpu().numpy())
    assert (sad == o_sad).all() and (score == o_score).all() and (sel == o_sel).all()


def test_extremes_all_black_white_and_identical():
    T, H, W = 6, 64, 128
    frames = torch.zeros((T, H, W), dtype=torch.uint8, device=DEV)
    frames[1] = 255; frames[3] = 255; frames[4] = 255
    sad, sel, score, mafd = _gpu_all(frames)
    assert sad.tolist() == [0, 255 * H * W, 255 * H * W, 255 * H * W, 0, 255 * H * W]
    o_sad, o_sel, o_score, _ = _oracle_all(frames.cpu().numpy())
    assert (sel == o_sel).all() and (score == o_score).all()
    # back-to-back cuts: the second is suppressed by |mafd - prev_mafd| (f_select.c)
    assert sel.tolist() == [0, 1, 0, 0, 0, 1]


def test_empty_and_single_frame_batches():
    sc = scene.SceneScorer(32, 32, 8, DEV)
    sad, mafd, score, sel = sc.score_batch(torch.empty((0, 32, 32), dtype=torch.uint8, device=DEV))
    assert sad.numel() == 0 and sel.numel() == 0
    one = torch.randint(0, 256, (1, 32, 32), dtype=torch.uint8, device=DEV)
    sad, mafd, score, sel = sc.score_batch(one, carry=False)
    torch.cuda.synchronize()
    assert sad.item() == 0 and score.item() == 0.0 and sel.item() == 0
    # one-frame batches through the carried state: frame by frame == the whole stream
    frames = torch.randint(0, 256, (5, 32, 32), dtype=torch.uint8, device=DEV)
    frames[3:] //= 8
    o_sad, o_sel, o_score, _ = _oracle_all(frames.cpu().numpy())
    sc.reset()
    for t in range(5):
        sad, mafd, score, sel = sc.score_batch(frames[t:t + 1])
        assert sc.fetch_cuts() == ([0] if o_sel[t] else [])
        assert int(sad.item()) == int(o_sad[t]) and score.item() == o_score[t]


def test_synthetic_scenes_cuts_match_oracle_and_layout():
    T, H, W = 900, 480, 854
    frames, layout_cuts = synth.synth_luma(T, H, W, device=DEV, seed=synth.FRAME_SEED)
    sad, sel, score, mafd = _gpu_all(frames)
    o_sad, o_sel, o_score, o_mafd = _oracle_all(frames.cpu().numpy())
    assert (sad == o_sad).all() and (sel == o_sel).all() and (score == o_score).all()
    got = np.flatnonzero(sel).tolist()
    # every selected frame is a scene boundary of the generator (or its flash frame)...
    assert len(got) >= 3
    # ...and every plain boundary that is not back-to-back with another is selected
    for c in layout_cuts:
        if (c - 1) not in layout_cuts and (c + 1) not in layout_cuts and c not in got:
            # the fade scene's entry can be softened by its ramp; everything else must be a cut
            assert o_mafd[c] <= 30.0 or abs(o_mafd[c] - o_mafd[c - 1]) <= 30.0


def test_streaming_batches_equal_whole_stream():
    T, H, W = 300, 120, 160
    frames, _ = synth.synth_luma(T, H, W, device=DEV, seed=11, min_scene=10, max_scene=40)
    whole = _gpu_all(frames)
    sc = scene.SceneScorer(H, W, 64, DEV)
    sads, sels, scores = [], [], []
    for s in range(0, T, 50):   # batches that are not multiples of the kernel's time chunk
        part = frames[s:s + 50]
        sad, mafd, score, sel = sc.score_batch(part)
        sads.append(sad.cpu().numpy().view(np.uint64).copy()); sels.append(sel.cpu().numpy().copy())
        scores.append(score.cpu().numpy().copy())
    assert (np.concatenate(sads) == whole[0]).all()
    assert (np.concatenate(sels) == whole[1]).all()
    assert (np.concatenate(scores) == whole[2]).all()


def test_detect_scene_cuts_yields_reference_parser_values():
    T, H, W = 240, 96, 128
    frames, _ = synth.synth_luma(T, H, W, device=DEV, seed=5, min_scene=20, max_scene=50, adversarial=False)
    got = list(scene.detect_scene_cuts(frames, time_base=(1, 30), batch=64))
    o_sad, o_sel, _, _ = _oracle_all(frames.cpu().numpy())
    idx = np.flatnonzero(o_sel).tolist()
    assert [n for n, _ in got] == idx
    for n, ts in got:
        assert ts == oracle.pts_time_value(n, 1, 30, 0)
        assert scene.parse_showinfo_line(
            f"[Parsed_showinfo_1 @ 0x1] n:{len(idx):4d} pts:{n:7d} pts_time:{oracle.fmt_pts_time(n, 1, 30, 0):<7s} pos: 0") == ts


def test_standalone_sad_and_select_entry_points():
    T, H, W = 70, 64, 96
    frames = torch.randint(0, 256, (T, H, W), dtype=torch.uint8, device=DEV)
    sc = scene.SceneScorer(H, W, T, DEV)
    sad = sc.luma_sad(frames).clone()
    sel, score, mafd = scene.scene_select(sad, H, W, 0.3)
    torch.cuda.synchronize()
    o_sad, o_sel, o_score, o_mafd = _oracle_all(frames.cpu().numpy())
    assert (sad.cpu().numpy().view(np.uint64) == o_sad).all()
    assert (sel.cpu().numpy() == o_sel).all() and (score.cpu().numpy() == o_score).all()
    assert (mafd.cpu().numpy() == o_mafd).all()
    # continuing a stream: the predecessor's mafd is read from DEVICE memory (no by-value carry)
    sad_all = torch.from_numpy(o_sad.view(np.int64)).to(DEV)
    sel2, score2, _ = scene.scene_select(sad_all[20:].contiguous(), H, W, 0.3, prev_mafd=mafd[19:20].clone())
    torch.cuda.synchronize()
    assert (sel2.cpu().numpy() == o_sel[20:]).all() and (score2.cpu().numpy() == o_score[20:]).all()


def test_shape_variants_give_identical_results():
    """The kernel shape is a per-call argument (no global knob): every shape, same bits - also
    when the batch continues a stream through the device-resident state."""
    T, H, W = 150, 270, 480
    frames = torch.randint(0, 256, (T, H, W), dtype=torch.uint8, device=DEV)
    frames[70:] //= 3
    o_sad, o_sel, _, _ = _oracle_all(frames.cpu().numpy())
    sc = scene.SceneScorer(H, W, T, DEV)
    for U in (1, 2, 4, 8):
        for tc in (8, 64, 128, 256):
            for nt in (False, True):
                sad, _, _, sel = sc.score_batch(frames, carry=False, shape=_lib.shape(U, tc, nt))
                assert (sad.cpu().numpy().view(np.uint64) == o_sad).all(), (U, tc, nt)
                sc.reset()
                sc.score_batch(frames[:61], shape=_lib.shape(U, tc, nt))
                head = sc.fetch_cuts()
                sc.score_batch(frames[61:], shape=_lib.shape(U, tc, nt))
                got = head + [61 + i for i in sc.fetch_cuts()]
                assert got == np.flatnonzero(o_sel).tolist(), (U, tc, nt)
    with pytest.raises(RuntimeError, match="shape"):
        sc.score_batch(frames, carry=False, shape=_lib.shape(3, 64))


def test_chained_batches_in_one_hip_graph_equal_the_whole_stream():
    """the carried state (previous frame, previous mafd) lives on the device, so a
    chain of micro-batches has no host round trip in it and is captured in ONE HIP graph; replaying
    the graph on new frames equals the whole-stream oracle result."""
    T, H, W, NB = 32, 270, 480, 4
    sc = scene.SceneScorer(H, W, T, DEV)
    static = torch.zeros((NB * T, H, W), dtype=torch.uint8, device=DEV)
    cuts_out = torch.zeros((NB, 1 + sc.cuts_cap), dtype=torch.int32, device=DEV)
    sel_out = torch.zeros((NB, T), dtype=torch.uint8, device=DEV)
    g = torch.Generator(device=DEV); g.manual_seed(7)

    def chain():
        sc.reset()
        for b in range(NB):
            _, _, _, sel = sc.score_batch(static[b * T:(b + 1) * T])
            cuts_out[b].copy_(sc.cuts)
            sel_out[b].copy_(sel)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain()                                        # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        chain()
    for trial in range(3):
        frames, _ = synth.synth_luma(NB * T, H, W, device=DEV, seed=40 + trial, min_scene=9, max_scene=30)
        static.copy_(frames)
        graph.replay()
        torch.cuda.synchronize()
        _, o_sel, _, _ = _oracle_all(frames.cpu().numpy())
        assert (sel_out.cpu().numpy().reshape(-1) == o_sel).all(), trial
        got = []
        ch = cuts_out.cpu().numpy()
        for b in range(NB):
            got += [b * T + int(i) for i in ch[b, 1:1 + ch[b, 0]]]
        assert got == np.flatnonzero(o_sel).tolist() and len(got) >= 3


def test_bad_arguments_raise():
    sc = scene.SceneScorer(32, 32, 4, DEV)
    with pytest.raises(RuntimeError):
        sc.score_batch(torch.zeros((5, 32, 32), dtype=torch.uint8, device=DEV))   # > max_batch
    with pytest.raises(RuntimeError):
        sc.score_batch(torch.zeros((2, 32, 32), dtype=torch.float32, device=DEV))
    lib = _lib.load()
    rc = lib.tvz_luma_sad_u8(None, 4, 32, 32, 1024, 32, None, None, 0, None)
    assert rc != 0 and b"NULL" in lib.tvz_last_error()
    with pytest.raises(RuntimeError, match="libtvz error"):
        _lib.check(rc)


def test_full_size_property_linearity_of_sad():
    """BASELINE config size (1080p) property check without the oracle: SAD against a constant
    shift is exactly shift * H * W, and sad(a,b) == sad(b,a)."""
    H, W = 1080, 1920
    base = torch.randint(10, 200, (1, H, W), dtype=torch.uint8, device=DEV)
    frames = torch.cat([base, base + 7, base, base + 31, base + 31], dim=0)
    sad, sel, score, mafd = _gpu_all(frames)
    assert sad.tolist() == [0, 7 * H * W, 7 * H * W, 31 * H * W, 0]
    assert mafd.tolist() == [0.0, 7.0, 7.0, 31.0, 0.0]
    assert sel.tolist() == [0, 0, 0, 0, 0]   # |31-7| = 24 <= 30: not a cut


@pytest.mark.parametrize("shape,bitdepth", [((20, 270, 480), 10), ((9, 33, 51), 10), ((6, 1080, 1920), 10),
                                            ((12, 64, 64), 12), ((5, 17, 3), 16), ((70, 48, 80), 16)])
def test_16bit_luma_matches_oracle(shape, bitdepth):
    """yuv420p10-style planes: ffmpeg's ff_scene_sad16_c + mafd / 2^(bitdepth-8)."""
    rng = np.random.default_rng(shape[0] * 31 + bitdepth)
    f = rng.integers(0, 1 << bitdepth, size=shape, dtype=np.uint16)
    f[3:] = np.clip(f[3:].astype(np.int64) // 5 + (600 if bitdepth == 10 else 9000), 0, (1 << bitdepth) - 1).astype(np.uint16)
    T, H, W = shape
    sc = scene.SceneScorer(H, W, T, DEV, 0.3, bitdepth=bitdepth)
    d = torch.from_numpy(f.view(np.int16)).to(DEV)
    sad, mafd, score, sel = sc.score_batch(d)
    torch.cuda.synchronize()
    o_sad = oracle.luma_sad(f)
    o_sel, o_score, o_mafd, _ = oracle.scene_select(o_sad, H, W, 0.3, bitdepth=bitdepth)
    assert (sad.cpu().numpy().view(np.uint64) == o_sad).all()
    assert (mafd.cpu().numpy() == o_mafd).all() and (score.cpu().numpy() == o_score).all()
    assert (sel.cpu().numpy() == o_sel).all()
    # chunked == whole, through the carried previous frame
    sc2 = scene.SceneScorer(H, W, 4, DEV, 0.3, bitdepth=bitdepth)
    sels = []
    for s0 in range(0, T, 4):
        part = d[s0:s0 + 4]
        _, _, _, s_ = sc2.score_batch(part)
        sels.append(s_.cpu().numpy().copy())
    assert (np.concatenate(sels) == o_sel).all()


def test_16bit_padded_rows_generic_path():
    rng = np.random.default_rng(8)
    big = rng.integers(0, 1024, size=(7, 40, 70), dtype=np.uint16)
    d_big = torch.from_numpy(big.view(np.int16)).to(DEV)
    view = d_big[:, 3:33, 5:58]                      # padded rows, 2-byte aligned only
    sc = scene.SceneScorer(30, 53, 7, DEV, 0.3, bitdepth=10)
    sad, _, _, sel = sc.score_batch(view)
    torch.cuda.synchronize()
    o_sad = oracle.luma_sad(big[:, 3:33, 5:58])
    assert (sad.cpu().numpy().view(np.uint64) == o_sad).all()


def test_fuzz_shapes_strides_chunking():
    """Seeded fuzz: random sizes (flat and generic paths), padded/unaligned views, random chunking
    with carried state, 8- and 16-bit: always bit-exact against the oracle."""
    rng = np.random.default_rng(424242)
    for trial in range(40):
        H, W, T = int(rng.integers(1, 97)), int(rng.integers(1, 130)), int(rng.integers(1, 200))
        if trial % 5 == 0:
            H, W = int(rng.integers(1, 40)) * 2, int(rng.integers(1, 40)) * 8      # flat-eligible
        s16 = trial % 3 == 2
        bd = int(rng.choice([10, 12, 16])) if s16 else 8
        pad_h, pad_w, off_h, off_w = (int(x) for x in rng.integers(0, 5, 4))
        dt = np.uint16 if s16 else np.uint8
        big = rng.integers(0, 1 << bd, size=(T, H + pad_h + off_h, W + pad_w + off_w)).astype(dt)
        big[T // 2:] = (big[T // 2:] // 3).astype(dt)                 # a level change: real cuts
        view_np = big[:, off_h:off_h + H, off_w:off_w + W]
        d_big = torch.from_numpy(big.view(np.int16) if s16 else big).to(DEV)
        d_view = d_big[:, off_h:off_h + H, off_w:off_w + W]
        o_sad = oracle.luma_sad(view_np)
        o_sel, o_score, o_mafd, _ = oracle.scene_select(o_sad, H, W, 0.3, bitdepth=bd)
        step = int(rng.integers(1, T + 1))
        sc = scene.SceneScorer(H, W, step, DEV, 0.3, bitdepth=bd)
        sads, sels, scores = [], [], []
        for s0 in range(0, T, step):
            part = d_view[s0:s0 + step]
            sad, mafd, score, sel = sc.score_batch(part)
            sads.append(sad.cpu().numpy().view(np.uint64).copy())
            sels.append(sel.cpu().numpy().copy())
            scores.append(score.cpu().numpy().copy())
        assert (np.concatenate(sads) == o_sad).all(), (trial, H, W, T, step, bd)
        assert (np.concatenate(scores) == o_score).all() and (np.concatenate(sels) == o_sel).all(), trial


def test_scene_path_is_graph_capturable():
    """The scene entry points enqueue launches only (no allocation, no synchronisation), so a
    caller can capture a micro-batch step into a HIP graph and replay it on new frame contents."""
    T, H, W = 64, 270, 480
    sc = scene.SceneScorer(H, W, T, DEV)
    static = torch.zeros((T, H, W), dtype=torch.uint8, device=DEV)
    g = torch.Generator(device=DEV); g.manual_seed(1)
    a = torch.randint(0, 256, (T, H, W), dtype=torch.uint8, device=DEV, generator=g)
    b = torch.randint(0, 256, (T, H, W), dtype=torch.uint8, device=DEV, generator=g)
    b[T // 2:] //= 4
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        static.copy_(a)
        sc.score_batch(static, carry=False)           # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sc.score_batch(static, carry=False)
    for frames in (a, b, a):
        static.copy_(frames)
        graph.replay()
        torch.cuda.synchronize()
        o_sad, o_sel, o_score, _ = _oracle_all(frames.cpu().numpy())
        assert (sc.sad[:T].cpu().numpy().view(np.uint64) == o_sad).all()
        assert (sc.selected[:T].cpu().numpy() == o_sel).all() and (sc.score[:T].cpu().numpy() == o_score).all()


def test_detect_scene_cuts_from_a_file_path(tmp_path):
    from tvidz_amd import feeder
    T, H, W = 150, 96, 128
    frames, _ = synth.synth_luma(T, H, W, device=DEV, seed=9, min_scene=15, max_scene=40, adversarial=False)
    p = str(tmp_path / "clip.y4m")
    feeder.write_y4m(p, frames.cpu().numpy(), fps=(25, 1), chroma="420jpeg")
    got = list(scene.detect_scene_cuts(p, batch=64))
    _, o_sel, _, _ = _oracle_all(frames.cpu().numpy())
    assert got == [(int(i), oracle.pts_time_value(int(i), 1, 25, 0)) for i in np.flatnonzero(o_sel)]
    assert len(got) >= 2


@pytest.mark.parametrize("splits", [(300,), (100, 200), (1, 149, 150), (75, 75, 75, 75), (0, 120, 0, 180), (299, 1)])
def test_one_long_video_in_time_chunks_with_a_one_frame_halo(splits):
    """SURVEY 8e, scene scoring: a single long video over several GPUs = time chunks, each scored on its
    device against a one-frame halo, the SADs concatenated before the diff step.  One GPU here (every
    chunk on cuda:0: the data path - halo copy, per-chunk SAD, concatenation, ONE epilogue - is the same);
    the selected frames equal the oracle's over the whole video, cuts AT chunk boundaries included."""
    T, H, W = 300, 96, 160
    frames, _ = synth.synth_luma(T, H, W, device=DEV, seed=11, min_scene=9, max_scene=40)
    f_np = frames.cpu().numpy()
    # force cuts exactly at the boundaries the splits create (a level jump the filter must see across chunks)
    for b in np.cumsum(splits)[:-1]:
        if 0 < b < T:
            f_np[b:] = (f_np[b:].astype(np.int16) + 97).astype(np.uint8)
    frames = torch.from_numpy(f_np).to(DEV)
    _, sel, _, _ = _oracle_all(f_np)
    chunks, at = [], 0
    for n in splits:
        chunks.append(frames[at:at + n])
        at += n
    assert at == T
    got = scene.scene_cuts_chunked(chunks).tolist()
    assert got == np.flatnonzero(sel).tolist() and len(got) >= 5
    whole = [i for i, _ in scene.detect_scene_cuts(frames, batch=64)]
    assert got == whole
