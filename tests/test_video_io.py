"""frame / video I/O helpers of scripts/sampling/util.py:288-382, 689-762 against the outputs of the
reference's own load_img / load_video_keyframes (tests/golden/video_io.npz), plus GIF round trips and the depth
hint normalisations of the (pass-through) depth encoders."""
import os

import numpy as np
import pytest
import torch


def _frames(n=23, h=40, w=56):           # == tests/golden/make_golden.py: video_fixture_frames
    rs = np.random.RandomState(5)
    yy, xx = np.mgrid[0:h, 0:w]
    return [np.stack([(xx * 4 + 7 * i) % 256, (yy * 5 + 3 * i) % 256, rs.randint(0, 256, (h, w))], -1).astype(np.uint8)
            for i in range(n)]


def test_keyframe_loading_matches_reference_functions(golden_dir, tmp_path):
    from PIL import Image
    from scripts.sampling.util import keyframe_indices, load_img, load_video_keyframes
    z = np.load(os.path.join(golden_dir, "video_io.npz"))
    for i, fr in enumerate(_frames()):
        Image.fromarray(fr).save(os.path.join(tmp_path, f"frame_{i:04d}.png"))
    d = str(tmp_path)
    assert np.array_equal(load_video_keyframes(d, 20, 3, 5, size=(32, 48)).numpy(), z["dir_20_3_5_resized"])
    assert np.array_equal(load_video_keyframes(d, 20, 10, 9).numpy(), z["dir_20_10_9"])
    assert np.array_equal(load_video_keyframes(d, 8, 3, 6).numpy(), z["dir_8_3_6"])
    assert np.array_equal(load_img(os.path.join(d, "frame_0003.png"), (24, 40)).numpy(), z["img_resized"])
    assert keyframe_indices(23, 20, 10, 9).tolist() == [0, 2, 4, 6, 8, 10, 12, 14, 16]
    assert keyframe_indices(23, 20, 3, 5).tolist() == np.linspace(0, 22, 5).astype(int).tolist()    # too short: linspace
    with pytest.raises(NotImplementedError):
        load_video_keyframes("clip.mp4", 20, 3, 5)
    with pytest.raises(ValueError):
        load_video_keyframes("clip.avi", 20, 3, 5)


def test_gif_round_trip_and_grid(tmp_path):
    from PIL import Image
    from scripts.sampling.util import load_video_keyframes, perform_save_locally_video
    frames = np.stack(_frames(6, 16, 24)).astype(np.float32) / 255.0                  # (T, H, W, 3)
    frames = np.round(frames * 4) / 4                                                   # few colours: GIF palette is exact
    samples = torch.from_numpy(frames).permute(3, 0, 1, 2)[None]                         # (1, 3, T, H, W) in [0, 1]
    paths = perform_save_locally_video(str(tmp_path), samples, fps=4, return_savepaths=True)
    assert paths == [os.path.join(str(tmp_path), "gif", "animation-0000.gif")] and os.path.exists(paths[0])
    grid = np.array(Image.open(os.path.join(str(tmp_path), "grid", "grid-0000.png")))
    assert grid.shape == (16, 6 * 24, 3)
    back = load_video_keyframes(paths[0], 4, 4, 6)                                       # every frame
    assert back.shape == (6, 3, 16, 24)
    want = (torch.from_numpy((255.0 * frames).astype(np.uint8)).permute(0, 3, 1, 2).float() / 255.0) * 2 - 1
    assert torch.allclose(back, want, atol=1.5 / 255 * 2)
    with pytest.raises(NotImplementedError):
        perform_save_locally_video(str(tmp_path), samples, fps=4, savetype="mp4")


def test_depth_hint_normalisation():
    """DepthMidasEncoder / DepthZoeEncoder accept the raw depth of their (absent) networks and apply the reference's
    normalisation (encoders/modules.py:1376-1386 and 1324-1336)."""
    from sgm.modules.encoders.modules import DepthMidasEncoder, DepthZoeEncoder
    g = torch.Generator().manual_seed(1)
    raw = torch.rand(2, 1, 3, 8, 12, generator=g) * 7 + 1
    m = DepthMidasEncoder()(raw)
    d = raw - raw.min()
    d = d / d.max()
    assert m.shape == (2, 3, 3, 8, 12) and torch.allclose(m, -(d * 2 - 1).repeat(1, 3, 1, 1, 1))
    assert float(m.max()) == 1.0 and float(m.min()) == -1.0
    zo = DepthZoeEncoder()(raw)
    flat = raw.view(2, -1)
    n = flat.shape[1]
    vmin, vmax = flat.sort(dim=1).values[:, int(0.02 * n) - 1], flat.sort(dim=1).values[:, int(0.85 * n) - 1]
    want = (((raw - vmin[:, None, None, None, None]) / (vmax - vmin)[:, None, None, None, None]).clamp(0, 1) * 2 - 1)
    assert torch.allclose(zo, want.repeat(1, 3, 1, 1, 1))
    hint = (torch.rand(1, 1, 2, 8, 8) * 2 - 1).repeat(1, 3, 1, 1, 1)
    assert DepthMidasEncoder()(hint) is hint                                           # finished hint: passed through
    with pytest.raises(NotImplementedError, match="RGB"):                              # three different channels: not a hint
        DepthMidasEncoder()(torch.rand(1, 3, 2, 8, 8) * 2 - 1)
    with pytest.raises(NotImplementedError):
        DepthMidasEncoder()(["video.mp4"])
